#!/usr/bin/env python3
"""k_bgzf_inflate builds side by side on one box: the product library and variants made by tools/build_variant.sh (INF_SORTED_GLOBAL,
INF_WAVES), each at several members-per-wave settings, on the chr19 40x HiFi BAM with realistic SEQ / QUAL -- kernel time between
events with the file already on the device, output checked against the stream once per build.
usage: exp_inflate_variants.py [scale] name[@lanes] ..."""
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
TMP = "/tmp/gci_inf_exp"


def child():
    import torch
    from gci_amd.device import Engine
    raw, pos, isz, stream = (np.load(TMP + "_%s.npy" % k) for k in ("raw", "pos", "isz", "stream"))
    e = Engine(0)
    d_raw = e.upload_padded(raw)
    d = e.bgzf_inflate(None, pos, isz, d_raw=d_raw)
    ok = bool(np.array_equal(d.cpu().numpy(), stream))
    best = {}
    for crc in (False, True):
        ts = []
        for _ in range(4):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            d = e.bgzf_inflate(None, pos, isz, check_crc=crc, d_raw=d_raw)
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        best[crc] = min(ts)
    print("%-14s inflate %.1f ms (%.1f GB/s out)   with crc %.1f ms   output %s" % (
        os.environ.get("EXP_NAME", "?"), best[False], stream.shape[0] / best[False] / 1e6, best[True], "ok" if ok else "WRONG"), flush=True)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        return child()
    from gci_amd import hostio, synth
    from gci_amd.formats import bam as bamfmt
    import tempfile
    scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
    rs = synth.simulate_reads((("chr19", int(61_707_364 * scale)),), 40, "hifi", seed=synth.seed_for(2, 0))
    stream, _ = synth.to_bam_stream(rs, seq_qual="random", seed=7)
    p = os.path.join(tempfile.mkdtemp(), "x.bam")
    bamfmt.write_bam_stream(p, stream, level=1, threads=hostio.default_threads())
    raw = np.fromfile(p, dtype=np.uint8)
    pos, isz = hostio.bgzf_blocks(raw)
    for k, v in (("raw", raw), ("pos", pos), ("isz", isz), ("stream", stream)):
        np.save(TMP + "_%s.npy" % k, v)
    print("file: %d members, %.2f GB -> %.2f GB" % (isz.shape[0], raw.shape[0] / 1e9, stream.shape[0] / 1e9), flush=True)
    for spec in sys.argv[2:] or ["product"]:
        name, _, lanes = spec.partition("@")
        env = dict(os.environ, EXP_NAME=spec)
        if name != "product":
            env["GCI_LIB_PATH"] = os.path.join(ROOT, "gci_amd", "csrc", "libgci_hip_%s.so" % name)
        if lanes:
            env["GCI_INFLATE_LANES"] = lanes
        subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, check=False)


if __name__ == "__main__":
    main()
