#!/usr/bin/env python3
"""gci_bgzf_inflate_device as the library runs it (GCI_INFLATE=wave default | lane) on a HiFi BAM with SEQ / QUAL of realistic
entropy: output against the known inflated stream, how the members fared with the wave decoder, time per call (HIP events around
the call, inputs resident).  Usage: inflate_product.py [scale of chr19] [repeats]"""
import os, sys, tempfile, time
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import numpy as np, torch
from gci_amd import synth, hostio
from gci_amd.device import Engine
from gci_amd.formats import bam as bamfmt

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.25
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
kind = os.environ.get("READS", "hifi")                    # READS=ont: ONT reads (longer, noisier qualities: members that deflate less)
rs = synth.simulate_reads((("chr19", int(61_707_364 * scale)),), 40, kind, seed=synth.seed_for(2, 0))
stream, _ = synth.to_bam_stream(rs, seq_qual="random", seed=7)
p = os.path.join(tempfile.mkdtemp(), "x.bam")
bamfmt.write_bam_stream(p, stream, level=int(os.environ.get("BAM_LEVEL", "1")), threads=hostio.default_threads())
raw = np.fromfile(p, dtype=np.uint8)
pos, isz = hostio.bgzf_blocks(raw)
e = Engine(0)
d_raw = e.upload_padded(raw)
torch.cuda.synchronize()
ms = []
for _ in range(reps):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(e.stream):
        a.record(e.stream)
        out = e.bgzf_inflate(None, pos, isz, check_crc=bool(int(os.environ.get("CHECK_CRC", "1"))), d_raw=d_raw)
        b.record(e.stream)
    torch.cuda.synchronize()
    ms.append(a.elapsed_time(b))
ok = np.array_equal(out.cpu().numpy(), stream)
print("reads %s, level %s, mode %s: %d members, %.1f MB -> %.1f MB; calls (tables up + inflate + crc) %s ms -> %.1f GB/s out; equal to the stream: %s; %s" % (
    kind, os.environ.get("BAM_LEVEL", "1"), os.environ.get("GCI_INFLATE", "wave"), isz.shape[0], raw.shape[0] / 1e6, stream.shape[0] / 1e6, ["%.2f" % x for x in ms],
    stream.shape[0] / 1e6 / min(ms), ok, e.inflate_stats()), flush=True)
if not ok:
    got = out.cpu().numpy()
    off = np.concatenate([[0], np.cumsum(isz)]).astype(np.int64)
    badm = [m for m in range(isz.shape[0]) if not np.array_equal(got[off[m]:off[m + 1]], stream[off[m]:off[m + 1]])]
    print("members that differ: %d, first %s" % (len(badm), badm[:10]))
    m = badm[0]
    d = np.flatnonzero(got[off[m]:off[m + 1]] != stream[off[m]:off[m + 1]])
    print("member %d: isize %d, %d bytes differ, first at %s" % (m, isz[m], d.shape[0], d[:20]))
    sys.exit(1)
