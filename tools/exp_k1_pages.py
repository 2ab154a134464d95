"""Where the paged record filter's time goes: the product build against builds with one part of the per-record work
compiled out (tools/build_variant.sh ... -DPGX_NO_*; results of those are wrong, timing only).  One box, one call.
usage: python tools/exp_k1_pages.py [scale] [variant ...]"""
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
TMP = "/tmp/gci_k1_exp"


def child():
    import torch
    from gci_amd.device import Engine
    stream, offs = np.load(TMP + "_s.npy"), np.load(TMP + "_o.npy")
    eng = Engine(0)
    d_s, d_o = eng.to_device(stream), eng.to_device(offs)
    ref_sel = eng.to_device(np.arange(25, dtype=np.int32))
    out = {}
    for what in ("pages", "stream"):
        if what == "pages":
            pg = eng.bam_pages(d_s, d_o, False)
            run = lambda: eng.bam_filter_pages(pg, ref_sel, 30, 50, 0.1, 0.9, check=False)          # noqa: E731
        else:
            run = lambda: eng.bam_filter(d_s, d_o, ref_sel, 30, 50, 0.1, 0.9, heads=True, check=False)   # noqa: E731
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            run()
        b.record()
        torch.cuda.synchronize()
        out[what] = a.elapsed_time(b) / 10
    print("%-10s pages %.3f ms   stream %.3f ms   (%d records, %.2f GB heads)" % (
        os.environ.get("GCI_VARIANT", "product"), out["pages"], out["stream"], offs.shape[0], stream.shape[0] / 1e9), flush=True)


def main():
    scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.3
    variants = sys.argv[2:] or ["product", "nocigar", "noaux", "nohash", "nodecide", "none"]
    from gci_amd import workloads
    t0 = time.time()
    inp = workloads.genome_dual(scale, 40.0)
    f = inp.files[0]
    np.save(TMP + "_s.npy", f.stream)
    np.save(TMP + "_o.npy", f.offsets)
    print("input: %d records, generated in %.0f s" % (f.offsets.shape[0], time.time() - t0), flush=True)
    del inp
    for v in variants:
        env = dict(os.environ, GCI_VARIANT=v)
        v, _, pb = v.partition("@")                              # name@page_bytes
        if pb:
            env["GCI_PAGE_BYTES"] = pb
        if v != "product":
            env["GCI_LIB_PATH"] = os.path.join(ROOT, "gci_amd", "csrc", "libgci_hip_%s.so" % v)
        subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, check=False)


if __name__ == "__main__":
    if "--child" in sys.argv:
        child()
    else:
        main()
