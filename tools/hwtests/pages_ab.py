#!/usr/bin/env python3
"""A/B of builds of the record-pages kernels (gci_bam_pages_size / _write) on one box: the heads stream of a 40x HiFi file at `scale` of
the genome is made once, every library named on the command line (GCI_LIB_PATH) times the two calls over it in a process of its own.
usage: pages_ab.py SCALE lib1.so lib2.so ...   (a name without a slash: gci_amd/csrc/libgci_hip_<name>.so; "product": the product build)"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np

if os.environ.get("PAGES_AB_CHILD"):
    import ctypes
    from gci_amd import hbm
    from gci_amd.device import Engine
    s, o = np.load(os.environ["PAGES_AB_CHILD"] + ".s.npy", mmap_mode="r"), np.load(os.environ["PAGES_AB_CHILD"] + ".o.npy")
    eng = Engine(0, backend="native")
    T = eng.T
    d_s, d_o = eng.to_device(np.asarray(s)), eng.to_device(o)
    pg = eng.bam_pages(d_s, d_o, False)
    eng.sync()
    lib, ctx = eng.lib, eng.ctx
    n, nb = int(o.shape[0]), int(s.shape[0])
    res = {}
    for what in ("size", "write"):
        a, b = T.Event(enable_timing=True), T.Event(enable_timing=True)
        reps = 10
        a.record(eng.stream)
        for _ in range(reps):
            if what == "size":
                h = (ctypes.c_uint64 * 3)()
                lib.gci_bam_pages_size(ctx, ctypes.c_void_p(d_s.data_ptr()), nb, ctypes.c_void_p(d_o.data_ptr()), n, 0, pg.page_bytes, h)
            else:
                lib.gci_bam_pages_write(ctx, ctypes.c_void_p(d_s.data_ptr()), nb, ctypes.c_void_p(d_o.data_ptr()), n, 0, ctypes.c_void_p(pg.buf.data_ptr()), int(pg.buf.shape[0]))
        b.record(eng.stream)
        b.synchronize()
        res[what] = a.elapsed_time(b) / reps
    gb = (nb + int(pg.buf.shape[0])) / 1e9
    print("%-28s records %d, pages %d: size %.3f ms, write %.3f ms (%.2f GB moved: %.2f TB/s)" % (
        os.environ.get("PAGES_AB_LABEL", "?"), n, pg.n_pages, res["size"], res["write"], gb, gb / res["write"] / 1e3), flush=True)
    sys.exit(0)

from gci_amd import workloads, synth
scale = float(sys.argv[1])
inp = workloads.genome_dual(scale, 40.0, contigs=synth.CHM13, n_files=1)
base = "/dev/shm/pages_ab_%d" % os.getpid()
np.save(base + ".s.npy", inp.files[0].stream)
np.save(base + ".o.npy", inp.files[0].offsets)
try:
    for name in sys.argv[2:]:
        path = None if name == "product" else name if "/" in name else os.path.join(ROOT, "gci_amd", "csrc", "libgci_hip_%s.so" % name)
        env = dict(os.environ, PAGES_AB_CHILD=base, PAGES_AB_LABEL=name, PYTHONPATH=ROOT)
        if path:
            env["GCI_LIB_PATH"] = path
        r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True)
        print(r.stdout.strip() or r.stderr[-600:], flush=True)
finally:
    for e in (".s.npy", ".o.npy"):
        os.remove(base + e)
