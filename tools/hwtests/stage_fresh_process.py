#!/usr/bin/env python3
"""The staging ring over a file that ANOTHER process wrote and read before: is the slow first pass (20 - 25 GB/s against 50) a property of
the file's pages or of the process that reads them?  Usage: stage_fresh_process.py make|read [threads]"""
import ctypes, os, sys, time
HERE = os.path.dirname(os.path.abspath(__file__)); sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import numpy as np
p, N = "/dev/shm/_fresh.bin", 16 << 30
if sys.argv[1] == "make":
    blk = np.random.default_rng(3).integers(0, 256, 64 << 20, dtype=np.uint8).tobytes()
    with open(p, "wb") as f:
        for _ in range(N // len(blk)): f.write(blk)
    sys.exit(0)
import torch
from gci_amd.device import Engine
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 12
e = Engine(0); lib = e.lib
dst = torch.empty(N, dtype=torch.uint8, device="cuda"); st = torch.cuda.Stream()
h = ctypes.c_void_p(); assert lib.gci_stage_create(e.ctx, 64 << 20, 4, threads, ctypes.byref(h)) == 0
out = []
for _ in range(int(sys.argv[3]) if len(sys.argv) > 3 else 2):
    raw = np.memmap(p, dtype=np.uint8, mode="r")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    assert lib.gci_stage_send(e.ctx, h, ctypes.c_void_p(raw.ctypes.data), N, ctypes.c_void_p(dst.data_ptr()), ctypes.c_void_p(st.cuda_stream), int(os.environ.get("FORGET", "1")), 1) == 0
    st.synchronize(); out.append(N / (time.perf_counter() - t0) / 1e9); del raw
print("process %d, %d threads, forget %s: passes %s GB/s" % (os.getpid(), threads, os.environ.get("FORGET", "1"), ", ".join("%.1f" % x for x in out)), flush=True)
