#!/usr/bin/env python3
"""How long the arena's slabs take to make: gci_dev_reserve of 4 / 16 / 32 / 64 / 96 GB in fresh processes (hipMalloc of one slab),
and the same in slabs of 16 GB (GCI_ARENA_SLAB_GB caps the geometric growth; reserve asks for what is missing in one piece)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r"""
import ctypes, sys, time
sys.path.insert(0, %r)
from gci_amd import _lib
lib = _lib.load()
n = ctypes.c_int(0); lib.gci_dev_count(ctypes.byref(n)); lib.gci_dev_mem_info(0, None, None)
gb, pieces = int(sys.argv[1]), int(sys.argv[2])
t0 = time.perf_counter()
got = ctypes.c_uint64(0)
for k in range(pieces):
    lib.gci_dev_reserve(0, (gb << 30) * (k + 1) // pieces, ctypes.byref(got))
t1 = time.perf_counter()
p = ctypes.c_void_p()
lib.gci_dev_malloc(0, 1 << 30, ctypes.byref(p))
t2 = time.perf_counter()
print("reserve %%3d GB in %%d piece(s): %%.3f s (reserved %%.1f GB); a 1 GB block out of it: %%.6f s" %% (gb, pieces, t1 - t0, got.value / 2**30, t2 - t1), flush=True)
""" % ROOT
for gb, pieces in ((4, 1), (16, 1), (32, 1), (64, 1), (96, 1), (64, 4), (64, 16)):
    r = subprocess.run([sys.executable, "-c", CHILD, str(gb), str(pieces)], capture_output=True, text=True)
    print(r.stdout.strip() or r.stderr[-400:], flush=True)
