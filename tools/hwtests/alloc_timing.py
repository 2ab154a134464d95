#!/usr/bin/env python3
"""How long device allocations take on this box (the first run of an ingestion makes ~30 GB of them with nothing queued on the
device to hide behind: DESIGN.md section 8): torch.empty and gci_malloc of 1 / 4 / 8 GB, first and second time, on an idle device
and beside a pinned H2D copy; uname and the amdgpu module's version beside it, since the boxes of the pool differ."""
import os, sys, time, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ctypes, torch
from gci_amd.device import Engine
print("kernel:", os.uname().release, "| amdgpu:", (subprocess.run("cat /sys/module/amdgpu/version 2>/dev/null || modinfo -F version amdgpu 2>/dev/null", shell=True, capture_output=True, text=True).stdout.strip() or "?"))
e = Engine(0)
torch.cuda.synchronize()
def t(f):
    t0 = time.perf_counter(); r = f(); torch.cuda.synchronize(); return r, (time.perf_counter() - t0) * 1e3
for gb in (1, 4, 8):
    n = gb << 30
    a, ms1 = t(lambda: torch.empty(n, dtype=torch.uint8, device="cuda"))
    del a; torch.cuda.empty_cache()
    a, ms2 = t(lambda: torch.empty(n, dtype=torch.uint8, device="cuda"))
    p = ctypes.c_void_p()
    _, ms3 = t(lambda: e.lib.gci_malloc(e.ctx, ctypes.c_uint64(n), ctypes.byref(p)))
    print("%d GB: torch.empty %.1f ms, again after empty_cache %.1f ms, gci_malloc %.1f ms" % (gb, ms1, ms2, ms3), flush=True)
    e.lib.gci_free(e.ctx, p); del a; torch.cuda.empty_cache()
# beside a pinned copy
src = torch.empty(4 << 30, dtype=torch.uint8).pin_memory()
dst = torch.empty(4 << 30, dtype=torch.uint8, device="cuda")
st = torch.cuda.Stream()
torch.cuda.synchronize()
t0 = time.perf_counter()
with torch.cuda.stream(st):
    dst.copy_(src, non_blocking=True)
a = torch.empty(8 << 30, dtype=torch.uint8, device="cuda")
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("8 GB torch.empty beside a 4 GB pinned H2D copy: the allocation returned after %.1f ms, the copy was over after %.1f ms (alone: ~75 ms)" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3))
# first touch: is a fresh allocation slower to write than one that has been written before?  (demand paging -- XNACK / retry faults --
# would make the FIRST kernel that touches a buffer pay for its pages; the boxes of the pool may differ in that)
print("xnack / retry:", subprocess.run("(rocminfo 2>/dev/null | grep -i -m2 xnack); echo HSA_XNACK=$HSA_XNACK; cat /sys/module/amdgpu/parameters/noretry 2>/dev/null",
                                       shell=True, capture_output=True, text=True).stdout.replace("\n", " | "))
del a, dst
torch.cuda.empty_cache()
for gb in (8, 8):
    b = torch.empty(gb << 30, dtype=torch.uint8, device="cuda")
    _, f1 = t(lambda: b.fill_(1))
    _, f2 = t(lambda: b.fill_(2))
    print("%d GB fresh buffer: first fill %.1f ms (%.0f GB/s), second fill %.1f ms (%.0f GB/s)" % (gb, f1, gb * 1.074 / f1 * 1e3, f2, gb * 1.074 / f2 * 1e3), flush=True)
    del b
    torch.cuda.empty_cache()
