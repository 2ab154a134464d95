#!/usr/bin/env python3
"""Two bench steps in flight on two streams (two library contexts, no cross-stream dependency) vs one."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from gci_amd import _lib
from gci_amd.device import Engine
torch.cuda.set_device(0)
e0 = Engine(0)
s1 = torch.cuda.Stream()
with torch.cuda.stream(s1):
    e1 = Engine(0, stream=s1)
w0 = bench.Workload(e0, 0, 1, bench.CHR19_LEN, 40.0)
with torch.cuda.stream(s1):
    w1 = bench.Workload(e1, 0, 1, bench.CHR19_LEN, 40.0)
    for _ in range(3):
        w1.step()
for _ in range(3):
    w0.step()
torch.cuda.synchronize()
N = 40
for mode in ("one", "two", "one", "two"):
    torch.cuda.synchronize(); t = time.perf_counter()
    if mode == "one":
        for _ in range(N):
            w0.step()
    else:
        for k in range(N):
            if k & 1:
                with torch.cuda.stream(s1):
                    w1.step()
            else:
                w0.step()
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(mode, "%.1f us/step" % (dt / N * 1e6), flush=True)
# kernel durations under overlap
for e in (e0, e1):
    e.profile_enable(1 << _lib.PROF_DEPTH_SCAN); e.profile_read()
for k in range(N):
    if k & 1:
        with torch.cuda.stream(s1):
            w1.step()
    else:
        w0.step()
torch.cuda.synchronize()
print("k_tile_build under overlap:", {i: {k: round(ms / n * 1e3, 1) for k, (ms, n) in e.profile_read().items()} for i, e in enumerate((e0, e1))})
assert w0.check() and w1.check()
