#!/usr/bin/env python3
"""What a device allocation costs on this box: hipMalloc / hipFree through torch's allocator (empty_cache between), by size."""
import time, torch
torch.cuda.init()
x = torch.empty(1, device="cuda"); torch.cuda.synchronize()
for gb in (0.25, 1, 2, 4, 8, 16):
    n = int(gb * (1 << 30))
    ts = []
    for _ in range(3):
        torch.cuda.empty_cache(); torch.cuda.synchronize()
        t0 = time.perf_counter(); a = torch.empty(n, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize(); t1 = time.perf_counter()
        a.zero_(); torch.cuda.synchronize(); t2 = time.perf_counter()
        del a; torch.cuda.empty_cache(); torch.cuda.synchronize(); t3 = time.perf_counter()
        ts.append((t1 - t0, t2 - t1, t3 - t2))
    print("%.2f GiB: malloc %.1f ms, first touch (fill) %.1f ms, free %.1f ms" % (gb, min(t[0] for t in ts) * 1e3, min(t[1] for t in ts) * 1e3, min(t[2] for t in ts) * 1e3))
