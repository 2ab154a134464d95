#!/usr/bin/env python3
"""Host -> device copy rate of this box: pageable and pinned, one stream and two, 256 MB and 2 GB (what bounds number (2) of
SURVEY.md section 8(d): 5.3 GB of heads streams per step)."""
import time
import torch

dev = torch.device("cuda:0")
for mb in (256, 2048):
    n = mb << 20
    page = torch.empty(n, dtype=torch.uint8).fill_(1)
    pin = torch.empty(n, dtype=torch.uint8, pin_memory=True).fill_(1)
    dst = torch.empty(n, dtype=torch.uint8, device=dev)
    dst2 = torch.empty(n, dtype=torch.uint8, device=dev)
    for name, src in (("pageable", page), ("pinned", pin)):
        for _ in range(2):
            dst.copy_(src, non_blocking=True); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            dst.copy_(src, non_blocking=True)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        print("%5d MB %-8s one stream : %.1f GB/s" % (mb, name, n / dt / 1e9))
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    h = n // 2
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        with torch.cuda.stream(s1): dst[:h].copy_(pin[:h], non_blocking=True)
        with torch.cuda.stream(s2): dst2[:h].copy_(pin[h:], non_blocking=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print("%5d MB pinned   two streams: %.1f GB/s" % (mb, n / dt / 1e9))
    # device -> host
    t0 = time.perf_counter()
    for _ in range(3):
        pin.copy_(dst, non_blocking=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print("%5d MB D2H pinned           : %.1f GB/s" % (mb, n / dt / 1e9))
