#!/usr/bin/env python3
"""How fast the bytes of a file in the page cache (tmpfs) reach a pinned host buffer, by method and thread count: what feeds the
device's inflate at genome size (19 GB/s of file bytes keep it busy).  mmap + memcpy against pread straight into the pinned slot."""
import mmap, os, sys, time
from concurrent.futures import ThreadPoolExecutor
import numpy as np, torch
GB = float(sys.argv[1]) if len(sys.argv) > 1 else 6.0
p = "/dev/shm/_staging_bw.bin"
n = int(GB * (1 << 30))
with open(p, "wb") as f:
    blk = np.random.default_rng(1).integers(0, 256, 64 << 20, dtype=np.uint8).tobytes()
    for _ in range(n // len(blk)):
        f.write(blk)
n = os.path.getsize(p)
SLOT = 64 << 20
slots = [torch.empty(SLOT, dtype=torch.uint8).pin_memory() for _ in range(4)]
views = [s.numpy() for s in slots]
dev = torch.empty(n, dtype=torch.uint8, device="cuda")
stream = torch.cuda.Stream()
def run(method, T):
    pool = ThreadPoolExecutor(T)
    raw = np.memmap(p, dtype=np.uint8, mode="r") if method == "mmap" else None
    fd = os.open(p, os.O_RDONLY)
    free_at = [None] * 4
    torch.cuda.synchronize(); t0 = time.perf_counter()
    k = 0
    for a in range(0, n, SLOT):
        b = min(n, a + SLOT)
        if free_at[k] is not None: free_at[k].synchronize()
        step = (-(-(b - a) // T) + 4095) // 4096 * 4096
        if method == "mmap":
            jobs = [pool.submit(np.copyto, views[k][x - a:min(b, x + step) - a], raw[x:min(b, x + step)]) for x in range(a, b, step)]
        else:
            mv = memoryview(views[k])
            jobs = [pool.submit(os.preadv, fd, [mv[x - a:min(b, x + step) - a]], x) for x in range(a, b, step)]
        for j in jobs: j.result()
        with torch.cuda.stream(stream):
            dev[a:b].copy_(slots[k][:b - a], non_blocking=True)
            ev = torch.cuda.Event(); ev.record(stream)
        free_at[k] = ev
        k = (k + 1) % 4
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    os.close(fd); pool.shutdown()
    del raw
    return n / dt / 1e9
for method in ("mmap", "pread"):
    for T in (4, 8, 12, 16):
        print("%-6s %2d threads: %.1f GB/s" % (method, T, max(run(method, T) for _ in range(2))), flush=True)
t0 = time.perf_counter(); a = np.memmap(p, dtype=np.uint8, mode="r"); dev.copy_(torch.from_numpy(np.asarray(a))); torch.cuda.synchronize()
print("pageable copy of the mapping: %.1f GB/s" % (n / (time.perf_counter() - t0) / 1e9))
os.remove(p)
