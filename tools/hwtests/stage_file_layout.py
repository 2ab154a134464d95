#!/usr/bin/env python3
"""Does it matter to the staging ring WHO wrote the file?  A 16 GB tmpfs file written by one process against the same file written by
16 processes side by side (as workloads.write_bgzf_from_heads writes a BAM: its pages end up on whatever NUMA node each writer ran
on), read through gci_stage_send (12 threads, 64 MB slots); and the threads' own placement (numactl is not in the image: taskset)."""
import ctypes, os, sys, time, multiprocessing as mp
HERE = os.path.dirname(os.path.abspath(__file__)); sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import numpy as np, torch
from gci_amd.device import Engine
N = 16 << 30
def write_part(args):
    p, k, parts = args
    blk = np.random.default_rng(k).integers(0, 256, 64 << 20, dtype=np.uint8).tobytes()
    fd = os.open(p, os.O_WRONLY)
    per = N // parts
    for off in range(k * per, (k + 1) * per, len(blk)): os.pwrite(fd, blk, off)
    os.close(fd)
def make(p, parts):
    with open(p, "wb") as f: f.truncate(N)
    if parts == 1: write_part((p, 0, 1))
    else:
        with mp.get_context("fork").Pool(parts) as pool: pool.map(write_part, [(p, k, parts) for k in range(parts)])
e = Engine(0); lib = e.lib
dst = torch.empty(N, dtype=torch.uint8, device="cuda"); st = torch.cuda.Stream()
h = ctypes.c_void_p(); assert lib.gci_stage_create(e.ctx, 64 << 20, 4, 12, ctypes.byref(h)) == 0
def send(p):
    raw = np.memmap(p, dtype=np.uint8, mode="r")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    assert lib.gci_stage_send(e.ctx, h, ctypes.c_void_p(raw.ctypes.data), N, ctypes.c_void_p(dst.data_ptr()), ctypes.c_void_p(st.cuda_stream), 1, 1) == 0
    st.synchronize(); dt = time.perf_counter() - t0; del raw
    return N / dt / 1e9
try:
    print("cpus allowed:", len(os.sched_getaffinity(0)), "numa nodes:", len([d for d in os.listdir("/sys/devices/system/node") if d.startswith("node")]))
except Exception as ex: print(ex)
for parts in (1, 16):
    p = "/dev/shm/_layout_%d.bin" % parts
    make(p, parts)
    print("written by %2d process(es): passes %s GB/s" % (parts, ", ".join("%.1f" % send(p) for _ in range(3))), flush=True)
    os.remove(p)
