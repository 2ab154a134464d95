#!/usr/bin/env python3
"""How fast the library's staging ring (gci_stage_send / gci_stage_send_fd) brings a file in tmpfs to the device, by thread count, slot
size and method, the device otherwise idle -- the host side of the uploads in isolation."""
import ctypes, os, sys, time
HERE = os.path.dirname(os.path.abspath(__file__)); sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import numpy as np, torch
from gci_amd.device import Engine
GB = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
p = "/dev/shm/_stage_bw.bin"
blk = np.random.default_rng(1).integers(0, 256, 64 << 20, dtype=np.uint8).tobytes()
with open(p, "wb") as f:
    for _ in range(int(GB * (1 << 30)) // len(blk)): f.write(blk)
n = os.path.getsize(p)
e = Engine(0); lib = e.lib
dst = torch.empty(n, dtype=torch.uint8, device="cuda"); st = torch.cuda.Stream()
def run(method, threads, slot_mb, slots=4):
    h = ctypes.c_void_p()
    assert lib.gci_stage_create(e.ctx, slot_mb << 20, slots, threads, ctypes.byref(h)) == 0
    best = 0.0
    for _ in range(2):
        raw = np.memmap(p, dtype=np.uint8, mode="r"); fd = os.open(p, os.O_RDONLY)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        if method == "mmap":
            rc = lib.gci_stage_send(e.ctx, h, ctypes.c_void_p(raw.ctypes.data), n, ctypes.c_void_p(dst.data_ptr()), ctypes.c_void_p(st.cuda_stream), 1, 1)
        else:
            rc = lib.gci_stage_send_fd(e.ctx, h, fd, 0, n, ctypes.c_void_p(dst.data_ptr()), ctypes.c_void_p(st.cuda_stream), 1)
        assert rc == 0
        st.synchronize(); best = max(best, n / (time.perf_counter() - t0) / 1e9)
        os.close(fd); del raw
    lib.gci_stage_free(h)
    return best
for method in ("mmap", "pread"):
    print(method + ": " + ", ".join("%d thr %.1f" % (t, run(method, t, 64)) for t in (4, 8, 12, 16, 24, 32)) + " GB/s (64 MB slots); 16 thr, 128 MB slots: %.1f; 16 thr, 8 x 32 MB: %.1f" % (run(method, 16, 128), run(method, 16, 32, 8)), flush=True)
os.remove(p)
