"""What a plain fill of N bytes reaches on this box, by method: torch zero_() / fill_() on uint8, int32, int64 views, hipMemsetAsync
through the library (gci_memset) -- the ceiling bench.py holds k_tile_build against (roofline.fill_ceiling_gbs)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gci_amd.device import Engine
import ctypes
e = Engine(0)
def timed(fn, nbytes, reps=5):
    for _ in range(2): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(e.stream)
    for _ in range(reps): fn()
    b.record(e.stream); torch.cuda.synchronize()
    return nbytes / (a.elapsed_time(b) / reps * 1e-3) / 1e9
for gb in (4, 12, 22):
    n = int(gb * 1e9) // 16 * 16
    with torch.cuda.stream(e.stream):
        x = torch.empty(n, dtype=torch.uint8, device=e.device)
        r = {"uint8 zero_": timed(lambda: x.zero_(), n), "int32 zero_": timed(lambda: x.view(torch.int32).zero_(), n),
             "int32 fill_(7)": timed(lambda: x.view(torch.int32).fill_(7), n), "int64 fill_(7)": timed(lambda: x.view(torch.int64).fill_(7), n),
             "gci_memset": timed(lambda: e.lib.gci_memset(e.ctx, ctypes.c_void_p(x.data_ptr()), 0, n), n)}
    print("%2d GB: " % gb + ", ".join("%s %.0f GB/s" % kv for kv in r.items()), flush=True)
    del x; torch.cuda.empty_cache()
