#!/usr/bin/env python3
"""Device time of the two gci_depth_deflate_* passes on the chr19 / 40x track (and of the whole Engine.depth_deflate call)."""
import os, sys, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from gci_amd.device import Engine
e = Engine(0)
w = bench.Workload(e, 0, 1, bench.CHR19_LEN, 40.0)
w.step(); torch.cuda.synchronize()
track = w.track
for _ in range(2):
    torch.cuda.synchronize(); t = time.perf_counter(); blobs = e.depth_deflate(track); torch.cuda.synchronize()
    print("depth_deflate call: %.2f ms, %d bytes" % ((time.perf_counter() - t) * 1e3, sum(map(len, blobs))))
# the two kernels alone
L = bench.CHR19_LEN
MB = 64 * 4096
elem = np.arange(0, L, MB, dtype=np.uint64); cnt = np.minimum(MB, L - elem.astype(np.int64)).astype(np.uint32)
nm = len(elem)
d_elem, d_cnt = e.to_device(elem), e.to_device(cnt)
tb = torch.empty(nm * 64, dtype=torch.int32, device=e.device)
mb, crc, isz = (torch.empty(nm, dtype=torch.int32, device=e.device) for _ in range(3))
p = lambda t: ctypes.c_void_p(t.data_ptr())
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
for rep in range(3):
    ev[0].record()
    e._chk(e.lib.gci_depth_deflate_size(e.ctx, p(track), p(d_elem), p(d_cnt), nm, p(tb), p(mb), p(crc), p(isz)), "size")
    ev[1].record()
    sizes = mb.cpu().numpy().view(np.uint32).astype(np.uint64)
    offs = np.zeros(nm + 1, dtype=np.uint64); np.cumsum(sizes, out=offs[1:])
    out = torch.empty(int(offs[nm]), dtype=torch.uint8, device=e.device); d_off = e.to_device(offs[:nm].copy())
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    e._chk(e.lib.gci_depth_deflate_write(e.ctx, p(track), p(d_elem), p(d_cnt), nm, p(tb), p(crc), p(isz), p(d_off), p(out), int(offs[nm])), "write")
    b.record(); torch.cuda.synchronize()
    print("size pass %.1f us, write pass %.1f us, members %d" % (ev[0].elapsed_time(ev[1]) * 1e3, a.elapsed_time(b) * 1e3, nm))
