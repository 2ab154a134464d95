"""Phase timestamps (shader clock) inside k_tile_build's sparse path, built with -DGCI_TILE_TRACE into a separate .so."""
import sys, subprocess, os, ctypes, numpy as np, torch
sys.path.insert(0, '.')
from gci_amd import build, synth, _lib
so = "/tmp/libgci_ttrace.so"
subprocess.run([build.hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DGCI_TILE_TRACE", "-o", so] + build.SOURCES + ["-lz", "-lpthread"], check=True)
_lib.LIB_PATH = so
from gci_amd.device import Engine, JoinInput
from gci_amd._lib import BuildOpts
e = Engine(0)
L = 61_707_364
rs = synth.simulate_reads((("chr19", L),), 40, "hifi", seed=synth.seed_for(2, 0))
stream, offs = synth.to_bam_stream(rs)
e.set_layout([L])
d_bam, d_off = e.to_device(stream), e.to_device(offs)
recs = e.bam_filter(d_bam, d_off, e.to_device(np.zeros(1, np.int32)), 30, 50, 0.1, 0.9)
ivl, cnt = e.name_join([JoinInput(recs, d_bam, d_off, 36)], 0.9)
track = e.new_track()
nt = (L + 4095) // 4096
trace = torch.zeros(nt * 8, dtype=torch.int64, device=e.device)
os.environ["GCI_TILE_TRACE_PTR"] = str(trace.data_ptr())
p = lambda t: ctypes.c_void_p(t.data_ptr())
toff = torch.zeros(2, dtype=torch.int64, device=e.device)
o = BuildOpts(); o.flank = 15; o.want_text = 1; o.d_contig_text_off = toff.data_ptr()
e._chk(e.lib.gci_depth_build_begin(e.ctx, p(ivl), p(cnt), int(ivl.shape[0]), ctypes.byref(o)), "b")
text = torch.empty(int(toff[1].item()) + 64, dtype=torch.uint8, device=e.device)
for _ in range(4):
    e._chk(e.lib.gci_depth_build_begin(e.ctx, p(ivl), p(cnt), int(ivl.shape[0]), ctypes.byref(o)), "b")
    e._chk(e.lib.gci_depth_build_finish(e.ctx, p(track), p(text), int(text.shape[0])), "f")
torch.cuda.synchronize()
tr = trace.cpu().numpy().reshape(nt, 8)
t0 = tr[:, 0].min()
print("tiles", nt, "kernel span (cycles)", int(tr[:, 7].max() - t0))
rel = tr - tr[:, :1]
names = {1: "loads done", 2: "sorted", 3: "depth issued", 4: "text edges issued", 7: "text issued / end"}
for i in (1, 2, 3, 4, 7):
    v = rel[:, i]
    print("phase %d %-20s median %8.0f  p90 %8.0f  max %8.0f cycles since wave start" % (i, names[i], np.median(v), np.percentile(v, 90), v.max()))
starts = np.sort(tr[:, 0] - t0)
print("wave start times: p10 %d p50 %d p90 %d max %d" % tuple(np.percentile(starts, [10, 50, 90, 100]).astype(int)))
ends = np.sort(tr[:, 7] - t0)
print("wave end times:   p10 %d p50 %d p90 %d max %d" % tuple(np.percentile(ends, [10, 50, 90, 100]).astype(int)))
