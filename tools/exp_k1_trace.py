"""Phase timestamps (shader clock) inside k_bam_filter, built with -DGCI_K1_TRACE into a separate .so."""
import sys, subprocess, os, ctypes, numpy as np, torch
sys.path.insert(0, '.')
from gci_amd import build, synth, _lib
so = "/tmp/libgci_trace.so"
subprocess.run([build.hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DGCI_K1_TRACE", "-DGCI_K1_WALL", "-o", so] + build.SOURCES, check=True)
_lib.LIB_PATH = so
from gci_amd.device import Engine
e = Engine(0)
L = 61_707_364
rs = synth.simulate_reads((("chr19", L),), 40, "hifi", seed=synth.seed_for(2, 0))
if len(sys.argv) > 1: rs = rs.take(np.arange(0, len(rs), int(sys.argv[1])))
stream, offs = synth.to_bam_stream(rs)
e.set_layout([L])
d_bam, d_off = e.to_device(stream), e.to_device(offs)
sel = e.to_device(np.zeros(1, np.int32))
out = torch.empty((len(rs), 32), dtype=torch.uint8, device=e.device)
nb = (len(rs) + 31) // 32
trace = torch.zeros(nb * 16, dtype=torch.int64, device=e.device)
os.environ["GCI_K1_TRACE_PTR"] = str(trace.data_ptr())
for ab in (0,):
    for _ in range(4):
        e.bam_filter(d_bam, d_off, sel, 30, 50, 0.1, 0.9, out=out, check=False)
    e.profile_enable(1); e.profile_read()
    for _ in range(10):
        e.bam_filter(d_bam, d_off, sel, 30, 50, 0.1, 0.9, out=out, check=False)
    pr = e.profile_read()
    print("ablate", ab, "k_bam_filter us", round(pr["k_bam_filter"][0] / pr["k_bam_filter"][1] * 1e3, 1))
os.environ["GCI_K1_ABLATE"] = "0"
for _ in range(3):
    e.bam_filter(d_bam, d_off, sel, 30, 50, 0.1, 0.9, out=out, check=False)
torch.cuda.synchronize()
tr = trace.cpu().numpy().reshape(nb, 16)
t0 = tr[:, 0].min()
rel = (tr - tr[:, :1])
ok = tr[:, 7] > 0
print("blocks", nb, "kernel span (cycles)", int(tr[:, 7].max() - t0))
for i in (1, 2, 8, 9, 10, 3, 4, 5, 6, 7):
    v = rel[ok][:, i]
    print("phase %d: median %8.0f  p90 %8.0f  max %8.0f cycles since block start" % (i, np.median(v), np.percentile(v, 90), v.max()))
# slots 14 / 15: wall_clock64() (100 MHz, one time base for the whole device) at block start / end
w0, w1 = tr[:, 14], tr[:, 15]
base = w0.min()
st, en = (w0 - base) / 100.0, (w1 - base) / 100.0
print("wall clock (us): block starts p1 %.1f p10 %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f | ends p50 %.1f p99 %.1f max %.1f | life p50 %.1f p90 %.1f" % (
    *np.percentile(st, [1, 10, 50, 90, 99, 100]), *np.percentile(en, [50, 99, 100]), *np.percentile(en - st, [50, 90])))
hist, edges = np.histogram(st, bins=12)
print("start histogram (us):", [(round(float(a), 1), int(h)) for a, h in zip(edges[:-1], hist)])
