"""Phase timestamps (shader clock) inside k_bam_filter on a genome-geometry HEADS stream (a quarter of configs[2]'s first file),
built with -DGCI_K1_TRACE into a separate .so: where a workgroup's time goes at the size that matters."""
import sys, subprocess, os, numpy as np, torch
sys.path.insert(0, '.')
from gci_amd import build, workloads, _lib
so = "/tmp/libgci_trace.so"
subprocess.run([build.hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DGCI_K1_TRACE", "-DGCI_K1_WALL", "-o", so] + build.SOURCES + ["-lz", "-lpthread"], check=True)
_lib.LIB_PATH = so
from gci_amd.device import Engine
inp = workloads.genome_dual(float(sys.argv[1]) if len(sys.argv) > 1 else 0.25, 40.0, n_files=1)
f = inp.files[0]
e = Engine(0)
e.set_layout(inp.lengths)
d_bam, d_off = e.to_device(f.stream), e.to_device(f.offsets)
sel = e.to_device(np.arange(len(inp.contigs), dtype=np.int32))
n = int(f.offsets.shape[0])
out = torch.empty((n, 32), dtype=torch.uint8, device=e.device)
nb = (n + 31) // 32
trace = torch.zeros(nb * 16, dtype=torch.int64, device=e.device)
os.environ["GCI_K1_TRACE_PTR"] = str(trace.data_ptr())
for _ in range(4):
    e.bam_filter(d_bam, d_off, sel, 30, 50, 0.1, 0.9, out=out, check=False, heads=True)
e.profile_enable(1); e.profile_read()
for _ in range(10):
    e.bam_filter(d_bam, d_off, sel, 30, 50, 0.1, 0.9, out=out, check=False, heads=True)
pr = e.profile_read()
print("records", n, "k_bam_filter us (traced build)", round(pr["k_bam_filter"][0] / pr["k_bam_filter"][1] * 1e3, 1))
torch.cuda.synchronize()
tr = trace.cpu().numpy().reshape(nb, 16)
rel = (tr - tr[:, :1])
ok = tr[:, 7] > 0
print("blocks", nb, "with a full fast path in thread 0:", int(ok.sum()))
prev = 0
for i, what in ((1, "offsets loaded, head loads issued"), (2, "barrier (heads staged)"), (8, "core fields"), (9, "aux loads issued"), (10, "CIGAR tail from global"),
                (3, "staged CIGAR"), (4, "LDS fence"), (5, "aux walk"), (6, "name hash"), (7, "decision + store")):
    v = rel[ok][:, i]
    print("phase %2d %-36s median %7.0f (+%5.0f)  p90 %7.0f cycles since block start" % (i, what, np.median(v), np.median(v) - prev, np.percentile(v, 90)))
    prev = np.median(v)
w0, w1 = tr[:, 14], tr[:, 15]
base = w0.min()
st, en = (w0 - base) / 100.0, (w1 - base) / 100.0
print("wall clock (us): block starts p50 %.1f p99 %.1f | ends max %.1f | life p50 %.2f p90 %.2f" % (*np.percentile(st, [50, 99]), en.max(), *np.percentile(en - st, [50, 90])))
