#!/usr/bin/env python3
"""The PAF filter of filter() (GCI.py:211-254) on the GPU (gci_paf_filter_device) against the native host filter
(gci_paf_filter): wall time for a whole-chromosome-scale PAF, same records out.  Usage: exp_paf.py [n_lines]"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gci_amd import hostio
from gci_amd.device import Engine, REC_DTYPE
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
rng = np.random.default_rng(5)
targets = ["chr%d" % i for i in range(1, 26)]
q = rng.integers(0, int(n * 0.95), n)
qlen = rng.integers(8000, 25000, n)
qs = rng.integers(0, 200, n); qe = qlen - rng.integers(0, 200, n)
t = rng.integers(0, 26, n)
ts = rng.integers(0, 100_000_000, n); aln = qe - qs
nm = (aln * rng.choice([0.88, 0.95, 0.999], n)).astype(np.int64)
mq = rng.choice([0, 20, 40, 60], n, p=[0.05, 0.05, 0.1, 0.8])
names = (["chr%d" % i for i in range(1, 26)] + ["chrUn"])
lines = ["m64011_190830_220126/%d/ccs\t%d\t%d\t%d\t+\t%s\t150000000\t%d\t%d\t%d\t%d\t%d\ttp:A:P\tcm:i:2000\ts1:i:17000\tdv:f:0.0010\trl:i:50" % (
    q[i], qlen[i], qs[i], qe[i], names[t[i]], ts[i], ts[i] + aln[i], nm[i], aln[i], mq[i]) for i in range(n)]
tmp = tempfile.mkdtemp()
p = os.path.join(tmp, "big.paf")
open(p, "w").write("\n".join(lines) + "\n")
print("PAF: %d lines, %.1f MB" % (n, os.path.getsize(p) / 1e6), flush=True)
e = Engine(0)
e.paf_filter([p], targets, 30, 50, 0.9)
for k in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = e.paf_filter([p], targets, 30, 50, 0.9)
    torch.cuda.synchronize(); dt_gpu = time.perf_counter() - t0
t0 = time.perf_counter(); host = hostio.paf_filter([p], targets, 30, 50, 0.9); dt_host = time.perf_counter() - t0
a = out[0].recs.cpu().numpy().reshape(-1).view(REC_DTYPE); b = host[0][0].reshape(-1).view(REC_DTYPE)
key = lambda x: sorted(zip(x["name_hash"].tolist(), x["contig"].tolist(), x["start"].tolist(), x["end"].tolist(), x["qlen"].tolist(), x["flags"].tolist()))
print("gpu (incl. file read + H2D) %.3f s = %.1f M lines/s; host (%d threads) %.3f s = %.1f M lines/s; %d queries; same records: %s" % (
    dt_gpu, n / dt_gpu / 1e6, hostio.default_threads(), dt_host, n / dt_host / 1e6, a.shape[0], key(a) == key(b)))
