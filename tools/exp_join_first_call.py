#!/usr/bin/env python3
"""Why the command line's join takes seconds at genome size when its kernels take milliseconds: the FIRST gci_name_join_count
of a process against the second, on the join input the command line builds (runs concatenated).  Usage: [scale] [runs]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gci_amd import workloads, synth, pipeline
from gci_amd.device import Engine, JoinInput
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.25
n_runs = int(sys.argv[2]) if len(sys.argv) > 2 else 6
inp = workloads.genome_dual(scale, 40.0, contigs=synth.CHM13)
eng = Engine(0)
eng.set_layout(inp.lengths)
ref_sel = eng.to_device(np.arange(len(inp.names), dtype=np.int32))
def t(label, fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize()
    print("%-50s %.4f s" % (label, time.perf_counter() - t0), flush=True); return r
ins = []
for f in inp.files:
    d_s, d_o = eng.to_device(f.stream), eng.to_device(f.offsets)
    n = int(f.offsets.shape[0])
    parts = []
    # the records in `n_runs` runs, each paged and filtered on its own, kept and concatenated as the command line does
    bounds = [n * k // n_runs for k in range(n_runs + 1)]
    first = int(f.offsets[0])
    for k in range(n_runs):
        a, b = bounds[k], bounds[k + 1]
        lo = int(f.offsets[a]); hi = int(f.offsets[b]) if b < n else int(f.stream.shape[0])
        sub = d_s[lo:hi]
        off = (d_o[a:b] - lo).contiguous()
        ji = pipeline._filter_stream(eng, sub, off, False, ref_sel, (30, 50, 0.1, 0.9), rec_idx_base=a)
        parts.append(pipeline._keep_part(eng, ji))
    ins.append(t("concat parts of a file", lambda: pipeline._concat_parts(eng, parts)))
for k in range(3):
    ivl, cnt = t("name_join call %d (%d records)" % (k + 1, sum(int(i.recs.shape[0]) for i in ins)), lambda: eng.name_join(ins, 0.9, count_flank=15))
print("intervals", int(cnt.item()))
single = []
for f in inp.files:
    d_s, d_o = eng.to_device(f.stream), eng.to_device(f.offsets)
    single.append(pipeline._filter_stream(eng, d_s, d_o, False, ref_sel, (30, 50, 0.1, 0.9)))
for k in range(2):
    ivl2, cnt2 = t("name_join, one part per file, call %d" % (k + 1), lambda: eng.name_join(single, 0.9, count_flank=15))
print("intervals", int(cnt2.item()))
