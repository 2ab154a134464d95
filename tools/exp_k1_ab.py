#!/usr/bin/env python3
"""K1 alone on a genome-geometry heads stream (GCI_LIB_PATH selects the build): average of 10 launches between HIP events,
and a checksum of the records so that builds can be compared.  Usage: exp_k1_ab.py [scale]"""
import os, sys, time, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gci_amd import workloads
from gci_amd.device import Engine
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.25
inp = workloads.genome_dual(scale, 40.0, n_files=1)
f = inp.files[0]
eng = Engine(0)
eng.set_layout(inp.lengths)
d_s, d_o = eng.to_device(f.stream), eng.to_device(f.offsets)
sel = eng.to_device(np.arange(len(inp.contigs), dtype=np.int32))
recs = eng.bam_filter(d_s, d_o, sel, 30, 50, 0.1, 0.9, heads=True)
crc = zlib.crc32(recs.cpu().numpy().tobytes())
times = []
for _ in range(7):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(20):
        eng.bam_filter(d_s, d_o, sel, 30, 50, 0.1, 0.9, heads=True, check=False)
    b.record(); torch.cuda.synchronize()
    times.append(a.elapsed_time(b) * 50)
print("%s: %d records, us per call min %.1f median %.1f max %.1f, crc %08x" % (os.environ.get("GCI_LIB_PATH", "default"), f.offsets.shape[0],
                                                                             min(times), sorted(times)[3], max(times), crc))
