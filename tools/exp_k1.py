"""Experiment: K1 time vs layout (real stream with SEQ/QUAL vs compact l_seq = 0) and vs record count."""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from dataclasses import replace
from gci_amd import synth
from gci_amd.device import Engine
e = Engine(0)
L = 61_707_364
rs = synth.simulate_reads((("chr19", L),), 40, "hifi", seed=synth.seed_for(2, 0))
def run(rs, label):
    stream, offs = synth.to_bam_stream(rs)
    d_bam, d_off = e.to_device(stream), e.to_device(offs)
    sel = e.to_device(np.zeros(1, np.int32))
    out = torch.empty((len(rs), 32), dtype=torch.uint8, device=e.device)
    for _ in range(4):
        e.bam_filter(d_bam, d_off, sel, 30, 50, 0.1, 0.9, out=out, check=False)
    e.profile_enable(1); e.profile_read()
    for _ in range(10):
        e.bam_filter(d_bam, d_off, sel, 30, 50, 0.1, 0.9, out=out, check=False)
    p = e.profile_read()
    print("%-34s records %7d bytes %11d  k_bam_filter %.1f us" % (label, len(rs), stream.shape[0], p["k_bam_filter"][0] / p["k_bam_filter"][1] * 1e3))
run(rs, "with seq/qual")
run(replace(rs, l_seq=np.zeros_like(rs.l_seq)), "l_seq=0 compact")
run(rs.take(np.arange(0, len(rs), 2)), "every 2nd record")
run(rs.take(np.arange(0, len(rs), 4)), "every 4th record")
run(rs.take(np.arange(0, len(rs), 16)), "every 16th record")
