"""Experiment: where does K1's time go?  (a) real stream with SEQ/QUAL, (b) same records with l_seq = 0."""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from gci_amd import synth, _lib
from gci_amd.device import Engine
e = Engine(0)
L = 61_707_364
rs = synth.simulate_reads((("chr19", L),), 40, "hifi", seed=synth.seed_for(2, 0))
def run(rs, label):
    stream, offs = synth.to_bam_stream(rs)
    d_bam, d_off = e.to_device(stream), e.to_device(offs)
    sel = e.to_device(np.zeros(1, np.int32))
    out = torch.empty((len(rs), 32), dtype=torch.uint8, device=e.device)
    e.profile_enable(1)
    for _ in range(12):
        e.bam_filter(d_bam, d_off, sel, 30, 50, 0.1, 0.9, out=out, check=False)
    p = e.profile_read()
    print(label, "bytes", stream.shape[0], "k_bam_filter us", p["k_bam_filter"][0] / p["k_bam_filter"][1] * 1e3)
run(rs, "with seq/qual")
from dataclasses import replace
rs2 = replace(rs, l_seq=np.zeros_like(rs.l_seq))
run(rs2, "l_seq=0 compact")
# fewer records
rs3 = rs.take(np.arange(0, len(rs), 4))
run(rs3, "every 4th record, with seq/qual")
