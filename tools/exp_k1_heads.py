#!/usr/bin/env python3
"""K1 over the whole inflated stream vs over the heads stream (chr19 / 40x HiFi and a 15x ONT set): HIP-event time per launch."""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gci_amd import synth, hostio, _lib
from gci_amd.device import Engine
from gci_amd.formats import bam as bamfmt
e = Engine(0)
for kind, cov in (("hifi", 40), ("ont", 15)):
    rs = synth.simulate_reads(synth.CHR19, cov, kind, seed=synth.seed_for(2, 0), **({"long_cigar_frac": 0.001} if kind == "ont" else {}))
    stream, offs = synth.to_bam_stream(rs)
    p = os.path.join(tempfile.mkdtemp(), "a.bam")
    bamfmt.write_bam_stream(p, stream, level=1, threads=hostio.default_threads())
    hd = hostio.bam_heads(np.fromfile(p, dtype=np.uint8))
    ref_sel = e.to_device(np.zeros(1, dtype=np.int32))
    res = {}
    for name, s, o, heads in (("full", stream, offs, False), ("heads", hd.stream, hd.offsets, True)):
        d_s, d_o = e.to_device(s), e.to_device(o)
        out = None
        e.profile_enable(1 << _lib.PROF_BAM_FILTER if hasattr(_lib, "PROF_BAM_FILTER") else 1); e.profile_read()
        for _ in range(12):
            out = e.bam_filter(d_s, d_o, ref_sel, 30, 50, 0.1, 0.9, heads=heads, out=out)
        pr = e.profile_read()
        res[name] = (round(list(pr.values())[0][0] / list(pr.values())[0][1] * 1e3, 1), int(s.shape[0]), out.clone())
        e.profile_enable(0)
    assert torch.equal(res["full"][2], res["heads"][2])
    print(kind, {k: v[:2] for k, v in res.items()}, flush=True)
    hd.close()
