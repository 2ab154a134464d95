"""K1 on ONT-like records (long CIGARs) and on records that all take the slow path."""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from gci_amd import synth
from gci_amd.device import Engine, REC_DTYPE
from oracle import gci_oracle as O
e = Engine(0)
def run(rs, label):
    stream, offs = synth.to_bam_stream(rs)
    d_bam, d_off = e.to_device(stream), e.to_device(offs)
    sel = e.to_device(np.zeros(1, np.int32))
    out = torch.empty((len(rs), 32), dtype=torch.uint8, device=e.device)
    for _ in range(3):
        e.bam_filter(d_bam, d_off, sel, 30, 50, 0.1, 0.9, out=out, check=False)
    e.profile_enable(1); e.profile_read()
    for _ in range(8):
        e.bam_filter(d_bam, d_off, sel, 30, 50, 0.1, 0.9, out=out, check=False)
    p = e.profile_read()
    want = O.bam_filter_arrays(stream, offs, np.zeros(1, np.int32), 30, 50, 0.1, 0.9)
    got = out.cpu().numpy().reshape(-1).view(REC_DTYPE)
    ok = np.array_equal(got["flags"] & 1, want["passed"]) and np.array_equal(got["end"][want["passed"] == 1], want["end"][want["passed"] == 1])
    n_ops = np.diff(rs.cigar_off)
    print("%-30s records %7d  ops mean %7.0f max %7d  >512: %6d  k_bam_filter(+slow) %.1f us  parity %s" % (
        label, len(rs), n_ops.mean(), n_ops.max(), (n_ops > 512).sum(), p["k_bam_filter"][0] / p["k_bam_filter"][1] * 1e3, ok))
L = 61_707_364
run(synth.simulate_reads((("chr19", L),), 40, "ont", seed=5), "ONT 40x chr19")
run(synth.simulate_reads((("chr19", L),), 40, "ont", seed=5, long_cigar_frac=0.05), "ONT 40x, 5% >65535 ops")
