import sys, numpy as np
sys.path.insert(0, '.')
from gci_amd import synth
from gci_amd.device import Engine, REC_DTYPE
from oracle import gci_oracle as O
rs = synth.simulate_reads((("a", 1_500_000), ("b", 700_000), ("c", 40_000)), 15, "ont", seed=13, long_cigar_frac=0.01)
stream, offs = synth.to_bam_stream(rs)
ref_sel = np.arange(3, dtype=np.int32)
want = O.bam_filter_arrays(stream, offs, ref_sel, 30, 50, 0.1, 0.9)
e = Engine(0)
got = e.bam_filter(e.to_device(stream), e.to_device(offs), e.to_device(ref_sel), 30, 50, 0.1, 0.9).cpu().numpy().reshape(-1).view(REC_DTYPE)
p = want["passed"].astype(bool); g = (got["flags"] & 1).astype(bool)
bad = np.flatnonzero(p != g)
n_ops = np.diff(rs.cigar_off)
print("n", len(rs), "bad", bad.shape[0])
print("bad n_ops", np.sort(n_ops[bad])[:10], "min; good pass n_ops max", n_ops[p & g].max() if (p&g).any() else None)
print("bad nm", np.sort(rs.nm[bad])[:10], "good nm max", rs.nm[p & g].max())
print("bad l_seq min", rs.l_seq[bad].min(), "good l_seq max", rs.l_seq[p&g].max())
print("name len", np.char.str_len(rs.names)[bad][:5])
print("bad nm_last frac", rs.nm_last[bad].mean(), "all", rs.nm_last.mean(), "good", rs.nm_last[p&g].mean())
tot = rs.op_totals()
print("bad S", tot[bad][:8, 4], "flag", rs.flag[bad][:8], "mapq", rs.mapq[bad][:8])
for (cp, ip) in [(1.0, -1.0), (0.1, -1.0), (1.0, 0.9)]:
    want = O.bam_filter_arrays(stream, offs, ref_sel, 30, 50, cp, ip)
    got = e.bam_filter(e.to_device(stream), e.to_device(offs), e.to_device(ref_sel), 30, 50, cp, ip).cpu().numpy().reshape(-1).view(REC_DTYPE)
    p2 = want["passed"].astype(bool); g2 = (got["flags"] & 1).astype(bool)
    print(cp, ip, "bad", (p2 != g2).sum(), "gpu pass", g2.sum(), "want", p2.sum())
    b2 = np.flatnonzero(p2 & g2)
    print("   end mismatch", (got["end"][b2] != want["end"][b2]).sum(), "qlen mismatch", (got["qlen"][b2] != want["qlen"][b2]).sum())
