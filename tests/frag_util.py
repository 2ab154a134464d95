"""A fragmented assembly as command-line input (tests/test_gpu_e2e.py, tests/test_gpu_dist.py): 2 500 scaffolds of 2 - 40 kb (a
header of thousands of references, thousands of one-tile contigs, scaffolds shorter than a read), both read types with a
BAM and a PAF each, query names of up to 230 characters (the record filter's slow path and the PAF tokeniser), gaps at
scaffold ends."""
import os

import numpy as np

from gci_amd import synth
from gci_amd.formats import paf as paffmt


def write_inputs(inp: str):
    """-> (contigs, argv tail after the program name and before -d)."""
    rng = np.random.default_rng(77)
    lens = rng.integers(2_000, 40_000, 2_500)
    contigs = tuple(("scf%05d_len%d" % (i, int(l)), int(l)) for i, l in enumerate(lens))
    gaps = {contigs[i][0]: [(0, 50)] for i in range(0, 2_500, 97)}
    gaps[contigs[3][0]] = [(contigs[3][1] - 10, contigs[3][1])]

    def long_names(rs, seed):
        r = np.random.default_rng(seed)
        names = rs.names.astype(object)
        for i in r.choice(len(rs), size=len(rs) // 20, replace=False):
            names[i] = names[i] + b"/" + b"x" * int(r.integers(150, 200))
        rs.names = names.astype("S")
        return rs

    h = long_names(synth.simulate_reads(contigs, 20, "hifi", seed=1201), 1)
    h2 = synth.perturb(h, 1202)
    n = long_names(synth.simulate_reads(contigs, 15, "ont", seed=1203, long_cigar_frac=0.0), 2)
    n2 = synth.perturb(n, 1204)
    os.makedirs(inp, exist_ok=True)
    ref = os.path.join(inp, "ref.fa")
    synth.write_reference_fasta(ref, contigs, gaps)
    files = {"h.bam": h, "h.paf": synth.to_paf_lines(h2, 5, 0.05), "n.bam": n, "n.paf": synth.to_paf_lines(n2, 6, 0.05)}
    for name, obj in files.items():
        if name.endswith(".bam"):
            synth.write_bam_file(os.path.join(inp, name), obj, level=1, threads=4)
        else:
            paffmt.write(os.path.join(inp, name), obj)
    p = lambda f: os.path.join(inp, f)      # noqa: E731
    return contigs, ["-r", ref, "--hifi", p("h.bam"), p("h.paf"), "--nano", p("n.paf"), p("n.bam"), "-t", "4", "-ts", "1"]
