"""CPU checks of the genome-scale test infrastructure: the oracle's heads-stream mode and its exact restriction of the
join to chosen contigs (used by bench.py and tests/test_gpu_genome.py at BASELINE configs[2] size), and the
determinism of the workload generator."""
import numpy as np

from gci_amd import synth, workloads

FILTER = (30, 50, 0.1, 0.9)


def test_oracle_heads_mode_equals_full_stream(oracle):
    for kind, seed in (("hifi", 5), ("ont", 3)):
        rs = synth.simulate_reads((("a", 300_000), ("b", 200_000)), 20, kind, seed=seed)
        full, o_full = synth.to_bam_stream(rs)
        heads, o_heads = synth.to_bam_stream(rs, heads=True)
        assert heads.shape[0] < full.shape[0] // 4
        want = oracle.bam_file_dict(full, o_full, ["a", "b"], ["a", "b"], *FILTER)
        got = oracle.bam_file_dict(heads, o_heads, ["a", "b"], ["a", "b"], *FILTER, heads=True)
        assert got == want and len(want[0]) > 50


def test_genome_workload_and_join_restricted_to_contigs(oracle):
    inp = workloads.genome_dual(0.01, procs=2)
    again = workloads.genome_dual(0.01, procs=1)
    for a, b in zip(inp.files, again.files):
        assert np.array_equal(a.stream, b.stream) and np.array_equal(a.offsets, b.offsets)
        assert a.aligned_bases == b.aligned_bases == int(a.aligned_per_contig.sum())
    names = inp.names
    assert len(names) == 25 and len(inp.files) == 2
    bams = [(f.stream, f.offsets, names) for f in inp.files]
    dicts, hq = [], set()
    for s, o, r in bams:
        d, h = oracle.bam_file_dict(s, o, r, names, *FILTER, heads=True)
        dicts.append(d)
        hq |= h
    full = oracle.name_join(dicts, hq, 0.9)
    # the second aligner's view drops / moves reads (deleted by the join) and re-draws MAPQ (high-quality names of
    # one file only are kept or re-added)
    assert len(full) > 1000 and any(q not in full for q in dicts[0]) and any(q not in dicts[0] for q in full)
    for chosen in (["chr1", "chr14", "chrM"], ["chr22"]):
        sub = oracle.file1_on_contigs(bams, names, chosen, *FILTER, 0.9, heads=True)
        assert sub == {q: s for q, s in full.items() if s[0] in chosen}
