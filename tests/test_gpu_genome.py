"""BASELINE configs[2] at full size through the whole device path, against the oracle:
CHM13 geometry (25 contigs, 3.117 Gb), two 40x HiFi alignment files of the same reads (the second perturbed the way
another aligner's output differs) as heads streams -> K1 x 2 -> `-op` join -> fused depth build (track, sums, issue
runs, text), compared on chr1 (the longest contig), chr14 (starts past 2^31 elements of the track) and chrM (a
contig shorter than a tile) with the oracle's exact restriction of the join to those contigs
(oracle.file1_on_contigs), plus the genome-wide sum of depth against the clipped join intervals.

GCI_TEST_GENOME_SCALE (default 1.0) shrinks every contig for a quick run."""
import os

import numpy as np
import pytest

from gci_amd import pipeline, workloads
from gci_amd.device import JoinInput

pytestmark = pytest.mark.gpu

FILTER = (30, 50, 0.1, 0.9)
OVLP, FLANK = 0.9, 15


@pytest.fixture(scope="module")
def genome():
    return workloads.genome_dual(float(os.environ.get("GCI_TEST_GENOME_SCALE", "1.0")))


@pytest.mark.parametrize("join_mode", ["auto"])
def test_genome_dual_bam_full_path_matches_oracle(engine, oracle, genome, join_mode):
    import torch
    inp = genome
    names, lens = inp.names, inp.lengths
    engine.set_layout(lens)
    assert engine.offsets[names.index("chr14")] + lens[names.index("chr14")] > 2 ** 31 or sum(lens) < 2 ** 31
    ref_sel = engine.to_device(np.arange(len(names), dtype=np.int32))
    ins = []
    for f in inp.files:
        # what the command line and bench.py run: record pages made on the device from the heads stream, the paged filter
        d_s, d_o = engine.to_device(f.stream), engine.to_device(f.offsets)
        pages = engine.bam_pages(d_s, d_o, False)
        del d_s, d_o
        recs, noff = engine.bam_filter_pages(pages, ref_sel, *FILTER)
        ins.append(JoinInput(recs, pages.buf, noff, 0))
    ivl, cnt = engine.name_join(ins, OVLP, count_flank=FLANK)
    track = engine.new_track()
    fused = engine.depth_build_fused(ivl, cnt, FLANK, track, want_text=True, want_sums=True, issue=(-1.0, 0.0, FLANK),
                                     counted=True)
    K = int(cnt.item())
    # genome-wide: the sum of depth is the sum of the clipped interval lengths
    iv = ivl[:K].cpu().numpy().astype(np.int64)
    L = np.asarray(lens, dtype=np.int64)[iv[:, 0]]
    a, b = np.clip(iv[:, 1] + FLANK, 0, L), np.clip(iv[:, 2] - FLANK + 1, 0, L)
    assert int(np.maximum(b - a, 0).sum()) == int(np.asarray(fused["sums"]).sum())

    chosen = ["chr1", "chr14", "chrM"]
    file1 = oracle.file1_on_contigs([(f.stream, f.offsets, names) for f in inp.files], names, chosen, *FILTER, OVLP, heads=True)
    # the join itself: the same interval multiset on the chosen contigs
    for c in chosen:
        ci = names.index(c)
        got = iv[iv[:, 0] == ci][:, 1:3]
        want = np.asarray(sorted((s[1], s[2]) for s in file1.values() if s[0] == c), dtype=np.int64).reshape(-1, 2)
        got = got[np.lexsort((got[:, 1], got[:, 0]))]
        assert np.array_equal(got, want), c
    tl = {c: lens[names.index(c)] for c in chosen}
    toff = fused["text_off"]
    for c in chosen:                                     # one contig at a time: chr1 is 2 GB as int64
        ci = names.index(c)
        want = oracle.depth_build({q: s for q, s in file1.items() if s[0] == c}, {c: tl[c]}, FLANK)[c]
        o = engine.offsets[ci]
        got = track[o:o + tl[c]].cpu().numpy()
        assert np.array_equal(got, want), c
        assert int(fused["sums"][ci]) == int(want.sum()), c
        bed = oracle.collapse_depth_range({c: want}, -1, 0, FLANK, 0)[c]
        lo, hi = pipeline._slice_bound(FLANK, tl[c]), pipeline._slice_bound(tl[c] - FLANK, tl[c])
        assert pipeline._issues_from_runs(fused["runs"][ci], max(0, hi - lo), tl[c], FLANK, 0) == bed, c
        text = fused["text"][int(toff[ci]):int(toff[ci + 1])].cpu().numpy().tobytes()
        assert text == oracle.depth_text_contig(want), c
        del want, got, text
    del fused, track
    torch.cuda.empty_cache()


@pytest.mark.parametrize("config", [4, 5])
def test_two_read_types_bam_and_paf_match_oracle(engine, oracle, config):
    """BASELINE configs[3] (CHM13, --hifi + --nano, per read type one BAM + one PAF) and configs[4] (diploid mat + pat, 46
    contigs, HiFi 100x + ONT, 20 N gaps, -R regions) through the device path of one GPU -- bench.py's TwoTypeWorkload: K2 on
    the device, the paged record filter on HiFi and on ONT (CIGARs in the blob, chunk queue), PAF-then-BAM joins, two builds,
    gap masks, per-base max, three issue scans, region scans -- against the oracle's exact restriction to whole contigs
    (oracle.file1_on_contigs_mixed), tracks, issue lists and region lists.

    configs[3] runs at FULL size (25 contigs of CHM13, 3.1 Gb, HiFi + ONT with their PAFs: ~150 s, most of it the host-side
    generation); configs[4] at 0.1 of every contig (GCI_TEST_TWO_TYPE_SCALE overrides both; at 1.0 it takes ~170 s and passes:
    profiles/r06f_two_type_full_size_tests.txt; `python bench.py --workload diploid` runs it at full size)."""
    import bench
    scale = float(os.environ.get("GCI_TEST_TWO_TYPE_SCALE", "1.0" if config == 4 else "0.1"))
    inp = workloads.genome_two_type(config, scale, 40.0 if config == 4 else 100.0, 40.0 if config == 4 else 20.0)
    assert len(inp.contigs) == (25 if config == 4 else 46) and (inp.hifi.paf is not None) == (config == 4)
    w = bench.TwoTypeWorkload(engine, inp, "test")
    w.step()
    assert w.check()
    names = inp.names
    chosen = ["chr14", "chr22", "chrM"] if config == 4 else ["mat_chr14", "pat_chr21", "pat_chr22"]
    chosen += [c for c in (list(inp.gaps)[:1] + [r[0] for r in inp.regions[:1]]) if c not in chosen]
    ok, used = bench.parity_two_type(w, chosen)
    assert ok and len(used) >= 3
    # every long ONT CIGAR went through the blob and the chunk queue: ONT records pass the filter at all
    assert int(w.types[1]["count"].item()) > 0.3 * inp.nano.n_reads
    if config == 5:                                      # configs[4] names "-p windowed depth": the numeric front end on all contigs
        n3 = bench.plot_front_end_number(w)
        assert n3["parity_vs_oracle"] and n3["values"] > 46 and n3["contigs"] == 46
    del w
    import torch
    torch.cuda.empty_cache()
