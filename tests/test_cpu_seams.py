"""libgci_cpu.so (gci_amd/csrc/cpu/gci_cpu.cpp: include/gci_hip.h a second time, through g++, on host memory and host threads)
against the oracle, seam by seam -- the cases of tests/test_gpu_seams.py that concern the seam set, run here without a GPU.
Bit-exact: integer work and IEEE f64 comparisons."""
import numpy as np
import pytest

from gci_amd import cpu, synth
from gci_amd.formats import bam


@pytest.fixture(scope="module")
def eng():
    return cpu.CpuEngine(threads=4)


def name_hash(names):
    lib = cpu.load()
    return np.array([lib.gci_name_hash(n, len(n)) for n in names], dtype=np.uint64)


def test_the_library_exports_the_seam_set_of_the_header():
    """Every name bound here is declared in include/gci_hip.h (but for the one option call of its own) and exported by the library."""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, "include", "gci_hip.h")).read()
    declared = set(re.findall(r"\b(gci_[a-z0-9_]+)\s*\(", header))
    lib = cpu.load()
    for name, _, _ in cpu.EXPORTS:
        assert hasattr(lib, name)
        assert name in declared or name == "gci_cpu_option", name
    assert lib.gci_abi_version() == 1


def _filter_case(eng, oracle, rs, targets=None, mq=30, cut=50, cp=0.1, ip=0.9):
    stream, offs = synth.to_bam_stream(rs)
    refs = [n for n, _ in rs.contigs]
    targets = targets or refs
    tindex = {t: i for i, t in enumerate(targets)}
    ref_sel = np.array([tindex.get(r, -1) for r in refs], dtype=np.int32)
    want = oracle.bam_filter_arrays(stream, offs, ref_sel, mq, cut, cp, ip)
    got = eng.bam_filter(stream, offs, ref_sel, mq, cut, cp, ip)
    p = want["passed"].astype(bool)
    assert np.array_equal((got["flags"] & 1).astype(bool), p)
    assert np.array_equal((got["flags"] & 2) != 0, want["hq"].astype(bool))
    for f in ("contig", "start", "end", "qlen", "name_len"):
        assert np.array_equal(got[f][p], want[f][p]), f
    names = [bytes(stream[int(o):int(o) + int(n)]) for o, n in zip(want["name_off"][p], want["name_len"][p])]
    assert np.array_equal(got["name_hash"][p], name_hash(names))
    assert np.array_equal(got["rec_idx"], np.arange(len(rs)))
    # the same records without their SEQ / QUAL bytes (a heads stream): the same decisions
    from bam_util import heads_expected
    h_bytes, h_offs = heads_expected(stream, offs, bam.parse_header(stream).first_record)
    eng.heads(True)
    try:
        got_h = eng.bam_filter(np.frombuffer(h_bytes, dtype=np.uint8), h_offs, ref_sel, mq, cut, cp, ip)
    finally:
        eng.heads(False)
    assert np.array_equal(got_h, got)
    return int(p.sum())


@pytest.mark.parametrize("kind,seed,cov", [("hifi", 11, 8), ("ont", 13, 6)])
def test_bam_filter_matches_oracle(eng, oracle, kind, seed, cov):
    rs = synth.simulate_reads((("a", 600_000), ("b", 300_000), ("c", 40_000)), cov, kind, seed=seed, long_cigar_frac=0.01 if kind == "ont" else 0.0)
    assert _filter_case(eng, oracle, rs) > 50
    _filter_case(eng, oracle, rs, targets=["b"], mq=20, cut=40, cp=0.02, ip=0.995)          # --chrs and other thresholds


def test_bam_filter_randomised_records(eng, oracle):
    """Records built field by field (names of 1..254 bytes, 0..600 CIGAR operations with every code, NM anywhere among Z / H / B /
    scalar tags, SEQ '*', placed-unmapped, long-CIGAR placeholders with and without CG, contigs in and out of the selection): the
    same record and the same status as the oracle for the first offender, offenders removed one by one."""
    rng = np.random.default_rng(31)
    refs = [("c%d" % i, 3_000_000) for i in range(5)]
    recs = []
    for i in range(1200):
        n_ops = int(rng.choice([0, 1, 2, 3, 40, 70, 130, 600], p=[.02, .1, .1, .1, .2, .2, .18, .1]))
        ops = []
        for _ in range(n_ops):
            o = int(rng.choice([0, 7, 8, 1, 2, 3, 4, 5, 6], p=[.3, .3, .1, .1, .1, .02, .04, .02, .02]))
            ops.append((o, int(rng.integers(20, 300)) if o in (0, 7) else int(rng.integers(1, 4))))
        qlen = sum(l for o, l in ops if (bam.QUERY_CONSUMING >> o) & 1)
        l_seq = 0 if rng.random() < 0.05 else qlen
        name = bytes(rng.integers(33, 127, int(rng.choice([1, 5, 30, 40, 100, 185, 220, 254]))).astype(np.uint8)).decode()
        tags = []
        for _ in range(int(rng.integers(0, 7))):
            k = rng.integers(0, 6)
            tg = "X%s" % chr(int(rng.integers(97, 123)))
            if k == 0: tags.append((tg, "i", int(rng.integers(-5, 5))))
            elif k == 1: tags.append((tg, "Z", "s" * int(rng.integers(0, 200))))
            elif k == 2: tags.append((tg, "B:C", list(range(int(rng.integers(0, 60))))))
            elif k == 3: tags.append((tg, "A", "P"))
            elif k == 4: tags.append((tg, "f", 0.5))
            else: tags.append((tg, "B:I", [7] * int(rng.integers(0, 12))))
        nm_total = sum(l for o, l in ops if o in (1, 2, 8))
        if rng.random() < 0.97:
            nmv = max(0, nm_total + int(rng.integers(-2, 40)))
            typ = "C" if nmv < 256 and rng.random() < 0.7 else ("S" if nmv < 65536 and rng.random() < 0.5 else "i")
            tags.insert(int(rng.integers(0, len(tags) + 1)), ("NM", typ, nmv))
        flag = int(rng.choice([0, 16, 0x100, 0x800, 0x4, 0x1, 0x400]))
        mapq = int(rng.choice([0, 10, 29, 30, 49, 50, 60]))
        aux = bam.encode_aux(tags)
        if n_ops >= 2 and rng.random() < 0.05:
            real = ops
            rl = sum(l for o, l in real if (bam.REF_CONSUMING >> o) & 1)
            ops = [(4, l_seq), (3, max(rl, 1))]
            if rng.random() < 0.7:
                aux += bam.encode_aux([("CG", "B:I", [(l << 4) | o for o, l in real])])
        recs.append(bam.encode_record(int(rng.integers(-1, 5)), int(rng.integers(0, 2_000_000)), name, mapq, flag, ops, l_seq, aux))
    hdr = bam.encode_header([r for r, _ in refs], [l for _, l in refs])
    stream = np.frombuffer(hdr + b"".join(recs), dtype=np.uint8).copy()
    offs = bam.record_offsets(stream, bam.parse_header(stream).first_record)
    n_checked = 0
    for ref_sel, (cp, ip) in ((np.array([0, 1, 2, 3, 4], np.int32), (0.1, 0.9)), (np.array([-1, 0, -1, 1, -1], np.int32), (0.5, 0.5))):
        keep = np.ones(len(offs), dtype=bool)
        for _ in range(400):
            o_sub = offs[keep]
            try:
                want = oracle.bam_filter_arrays(stream, o_sub, ref_sel, 30, 50, cp, ip)
                break
            except oracle.OracleRecordError as e:
                with pytest.raises(cpu.CpuError) as g:
                    eng.bam_filter(stream, o_sub, ref_sel, 30, 50, cp, ip)
                assert (g.value.status, g.value.rec) == (e.status, e.rec)          # the FIRST failing record in file order
                keep[np.flatnonzero(keep)[e.rec]] = False
                n_checked += 1
        got = eng.bam_filter(stream, o_sub, ref_sel, 30, 50, cp, ip)
        p = want["passed"].astype(bool)
        assert np.array_equal((got["flags"] & 1).astype(bool), p)
        assert np.array_equal((got["flags"] & 2) != 0, want["hq"].astype(bool))
        for f in ("contig", "start", "end", "qlen", "name_len"):
            assert np.array_equal(got[f][p], want[f][p]), f
        assert p.sum() > 40
    assert n_checked > 3


def _join_from_sets(eng, sets, targets, ovlp=0.9):
    tindex = {t: i for i, t in enumerate(targets)}
    files = []
    for rs in sets:
        stream, offs = synth.to_bam_stream(rs)
        ref_sel = np.array([tindex.get(r, -1) for r, _ in rs.contigs], dtype=np.int32)
        files.append((eng.bam_filter(stream, offs, ref_sel, 30, 50, 0.1, 0.9), stream, offs, 36))
    return sorted(map(tuple, eng.name_join(files, ovlp)[["contig", "start", "end"]].tolist()))


@pytest.mark.parametrize("n_files", [1, 2, 3])
def test_name_join_matches_oracle(eng, oracle, n_files):
    contigs = (("a", 500_000), ("b", 250_000))
    targets = ["a", "b"]
    base = synth.simulate_reads(contigs, 15, "hifi", seed=21)
    sets = [base] + [synth.perturb(base, 100 + k) for k in range(1, n_files)]
    if n_files == 1:                                    # repeated names inside one file: the dict keeps the last record
        dup = base.take(np.arange(0, len(base), 7))
        dup.pos[:] = np.minimum(dup.pos + 1234, 200_000)
        dup.mapq[:] = 60
        sets = [synth.ReadSet.sorted(synth.concat(base, dup))]
    files, hq = [], set()
    for rs in sets:
        stream, offs = synth.to_bam_stream(rs)
        d, h = oracle.bam_file_dict(stream, offs, [n for n, _ in rs.contigs], targets, 30, 50, 0.1, 0.9)
        files.append(d)
        hq |= h
    want = sorted(({"a": 0, "b": 1}[v[0]], v[1], v[2]) for v in oracle.name_join(files, hq, 0.9).values())
    assert _join_from_sets(eng, sets, targets) == want and len(want) > 50


def _dict_input(d, hq, tindex):
    """A per-file dict name -> (target, start, end, qlen) as compact records + a names blob (what the PAF path hands the join)."""
    names = [q.encode() for q in d]
    recs = np.zeros(len(d), dtype=cpu.REC_DTYPE)
    recs["name_hash"] = name_hash(names)
    vals = list(d.values())
    recs["contig"] = [tindex[v[0]] for v in vals]
    recs["start"] = [v[1] for v in vals]
    recs["end"] = [v[2] for v in vals]
    recs["qlen"] = [v[3] for v in vals]
    recs["rec_idx"] = np.arange(len(d))
    recs["flags"] = [1 | (2 if q in hq else 0) for q in d]
    recs["name_len"] = [len(n) for n in names]
    off = np.concatenate([[0], np.cumsum([len(n) for n in names])]).astype(np.uint64)
    return recs, np.frombuffer(b"".join(names) or b"\0", dtype=np.uint8), off[:-1], 0


@pytest.mark.parametrize("n_files,seed", [(1, 0), (2, 1), (3, 2), (5, 4), (16, 6)])
def test_name_join_randomised_dicts(eng, oracle, n_files, seed):
    """The fold of GCI.py:279-299 on random per-file dicts from a small name pool: deletion, interval intersection, the ovlp / qlen of
    the CURRENT file's record, resurrection by a later file with three files and more."""
    rng = np.random.default_rng(100 + seed)
    pool = [("read%05d" % i) for i in range(2000)]
    targets = ["t0", "t1", "t2"]
    files = []
    for f in range(n_files):
        d = {}
        for q in rng.choice(pool, size=int(rng.integers(700, 1700)), replace=False):
            s = int(rng.integers(0, 50_000))
            ln = int(rng.integers(50, 20_000))
            d[str(q)] = (targets[int(rng.integers(0, 3)) if rng.random() < 0.15 else 0], s, s + ln, int(max(1, ln + rng.integers(-40, 400))))
        files.append(d)
    for f in range(1, n_files):
        for q in list(files[f])[::2]:
            if q in files[0]:
                t, s, e, ql = files[0][q]
                j = int(rng.integers(-30, 30))
                files[f][q] = (t, max(0, s + j), e + j, max(1, e - s + int(rng.integers(-5, 60))))
    hq = set(str(q) for q in rng.choice(pool, size=600, replace=False)) & set().union(*[set(d) for d in files])
    want = sorted((targets.index(v[0]), v[1], v[2]) for v in oracle.name_join(files, hq, 0.9).values())
    tindex = {t: i for i, t in enumerate(targets)}
    inputs = [_dict_input(d, hq, tindex) for d in files]
    got = sorted(map(tuple, eng.name_join(inputs, 0.9)[["contig", "start", "end"]].tolist()))
    assert got == want and len(want) > 100
    got2 = sorted(map(tuple, eng.name_join(inputs, 0.9, contig_map=np.array([0, -1, -1], dtype=np.int32))[["contig", "start", "end"]].tolist()))
    assert got2 == [w for w in want if w[0] == 0]
    # a record with qlen 0 that reaches the overlap test: ZeroDivisionError (GCI.py:292)
    if n_files >= 2:
        q = next(q for q in files[1] if q in files[0] and files[1][q][0] == files[0][q][0] and (q in hq or all(q in d for d in files)))
        bad = [dict(d) for d in files]
        bad[1][q] = bad[1][q][:3] + (0,)
        with pytest.raises(cpu.CpuError) as g:
            eng.name_join([_dict_input(d, hq, tindex) for d in bad], 0.9)
        assert g.value.status == -4


def test_depth_build_slice_semantics(eng, oracle):
    rng = np.random.default_rng(5)
    lengths = {"x": 10_000, "y": 4096, "z": 4097, "w": 50, "v": 123_457}
    targets = list(lengths)
    eng.set_layout([lengths[t] for t in targets])
    ivls = []
    for c, t in enumerate(targets):
        L = lengths[t]
        for _ in range(400):
            s = int(rng.integers(0, L))
            ivls.append((c, s, int(min(L + 40, s + rng.integers(1, max(2, L // 3))))))
    # NumPy's negative-stop wrap (e <= fl - 2), empty slices, reads hanging over the end
    ivls += [(3, 0, 10), (3, 0, 13), (3, 0, 14), (3, 20, 25), (0, 9_990, 10_050), (1, 0, 4096), (2, 4090, 4097)]
    arr = np.array([(c, s, e, 0) for c, s, e in ivls], dtype=np.int32).view(cpu.IVL_DTYPE).reshape(-1)
    for fl in (15, 0, 3):
        want = oracle.depth_build_py([(targets[c], s, e) for c, s, e in ivls], lengths, fl)
        track = eng.depth_build(arr, fl)
        for c, t in enumerate(targets):
            assert np.array_equal(eng.contig(track, c), want[t]), (t, fl)
        assert int(track.sum()) == sum(int(v.sum()) for v in want.values())              # padding between contigs stays zero
        assert np.array_equal(eng.depth_sum(track), np.array([want[t].sum() for t in targets]))


def _random_depth(rng, L):
    d = rng.poisson(3.0, L).astype(np.int64)
    for _ in range(max(3, L // 5000)):
        a = int(rng.integers(0, L))
        d[a:a + int(rng.integers(1, 400))] = 0
    d[:int(rng.integers(0, 40))] = 0
    d[L - int(rng.integers(1, 40)):] = 0
    return d


def _track_of(eng, depths):
    targets = list(depths)
    eng.set_layout([int(depths[t].shape[0]) for t in targets])
    track = eng.new_track()
    for c, t in enumerate(targets):
        eng.contig(track, c)[:] = depths[t]
    return track


def _collapse(eng, track, names, lengths, threshold, flank, start_pos):
    """collapse_depth_range (GCI.py:356-390) from the run boundaries: the drop rule `i > flank_len` and the coordinate shifts are the host's."""
    runs = eng.issue_runs(track, -1, threshold, flank)
    out = {}
    for c, t in enumerate(names):
        L, segs = lengths[c], []
        for a, b in runs[c]:                                  # relative to the window [flank, L - flank)
            last = b == L - 2 * flank
            if last or b > flank:                             # (a run that ends inside the first flank_len scanned bases is dropped)
                segs.append((a + flank + start_pos, b + flank + start_pos))
        out[t] = segs
    return out


@pytest.mark.parametrize("flank,threshold", [(15, 0), (0, 0), (2, 1), (40, 2)])
def test_issue_scan_matches_oracle(eng, oracle, flank, threshold):
    rng = np.random.default_rng(7 + flank)
    depths = {"a": _random_depth(rng, 200_001), "b": _random_depth(rng, 4096), "c": _random_depth(rng, 8193),
              "tiny": np.zeros(20, dtype=np.int64), "one": np.zeros(2 * flank + 1, dtype=np.int64),
              "full": np.full(5000, 9, dtype=np.int64), "zero": np.zeros(12_345, dtype=np.int64)}
    track = _track_of(eng, depths)
    got = _collapse(eng, track, list(depths), [int(v.shape[0]) for v in depths.values()], threshold, flank, 0)
    want = oracle.collapse_depth_range(depths, -1, threshold, flank, 0)
    assert got == want and sum(len(v) for v in want.values()) > 10


def test_issue_scan_kats_and_windows(eng, oracle):
    """SURVEY.md R10 known answers (produced by the reference's collapse_depth_range), and windows of a track (the -R regions)."""
    cases = [([0] * 10, 2, 0, [(2, 8)]), ([0, 0, 0, 0, 5, 5, 5, 5, 0, 0], 2, 0, []), ([0, 0, 0, 0, 0, 5, 5, 5, 0, 0], 2, 0, [(2, 5)]),
             ([0, 0, 5, 5, 5, 5, 5, 0, 0, 0], 2, 0, [(7, 8)]), ([0, 3, 0, 0, 3, 0], 0, 100, [(100, 101), (102, 104), (105, 106)])]
    for d, fl, sp, want in cases:
        track = _track_of(eng, {"k": np.array(d, dtype=np.int64)})
        assert _collapse(eng, track, ["k"], [len(d)], 0, fl, sp)["k"] == want
        assert oracle.collapse_contig(np.array(d), -1, 0, fl, sp) == want
    rng = np.random.default_rng(3)
    depths = {"a": _random_depth(rng, 50_000), "b": _random_depth(rng, 9_000)}
    track = _track_of(eng, depths)
    regions = [("a", 0, 50_000), ("a", 100, 20_000), ("b", 4000, 9000), ("a", 4095, 4097), ("b", 10, 10)]
    wins = [(int(eng.offsets[list(depths).index(t)]) + s, int(eng.offsets[list(depths).index(t)]) + e) for t, s, e in regions]
    runs = eng.issue_runs(track, -1, 0, 0, windows=wins)
    for (t, s, e), r in zip(regions, runs):
        # fl = 0: every run counts but one that ends at relative 0 (impossible) -- the window's own coordinates + its start
        assert [(a + s, b + s) for a, b in r] == oracle.collapse_contig(depths[t][s:e], -1, 0, 0, s), (t, s, e)
    sums = eng.range_sums(track, np.array(wins, dtype=np.int64))
    assert [int(x) for x in sums] == [int(depths[t][s:e].sum()) for t, s, e in regions]


def test_gap_mask_max2_text_sum(eng, oracle):
    rng = np.random.default_rng(9)
    h = {"a": rng.poisson(30, 70_001).astype(np.int64), "b": rng.integers(0, 120_000, 5000).astype(np.int64),
         "c": np.array([0, 9, 10, 99, 100, 999, 1000, 9999, 10000, 2**31 - 1], dtype=np.int64)}
    n = {k: rng.poisson(25, v.shape[0]).astype(np.int64) for k, v in h.items()}
    th = _track_of(eng, h)
    tn = _track_of(eng, n)
    two = eng.max2(th, tn)
    want2 = oracle.max2(h, n)
    names = list(h)
    for c, k in enumerate(names):
        assert np.array_equal(eng.contig(two, c), want2[k])
    gaps = {"a": [(10, 500), (69_990, 80_000), (-20, -5)], "b": [(0, 1)], "nope": [(1, 2)]}
    eng.gap_mask(two, [(names.index(k), a, b) for k, v in gaps.items() if k in names for a, b in v])
    oracle.merge_gaps_depths(want2, gaps)
    for c, k in enumerate(names):
        assert np.array_equal(eng.contig(two, c), want2[k])
    text, offs = eng.depth_text(two)
    assert text.tobytes() == b"".join(oracle.depth_text_contig(want2[k]) for k in names)
    assert [int(x) for x in offs] == list(np.cumsum([0] + [len(oracle.depth_text_contig(want2[k])) for k in names]))
    sums = eng.depth_sum(two)
    assert float(sums.sum()) / sum(v.shape[0] for v in want2.values()) == oracle.mean_depth(want2)


def test_results_do_not_depend_on_the_threads():
    a, b = cpu.CpuEngine(threads=1), cpu.CpuEngine(threads=7)
    rs = synth.simulate_reads((("a", 300_000), ("b", 90_000)), 10, "hifi", seed=3)
    stream, offs = synth.to_bam_stream(rs)
    sel = np.array([0, 1], dtype=np.int32)
    ra, rb = a.bam_filter(stream, offs, sel, 30, 50, 0.1, 0.9), b.bam_filter(stream, offs, sel, 30, 50, 0.1, 0.9)
    assert np.array_equal(ra, rb)
    ia, ib = a.name_join([(ra, stream, offs, 36)], 0.9), b.name_join([(rb, stream, offs, 36)], 0.9)
    assert sorted(map(tuple, ia.tolist())) == sorted(map(tuple, ib.tolist()))
    for e, iv in ((a, ia), (b, ib)):
        e.set_layout([300_000, 90_000])
    ta, tb = a.depth_build(ia, 15), b.depth_build(ib, 15)
    assert np.array_equal(ta, tb) and a.depth_text(ta)[0].tobytes() == b.depth_text(tb)[0].tobytes()
    assert np.array_equal(a.issue_keys(ta, -1, 0, 15), b.issue_keys(tb, -1, 0, 15))


def test_depth_text_as_gzip_members(eng):
    """gci_depth_deflate_size / _write behind the same header on the host (N2): the members decompress -- CRC and length checked by
    Python's gzip -- to exactly f'{depth}\\n' per base: runs of every length and line width, contigs that end inside a tile / a member,
    depths up to 2^31 - 1; the same cases as tests/test_gpu_seams.py::test_depth_text_as_gzip_members_from_the_gpu."""
    import gzip
    rng = np.random.default_rng(77)
    lens = [1, 3, 4095, 4096, 4097, 64 * 4096 - 1, 64 * 4096, 64 * 4096 + 5, 300_001, 1_000_003]
    eng.set_layout(lens)
    host = np.zeros(eng.total, dtype=np.int32)
    want = []
    for c, (off, L) in enumerate(zip(eng.offsets.tolist(), lens)):
        if c % 3 == 0:        # long runs of small depths, like a real track
            edges = np.sort(rng.choice(np.arange(1, max(L, 2)), size=min(L - 1, max(1, L // 900)), replace=False)) if L > 1 else np.zeros(0, int)
            vals = rng.integers(0, 130, edges.shape[0] + 1)
            d = np.repeat(vals, np.diff(np.concatenate(([0], edges, [L]))))
        elif c % 3 == 1:      # every run length from 1 up, every line width
            widths = rng.choice([0, 7, 42, 999, 1000, 65_536, 9_999_999, 123_456_789, 2_147_483_647], size=L)
            rl = rng.integers(1, 6, size=L)
            d = np.repeat(widths, rl)[:L]
        else:                 # no two neighbours equal
            d = (np.arange(L) * 7919 + c) % 1013
        host[off:off + L] = d
        want.append(("\n".join(map(str, d.tolist())) + "\n").encode())
    members = eng.depth_deflate(host)
    assert len(members) == len(lens)
    for c, blob in enumerate(members):
        assert blob[:4] == b"\x1f\x8b\x08\x00"
        assert gzip.decompress(blob) == want[c], c
        assert blob.count(b"\x1f\x8b\x08\x00\x00\x00\x00\x00\x00\xff") >= -(-lens[c] // (64 * 4096))
    assert len(members[9]) * 40 < len(want[9])
    text, off_t = eng.depth_text(host)
    whole = text.tobytes()
    for c in range(len(lens)):
        assert gzip.decompress(members[c]) == whole[int(off_t[c]):int(off_t[c + 1])]
    # two lines of two bytes (the second goes out as literals), three of them, and a run of 4096
    eng.set_layout([2, 3, 4096])
    t = np.concatenate([np.full(2, 7), np.full(3, 7), np.full(4096, 5)]).astype(np.int32)
    tr = np.zeros(eng.total, dtype=np.int32)
    for c, (off, L) in enumerate(zip(eng.offsets.tolist(), [2, 3, 4096])):
        tr[off:off + L] = [7, 7, 5][c]
    got = eng.depth_deflate(tr)
    assert [gzip.decompress(b) for b in got] == [b"7\n" * 2, b"7\n" * 3, b"5\n" * 4096]
