"""GPU parity, seam by seam: each C-ABI export against the oracle's restatement of the matching
reference function on the same seeded inputs (bit-exact: everything here is integer work or an
IEEE f64 comparison)."""
import numpy as np
import pytest
import torch

from bam_util import heads_expected
from gci_amd import synth
from gci_amd.device import JoinInput, REC_DTYPE, name_hash_np
from gci_amd import pipeline
from gci_amd._lib import GciError as GciErr

pytestmark = pytest.mark.gpu


def _recs_np(t):
    return t.cpu().numpy().reshape(-1).view(REC_DTYPE)


def _filter_case(engine, oracle, rs, targets=None, mq=30, cut=50, cp=0.1, ip=0.9):
    stream, offs = synth.to_bam_stream(rs)
    refs = [n for n, _ in rs.contigs]
    targets = targets or refs
    tindex = {t: i for i, t in enumerate(targets)}
    ref_sel = np.array([tindex.get(r, -1) for r in refs], dtype=np.int32)
    want = oracle.bam_filter_arrays(stream, offs, ref_sel, mq, cut, cp, ip)
    d_bam, d_off = engine.to_device(stream), engine.to_device(offs)
    full = engine.bam_filter(d_bam, d_off, engine.to_device(ref_sel), mq, cut, cp, ip).clone()
    got = _recs_np(full)
    # the same records without their SEQ / QUAL bytes (what the command line uploads): the same 32 bytes per record
    from gci_amd.formats import bam
    h_bytes, h_offs = heads_expected(stream, offs, bam.parse_header(stream).first_record)
    # ... as record pages (gci_bam_pages_*, gci_bam_filter_pages), made from either stream
    _via_pages(engine, d_bam, d_off, True, ref_sel, mq, cut, cp, ip, full, stream)
    h_np = np.frombuffer(h_bytes, dtype=np.uint8)
    _via_pages(engine, engine.to_device(h_np), engine.to_device(h_offs), False, ref_sel, mq, cut, cp, ip, full, h_np, page_bytes=8192)
    p = want["passed"].astype(bool)
    assert np.array_equal((got["flags"] & 1).astype(bool), p)
    assert np.array_equal(((got["flags"] & 2) != 0), want["hq"].astype(bool))
    for f in ("contig", "start", "end", "qlen"):
        assert np.array_equal(got[f][p], want[f][p]), f
    assert np.array_equal(got["name_len"][p], want["name_len"][p])
    names = [bytes(stream[int(o):int(o) + int(n)]) for o, n in zip(want["name_off"][p], want["name_len"][p])]
    assert np.array_equal(got["name_hash"][p], name_hash_np(names))
    assert np.array_equal(got["rec_idx"], np.arange(len(rs)))
    return stream, offs, d_bam, d_off, got, p.sum()


def _via_pages(engine, d_stream, d_off, has_seq, ref_sel, mq, cut, cp, ip, full, host_stream, page_bytes=0):
    """The paged record filter over the pages made of this stream: the same 32 bytes per record, and name offsets that
    point at the names inside the pages buffer."""
    pages = engine.bam_pages(d_stream, d_off, has_seq, page_bytes)
    recs, noff = engine.bam_filter_pages(pages, engine.to_device(ref_sel), mq, cut, cp, ip)
    same = recs.clone()
    same[:, 29] &= 3                                   # (GCI_REC_NAME16: where the name lies, not what the record is)
    assert torch.equal(same, full)
    got = _recs_np(recs)
    n16, npass = int(((got["flags"] & 4) != 0).sum()), int(((got["flags"] & 1) != 0).sum())
    assert not np.any((got["flags"] & 5) == 4) and n16 >= 0.8 * npass      # passing records whose name lies in a page carry it
    buf = pages.buf.cpu().numpy()
    no = noff.cpu().numpy()
    offs = d_off.cpu().numpy().view(np.uint64)
    for i in np.flatnonzero(got["flags"] & 1)[::37]:
        n = int(got["name_len"][i])
        assert bytes(buf[int(no[i]):int(no[i]) + n]) == bytes(host_stream[int(offs[i]) + 36:int(offs[i]) + 36 + n])
    return pages


@pytest.mark.parametrize("kind,seed,cov", [("hifi", 11, 20), ("hifi", 12, 20), ("ont", 13, 15)])
def test_bam_filter_matches_oracle(engine, oracle, kind, seed, cov):
    rs = synth.simulate_reads((("a", 1_500_000), ("b", 700_000), ("c", 40_000)), cov, kind, seed=seed,
                              long_cigar_frac=0.01 if kind == "ont" else 0.0)
    *_, n_pass = _filter_case(engine, oracle, rs)
    assert n_pass > 100
    # --chrs restriction and non-default thresholds
    _filter_case(engine, oracle, rs, targets=["b"], mq=20, cut=40, cp=0.02, ip=0.995)


def _dicts_from(oracle, rs, targets, **kw):
    stream, offs = synth.to_bam_stream(rs)
    refs = [n for n, _ in rs.contigs]
    return oracle.bam_file_dict(stream, offs, refs, targets, 30, 50, 0.1, 0.9), (stream, offs)


def _join_gpu(engine, sets, targets, ovlp=0.9):
    inputs = []
    tindex = {t: i for i, t in enumerate(targets)}
    for rs in sets:
        stream, offs = synth.to_bam_stream(rs)
        refs = [n for n, _ in rs.contigs]
        ref_sel = engine.to_device(np.array([tindex.get(r, -1) for r in refs], dtype=np.int32))
        d_bam, d_off = engine.to_device(stream), engine.to_device(offs)
        recs = engine.bam_filter(d_bam, d_off, ref_sel, 30, 50, 0.1, 0.9)
        inputs.append(JoinInput(recs, d_bam, d_off, 36))
    ivl, cnt = engine.name_join(inputs, ovlp)
    n = int(cnt.item())
    return sorted(map(tuple, ivl[:n, :3].cpu().numpy().tolist()))


@pytest.fixture(params=["classic", "partition"])
def join_mode(request, engine):
    """Both implementations of the join behind gci_name_join: the global open-addressing table and the radix-partitioned
    one with per-bucket tables in LDS (chosen by size in production; gci_join_mode forces one)."""
    engine.set_join_mode(request.param)
    yield request.param
    engine.set_join_mode("auto")


@pytest.fixture(params=["atomic", "radix"])
def events_mode(request, monkeypatch):
    """Both ways a depth build buckets its events by tile: one device-scope atomic per event, and (chosen by size in
    production; GCI_EVENTS forces one) the radix partition by tile range with the counting in LDS."""
    monkeypatch.setenv("GCI_EVENTS", request.param)
    return request.param


@pytest.mark.parametrize("n_files", [1, 2, 3])
def test_name_join_matches_oracle(engine, oracle, n_files, join_mode):
    contigs = (("a", 900_000), ("b", 500_000))
    targets = ["a", "b"]
    base = synth.simulate_reads(contigs, 25, "hifi", seed=21)
    sets = [base] + [synth.perturb(base, 100 + k) for k in range(1, n_files)]
    if n_files == 1:
        # repeated names inside one file: the dict keeps the last record
        dup = base.take(np.arange(0, len(base), 7))
        dup.pos[:] = np.minimum(dup.pos + 1234, 400_000)
        dup.mapq[:] = 60
        sets = [synth.ReadSet.sorted(_concat(base, dup))]
    files, hq = [], set()
    for rs in sets:
        (d, h), _ = _dicts_from(oracle, rs, targets)
        files.append(d)
        hq |= h
    want = oracle.name_join(files, hq, 0.9)
    tindex = {"a": 0, "b": 1}
    want = sorted((tindex[v[0]], v[1], v[2]) for v in want.values())
    got = _join_gpu(engine, sets, targets)
    assert got == want
    assert len(want) > 100


_concat = synth.concat


def test_depth_build_slice_semantics(engine, oracle, events_mode):
    rng = np.random.default_rng(5)
    lengths = {"x": 10_000, "y": 4096, "z": 4097, "w": 50, "v": 123_457}
    targets = list(lengths)
    engine.set_layout([lengths[t] for t in targets])
    ivls = []
    for c, t in enumerate(targets):
        L = lengths[t]
        for _ in range(400):
            s = int(rng.integers(0, L))
            e = int(min(L + 40, s + rng.integers(1, max(2, L // 3))))
            ivls.append((c, s, e))
    # NumPy negative-stop wrap (e <= fl - 2), empty slices, reads hanging over the end
    ivls += [(3, 0, 10), (3, 0, 13), (3, 0, 14), (3, 20, 25), (0, 9_990, 10_050), (1, 0, 4096), (2, 4090, 4097)]
    for fl in (15, 0, 3):
        want = oracle.depth_build_py([(targets[c], s, e) for c, s, e in ivls], lengths, fl)
        d = engine.to_device(np.array([(c, s, e, 0) for c, s, e in ivls], dtype=np.int32))
        track = engine.new_track()
        engine.depth_build(d, None, fl, track)
        tr = pipeline.DepthTracks(engine, lengths, track)
        for t in targets:
            assert np.array_equal(tr[t], want[t]), (t, fl)
        # padding between contigs stays zero
        assert int(track.sum().item()) == sum(int(v.sum()) for v in want.values())
        assert np.array_equal(tr.sums(), np.array([want[t].sum() for t in targets]))


def _random_depth(rng, L):
    d = rng.poisson(3.0, L).astype(np.int64)
    for _ in range(max(3, L // 5000)):
        a = int(rng.integers(0, L))
        d[a:a + int(rng.integers(1, 400))] = 0
    d[:int(rng.integers(0, 40))] = 0
    d[L - int(rng.integers(1, 40)):] = 0
    return d


def _upload_depths(engine, depths):
    targets = list(depths)
    lengths = {t: int(depths[t].shape[0]) for t in targets}
    offs = engine.set_layout([lengths[t] for t in targets])
    flat = np.zeros(max(engine.total, 1), dtype=np.int32)
    for o, t in zip(offs, targets):
        flat[o:o + lengths[t]] = depths[t]
    return pipeline.DepthTracks(engine, lengths, engine.to_device(flat))


@pytest.mark.parametrize("flank,threshold", [(15, 0), (0, 0), (2, 1), (40, 2)])
def test_issue_scan_matches_oracle(engine, oracle, flank, threshold):
    rng = np.random.default_rng(7 + flank)
    depths = {"a": _random_depth(rng, 200_001), "b": _random_depth(rng, 4096), "c": _random_depth(rng, 8193),
              "tiny": np.zeros(20, dtype=np.int64), "one": np.zeros(2 * flank + 1, dtype=np.int64),
              "full": np.full(5000, 9, dtype=np.int64), "zero": np.zeros(12_345, dtype=np.int64)}
    tr = _upload_depths(engine, depths)
    got = pipeline.collapse_depth_range(tr, -1, threshold, flank, 0)
    want = oracle.collapse_depth_range(depths, -1, threshold, flank, 0)
    assert got == want
    assert sum(len(v) for v in want.values()) > 10


def test_issue_scan_kats(engine, oracle):
    """SURVEY.md R10 known answers, produced by the reference's collapse_depth_range."""
    cases = [([0] * 10, 2, 0, [(2, 8)]), ([0, 0, 0, 0, 5, 5, 5, 5, 0, 0], 2, 0, []),
             ([0, 0, 0, 0, 0, 5, 5, 5, 0, 0], 2, 0, [(2, 5)]), ([0, 0, 5, 5, 5, 5, 5, 0, 0, 0], 2, 0, [(7, 8)]),
             ([0, 3, 0, 0, 3, 0], 0, 100, [(100, 101), (102, 104), (105, 106)])]
    for d, fl, sp, want in cases:
        tr = _upload_depths(engine, {"k": np.array(d, dtype=np.int64)})
        assert pipeline.collapse_depth_range(tr, -1, 0, fl, sp)["k"] == want
        assert oracle.collapse_contig(np.array(d), -1, 0, fl, sp) == want


def test_regions_scan(engine, oracle):
    rng = np.random.default_rng(3)
    depths = {"a": _random_depth(rng, 50_000), "b": _random_depth(rng, 9_000)}
    tr = _upload_depths(engine, depths)
    regions = [("a", 0, 50_000), ("a", 100, 20_000), ("b", 4000, 9500), ("a", 4095, 4097), ("b", 10, 10),
               ("a", 30_000, 20_000), ("a", -500, 50_000)]
    got = pipeline.collapse_regions(tr, regions, -1, 0)
    for (t, s, e), g in zip(regions, got):
        assert g == oracle.collapse_contig(depths[t][s:e], -1, 0, 0, s), (t, s, e)


def test_gap_mask_max2_text(engine, oracle):
    rng = np.random.default_rng(9)
    h = {"a": rng.poisson(30, 70_001).astype(np.int64), "b": rng.integers(0, 120_000, 5000).astype(np.int64),
         "c": np.array([0, 9, 10, 99, 100, 999, 1000, 9999, 10000, 2**31 - 1], dtype=np.int64)}
    n = {k: rng.poisson(25, v.shape[0]).astype(np.int64) for k, v in h.items()}
    th, tn = _upload_depths(engine, h), _upload_depths(engine, n)
    two = pipeline.merge_two_type_depth(th, tn, write=False)
    want2 = oracle.max2(h, n)
    for k in h:
        assert np.array_equal(two[k], want2[k])
    gaps = {"a": [(10, 500), (69_990, 80_000), (-20, -5)], "b": [(0, 1)], "nope": [(1, 2)]}
    pipeline.merge_gaps_depths(two, gaps)
    oracle.merge_gaps_depths(want2, gaps)
    for k in h:
        assert np.array_equal(two[k], want2[k])
    text, offs = engine.depth_text(two.track)
    host = text.cpu().numpy().tobytes()
    want_text = b"".join(oracle.depth_text_contig(want2[k]) for k in h)
    assert host == want_text
    assert [int(x) for x in offs] == list(np.cumsum([0] + [len(oracle.depth_text_contig(want2[k])) for k in h]))
    assert two.mean() == oracle.mean_depth(want2)


def test_fused_build_equals_seams_and_oracle(engine, oracle, events_mode):
    """gci_depth_build_begin/finish: depth, text, sums and issue runs from the fused passes equal the
    separate seams (and therefore the oracle) -- including contig ends on / off tile boundaries."""
    rng = np.random.default_rng(17)
    lengths = {"x": 50_000, "y": 4096, "z": 8193, "w": 31, "v": 300_001}
    targets = list(lengths)
    engine.set_layout([lengths[t] for t in targets])
    ivls = []
    for c, t in enumerate(targets):
        L = lengths[t]
        for _ in range(300 if L > 1000 else 5):
            s = int(rng.integers(0, L))
            ivls.append((c, s, int(min(L + 30, s + rng.integers(1, max(2, L // 4))))))
    ivls += [(1, 0, 4096), (2, 8000, 8193), (3, 0, 10), (0, 49_000, 50_000)]
    d = engine.to_device(np.array([(c, s, e, 0) for c, s, e in ivls], dtype=np.int32))
    for fl, thr in ((15, 0), (0, 2), (7, 1)):
        want = oracle.depth_build_py([(targets[c], s, e) for c, s, e in ivls], lengths, fl)
        track = engine.new_track()
        out = engine.depth_build_fused(d, None, fl, track, want_text=True, want_sums=True, issue=(-1, thr, fl))
        tr = pipeline.DepthTracks(engine, lengths, track)
        for t in targets:
            assert np.array_equal(tr[t], want[t]), (t, fl)
        assert np.array_equal(out["sums"], np.array([want[t].sum() for t in targets]))
        assert out["text"].cpu().numpy().tobytes() == b"".join(oracle.depth_text_contig(want[t]) for t in targets)
        tr._fresh_runs = ((-1.0, float(thr), fl), out["runs"])
        fused_bed = pipeline.collapse_depth_range(tr, -1, thr, fl, 0)
        tr.invalidate()
        assert fused_bed == pipeline.collapse_depth_range(tr, -1, thr, fl, 0) == oracle.collapse_depth_range(want, -1, thr, fl, 0)
        # a count held on the device limits how many intervals are used
        cnt = torch.tensor([len(ivls) // 2], dtype=torch.int32, device=engine.device)
        t2 = engine.new_track()
        engine.depth_build(d, cnt, fl, t2)
        half = oracle.depth_build_py([(targets[c], s, e) for c, s, e in ivls[:len(ivls) // 2]], lengths, fl)
        tr2 = pipeline.DepthTracks(engine, lengths, t2)
        for t in targets:
            assert np.array_equal(tr2[t], half[t])


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_fused_byproducts_sparse_and_dense_tiles(engine, oracle, seed, events_mode):
    """Pass 1 derives sums / text sizes / issue-run boundaries from the event list of a tile when the tile holds few
    events and from the dense difference array otherwise: low coverage with many short runs, runs crossing tile and
    window edges, pile-ups (> 63 events in a tile) next to empty tiles, contigs ending on and off tile boundaries."""
    rng = np.random.default_rng(100 + seed)
    lengths = {"a": 70_000, "b": 8192, "c": 4097, "d": 12_288, "e": 40, "f": 33_333, "g": 1, "h": 3, "i": 16, "j": 17, "k": 5}
    targets = list(lengths)
    engine.set_layout([lengths[t] for t in targets])
    ivls = []
    for c, t in enumerate(targets):
        L = lengths[t]
        for _ in range(max(2, L // 900)):                       # ~1-2x of short reads: depth 0/1/2/3 everywhere
            s = int(rng.integers(0, L))
            ivls.append((c, s, int(min(L + 5, s + rng.integers(1, 2500)))))
        for _ in range(L // 20_000):                            # pile-ups: > 63 events inside one tile
            s0 = int(rng.integers(0, max(1, L - 3000)))
            for _ in range(60):
                s = s0 + int(rng.integers(0, 2000))
                ivls.append((c, s, s + int(rng.integers(1, 800))))
    ivls += [(1, 0, 4096), (1, 4096, 8192), (3, 4095, 4097), (3, 8191, 8193), (0, 0, 1), (0, 69_999, 70_000), (4, 0, 40)]
    ivls += [(6, 0, 1), (7, 1, 3), (8, 0, 16), (8, 3, 9), (9, 0, 17), (9, 16, 17), (10, 2, 4)]     # tiny contigs: every edge case at once
    if seed == 2:       # three-digit depths (4-byte text pattern) over tiles with few events, next to 1- and 2-digit ones
        ivls += [(0, 5000, 60_000)] * 99 + [(0, 20_000, 40_000)] * 2 + [(5, 100, 33_000)] * 300 + [(5, 4090, 4100)] * 5
    if seed == 3:       # depth >= 1000 over tiles that hold no event of their own: four-digit text from the dense path
        ivls += [(5, 3000, 30_000)] * 1100 + [(5, 10_000, 10_010)] * 9000
    d = engine.to_device(np.array([(c, s, e, 0) for c, s, e in ivls], dtype=np.int32))
    for fl, lo, hi in ((0, -1, 0), (15, -1, 1), (3, 0, 2), (5000, -1, 0), (1, 1, 3)):
        want = oracle.depth_build_py([(targets[c], s, e) for c, s, e in ivls], lengths, fl)
        track = engine.new_track()
        out = engine.depth_build_fused(d, None, fl, track, want_text=True, want_sums=True, issue=(lo, hi, fl))
        tr = pipeline.DepthTracks(engine, lengths, track)
        for t in targets:
            assert np.array_equal(tr[t], want[t]), (t, fl)
        assert np.array_equal(out["sums"], np.array([want[t].sum() for t in targets]))
        assert out["text"].cpu().numpy().tobytes() == b"".join(oracle.depth_text_contig(want[t]) for t in targets)
        raw = engine.issue_scan(track, lo, hi, fl)               # stand-alone scan of the finished track
        for c in range(len(targets)):
            assert np.array_equal(np.asarray(out["runs"][c]), np.asarray(raw[c])), (targets[c], fl, lo, hi)
        tr._fresh_runs = ((float(lo), float(hi), fl), out["runs"])
        fused_bed = pipeline.collapse_depth_range(tr, lo, hi, fl, 0)
        tr.invalidate()
        scanned = pipeline.collapse_depth_range(tr, lo, hi, fl, 0)
        assert fused_bed == scanned == oracle.collapse_depth_range(want, lo, hi, fl, 0), (fl, lo, hi)
        if fl < 5000:
            assert sum(len(v) for v in scanned.values()) > 5


def test_dense_path_equals_event_list_path(engine, monkeypatch):
    """GCI_FORCE_DENSE=1 (read when a context is created) sends every tile through the dense difference-array kernels;
    the event-list kernels must produce the same track, text, sums and issue runs byte for byte."""
    from gci_amd.device import Engine
    rng = np.random.default_rng(77)
    lengths = [150_000, 4096, 9_000, 77]
    ivls = []
    for c, L in enumerate(lengths):
        for _ in range(max(3, L // 400)):
            s0 = int(rng.integers(0, L))
            ivls.append((c, s0, int(min(L + 3, s0 + rng.integers(1, 6000))), 0))
    ivls += [(0, 1000, 140_000, 0)] * 120                      # three-digit depths
    arr = np.array(ivls, dtype=np.int32)
    monkeypatch.setenv("GCI_FORCE_DENSE", "1")
    dense = Engine(0)
    monkeypatch.delenv("GCI_FORCE_DENSE")
    outs = []
    for eng in (engine, dense):
        eng.set_layout(lengths)
        track = eng.new_track()
        out = eng.depth_build_fused(eng.to_device(arr), None, 7, track, want_text=True, want_sums=True, issue=(-1, 2, 7))
        outs.append((track.cpu().numpy(), out["text"].cpu().numpy().tobytes(), out["text_off"], out["sums"],
                     [np.asarray(r) for r in out["runs"]]))
    # a text buffer that does not start 16-byte aligned (C callers): same bytes
    import ctypes
    from gci_amd._lib import BuildOpts
    for eng in (engine, dense):
        toff = torch.zeros(len(lengths) + 1, dtype=torch.int64, device=eng.device)
        o = BuildOpts(); o.flank = 7; o.want_text = 1; o.d_contig_text_off = toff.data_ptr()
        d_iv = eng.to_device(arr)
        eng._chk(eng.lib.gci_depth_build_begin(eng.ctx, ctypes.c_void_p(d_iv.data_ptr()), None, len(ivls), ctypes.byref(o)), "begin")
        total = int(toff[-1].item())
        for shift in (3, 8, 13):
            buf = torch.full((total + 64,), 0x55, dtype=torch.uint8, device=eng.device)
            trk = eng.new_track()
            eng._chk(eng.lib.gci_depth_build_begin(eng.ctx, ctypes.c_void_p(d_iv.data_ptr()), None, len(ivls), ctypes.byref(o)), "begin")
            eng._chk(eng.lib.gci_depth_build_finish(eng.ctx, ctypes.c_void_p(trk.data_ptr()), ctypes.c_void_p(buf.data_ptr() + shift), total), "finish")
            got = buf.cpu().numpy()
            assert got[shift:shift + total].tobytes() == outs[0][1], shift
            assert (got[:shift] == 0x55).all() and (got[shift + total:] == 0x55).all(), shift      # nothing outside [0, total)
    a, b = outs
    assert np.array_equal(a[0], b[0]) and a[1] == b[1]
    assert np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3])
    assert all(np.array_equal(x, y) for x, y in zip(a[4], b[4]))
    assert len(a[1]) > 400_000


def test_build_begin_twice_then_finish(engine, events_mode):
    """A build that is begun again before it was finished (the wrapper does that when the issue-key buffer was too
    small) must not see the first attempt's per-tile counters: the table is re-zeroed."""
    import ctypes
    from gci_amd._lib import BuildOpts
    rng = np.random.default_rng(5)
    lengths = [90_000, 5000]
    engine.set_layout(lengths)
    ivls = [(int(c), int(s0), int(s0 + l), 0) for c, s0, l in zip(rng.integers(0, 2, 700), rng.integers(0, 4000, 700), rng.integers(1, 9000, 700))]
    d_iv = engine.to_device(np.array(ivls, dtype=np.int32))
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    want = engine.new_track()
    engine.depth_build(d_iv, None, 3, want)
    o = BuildOpts(); o.flank = 3
    got = engine.new_track()
    for _ in range(3):
        engine._chk(engine.lib.gci_depth_build_begin(engine.ctx, p(d_iv), None, len(ivls), ctypes.byref(o)), "begin")
    engine._chk(engine.lib.gci_depth_build_finish(engine.ctx, p(got), None, 0), "finish")
    assert torch.equal(got, want) and int(want.sum().item()) > 0
    again = engine.new_track()
    engine.depth_build(d_iv, None, 3, again)                 # and the table is clean for the next build
    assert torch.equal(again, want)


def test_plot_front_end_matches_reference_kats_and_oracle(engine, oracle, capsys):
    """N3 (`-p` numeric front-end): pipeline.sliding_window_average_depth / pre_plot_base on tracks in HBM equal the
    reference's outputs (tests/golden/kats.json, made by the unmodified functions) and the oracle on larger random
    tracks -- positions and values bit for bit (float64)."""
    import json, os
    from golden_util import GOLDEN
    k = json.load(open(os.path.join(GOLDEN, "kats.json")))
    for c in k["sliding_window_average_depth"]:
        if not c["depth"]:
            continue
        tr = _upload_depths(engine, {"t": np.array(c["depth"], dtype=np.int64)})
        pos, val = pipeline.sliding_window_average_depth(tr, "t", c["ws"], c["max_depth"], 0)
        # the reference gets the slice and `start` separately: here start = 0 slices nothing, positions shift by c["start"]
        want_pos, want_val = oracle.sliding_window_average_depth(c["depth"], c["ws"], c["max_depth"], 0)
        assert pos == want_pos and val.tolist() == want_val.tolist() == c["val"]
    for c in k["pre_plot_base"]:
        trs = [_upload_depths(engine, {t: np.array(v, dtype=np.int64) for t, v in d.items()}) for d in c["depths"]]
        # every _upload_depths sets the layout anew: bind each track right before it is used
        av, y_frac, y_min, y_max = pipeline.pre_plot_base(trs, c["max_depths"], c["ws"], 0)
        assert (y_frac, y_min, y_max) == (c["y_frac"], c["y_min"], c["y_max"])
        for a, want in zip(av, c["series"]):
            for t, (p, v) in a.items():
                assert p == want[t][0] and v.tolist() == want[t][1]
    # larger tracks, region slices (start / end), windows longer than a tile, a region shorter than the window
    rng = np.random.default_rng(41)
    d = rng.poisson(6.0, 300_000).astype(np.int64)
    for _ in range(40):
        a0 = int(rng.integers(0, 300_000))
        d[a0:a0 + int(rng.integers(1, 3000))] = 0
    d[:17] = 0
    d[-5:] = 0
    depths = {"x": d, "y": rng.poisson(2.0, 5000).astype(np.int64)}
    tr = _upload_depths(engine, depths)
    for target, ws, md, s0, e0 in (("x", 50_000, 9.5, 0, None), ("x", 1000, 6.2, 0, None), ("x", 777, 100.0, 12_345, 250_001),
                                   ("y", 50_000, 3.0, 0, None), ("y", 64, 1.9, 100, 4000), ("x", 4096, 5.0, 4096, 8192)):
        pos, val = pipeline.sliding_window_average_depth(tr, target, ws, md, s0, e0)
        sl = depths[target][s0:e0]
        wp, wv = oracle.sliding_window_average_depth(sl, ws, md, s0)
        assert pos == wp and np.array_equal(val, wv), (target, ws)
        assert len(pos) > 3
    capsys.readouterr()


def test_fasta_n_runs_on_gpu(engine, oracle, tmp_path):
    """N4 (first half): get_Ns_ref's scan on the GPU equals the host parser and the oracle's regex on awkward files:
    CRLF, blank lines, lower-case n, runs at record starts / ends and across line ends and tile boundaries, records
    that end and begin with N, '>' inside a title, empty records, no final newline, a repeated id."""
    from gci_amd.formats import fasta
    rng = np.random.default_rng(3)

    def seq(n, p_n=0.02):
        s = rng.choice(np.frombuffer(b"ACGTacgt", dtype=np.uint8), n)
        for _ in range(max(1, int(n * p_n / 50))):
            a = int(rng.integers(0, n))
            s[a:a + int(rng.integers(1, 200))] = rng.choice(np.frombuffer(b"Nn", dtype=np.uint8))
        return s.tobytes()

    recs = [("chrA desc > with gt", b"NNN" + seq(20_000) + b"nn"), ("chrB", b"N" + seq(9_000) + b"N"), ("empty", b""),
            ("chrC", seq(4096 * 3, 0.2)), ("onlyN", b"N" * 5000), ("chrA", seq(300)), ("tail", seq(777))]
    cases = {}
    for name, width, eol in (("lf60", 60, b"\n"), ("crlf70", 70, b"\r\n"), ("w4096", 4096, b"\n"), ("w1", 1, b"\n")):
        out = bytearray()
        for rid, sq in recs:
            out += b">" + rid.encode() + eol
            for i in range(0, len(sq), width):
                out += sq[i:i + width] + eol
                if rng.random() < 0.01:
                    out += eol                                  # a blank line inside the record
        cases[name] = bytes(out)
    cases["no_final_newline"] = cases["lf60"].rstrip(b"\n")
    cases["blanks"] = cases["lf60"].replace(b"AC", b"A C", 50)
    cases["one_line"] = b">x\nNNACGTNNNN"
    cases["no_records"] = b"just text\n"
    cases["empty_file"] = b""
    for name, data in cases.items():
        p = str(tmp_path / (name + ".fa"))
        open(p, "wb").write(data)
        ids_h, runs_h = fasta.n_runs(p)
        ids_d, runs_d = fasta.n_runs_device(engine, p)
        assert ids_d == ids_h and runs_d == runs_h and list(runs_d) == list(runs_h), name
        if name in ("lf60", "crlf70"):                         # and the reference's regex on the assembled strings
            want = {}
            for rid, sq in recs:
                r = oracle.n_runs_of(sq.decode())
                if r:
                    want.setdefault(rid.split()[0], []).extend(r)
            assert runs_d == want, name
    assert sum(len(v) for v in fasta.n_runs_device(engine, str(tmp_path / "lf60.fa"))[1].values()) > 20


def test_counting_join_equals_join_then_count(engine, oracle, join_mode, events_mode):
    """gci_name_join_count + gci_depth_build_begin(counted = 1) produce the same track, text, sums and issue runs as
    gci_name_join + a plain build; a counted build over other intervals or another flank is refused; a plain build after
    an unused counting join starts from a clean table."""
    from gci_amd._lib import GciError
    contigs = (("cA", 200_000), ("cB", 50_000))
    a = synth.simulate_reads(contigs, 25, "hifi", seed=101)
    b = synth.perturb(a, 102)
    lengths = [l for _, l in contigs]
    engine.set_layout(lengths)
    files = []
    for rs in (a, b):
        stream, offs = synth.to_bam_stream(rs)
        d_bam, d_off = engine.to_device(stream), engine.to_device(offs)
        recs = engine.bam_filter(d_bam, d_off, engine.to_device(np.arange(2, dtype=np.int32)), 30, 50, 0.1, 0.9)
        files.append(JoinInput(recs, d_bam, d_off, 36))
    results = []
    for fused in (False, True):
        ivl, cnt = engine.name_join(files, 0.9, count_flank=15 if fused else None)
        track = engine.new_track()
        out = engine.depth_build_fused(ivl, cnt, 15, track, want_text=True, want_sums=True, issue=(-1, 0, 15), counted=fused)
        results.append((track.cpu().numpy(), out["text"].cpu().numpy().tobytes(), out["sums"].tolist(),
                        [np.asarray(r).tolist() for r in out["runs"]], int(cnt.item())))
    assert results[0][4] == results[1][4] > 100
    assert np.array_equal(results[0][0], results[1][0]) and results[0][1:4] == results[1][1:4]
    # counted build with another flank: refused
    ivl, cnt = engine.name_join(files, 0.9, count_flank=15)
    with pytest.raises(GciError):
        engine.depth_build_fused(ivl, cnt, 3, engine.new_track(), want_text=False, counted=True)
    # ... and the table is clean again for a plain build
    t2 = engine.new_track()
    engine.depth_build(ivl, cnt, 15, t2)
    assert np.array_equal(t2.cpu().numpy(), results[0][0])
    # counted build without a counting join before it: refused
    ivl, cnt = engine.name_join(files, 0.9)
    with pytest.raises(GciError):
        engine.depth_build_fused(ivl, cnt, 15, engine.new_track(), want_text=False, counted=True)


def test_cross_rank_name_check_kernels(engine):
    """gci_hash_bucket + gci_hash_conflicts with two simulated ranks on one GPU: unique names -> 0 conflicts;
    a name present on both ranks is found; a repeated name inside ONE rank is not a conflict; overflow counts."""
    rng = np.random.default_rng(23)

    def recs_for(names, passed=None):
        r = np.zeros(len(names), dtype=REC_DTYPE)
        r["name_hash"] = name_hash_np(names)
        r["flags"] = 1 if passed is None else passed
        return engine.to_device(r.view(np.uint8).reshape(len(names), 32))

    a = [b"readA/%d" % i for i in range(5000)]
    b = [b"readB/%d" % i for i in range(4000)]

    pairs = {}                                        # alternate=True: per (rank, cap) two arrays used in turn

    def conflicts(na, nb, cap=6000, pa=None, pb=None, alternate=False):
        world = 2
        outs = []
        for k, (names, p) in enumerate(((na, pa), (nb, pb))):
            if alternate:
                # the ping-pong contract of gci_hash_bucket: zero-filled once, afterwards each call clears the count
                # words of the array the next call fills (and garbage beyond the counts must not matter)
                if (k, cap) not in pairs:
                    pairs[(k, cap)] = [torch.zeros(world * (cap + 1), dtype=torch.int64, device=engine.device) for _ in range(2)]
                    for t in pairs[(k, cap)]:
                        t.view(world, cap + 1)[:, 1:] = 0x5A5A5A5A
                o, nxt = pairs[(k, cap)]
                engine.hash_bucket(recs_for(names, p), world, cap, o, nxt)
                pairs[(k, cap)] = [nxt, o]
            else:
                o = torch.full((world * (cap + 1),), 0x77, dtype=torch.int64, device=engine.device)
                engine.hash_bucket(recs_for(names, p), world, cap, o)
            outs.append(o.view(world, cap + 1))
        n = torch.zeros(1, dtype=torch.int32, device=engine.device)
        for me in range(world):                       # what rank `me` receives: bucket `me` of every source
            recv = torch.stack([outs[src][me] for src in range(world)]).reshape(-1).contiguous()
            engine.hash_conflicts(recv, world, cap, n)
        return int(n.item())

    assert conflicts(a, b) == 0
    assert conflicts(a, b + [a[17]]) == 1
    assert conflicts(a + [a[3], a[3]], b) == 0                       # duplicates inside one rank: the local join's business
    assert conflicts(a, b + [a[5], a[900], a[4999]]) == 3
    flags = np.ones(len(b) + 1, dtype=np.uint8); flags[-1] = 0
    assert conflicts(a, b + [a[17]], pb=flags) == 0                   # filtered records do not take part
    assert conflicts(a, b, cap=100) > 0                              # bucket overflow forces the fallback
    # call after call (the conflict tables alternate and wipe each other; the bucket arrays alternate too)
    for rnd in range(5):
        extra = [a[int(i)] for i in rng.choice(len(a), size=rnd, replace=False)]
        assert conflicts(a, b + extra, alternate=True) == rnd
        assert conflicts(a[:100 + rnd], b[:50], cap=3000, alternate=True) == 0        # another table size in between
    big = [b"x%d" % i for i in range(70_000)]                           # several workgroups per bucket kernel
    assert conflicts(big, [b"y%d" % i for i in range(65_000)] + big[:7], cap=80_000, alternate=True) == 7


@pytest.mark.parametrize("seed", [31, 32])
def test_bam_filter_randomised_records(engine, oracle, seed):
    """Differential test of K1 (staged fast path + slow path) against the oracle on records built field by
    field: names of 1..254 bytes, 0..6000 CIGAR ops with every op code, NM at any place among Z / H / B / scalar
    tags (inside and beyond the staged window), SEQ '*', placed-unmapped, long-CIGAR placeholders with and without
    a CG tag, contigs in and out of the selection."""
    from gci_amd.formats import bam
    rng = np.random.default_rng(seed)
    refs = [("c%d" % i, 3_000_000) for i in range(5)]
    recs = []
    for i in range(3000):
        n_ops = int(rng.choice([0, 1, 2, 3, 40, 70, 130, 600, 6000], p=[.02, .1, .1, .1, .2, .2, .15, .1, .03]))
        ops = []
        for _ in range(n_ops):
            o = int(rng.choice([0, 7, 8, 1, 2, 3, 4, 5, 6], p=[.3, .3, .1, .1, .1, .02, .04, .02, .02]))
            ops.append((o, int(rng.integers(20, 300)) if o in (0, 7) else int(rng.integers(1, 4))))
        qlen = sum(l for o, l in ops if (bam.QUERY_CONSUMING >> o) & 1)
        l_seq = 0 if rng.random() < 0.05 else qlen
        name = bytes(rng.integers(33, 127, int(rng.choice([1, 5, 30, 40, 100, 183, 184, 185, 186, 219, 220, 221, 254]))).astype(np.uint8)).decode()
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
        ref = int(rng.integers(-1, 5))
        aux = bam.encode_aux(tags)
        if n_ops >= 2 and rng.random() < 0.05:            # placeholder-looking CIGAR, with or without a CG tag
            real = ops
            rl = sum(l for o, l in real if (bam.REF_CONSUMING >> o) & 1)
            ops = [(4, l_seq), (3, max(rl, 1))]
            if rng.random() < 0.7:
                aux += bam.encode_aux([("CG", "B:I", [(l << 4) | o for o, l in real])])
        recs.append(bam.encode_record(ref, int(rng.integers(0, 2_000_000)), name, mapq, flag, ops, l_seq, aux))
    hdr = bam.encode_header([r for r, _ in refs], [l for _, l in refs])
    stream = np.frombuffer(hdr + b"".join(recs), dtype=np.uint8).copy()
    offs = bam.record_offsets(stream, bam.parse_header(stream).first_record)
    d_bam, d_off = engine.to_device(stream), engine.to_device(offs)
    h_bytes, h_offs = heads_expected(stream, offs, bam.parse_header(stream).first_record)
    d_heads = engine.to_device(np.frombuffer(h_bytes, dtype=np.uint8))
    n_checked = 0
    for ref_sel, (cp, ip) in ((np.array([0, 1, 2, 3, 4], np.int32), (0.1, 0.9)), (np.array([-1, 0, -1, 1, -1], np.int32), (0.5, 0.5))):
        # the reference raises on the first bad record; remove offenders one by one until the oracle is clean, checking
        # that the GPU reports the same status for the same record each time
        keep = np.ones(len(offs), dtype=bool)
        for _ in range(400):
            o_sub = offs[keep]
            try:
                want = oracle.bam_filter_arrays(stream, o_sub, ref_sel, 30, 50, cp, ip)
                break
            except oracle.OracleRecordError as e:
                with pytest.raises(GciErr) as g:
                    engine.bam_filter(d_bam, engine.to_device(o_sub), engine.to_device(ref_sel), 30, 50, cp, ip)
                assert g.value.status == e.status, (e.status, g.value.status)
                # the GPU reports the FIRST failing record in file order, like the oracle's loop
                assert g.value.rec == e.rec
                with pytest.raises(GciErr) as gh:                       # the same through the pages of the heads stream
                    engine.bam_filter_pages(engine.bam_pages(d_heads, engine.to_device(h_offs[keep]), False), engine.to_device(ref_sel),
                                            30, 50, cp, ip)
                assert (gh.value.status, gh.value.rec) == (e.status, e.rec)
                with pytest.raises(GciErr) as gp:                       # ... and through those of the whole stream
                    engine.bam_filter_pages(engine.bam_pages(d_bam, engine.to_device(o_sub), True), engine.to_device(ref_sel), 30, 50, cp, ip)
                assert (gp.value.status, gp.value.rec) == (e.status, e.rec)
                keep[np.flatnonzero(keep)[e.rec]] = False
                n_checked += 1
        full = engine.bam_filter(d_bam, engine.to_device(o_sub), engine.to_device(ref_sel), 30, 50, cp, ip).clone()
        got = _recs_np(full)
        h_np = np.frombuffer(h_bytes, dtype=np.uint8)
        for pb in (8192, 16384, 32768):                                   # record pages of every size, from either stream
            _via_pages(engine, d_bam, engine.to_device(o_sub), True, ref_sel, 30, 50, cp, ip, full, stream, page_bytes=pb)
        _via_pages(engine, d_heads, engine.to_device(h_offs[keep]), False, ref_sel, 30, 50, cp, ip, full, h_np)
        p = want["passed"].astype(bool)
        assert np.array_equal((got["flags"] & 1).astype(bool), p)
        assert np.array_equal((got["flags"] & 2) != 0, want["hq"].astype(bool))
        for f in ("contig", "start", "end", "qlen", "name_len"):
            assert np.array_equal(got[f][p], want[f][p]), f
        names = [bytes(stream[int(a):int(a) + int(n)]) for a, n in zip(want["name_off"][p], want["name_len"][p])]
        assert np.array_equal(got["name_hash"][p], name_hash_np(names))
        assert p.sum() > 100
    assert n_checked > 5


@pytest.mark.parametrize("n_files,seed", [(1, 0), (2, 1), (3, 2), (4, 3), (5, 4), (9, 5), (16, 6)])
def test_name_join_randomised_dicts(engine, oracle, n_files, seed, join_mode):
    """The fold of GCI.py:279-299 on random per-file dicts drawn from a small name pool (many names in several
    files, on the same / different contigs, high-quality or not): exercises deletion, interval intersection, the
    ovlp / qlen-of-the-current-file test and resurrection by a later file with >= 3 files."""
    rng = np.random.default_rng(100 + seed)
    pool = [("read%05d" % i) for i in range(4000)]
    targets = ["t0", "t1", "t2"]
    files, hq = [], set()
    for f in range(n_files):
        d = {}
        for q in rng.choice(pool, size=int(rng.integers(1500, 3500)), replace=False):
            s = int(rng.integers(0, 50_000))
            ln = int(rng.integers(50, 20_000))
            qlen = int(max(1, ln + rng.integers(-40, 400)))
            d[str(q)] = (targets[int(rng.integers(0, 3)) if rng.random() < 0.15 else 0], s, s + ln, qlen)
        files.append(d)
    # make most shared names agree roughly with file 0 so that survivors exist
    for f in range(1, n_files):
        for q in list(files[f])[::2]:
            if q in files[0]:
                t, s, e, ql = files[0][q]
                j = int(rng.integers(-30, 30))
                files[f][q] = (t, max(0, s + j), e + j, max(1, e - s + int(rng.integers(-5, 60))))
    hq = set(str(q) for q in rng.choice(pool, size=1200, replace=False))
    # the reference's high-quality set only holds names that occur in some file
    hq &= set().union(*[set(d) for d in files])
    want = oracle.name_join(files, hq, 0.9)
    want = sorted((targets.index(v[0]), v[1], v[2]) for v in want.values())

    inputs = []
    tindex = {t: i for i, t in enumerate(targets)}
    for d in files:
        inputs.append(pipeline._paf_join_input(engine, d, hq, tindex))
    ivl, cnt = engine.name_join(inputs, 0.9)
    got = sorted(map(tuple, ivl[:int(cnt.item()), :3].cpu().numpy().tolist()))
    assert got == want
    assert len(want) > 300
    # a contig map that keeps only t0 (multi-GPU ownership filter) and renumbers it
    cmap = engine.to_device(np.array([0, -1, -1], dtype=np.int32))
    ivl2, cnt2 = engine.name_join(inputs, 0.9, contig_map=cmap)
    got2 = sorted(map(tuple, ivl2[:int(cnt2.item()), :3].cpu().numpy().tolist()))
    assert got2 == [w for w in want if w[0] == 0]


def test_depth_text_as_gzip_members_from_the_gpu(engine):
    """gci_depth_deflate_size / _write: the gzip members the device writes straight from the track decompress (CRC and
    length checked by Python's gzip) to exactly f'{depth}\\n' per base -- runs of every length and line width, contigs
    that end inside a tile / a member, depths up to 2^31 - 1."""
    import gzip
    rng = np.random.default_rng(77)
    lens = [1, 3, 4095, 4096, 4097, 64 * 4096 - 1, 64 * 4096, 64 * 4096 + 5, 300_001, 1_000_003]
    engine.set_layout(lens)
    track = engine.new_track()
    host = np.zeros(engine.total, dtype=np.int32)
    want = []
    for c, (off, L) in enumerate(zip(engine.offsets, lens)):
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
        want.append(b"".join(b"%d\n" % int(x) for x in d.tolist()) if L < 400_000 else ("\n".join(map(str, d.tolist())) + "\n").encode())
    track.copy_(torch.from_numpy(host))
    members = [bytes(b) for b in engine.depth_deflate(track)]
    assert len(members) == len(lens)
    for c, blob in enumerate(members):
        assert blob[:4] == b"\x1f\x8b\x08\x00"
        assert gzip.decompress(blob) == want[c], c
        assert blob.count(b"\x1f\x8b\x08\x00\x00\x00\x00\x00\x00\xff") >= -(-lens[c] // (64 * 4096))
    # the long-run contigs compress about a hundredfold
    assert len(members[9]) * 40 < len(want[9])
    # and against the text kernels on the same track
    text, off_t = engine.depth_text(track)
    whole = text.cpu().numpy().tobytes()
    for c in range(len(lens)):
        assert gzip.decompress(members[c]) == whole[int(off_t[c]):int(off_t[c + 1])]
    # byte for byte what libgci_cpu.so -- the same header through g++, the text written out and its CRC taken byte by byte -- makes
    # of the same track: the token rules and the GF(2) arithmetic of the device's CRC against an implementation without either trick
    from gci_amd import cpu
    ce = cpu.CpuEngine(threads=8)
    ce.set_layout(lens)
    assert [bytes(b) for b in ce.depth_deflate(host)] == members
    ce.close()


def test_gzip_members_from_the_run_lists_the_build_keeps(engine):
    """gci_build_opts.want_runs: k_tile_build writes down the constant-depth segments it holds in registers and the deflate
    passes take them instead of reading the track.  Same bytes as from the track itself -- over tiles with empty segments,
    neighbours of equal depth (an interval ends where another begins), dense tiles (pile-ups: more events than the sparse
    path takes, left to the walk), contigs that end inside a tile and contigs without any interval; a gap mask between the
    build and the deflate drops the lists (the members then show the masked track)."""
    import gzip
    rng = np.random.default_rng(4242)
    lens = [5, 4096, 4097, 70_000, 64 * 4096 + 17, 900_000, 12_345]
    engine.set_layout(lens)
    rows = []
    for c, L in enumerate(lens):
        if c == 6:
            continue                                         # a contig nothing aligns to
        n = max(3, L // 400)
        a = rng.integers(0, max(1, L - 1), n)
        b = np.minimum(L - 1, a + rng.integers(0, 3000, n))
        rows += [(c, int(x), int(y), 0) for x, y in zip(a, b)]
        # abutting intervals (the second begins where the first ended) and a pile-up in one tile
        for k in range(0, min(L, 60_000) - 200, 997):
            rows += [(c, k, k + 99, 0), (c, k + 100, k + 180, 0)]
        if L > 9000:
            st = rng.integers(4100, 8100, 300)
            rows += [(c, int(x), int(x) + 7, 0) for x in st]
    ivl = engine.to_device(np.asarray(rows, dtype=np.int32).reshape(-1, 4))
    plain = engine.new_track()
    engine.depth_build(ivl, None, 0, plain)
    want = [bytes(b) for b in engine.depth_deflate(plain)]          # lists made from the track (k_depth_runs)
    host = plain.cpu().numpy()
    for c, (off, L) in enumerate(zip(engine.offsets, lens)):
        assert gzip.decompress(want[c]) == ("\n".join(map(str, host[off:off + L].tolist())) + "\n").encode()
    track = engine.new_track()
    engine.depth_build_fused(ivl, None, 0, track, want_text=False, want_runs=True)
    assert torch.equal(track, plain)
    got = [bytes(b) for b in engine.depth_deflate(track, from_build=True)]
    assert got == want
    # the lists serve ONE size / write pair: a second statement "from the build" is refused, the plain call walks the track
    with pytest.raises(GciErr):
        engine.depth_deflate(track, from_build=True)
    assert [bytes(b) for b in engine.depth_deflate(track)] == want
    # ... then with flanks
    engine.depth_build(ivl, None, 15, plain)                        # (first: any build through this context drops the lists of the one before)
    engine.depth_build_fused(ivl, None, 15, track, want_text=False, want_runs=True)
    assert torch.equal(track, plain)
    assert [bytes(b) for b in engine.depth_deflate(track, from_build=True)] == [bytes(b) for b in engine.depth_deflate(plain)]
    # the lists are never taken on a pointer match alone: a write the context cannot see -- another context's gap
    # mask, here -- between the build and a deflate that does NOT claim "from the build" shows in the members
    engine.depth_build_fused(ivl, None, 0, track, want_text=False, want_runs=True)
    from gci_amd.device import Engine
    other = Engine(0)
    other.set_layout(lens)
    gaps_rows = np.asarray([(3, 100, 9000, 0), (5, 0, 4096, 0)], dtype=np.int32).reshape(-1, 4)
    other.gap_mask(track, other.to_device(gaps_rows))
    other.sync()
    unseen = [bytes(b) for b in engine.depth_deflate(track)]
    assert gzip.decompress(unseen[3]) == ("\n".join(map(str, track.cpu().numpy()[engine.offsets[3]:engine.offsets[3] + lens[3]].tolist())) + "\n").encode()
    other.close()
    # a mask through this context between build and deflate: the lists are dropped, claiming them is refused
    engine.depth_build_fused(ivl, None, 0, track, want_text=False, want_runs=True)
    gaps = engine.to_device(gaps_rows)
    engine.gap_mask(track, gaps)
    masked = track.cpu().numpy()
    with pytest.raises(GciErr):
        engine.depth_deflate(track, from_build=True)
    after = [bytes(b) for b in engine.depth_deflate(track)]
    for c, (off, L) in enumerate(zip(engine.offsets, lens)):
        assert gzip.decompress(after[c]) == ("\n".join(map(str, masked[off:off + L].tolist())) + "\n").encode()
    assert after[3] != want[3]


def _forged_input(engine, names, hashes, contig, start, end, qlen, hq):
    """A join input whose name hashes are given instead of computed (collision tests)."""
    n = len(names)
    r = np.zeros(n, dtype=REC_DTYPE)
    r["name_hash"] = np.asarray(hashes, dtype=np.uint64)
    r["contig"], r["start"], r["end"], r["qlen"] = contig, start, end, qlen
    r["rec_idx"] = np.arange(n)
    r["flags"] = 1 | (2 * np.asarray(hq, dtype=np.uint8))
    r["name_len"] = [len(x) for x in names]
    off = np.zeros(n + 1, dtype=np.int64)
    np.cumsum([len(x) for x in names], out=off[1:])
    blob = np.frombuffer(b"".join(names) or b"\x00", dtype=np.uint8)
    return JoinInput(engine.to_device(r.view(np.uint8).reshape(n, 32)), engine.to_device(blob), engine.to_device(off), 0)


def test_name_join_hash_collisions_and_repeats(engine, oracle, join_mode):
    """Names are confirmed on their bytes: reads whose 64-bit hashes (and lengths) coincide stay different reads, on
    both join paths; a name repeated thousands of times in one file keeps its last record (contig order first)."""
    rng = np.random.default_rng(77)
    n = 3000
    names = [b"coll%06d" % i for i in range(n)]                   # same length, forged to share hashes in groups of 3
    hashes = (np.arange(n, dtype=np.uint64) // np.uint64(3)) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(12345)
    files = []
    for f in range(2):
        d = {}
        for i in rng.permutation(n)[:2400]:
            s = 30 * int(i) + int(rng.integers(0, 40))             # the two files mostly agree on where a read lies
            d[names[i].decode()] = ("t0" if rng.random() < 0.95 else "t1", s, s + 9000 + f, 9000)
        files.append(d)
    hq = set(q for q in files[0] if rng.random() < 0.3) | set(q for q in files[1] if rng.random() < 0.3)
    want = oracle.name_join(files, hq, 0.9)
    want = sorted((0 if v[0] == "t0" else 1, v[1], v[2]) for v in want.values())
    hmap = {nm.decode(): int(h) for nm, h in zip(names, hashes)}
    inputs = []
    for d in files:
        qs = list(d)
        inputs.append(_forged_input(engine, [q.encode() for q in qs], [hmap[q] for q in qs],
                                    [0 if d[q][0] == "t0" else 1 for q in qs], [d[q][1] for q in qs], [d[q][2] for q in qs],
                                    [d[q][3] for q in qs], [q in hq for q in qs]))
    ivl, cnt = engine.name_join(inputs, 0.9)
    got = sorted(map(tuple, ivl[:int(cnt.item()), :3].cpu().numpy().tolist()))
    assert got == want and len(want) > 500
    # one name 20000 times (on both contigs, shuffled) among 500 others, single file: the dict keeps the last record
    # of the later contig
    rep = [b"again"] * 20000 + [b"other%04d" % i for i in range(500)]
    contig = np.concatenate([rng.integers(0, 2, 20000), np.zeros(500, dtype=np.int64)])
    start = np.arange(len(rep))
    one = _forged_input(engine, rep, name_hash_np(rep), contig, start, start + 100, np.full(len(rep), 100), np.zeros(len(rep)))
    ivl, cnt = engine.name_join([one], 0.9)
    got = sorted(map(tuple, ivl[:int(cnt.item()), :3].cpu().numpy().tolist()))
    last1 = int(np.flatnonzero(contig[:20000] == 1)[-1])
    assert got == sorted([(1, last1, last1 + 100)] + [(0, 20000 + i, 20100 + i) for i in range(500)])


def test_partitioned_join_bucket_overflow_falls_back(engine, monkeypatch):
    """More distinct names in one bucket than its LDS table holds (only forged hashes get there): the partitioned join
    reports GCI_E_CAPACITY, Engine.name_join() retries on the classic table."""
    from gci_amd import _lib
    n = 5000
    names = [b"ovf%05d" % i for i in range(n)]
    hashes = np.uint64(0xABCD) << np.uint64(32) | np.arange(n, dtype=np.uint64)          # same bucket and slot bits
    start = np.arange(n)
    one = _forged_input(engine, names, hashes, np.zeros(n), start, start + 50, np.full(n, 50), np.zeros(n))
    engine.set_join_mode("partition")
    try:
        with pytest.raises(GciErr) as e:
            engine.name_join([one], 0.9, fallback=False)
        assert e.value.status == _lib.GCI_E_CAPACITY
        ivl, cnt = engine.name_join([one], 0.9)
    finally:
        engine.set_join_mode("auto")
    assert sorted(map(tuple, ivl[:int(cnt.item()), :3].cpu().numpy().tolist())) == [(0, i, i + 50) for i in range(n)]


@pytest.mark.parametrize("counted", [False, True])
def test_fused_build_grows_the_issue_key_buffer(engine, oracle, counted, events_mode):
    """More issue-run boundaries than the key buffer holds (a fragmented, low-coverage assembly): depth_build_fused
    grows the buffer and runs its first pass again -- after a counting join, too, whose per-tile counts the first
    attempt has used up.  Depth, sums, text and runs must come out as from the oracle."""
    L = 400_000
    contigs = (("frag", L), ("tail", 30_000))
    lengths = [l for _, l in contigs]
    engine.set_layout(lengths)
    rng = np.random.default_rng(909)
    starts = np.sort(rng.choice(np.arange(0, L - 400, 700), size=500, replace=False))     # 500 short reads: ~1000 boundaries
    ivl_np = np.zeros((starts.shape[0], 4), dtype=np.int32)
    ivl_np[:, 1], ivl_np[:, 2] = starts, starts + rng.integers(100, 600, starts.shape[0])
    want = oracle.depth_build_py([("frag", int(s), int(e)) for _, s, e, _ in ivl_np.tolist()], dict(contigs), 15)
    bed = oracle.collapse_depth_range(want, -1, 0, 15, 0)
    assert len(bed["frag"]) > 300
    if counted:
        names = [b"r%05d" % i for i in range(ivl_np.shape[0])]
        one = _forged_input(engine, names, name_hash_np(names), ivl_np[:, 0], ivl_np[:, 1], ivl_np[:, 2],
                            np.full(len(names), 500), np.zeros(len(names)))
        ivl, cnt = engine.name_join([one], 0.9, count_flank=15)
    else:
        ivl, cnt = engine.to_device(ivl_np), None
    track = engine.new_track()
    out = engine.depth_build_fused(ivl, cnt, 15, track, want_text=True, want_sums=True, issue=(-1, 0, 15), counted=counted,
                                   key_cap=64)
    tr = pipeline.DepthTracks(engine, dict(contigs), track)
    for t in ("frag", "tail"):
        assert np.array_equal(tr[t], want[t]), t
    assert out["sums"].tolist() == [int(want[t].sum()) for t in ("frag", "tail")]
    assert out["text"].cpu().numpy().tobytes() == b"".join(oracle.depth_text_contig(want[t]) for t in ("frag", "tail"))
    got = {t: pipeline._issues_from_runs(out["runs"][c], lengths[c] - 30, lengths[c], 15, 0) for c, t in enumerate(("frag", "tail"))}
    assert got == bed


def test_depth_gz_skips_zero_length_contigs(engine, tmp_path):
    """write_depth (GCI.py:99-143) writes nothing at all for a contig of length 0 -- its chunk loop never runs, so not
    even the '>' line appears."""
    import gzip
    lengths = {"a": 5000, "empty": 0, "b": 4100}
    engine.set_layout(list(lengths.values()))
    track = engine.new_track()
    track.zero_()
    tr = pipeline.DepthTracks(engine, lengths, track)
    pipeline.write_depth(str(tmp_path), "z", tr, 1)
    text = gzip.open(str(tmp_path / "z.depth.gz"), "rb").read()
    assert text == b">a\n" + b"0\n" * 5000 + b">b\n" + b"0\n" * 4100


def test_paged_filter_of_tag_heavy_records(engine, oracle):
    """A file in which EVERY record carries ~800 bytes of tags behind a CIGAR of some 60 operations (HiFi with MD / cs / SA): none
    fits a page record, so each one keeps its CIGAR in the blob in a piece of 240 bytes and every passing one goes through the
    chunk queue -- whose capacity round 3 derived from blob bytes / 512 and refused such a file (GCI_E_CAPACITY)."""
    from gci_amd.formats import bam
    rng = np.random.default_rng(77)
    refs = [("t0", 4_000_000), ("t1", 1_000_000)]
    recs = []
    for i in range(6000):
        ops = []
        for k in range(int(rng.integers(40, 100))):
            ops.append((7, int(rng.integers(100, 400))) if k % 2 == 0 else (int(rng.choice([1, 2, 8])), int(rng.integers(1, 3))))
        if rng.random() < 0.1:
            ops.insert(0, (4, int(rng.integers(10, 4000))))
        qlen = sum(l for o, l in ops if (bam.QUERY_CONSUMING >> o) & 1)
        nm = sum(l for o, l in ops if o in (1, 2, 8))
        tags = [("NM", "i", nm), ("MD", "Z", "A7" * int(rng.integers(150, 250))), ("cs", "Z", ":9*ag" * int(rng.integers(60, 90))),
                ("SA", "Z", "t1,100,+,50S900M,60,3;")]
        if i % 50 == 0:
            tags = tags[1:] + tags[:1]                 # NM behind the long tags
        recs.append(bam.encode_record(int(rng.integers(0, 2)), int(rng.integers(0, 900_000)), "tagged/%d/ccs" % i,
                                      int(rng.choice([60, 60, 60, 20])), int(rng.choice([0, 16, 0x800])), ops, qlen, bam.encode_aux(tags)))
    hdr = bam.encode_header([r for r, _ in refs], [l for _, l in refs])
    stream = np.frombuffer(hdr + b"".join(recs), dtype=np.uint8).copy()
    offs = bam.record_offsets(stream, bam.parse_header(stream).first_record)
    ref_sel = np.array([0, 1], np.int32)
    want = oracle.bam_filter_arrays(stream, offs, ref_sel, 30, 50, 0.1, 0.9)
    d_bam, d_off = engine.to_device(stream), engine.to_device(offs)
    full = engine.bam_filter(d_bam, d_off, engine.to_device(ref_sel), 30, 50, 0.1, 0.9).clone()
    pages = _via_pages(engine, d_bam, d_off, True, ref_sel, 30, 50, 0.1, 0.9, full, stream)
    blob = int(pages.buf.shape[0]) - pages.blob_off - 16
    assert 0 < blob < 6000 * 512                       # the bound of round 3 (blob / 512 items) would have been < the records queued
    got = _recs_np(full)
    p = want["passed"].astype(bool)
    assert np.array_equal((got["flags"] & 1).astype(bool), p) and p.sum() > 2000
    for f in ("contig", "start", "end", "qlen"):
        assert np.array_equal(got[f][p], want[f][p]), f


@pytest.mark.parametrize("flank,threshold,with_gaps", [(15, 0, True), (15, 0, False), (0, 1, True), (40, 2, True)])
def test_two_type_tail_in_one_pass_equals_the_seams(engine, oracle, flank, threshold, with_gaps):
    """gci_two_type_tail -- N-run masks of both tracks, their per-base maximum and the issue runs of all three in ONE pass over
    the two tracks -- against the oracle's merge_gaps_depths / max2 / collapse_depth_range, through the pipeline's lazy masks
    (merge_gaps_depths(lazy=True) + merge_two_type_depth(issue_hint=...)): runs that start or end on tile borders, at the window
    edges, inside and at the borders of N runs, N runs that span tiles and contigs that are shorter than a tile or than 2 flanks."""
    rng = np.random.default_rng(100 + flank)
    shapes = {"a": 200_001, "b": 4096, "c": 8193, "tiny": 20, "one": 2 * flank + 1, "d": 12_345}
    h = {k: _random_depth(rng, L) for k, L in shapes.items()}
    n = {k: _random_depth(rng, L) for k, L in shapes.items()}
    for t in (h, n):                                  # low-depth runs right at tile borders and across them
        t["a"][4090:4100] = 0; t["a"][8192:8200] = 0; t["a"][12280:12288] = 0
    h["a"][100_000:100_050] = 0; n["a"][100_020:100_090] = 0      # a run of the maximum that is a run of neither alone in full
    gaps = {"a": [(4000, 4097), (8190, 12_300), (150_000, 150_001), (199_990, 250_000), (-30, -10)], "c": [(0, 5), (8192, 8193)],
            "tiny": [(3, 9)], "nope": [(1, 2)]} if with_gaps else None
    th, tn = _upload_depths(engine, h), _upload_depths(engine, n)
    pipeline.merge_gaps_depths(th, gaps, lazy=True)
    pipeline.merge_gaps_depths(tn, gaps, lazy=True)
    assert (th._pending_gaps is not None) == with_gaps
    two = pipeline.merge_two_type_depth(th, tn, write=False, issue_hint=(-1, threshold, flank))
    assert th._pending_gaps is None and tn._pending_gaps is None and two._fresh_runs is not None
    pipeline.merge_gaps_depths(two, gaps)             # the reference masks the merged track as well (GCI.py:1019): nothing left to do
    assert two._fresh_runs is not None
    oracle.merge_gaps_depths(h, gaps)
    oracle.merge_gaps_depths(n, gaps)
    m = oracle.max2(h, n)
    oracle.merge_gaps_depths(m, gaps)
    for tr, want in ((th, h), (tn, n), (two, m)):
        for k in shapes:
            assert np.array_equal(tr[k], want[k]), k
        assert pipeline.collapse_depth_range(tr, -1, threshold, flank, 0) == oracle.collapse_depth_range(want, -1, threshold, flank, 0)
        fresh = tr._fresh_runs
        tr.invalidate()                                # ... and the same from a scan of the finished track
        assert pipeline.collapse_depth_range(tr, -1, threshold, flank, 0) == oracle.collapse_depth_range(want, -1, threshold, flank, 0)
        assert fresh is not None
    assert sum(len(v) for v in oracle.collapse_depth_range(m, -1, threshold, flank, 0).values()) > 5
    assert two.mean() == oracle.mean_depth(m)
    # the per-contig sums came out of the same pass (the mean depth of a `-p` run): equal to a pass of their own
    th2, tn2 = _upload_depths(engine, h), _upload_depths(engine, n)
    pipeline.merge_gaps_depths(th2, gaps, lazy=True)
    pipeline.merge_gaps_depths(tn2, gaps, lazy=True)
    two2 = pipeline.merge_two_type_depth(th2, tn2, write=False, issue_hint=(-1, threshold, flank))
    for tr, want in ((th2, h), (tn2, n), (two2, m)):
        assert tr._fresh_sums is not None and [int(x) for x in tr.sums()] == [int(want[k].sum()) for k in shapes]
        assert tr.mean() == oracle.mean_depth(want)
        assert [int(x) for x in engine.depth_sum(tr.track)] == [int(want[k].sum()) for k in shapes]


# ---- the two divisions of GCI.py:165 at their thresholds, through the kernel the product runs (round 6) -------------------------
def _threshold_records():
    """Records whose clip quotient S / (M + I + S) and identity quotient (M - mm) / (M + I + D) cover what single precision can
    and cannot tell apart: small exact fractions, denominators beyond 2^24 (inexact as f32), beyond 2^30 (the 32-bit path of the
    kernel is left), one-operation and many-operation CIGARs, quotients of 0 and 1."""
    from test_bam_decode import op, rec_bytes
    import struct
    shapes = [  # (S, M, I, D, mismatches under M)
        (10, 90, 0, 0, 0), (10, 90, 0, 0, 9), (11, 89, 0, 0, 10), (1, 9, 0, 0, 1), (0, 100, 0, 0, 10), (0, 100, 0, 0, 0),
        (100, 900, 0, 0, 100), (333, 2667, 30, 0, 270), (1000, 9000, 100, 100, 720), (1801, 16203, 7, 9, 1620),
        (1 << 20, 9 << 20, 0, 0, 1 << 20), ((1 << 20) + 1, 9 << 20, 0, 0, (1 << 20) - 1), (1677722, 15099494, 3, 5, 1509949),
        (16777217, 150994943, 11, 13, 15099494), (26843546, 241591910, 0, 1, 24159191),
        (100, 0, 0, 0, 0), (0, 1, 0, 0, 0), (0, 1, 0, 0, 1), (5, 45, 0, 0, 5), (7, 63, 1, 1, 5), (12345, 111105, 17, 19, 11112),
        (99999, 900001, 0, 0, 90000), (50000, 450000, 0, 0, 45001), (3, 30, 0, 0, 3), (2, 17, 1, 0, 2), (214748364, 1932735283, 0, 0, 193273528),
    ]
    out = []
    for k, (S, M, I, D, mm) in enumerate(shapes):
        ops = []
        if S:
            for piece in _pieces(S):
                ops.append(op(piece, "S"))
        left = M
        parts = _pieces(M) if M else []
        for j, piece in enumerate(parts):
            ops.append(op(piece, "M" if (k + j) % 2 else "="))
            if j == 0 and I:
                ops.append(op(I, "I"))
            if j == 0 and D:
                ops.append(op(D, "D"))
        if not parts:
            if I:
                ops.append(op(I, "I"))
            if D:
                ops.append(op(D, "D"))
        nm = I + D + mm
        aux = (b"NMC" + bytes([nm])) if nm < 256 else (b"NMI" + struct.pack("<I", nm))
        out.append((rec_bytes(0, 100 + k, b"thr/%d/ccs" % k, 60, 0, ops, 50, aux), (S, M + I + S, M + I + D - nm, M + I + D)))
    return out


def _pieces(n, cap=(1 << 28) - 1):
    """n as operation lengths (an operation holds 28 bits): one, or a few."""
    out = []
    while n > cap:
        out.append(cap)
        n -= cap
    if n > (1 << 24) + 5:                      # (two operations where one would do: the lean path sums, the full path sums)
        out += [n - (1 << 24) - 3, (1 << 24) + 3]
    elif n:
        out.append(n)
    return out


def test_hand_assembled_records_through_the_paged_filter(engine):
    """tests/test_bam_decode.py's hand-assembled records -- the pysam-independent anchors of R1, "10 % clip passes exactly" and
    "identity 0.9 exactly" among them -- through the kernel the command line and bench.py run (gci_bam_pages_* +
    gci_bam_filter_pages), from the whole stream and from the heads stream."""
    from test_bam_decode import HAND, stream_of
    from gci_amd.formats import bam
    s, offs, h = stream_of([r for _, r, _ in HAND])
    h_bytes, h_offs = heads_expected(s, offs, bam.parse_header(s).first_record)
    sel = engine.to_device(np.array([0, 1], np.int32))
    for stream, o, has_seq in ((s, offs, True), (np.frombuffer(h_bytes, dtype=np.uint8), h_offs, False)):
        for pb in (0, 8192):
            pages = engine.bam_pages(engine.to_device(stream), engine.to_device(o), has_seq, pb)
            recs = _recs_np(engine.bam_filter_pages(pages, sel, 30, 50, 0.1, 0.9)[0])
            for i, (desc, _, want) in enumerate(HAND):
                if want is None:
                    continue
                f = int(recs["flags"][i])
                got = (f & 1, (f >> 1) & 1, int(recs["start"][i]), int(recs["end"][i]), int(recs["qlen"][i])) if f & 1 else (0, 0, 0, 0, 0)
                assert got == want, (desc, has_seq, pb)


@pytest.mark.parametrize("which", ["clip", "identity"])
def test_filter_thresholds_at_and_around_every_quotient(engine, oracle, which):
    """GCI.py:165 compares two f64 quotients with -cp / -ip.  The paged filter decides them in single precision unless the quotient
    lies within 1e-4 of the threshold (k_filter.hip: ratio_cmp and the lean path's copy): here every record's f64 quotient q is
    taken as the threshold itself, one ulp either side of it, and q +- 1e-7, 1e-6, 1e-5, 9.9e-5, 1.01e-4, 1e-3 -- i.e. exactly
    on, just inside and just outside the band in which the kernel must fall back to the f64 division -- through the stream
    kernel, the pages of the whole stream and the pages of the heads stream, against the oracle's division, all records each time."""
    from test_bam_decode import stream_of
    from gci_amd.formats import bam
    items = _threshold_records()
    s, offs, h = stream_of([r for r, _ in items], refs=(("chr1", 2_000_000_000),))
    ref_sel = np.array([0], np.int32)
    sel = engine.to_device(ref_sel)
    h_bytes, h_offs = heads_expected(s, offs, bam.parse_header(s).first_record)
    d_s, d_o = engine.to_device(s), engine.to_device(offs)
    pages = [engine.bam_pages(d_s, d_o, True), engine.bam_pages(engine.to_device(np.frombuffer(h_bytes, dtype=np.uint8)), engine.to_device(h_offs), False, 8192)]
    seen, n_flips, n_errors = set(), 0, 0
    for _, (a1, b1, a2, b2) in items:
        a, b = (a1, b1) if which == "clip" else (a2, b2)
        if b == 0:
            continue
        q = a / b
        cands = [q, np.nextafter(q, np.inf), np.nextafter(q, -np.inf)]
        for d in (1e-7, 1e-6, 1e-5, 9.9e-5, 1.01e-4, 1e-3):
            cands += [q + d, q - d, q * (1 + d), q * (1 - d)]
        for c in cands:
            c = float(c)
            if c in seen:
                continue
            seen.add(c)
            cp, ip = (c, 0.5) if which == "clip" else (0.75, c)
            try:
                want = oracle.bam_filter_arrays(s, offs, ref_sel, 30, 50, cp, ip)["passed"].astype(bool)
            except oracle.OracleRecordError as e:          # (the all-clipped record passes the clip test: ZeroDivisionError behind it)
                for call in ([lambda: engine.bam_filter(d_s, d_o, sel, 30, 50, cp, ip)] +
                             [lambda pg=pg: engine.bam_filter_pages(pg, sel, 30, 50, cp, ip) for pg in pages]):
                    with pytest.raises(GciErr) as g:
                        call()
                    assert (g.value.status, g.value.rec) == (e.status, e.rec), (which, c)
                n_errors += 1
                continue
            got_stream = (_recs_np(engine.bam_filter(d_s, d_o, sel, 30, 50, cp, ip))["flags"] & 1).astype(bool)
            assert np.array_equal(got_stream, want), ("stream", which, c)
            for k, pg in enumerate(pages):
                got = (_recs_np(engine.bam_filter_pages(pg, sel, 30, 50, cp, ip)[0])["flags"] & 1).astype(bool)
                assert np.array_equal(got, want), ("pages", k, which, c, np.flatnonzero(got != want))
            n_flips += int(want.sum())
    assert len(seen) > 300 and n_flips > 1000 and (n_errors > 0) == (which == "clip")
    # the twin in pure Python (the reference's own expression, GCI.py:165) on the thresholds that sit exactly on a quotient
    for i, (_, (a1, b1, a2, b2)) in enumerate(items):
        if b1 and b2:
            rec = bam.decode_record(s, int(offs[i]))
            for cp, ip in ((a1 / b1, a2 / b2), (np.nextafter(a1 / b1, -np.inf), a2 / b2), (a1 / b1, np.nextafter(a2 / b2, np.inf))):
                want = oracle.bam_filter_record_py(rec, h.references, list(h.references), 30, 50, float(cp), float(ip)) is not None
                got = bool(_recs_np(engine.bam_filter_pages(pages[0], sel, 30, 50, float(cp), float(ip))[0])["flags"][i] & 1)
                assert got == want, (i, cp, ip)
