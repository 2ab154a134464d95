"""GPU end-to-end parity: the drop-in CLI against files the UNMODIFIED reference wrote
(tests/golden/*), the reference's own MH63 example through the HIP kernels, hand-assembled
records through K1, reference error behaviour, and size-independent properties at full size."""
import gzip
import os
import struct

import numpy as np
import pytest
import torch

from golden_util import CASES, GOLDEN, cli_args, expected, images, manifest, read_outputs
from gci_amd import pipeline, synth
from gci_amd._lib import GciError
from gci_amd.device import JoinInput, REC_DTYPE

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", CASES)
def test_cli_reproduces_reference_files(engine, case, tmp_path, capsys):
    from gci_amd import cli
    out = str(tmp_path / "out")
    pipeline._ENGINE = engine
    cli.main(cli_args(case, out))
    got, want = read_outputs(out), expected(case)
    assert sorted(got) == sorted(want)
    for fn in want:
        assert got[fn] == want[fn], fn
    # -p: the figures, pixel for pixel (the numbers come from the GPU, the drawing is matplotlib as in the reference)
    got_img, want_img = images(out), images(os.path.join(GOLDEN, case, "expected"))
    assert sorted(got_img) == sorted(want_img)
    for fn in want_img:
        assert got_img[fn].shape == want_img[fn].shape and np.array_equal(got_img[fn], want_img[fn]), fn
    # R14: the transcript of GCI() equals the reference's (first line = main()'s echo of the arguments)
    stdout = capsys.readouterr().out
    first, _, rest = stdout.partition("\n")
    assert first.startswith("Used arguments:{")
    inp = os.path.join(GOLDEN, case, "inputs")
    assert rest.replace(out, "{OUT}").replace(inp, "{IN}") == manifest(case)["stdout"]
    # refuses to overwrite without -f, like the reference
    with pytest.raises(SystemExit) as e:
        cli.main(cli_args(case, out))
    assert "exists" in str(e.value) and "--force" in str(e.value)
    cli.main(cli_args(case, out) + ["-f"])
    assert read_outputs(out) == got


def test_first_bam_is_ingested_on_a_helper_thread_beside_the_n_scan(engine, tmp_path, monkeypatch, capsys):
    """The command line starts the ingestion of its first BAM file on a helper thread before it scans the assembly for N runs
    (pipeline.start_ingest_ahead; the N scan then runs on a context of its own) and filter() takes the result over: same files
    as with GCI_INGEST_AHEAD=0, the first file's bam_join_input() off the main thread, nothing left behind."""
    import threading
    from gci_amd import cli
    case = "c3_two_bam" if "c3_two_bam" in CASES else CASES[0]
    pipeline._ENGINE = engine
    seen = []
    real = pipeline.bam_join_input

    def spy(eng, path, *a, **kw):
        seen.append((os.path.basename(path), threading.current_thread() is threading.main_thread(), eng is engine))
        return real(eng, path, *a, **kw)

    monkeypatch.setattr(pipeline, "bam_join_input", spy)
    outs = {}
    for knob in ("1", "0"):
        monkeypatch.setenv("GCI_INGEST_AHEAD", knob)
        del seen[:]
        out = str(tmp_path / ("out" + knob))
        cli.main(cli_args(case, out))
        outs[knob] = read_outputs(out)
        assert not pipeline._INGEST_AHEAD and not pipeline._TABLES
        assert all(on_engine for _, _, on_engine in seen)
        first_on_main = seen[0][1]
        assert first_on_main == (knob == "0") and all(on_main for _, on_main, _ in seen[1:])
    capsys.readouterr()
    assert outs["1"] == outs["0"] == {k: v for k, v in expected(case).items()}


def test_mh63_example_through_gpu(engine, oracle):
    """example/MH63.depth.gz -> MH63.0.depth.bed + MH63.gci, byte for byte, with the scan and the
    text on the GPU (396 Mb, 12 contigs)."""
    d = os.path.join(GOLDEN, "MH63")
    text = gzip.open(os.path.join(d, "MH63.depth.gz"), "rb").read()
    depths = oracle.parse_depth_text(text)
    targets = list(depths)
    tl = {t: int(depths[t].shape[0]) for t in targets}
    offs = engine.set_layout([tl[t] for t in targets])
    flat = np.zeros(engine.total, dtype=np.int32)
    for o, t in zip(offs, targets):
        flat[o:o + tl[t]] = depths[t]
    tr = pipeline.DepthTracks(engine, tl, engine.to_device(flat))
    merged = pipeline.collapse_depth_range(tr, -1, 0, 15, 0)
    bed = "".join(f"{t}\t{s}\t{e}\n" for t, v in merged.items() for s, e in v)
    assert bed == open(os.path.join(d, "MH63.0.depth.bed")).read()
    from gci_amd import score
    assert score.index_text(tl, [merged], ["HiFi"]) == open(os.path.join(d, "MH63.gci")).read()
    t_dev, toff = engine.depth_text(tr.track)
    host = t_dev.cpu().numpy()
    rebuilt = b"".join((">%s\n" % t).encode() + host[int(toff[c]):int(toff[c + 1])].tobytes() for c, t in enumerate(targets))
    assert rebuilt == text
    assert tr.mean() == oracle.mean_depth(depths)


def test_hand_assembled_records_through_k1(engine):
    from test_bam_decode import HAND, stream_of
    s, offs, h = stream_of([r for _, r, _ in HAND])
    recs = engine.bam_filter(engine.to_device(s), engine.to_device(offs), engine.to_device(np.array([0, 1], np.int32)),
                             30, 50, 0.1, 0.9).cpu().numpy().reshape(-1).view(REC_DTYPE)
    for i, (desc, _, want) in enumerate(HAND):
        if want is None:
            continue
        f = int(recs["flags"][i])
        got = (f & 1, (f >> 1) & 1, int(recs["start"][i]), int(recs["end"][i]), int(recs["qlen"][i]))
        assert got == want, desc


def test_reference_errors_surface_as_the_same_exceptions(engine, tmp_path):
    from test_bam_decode import op, rec_bytes
    from gci_amd.formats import bam
    good = rec_bytes(0, 10, b"ok", 60, 0, [op(100, "M")], 100, b"NMC\x00")
    for bad, exc in ((rec_bytes(0, 10, b"nonm", 60, 0, [op(100, "M")], 100, b"ASi\x00\x00\x00\x00"), KeyError),
                     (rec_bytes(0, 10, b"zd", 60, 0, [op(100, "H")], 0, b"NMC\x00"), ZeroDivisionError)):
        p = str(tmp_path / ("%s.bam" % exc.__name__))
        bam.write_bam(p, ["chr1"], [100000], [good, bad, good])
        with pytest.raises(exc):
            pipeline.filter([], [p], prefix="x", directory=str(tmp_path), engine=engine, write=False)
    # a PAF line the reference dies on: Python's own exception for that line (type and message), GCI.py:217-229
    ok = "q\t100\t0\t50\t+\tchr1\t100000\t0\t50\t50\t50\t60\n"
    for k, (line, exc, msg) in enumerate((("q2\t100\t0\t50\t+\tchr1\t100000\t0\t50\tfifty\t50\t60\n", ValueError, "invalid literal for int\\(\\) with base 10: 'fifty'"),
                                          ("q2\t100\t0\t50\t+\tchr1\t100000\n", IndexError, "list index out of range"),
                                          ("q2\t100\t0\t50\t+\tchr1\t100000\t0\t50\t50\t0\t60\n", ZeroDivisionError, "division by zero"))):
        pf = tmp_path / ("bad%d.paf" % k)
        pf.write_text(ok + line + ok)
        pg = str(tmp_path / "good_for_paf.bam")
        bam.write_bam(pg, ["chr1"], [100000], [good])
        with pytest.raises(exc, match=msg):
            pipeline.filter([str(pf)], [pg], prefix="x", directory=str(tmp_path), engine=engine, write=False)
    # qlen == 0 on the second file of a join -> ZeroDivisionError at GCI.py:292
    a = rec_bytes(0, 10, b"q", 60, 0, [op(100, "M")], 100, b"NMC\x00")
    b = rec_bytes(0, 10, b"q", 60, 0, [op(100, "M")], 0, b"NMC\x00")
    pa, pb = str(tmp_path / "a.bam"), str(tmp_path / "b.bam")
    bam.write_bam(pa, ["chr1"], [100000], [a])
    bam.write_bam(pb, ["chr1"], [100000], [b])
    with pytest.raises(ZeroDivisionError):
        pipeline.filter([], [pa, pb], prefix="x", directory=str(tmp_path), engine=engine, write=False)


def test_full_size_properties_chr19(engine, oracle):
    """BASELINE configs[1] size (61.7 Mb, 40x): properties that do not need the oracle to finish the
    whole thing -- sum of depth == sum of trimmed interval lengths (a checksum of checksums),
    text round trip, linearity of the depth build, idempotence of the gap mask, max2 with self."""
    L = 61_707_364
    rs = synth.simulate_reads((("chr19", L),), 40, "hifi", seed=synth.seed_for(2, 0))
    stream, offs = synth.to_bam_stream(rs)
    engine.set_layout([L])
    d_bam, d_off = engine.to_device(stream), engine.to_device(offs)
    recs = engine.bam_filter(d_bam, d_off, engine.to_device(np.zeros(1, np.int32)), 30, 50, 0.1, 0.9)
    ivl, cnt = engine.name_join([JoinInput(recs, d_bam, d_off, 36)], 0.9)
    n = int(cnt.item())
    track = engine.new_track()
    engine.depth_build(ivl, cnt, 15, track)
    tr = pipeline.DepthTracks(engine, {"chr19": L}, track)
    h = ivl[:n].cpu().numpy().astype(np.int64)
    a = np.clip(h[:, 1] + 15, 0, L)
    b = np.clip(h[:, 2] - 15 + 1, 0, L)
    assert int(tr.sums()[0]) == int(np.maximum(b - a, 0).sum())
    # oracle on the same records (C, a few seconds at this size)
    want = oracle.bam_filter_arrays(stream, offs, np.zeros(1, np.int32), 30, 50, 0.1, 0.9)
    assert n == int(want["passed"].sum())
    # linearity: depth(A u B) == depth(A) + depth(B)
    half = n // 2
    t1, t2 = engine.new_track(), engine.new_track()
    engine.depth_build(ivl[:half].contiguous(), None, 15, t1)
    engine.depth_build(ivl[half:n].contiguous(), None, 15, t2)
    assert torch.equal(t1 + t2, track)
    # text round trip at full size
    text, toff = engine.depth_text(track)
    raw_text = text.cpu().numpy().tobytes()
    back = oracle.parse_depth_text(b">chr19\n" + raw_text)["chr19"]
    assert np.array_equal(back, tr["chr19"])
    # the gzip members the device writes for the same track (236 members, every CRC checked by gzip) hold the same text
    members = [bytes(b) for b in engine.depth_deflate(track)]
    assert len(members) == 1 and len(members[0]) * 50 < len(raw_text)
    assert gzip.decompress(members[0]) == raw_text
    # issue scan == oracle scan of the same depth
    assert pipeline.collapse_depth_range(tr, -1, 0, 15, 0)["chr19"] == oracle.collapse_contig(back, -1, 0, 15, 0)
    # gap mask idempotent, max2(x, x) == x
    gaps = {"chr19": [(1_000_000, 1_200_000), (L - 10, L + 50)]}
    pipeline.merge_gaps_depths(tr, gaps)
    once = tr.track.clone()
    pipeline.merge_gaps_depths(tr, gaps)
    assert torch.equal(once, tr.track)
    assert torch.equal(engine.max2(tr.track, tr.track), tr.track)
    assert int(tr.track[1_000_000:1_200_000].sum().item()) == 0


def test_empty_and_ragged_inputs(engine, oracle, tmp_path):
    from gci_amd.formats import bam
    # a BAM with a header and no records; contigs shorter than 2 * flank; a 1-base contig
    p = str(tmp_path / "empty.bam")
    bam.write_bam(p, ["c1", "tiny", "one"], [5000, 20, 1], [])
    depths, tl = pipeline.filter([], [p], prefix="e", directory=str(tmp_path), engine=engine, threads=1)
    assert tl == {"c1": 5000, "tiny": 20, "one": 1}
    host = depths.to_host()
    assert all(int(v.sum()) == 0 for v in host.values())
    got = pipeline.collapse_depth_range(depths, -1, 0, 15, 0)
    assert got == oracle.collapse_depth_range(host, -1, 0, 15, 0) == {"c1": [(15, 4985)], "tiny": [], "one": []}
    assert gzip.open(str(tmp_path / "e.depth.gz"), "rb").read() == oracle.depth_text(host)


def test_tables_and_first_runs_made_ahead_give_the_same_join_input(engine, oracle, tmp_path, monkeypatch):
    """prefetch_member_tables(): the member tables of the files of a run (read with pread) and the upload of every file's first
    run are started on a helper thread, file after file; bam_join_input() takes them over.  Same records as without, the
    uploaders are the helper's, and nothing is left behind."""
    import threading
    from gci_amd.formats import bam
    contigs = (("a", 400_000), ("b", 150_000))
    targets, filt = ["a", "b"], (30, 50, 0.1, 0.9)
    paths, want = [], []
    for k, seed in enumerate((5, 6)):
        rs = synth.simulate_reads(contigs, 8, "hifi", seed=seed).sorted()
        stream, offs = synth.to_bam_stream(rs)
        p = str(tmp_path / ("f%d.bam" % k))
        bam.write_bam_stream(p, stream, level=1, threads=4)
        paths.append(p)
        want.append(oracle.bam_file_dict(stream, offs, targets, targets, *filt)[0])
    monkeypatch.setattr(pipeline, "GPU_INFLATE_MAX", 0)
    monkeypatch.setattr(pipeline, "BAM_CHUNK_BYTES", 400_000)
    monkeypatch.setattr(pipeline, "_ENGINE", engine)
    made = []
    init = pipeline._RunUploads.__init__

    def spy(self, *a, **kw):
        made.append(threading.current_thread() is threading.main_thread())
        init(self, *a, **kw)

    monkeypatch.setattr(pipeline._RunUploads, "__init__", spy)
    engine.set_layout([dict(contigs)[t] for t in targets])
    plain = [pipeline.bam_join_input(engine, p, targets, filt, threads=4) for p in paths]
    assert made == [True, True]
    del made[:]
    pipeline.prefetch_member_tables(paths)
    assert sorted(pipeline._TABLES) == sorted(paths)
    ahead = [pipeline.bam_join_input(engine, p, targets, filt, threads=4) for p in paths]
    assert made == [False, False] and not pipeline._TABLES
    for a, b, w in zip(plain, ahead, want):
        ra, rb = a.recs.cpu().numpy(), b.recs.cpu().numpy()
        assert ra.shape[0] == rb.shape[0] and np.array_equal(ra, rb)
        assert int((ra.reshape(-1).view(pipeline.REC_DTYPE)["flags"] & 1).sum()) >= len(w) > 0
    # a table nobody asks for with the device path: dropped, its uploader closed
    pipeline.prefetch_member_tables(paths[:1])
    ji = pipeline.bam_join_input(engine, paths[0], targets, filt, threads=4, ingest="heads")
    assert ji.recs.shape[0] == plain[0].recs.shape[0] and not pipeline._TABLES


@pytest.mark.parametrize("k1", ["pages", "stream"])
def test_streamed_bam_ingestion_equals_one_shot(engine, oracle, tmp_path, monkeypatch, k1):
    """A BAM larger than the chunk budget is streamed: groups of BGZF members, partial records carried over, K1 per
    chunk, the record pages of every run kept (k1 = pages, the default) or the names packed (GCI_K1=stream: the record
    filter over the inflated stream itself).  Same join input content, same depth, for chunk sizes that cut records at
    every phase."""
    from gci_amd.formats import bam
    monkeypatch.setattr(pipeline, "K1_MODE", k1)
    contigs = (("a", 600_000), ("b", 250_000))
    rs = synth.simulate_reads(contigs, 12, "hifi", seed=91)
    dup = rs.take(np.arange(0, len(rs), 11))                        # repeated names far apart in the file
    dup.pos[:] = np.minimum(dup.pos + 70_000, 200_000)
    rs = synth.concat(rs, dup).sorted()
    stream, offs = synth.to_bam_stream(rs)
    p = str(tmp_path / "s.bam")
    bam.write_bam_stream(p, stream, level=1, threads=4)
    targets = ["a", "b"]
    tl = dict(contigs)
    filt = (30, 50, 0.1, 0.9)
    want_d, hq = oracle.bam_file_dict(stream, offs, targets, targets, *filt)
    want = oracle.depth_build(oracle.name_join([want_d], hq, 0.9), tl, 15)
    engine.set_layout([tl[t] for t in targets])
    # inflated + walked on the device (ingest "gpu", the default) whole and run by run of members; the heads stream of the
    # host pipeline; the whole inflated stream in one upload, and streamed in chunks from the host
    cases = [(None, None, None), (None, "gpu", 0), (None, "heads", None), (None, "full", None),
             (9_000_001, None, None), (1_234_567, None, None), (300_000, None, None)]
    for chunk, ingest, gpu_max in cases:
        packed = chunk is not None
        if gpu_max is not None:                       # force the run-by-run device path with small runs
            for run in (1_234_567, 300_000):
                monkeypatch.setattr(pipeline, "GPU_INFLATE_MAX", gpu_max)
                monkeypatch.setattr(pipeline, "BAM_CHUNK_BYTES", run)
                ji = pipeline.bam_join_input(engine, p, targets, filt, threads=4, ingest=ingest)
                assert ji.name_delta == 0 and ji.recs.shape[0] == len(rs)
            monkeypatch.undo()
            monkeypatch.setattr(pipeline, "K1_MODE", k1)
            packed = True
        else:
            ji = pipeline.bam_join_input(engine, p, targets, filt, threads=4, chunk_bytes=chunk, ingest=ingest)
        assert ji.recs.shape[0] == len(rs)
        assert (ji.name_delta == 0) == (packed or k1 == "pages" or ingest == "heads")     # (a heads stream is read through its pages)
        ivl, cnt = engine.name_join([ji], 0.9)
        track = engine.new_track()
        engine.depth_build(ivl, cnt, 15, track)
        tr = pipeline.DepthTracks(engine, tl, track)
        for t in targets:
            assert np.array_equal(tr[t], want[t]), (chunk, ingest, t)
    # and through filter() with the environment knob, including a second (perturbed) file and the join
    rs2 = synth.perturb(rs, 92)
    p2 = str(tmp_path / "s2.bam")
    synth.write_bam_file(p2, rs2, threads=4)
    s2, o2 = synth.to_bam_stream(rs2)
    d2, h2 = oracle.bam_file_dict(s2, o2, targets, targets, *filt)
    want2 = oracle.depth_build(oracle.name_join([want_d, d2], hq | h2, 0.9), tl, 15)
    monkeypatch.setattr(pipeline, "BAM_CHUNK_BYTES", 2_000_000)
    monkeypatch.setenv("GCI_BAM_INGEST", "full")
    depths, _ = pipeline.filter([], [p, p2], prefix="st", directory=str(tmp_path), engine=engine, threads=4)
    for t in targets:
        assert np.array_equal(depths[t], want2[t]), t
    for knob, kw in (("gpu", {}), ("gpu", {"GPU_INFLATE_MAX": 0, "BAM_CHUNK_BYTES": 700_000}), ("heads", {})):
        monkeypatch.setenv("GCI_BAM_INGEST", knob)
        for k, v in kw.items():
            monkeypatch.setattr(pipeline, k, v)
        depths, _ = pipeline.filter([], [p, p2], prefix="s" + knob[0] + str(len(kw)), directory=str(tmp_path), engine=engine, threads=4)
        for t in targets:
            assert np.array_equal(depths[t], want2[t]), (knob, kw, t)


@pytest.mark.parametrize("layout,events", [("chm13", "atomic"), ("diploid", "radix")])
def test_genome_scale_layout_chm13(engine, oracle, layout, events, monkeypatch):
    """BASELINE configs[2] geometry: CHM13 (25 contigs, 3.117 Gb => 761 k tiles, > 2^31 elements in one track).
    Intervals are generated directly (5x); checks that no 32-bit index is hiding anywhere: per-contig sums equal the
    sum of trimmed interval lengths, the fused by-products equal the stand-alone kernels, three whole contigs
    (first, last = chrM, one in the middle past the 2^31st element) equal the oracle bit for bit, text sizes add up.
    "diploid" (configs[4] geometry): both haplotypes + a contig of exactly three tiles, 6.2 Gb => 1.5 M tiles, with the
    events bucketed by radix partition (tile ranges of 2048 tiles) instead of one atomic per event."""
    monkeypatch.setenv("GCI_EVENTS", events)
    contigs = synth.CHM13
    if layout == "diploid":
        contigs = tuple(("mat_" + n, l) for n, l in synth.CHM13[:24]) + (("blk", 3 * 4096),) + tuple(("pat_" + n, l) for n, l in synth.CHM13)
    lens = np.array([l for _, l in contigs], dtype=np.int64)
    assert lens.sum() > 2**31
    rng = np.random.default_rng(2025)
    n = 900_000
    c = np.searchsorted(np.cumsum(lens), rng.integers(0, lens.sum(), n), side="right").astype(np.int32)
    c[:200] = 24                                                     # some reads on chrM (16,569 bp) / the three-tile contig
    L = lens[c]
    span = np.minimum(np.clip(rng.normal(18_000, 2_500, n), 5_000, 30_000).astype(np.int64), np.maximum(L - 1, 1))
    s = (rng.random(n) * (L - span)).astype(np.int64)
    e = s + span
    e[::1000] = L[::1000]                                            # reads reaching the contig end exactly
    e[:50] = L[:50] + 100                                            # ... and beyond it: the slice [s + fl, e - fl + 1) stops at the end
    ivl = np.stack([c, s.astype(np.int32), e.astype(np.int32), np.zeros(n, np.int32)], axis=1).astype(np.int32)
    offs = engine.set_layout(lens.tolist())
    assert engine.total > 2**31 and offs[12] * 1 > 0
    d_ivl = engine.to_device(ivl)
    track = engine.new_track()
    fl = 15
    out = engine.depth_build_fused(d_ivl, None, fl, track, want_text=True, want_sums=True, issue=(-1, 0, fl))
    a = np.clip(s + fl, 0, L)
    b = np.clip(e - fl + 1, 0, L)
    want_sums = np.zeros(len(lens), dtype=np.int64)
    np.add.at(want_sums, c, np.maximum(b - a, 0))
    assert np.array_equal(out["sums"], want_sums)
    assert np.array_equal(engine.depth_sum(track), want_sums)
    tl = {nme: int(l) for nme, l in contigs}
    tr = pipeline.DepthTracks(engine, tl, track)
    names = [nme for nme, _ in contigs]
    for ci in ((0, 13, 24) if layout == "chm13" else (24, 30, len(contigs) - 1)):   # chr1, chr14 (starts beyond 2^31 elements), chrM
        assert ci != 13 or offs[ci] > 2**31
        sel = c == ci
        want = oracle.depth_build({i: (names[ci], int(s[i]), int(e[i])) for i in np.flatnonzero(sel)}, {names[ci]: tl[names[ci]]}, fl)[names[ci]]
        got = tr[names[ci]]
        assert np.array_equal(got, want), names[ci]
        t0, t1 = int(out["text_off"][ci]), int(out["text_off"][ci + 1])
        assert out["text"][t0:t1].cpu().numpy().tobytes() == oracle.depth_text_contig(want)
        tr._fresh_runs = ((-1.0, 0.0, fl), out["runs"])
        fused = pipeline.collapse_depth_range(tr, -1, 0, fl, 0)[names[ci]]
        assert fused == oracle.collapse_contig(want, -1, 0, fl, 0)
    tr.invalidate()
    standalone = pipeline.collapse_depth_range(tr, -1, 0, fl, 0)
    tr._fresh_runs = ((-1.0, 0.0, fl), out["runs"])
    assert standalone == pipeline.collapse_depth_range(tr, -1, 0, fl, 0)
    # stand-alone text kernels agree with the fused text
    text2, off2 = engine.depth_text(track)
    assert np.array_equal(off2, out["text_off"]) and torch.equal(text2, out["text"])
    # total text bytes = sum over bases of (digits + 1): recompute from the per-depth histogram of one big contig
    h = np.bincount(tr[names[1]])
    digits = np.array([len(str(v)) + 1 for v in range(h.shape[0])])
    assert int((h * digits).sum()) == int(out["text_off"][2] - out["text_off"][1])


def test_fragmented_assembly_two_types(engine, oracle, tmp_path, capsys):
    """A fragmented assembly (tests/frag_util.py) through the command line against the oracle's whole path."""
    import frag_util
    from gci_amd import cli
    from gci_amd.formats import bam as bamfmt
    from gci_amd.formats import fasta
    inp = str(tmp_path / "in")
    contigs, args = frag_util.write_inputs(inp)
    out = str(tmp_path / "out")
    cli.main(["GCI.py"] + args + ["-d", out])
    capsys.readouterr()

    def kind(names):
        d = {"paf": [], "bam": []}
        for nm in names:
            p = os.path.join(inp, nm)
            if nm.endswith(".bam"):
                stream, hdr, offs = bamfmt.read_bam(p, threads=4)
                d["bam"].append((stream, offs, list(hdr.references)))
            else:
                d["paf"].append(p)
        return d

    _, ns_bed = fasta.n_runs(os.path.join(inp, "ref.fa"))
    want = oracle.run_path(hifi=kind(["h.bam", "h.paf"]), nano=kind(["n.paf", "n.bam"]), references=[c for c, _ in contigs],
                           lengths=[l for _, l in contigs], ns_bed=ns_bed or None, threshold=1)
    got = read_outputs(out)
    assert sorted(got) == sorted(want)
    for fn in want:
        assert got[fn] == want[fn], fn
