"""Host-side product code (no GPU): interval algebra and score text against the reference's
known answers, the PAF path against the oracle, FASTA / depth-file formats, the run-closing
rules, LPT sharding."""
import gzip
import json
import os
import re

import numpy as np
import pytest

from bam_util import heads_expected
from golden_util import GOLDEN, expected, inputs, manifest
from gci_amd import score
from gci_amd.formats import depthfile, fasta


@pytest.fixture(scope="module")
def kats():
    return json.load(open(os.path.join(GOLDEN, "kats.json")))


def test_interval_algebra_kats(kats):
    for c in kats["interval_algebra"]:
        segs = [tuple(x) for x in c["segs"]]
        assert score.complement_merged_depth({"t": segs}, {"t": c["L"]}, c["fl"], c["start"], c["end"])["t"] == c["complement"]
        got = score.merge_merged_depth_bed({"t": segs}, {"t": c["L"]}, c["dp"], c["fl"], c["start"], c["end"])["t"]
        assert [list(x) for x in got] == c["merged"]
    for c in kats["compute_n50"]:
        assert score.compute_n50(c["lengths"]) == c["out"]
    for c in kats["score_repr"]:
        assert repr(score.gci_score(c["obs_n50"], c["exp_n50"], c["obs_n"], c["exp_n"])) == c["out"]


def test_index_text_reproduces_mh63_gci():
    d = os.path.join(GOLDEN, "MH63")
    bed = {}
    for line in open(os.path.join(d, "MH63.0.depth.bed")):
        t, s, e = line.split("\t")
        bed.setdefault(t, []).append((int(s), int(e)))
    want = open(os.path.join(d, "MH63.gci")).read()
    tl = {}
    for line in want.splitlines()[2:14]:
        f = line.split("\t")
        tl[f[0]] = int(f[1])
        bed.setdefault(f[0], [])
    bed = {t: bed[t] for t in tl}
    assert score.index_text(tl, [bed], ["HiFi"]) == want


def test_issue_closing_rules_match_reference_kats(kats):
    """pipeline._issues_from_runs turns raw maximal runs (what K8 emits) into the reference's
    intervals: checked here with runs computed by numpy from the KAT depth vectors."""
    from gci_amd.pipeline import _issues_from_runs, _slice_bound
    for c in kats["collapse_depth_range"]:
        d = np.array(c["depth"], dtype=np.int64)
        L, fl = d.shape[0], c["fl"]
        a, b = _slice_bound(fl, L), _slice_bound(L - fl, L)
        b = max(a, b)
        g = (d[a:b] > c["lo"]) & (d[a:b] <= c["hi"])
        e = np.diff(np.concatenate(([0], g.astype(np.int8), [0])))
        runs = np.stack([np.flatnonzero(e == 1), np.flatnonzero(e == -1)], axis=1)
        assert _issues_from_runs(runs, b - a, L, fl, c["sp"]) == [tuple(x) for x in c["out"]], c


def test_paf_filter_matches_oracle(oracle, tmp_path):
    from gci_amd.pipeline import paf_filter
    case = "c5_two_type"
    m = manifest(case)
    pafs = [p for p in inputs(case, m["hifi"] + m["nano"]) if p.endswith(".paf")]
    targets = ["mat_chr1", "pat_chr1", "mat_chr2"]
    for sel in (targets, targets[:2]):
        for args in ((30, 50, 0.9), (10, 60, 0.99)):
            got = paf_filter(pafs, sel, *args)          # two files: exercises the un-reset block table
            want = oracle.paf_filter(pafs, sel, *args)
            assert got[1] == want[1]
            assert [list(d.items()) for d in got[0]] == [list(d.items()) for d in want[0]]
    # a hand-made file: ties between targets broken by name, touching blocks merge, longest block wins
    p = tmp_path / "t.paf"
    rows = [("q1", 1000, 0, 400, "+", "tB", 9000, 100, 500, 400, 400, 60), ("q1", 1000, 400, 800, "+", "tB", 9000, 500, 900, 400, 400, 60),
            ("q1", 1000, 0, 800, "+", "tA", 9000, 2000, 2800, 800, 800, 60), ("q2", 500, 0, 100, "+", "tA", 9000, 10, 110, 95, 100, 5),
            ("q3", 500, 0, 100, "+", "tZ", 9000, 10, 110, 100, 100, 60), ("q4", 500, 0, 200, "-", "tA", 9000, 50, 250, 199, 200, 49),
            ("q4", 500, 300, 350, "-", "tA", 9000, 5000, 5050, 50, 50, 49)]
    p.write_text("".join("\t".join(map(str, r)) + "\n" for r in rows))
    got = paf_filter([str(p)], ["tA", "tB"], 30, 50, 0.9)
    want = oracle.paf_filter([str(p)], ["tA", "tB"], 30, 50, 0.9)
    assert got == want
    assert got[0][0]["q1"] == ("tB", 100, 900, 1000) and "q2" not in got[0][0] and "q3" not in got[0][0]
    assert got[0][0]["q4"] == ("tA", 50, 250, 500) and got[1] == {"q1"}


def test_native_paf_filter_randomised(oracle, tmp_path):
    """gci_paf_filter (native) against the plain-Python statement and the oracle: many queries over few targets so
    that rank ties and touching / nested blocks happen, three files (the block table is never reset), CRLF line ends,
    extra columns, names the targets list does not hold; and the lines the reference would raise on."""
    from gci_amd.pipeline import paf_filter
    from paf_ref import paf_filter_py
    from gci_amd._lib import GciError
    rng = np.random.default_rng(11)
    targets = ["t%d" % i for i in range(6)]
    paths = []
    for f in range(3):
        rows = []
        for _ in range(1500):
            q = "read_%d" % int(rng.integers(0, 400))
            qlen = int(rng.choice([1000, 2000, 5000]))
            qs = int(rng.integers(0, qlen // 100)) * 50
            qe = min(qlen, qs + int(rng.integers(1, 20)) * 50)
            t = str(rng.choice(targets + ["other"]))
            ts = int(rng.integers(0, 100)) * 100
            te = ts + (qe - qs)
            aln = qe - qs
            nm = int(aln * rng.choice([0.85, 0.9, 0.95, 1.0]))
            ts_s = str(ts) if len(str(ts)) < 2 or rng.random() > 0.1 else str(ts)[0] + "_" + str(ts)[1:]      # int('1_200') is 1200
            rows.append("\t".join(map(str, (q, qlen, qs, qe, "+-"[int(rng.integers(0, 2))], t, 100000, ts_s, te, nm, aln,
                                            int(rng.choice([0, 29, 30, 49, 50, 60])), "tp:A:P", "cm:i:5"))))
        p = tmp_path / ("f%d.paf" % f)
        p.write_bytes(("\r\n" if f == 1 else "\n").join(rows).encode() + (b"" if f == 2 else b"\n"))
        paths.append(str(p))
    for sel in (targets, targets[1:4]):
        for args in ((30, 50, 0.9), (0, 60, 0.0), (50, 30, 0.95)):
            got, py, want = paf_filter(paths, sel, *args), paf_filter_py(paths, sel, *args), oracle.paf_filter(paths, sel, *args)
            assert got[1] == py[1] == want[1]
            for a, b, c in zip(got[0], py[0], want[0]):
                assert list(a.items()) == list(b.items()) == list(c.items())
            assert len(got[0][2]) >= len(got[0][0]) > 20
    # files large enough to be cut into many byte ranges (one per thread and more): same dicts in the same order as the
    # line-by-line statement, whatever the thread count; and the first offending line is reported, not any
    big = []
    for f in range(2):
        rows = []
        for i in range(12_000):
            q = "m64011_190830_220126/%d/ccs" % int(rng.integers(0, 7000))
            qlen = int(rng.choice([9000, 15000, 21000]))
            qs = int(rng.integers(0, 50)) * 100
            qe = min(qlen, qs + int(rng.integers(10, 150)) * 100)
            t = str(rng.choice(targets + ["other"]))
            ts = int(rng.integers(0, 900)) * 100
            aln = qe - qs
            rows.append("\t".join(map(str, (q, qlen, qs, qe, "+", t, 100000, ts, ts + aln, int(aln * rng.choice([0.88, 0.93, 0.99])), aln,
                                            int(rng.choice([0, 29, 30, 49, 50, 60])), "tp:A:P"))))
        p = tmp_path / ("big%d.paf" % f)
        p.write_bytes(("\r\n" if f else "\n").join(rows).encode() + b"\n")
        assert p.stat().st_size > 600_000
        big.append(str(p))
    want_big = paf_filter_py(big, targets, 30, 50, 0.9)
    from gci_amd import hostio
    for th in (1, 3, 8):
        native = hostio.paf_filter(big, targets, 30, 50, 0.9, threads=th)
        ref1 = hostio.paf_filter(big, targets, 30, 50, 0.9, threads=1)
        for (r1, n1, o1), (r2, n2, o2) in zip(native, ref1):
            assert np.array_equal(r1, r2) and np.array_equal(n1, n2) and np.array_equal(o1, o2)
    got_big = paf_filter(big, targets, 30, 50, 0.9)
    assert got_big[1] == want_big[1]
    for a, b in zip(got_big[0], want_big[0]):
        assert list(a.items()) == list(b.items())
    lines = open(big[0], "rb").read().split(b"\n")
    lines[7000] = lines[7000].replace(b"\t", b" ", 20)                      # no tabs left: IndexError in the reference
    lines[9000] = b"x"
    (tmp_path / "bigbad.paf").write_bytes(b"\n".join(lines))
    for th in (1, 8):
        with pytest.raises(GciError) as err:
            hostio.paf_filter([str(tmp_path / "bigbad.paf")], targets, 30, 50, 0.9, threads=th)
        assert "line 7001" in str(err.value)
    bad = tmp_path / "bad.paf"
    for text in ("q\t100\t0\t50\t+\n", "q\t100\t0\t50\t+\tt0\t1000\t0\t50\t50\t0\t60\n",
                 "q\t100\t0\tx\t+\tt0\t1000\t0\t50\t50\t50\t60\n", "q\t100\t0\t50\t+\tt0\t1000\t0\t50\t50\t50\n"):
        bad.write_text(text)
        with pytest.raises(GciError):
            paf_filter([str(bad)], ["t0"], 30, 50, 0.9)
        with pytest.raises((IndexError, ValueError, ZeroDivisionError)):
            paf_filter_py([str(bad)], ["t0"], 30, 50, 0.9)
    bad.write_text("q\t100\t0\tx\t+\tother\t1000\t0\t50\t50\t50\t60\n")      # not a selected target: never parsed
    assert paf_filter([str(bad)], ["t0"], 30, 50, 0.9) == paf_filter_py([str(bad)], ["t0"], 30, 50, 0.9) == ([{}], set())


def test_fasta_n_runs_match_regex(tmp_path):
    p = tmp_path / "r.fa"
    p.write_bytes(b">c1 desc here\nACGTNNNN\nNNacgt\nnNnN\n>c2\nNNNN\n>c3\tx\nACGT\r\nAC GT\n>c4\n\nN\n")
    ids, runs = fasta.n_runs(str(p))
    assert ids == ["c1", "c2", "c3", "c4"]
    seqs = {"c1": "ACGTNNNNNNacgtnNnN", "c2": "NNNN", "c3": "ACGTACGT", "c4": "N"}
    want = {k: [(m.start(), m.end()) for m in re.finditer(r"(?i)N+", v)] for k, v in seqs.items()}
    assert runs == {k: v for k, v in want.items() if v}
    assert fasta.record_ids(str(p)) == ids
    # the native title index (gci_fasta_titles) sees the same records as the numpy twin
    for body in (p.read_bytes(), b"", b"ACGT\n", b">only_a_title", b">a\n>b x\nAC>GT\n>\nNN", b"\n>late\nN\n",
                 b">big\n" + b"ACGTN>" * 400_000 + b"\n>tail y z\nNN\n"):
        q = tmp_path / "t.fa"
        q.write_bytes(body)
        buf, spans = fasta.indexed(str(q))
        assert spans == fasta.record_spans(buf)
        assert fasta.record_ids_indexed(str(q)) == fasta.record_ids(str(q))
        fasta._INDEXED = None
    # the golden case's FASTA: gaps.bed written by the reference
    case = "c5_two_type"
    _, r2 = fasta.n_runs(os.path.join(GOLDEN, case, "inputs", "ref.fa"))
    text = "".join(f"{t}\t{a}\t{b}\n" for t, segs in r2.items() for a, b in segs)
    assert text.encode() == expected(case)["GCI.gaps.bed"]


def test_depth_file_round_trip(tmp_path, oracle):
    d = {"a": np.array([0, 1, 22, 333, 4444, 0], dtype=np.int64), "b": np.zeros(3, dtype=np.int64)}
    text = oracle.depth_text(d)
    path = str(tmp_path / "x.depth.gz")
    body = {"a": text[3:text.index(b">b")], "b": text[text.index(b">b") + 3:]}
    depthfile.write_depth_gz(path, [(k, memoryview(v)) for k, v in body.items()], threads=2)
    assert gzip.open(path, "rb").read() == text == oracle.depth_text_py(d)
    back = depthfile.read_depth_gz(path)
    assert list(back) == ["a", "b"] and all(np.array_equal(back[k], d[k]) for k in d)
    assert depthfile.sha256_of_text(path) == __import__("hashlib").sha256(text).hexdigest()


def test_lpt_sharding():
    from gci_amd import shard, synth
    lens = [l for _, l in synth.CHM13]
    owner = shard.lpt_assign(lens, 8)
    load = [sum(l for l, o in zip(lens, owner) if o == r) for r in range(8)]
    assert sum(load) == sum(lens) and max(load) <= 1.05 * sum(lens) / 8 + max(lens) * 0.0 + 2e7
    m, mine = shard.contig_map_for(owner, 3)
    assert [m[c] for c in mine] == list(range(len(mine))) and (m >= 0).sum() == len(mine)
    assert shard.record_slices(10, 4) == [(0, 3), (3, 6), (6, 9), (9, 10)] and shard.record_slices(0, 2) == [(0, 0), (0, 0)]
    assert shard.lpt_assign([5], 4) == [0]


def test_native_host_io_matches_python_twins(tmp_path):
    """gci_bgzf_inflate / gci_bam_record_offsets / gci_gzip_members (host_io.cpp) against the pure-Python formats."""
    import gzip as _gz
    from gci_amd import hostio, synth
    from gci_amd.formats import bam, bgzf
    rs = synth.simulate_reads((("a", 400_000), ("b", 90_000)), 10, "hifi", seed=5)
    stream, offs = synth.to_bam_stream(rs)
    p = str(tmp_path / "x.bam")
    bam.write_bam_stream(p, stream, level=1, threads=4)
    raw = open(p, "rb").read()
    a = bgzf.decompress(raw, threads=2, check_crc=True)
    b = hostio.bgzf_inflate(raw, threads=4, check_crc=True)
    assert np.array_equal(a, b) and np.array_equal(b, stream)
    assert np.array_equal(hostio.read_bgzf_file(p, threads=3), stream)
    o2, first = hostio.bam_record_offsets(b)
    assert first == bam.parse_header(stream).first_record and np.array_equal(o2, offs)
    bad = bytearray(raw); bad[200] ^= 0xFF                                  # a flipped byte inside the first member
    with pytest.raises(Exception):
        hostio.bgzf_inflate(bytes(bad), threads=2, check_crc=True)
    with pytest.raises(Exception):
        hostio.bgzf_inflate(raw[:-41], threads=2)                         # truncated member
    with pytest.raises(Exception):
        hostio.bam_record_offsets(stream[:-7])
    text = b"".join(b"%d\n" % (i % 977) for i in range(300_000))
    z = hostio.gzip_members(text, threads=4, chunk=100_000)
    assert _gz.decompress(z) == text and z.count(b"\x1f\x8b\x08") >= len(text) // 100_000
    assert hostio.gzip_members(b"") == b""


@pytest.mark.parametrize("kind,seed", [("hifi", 5), ("ont", 6)])
def test_bam_heads_stream(tmp_path, kind, seed):
    """gci_bam_heads (host_io.cpp): BGZF bytes -> header + records without SEQ / QUAL, for group sizes that cut the
    header and the records at every phase; contradictory records are marked, truncated / corrupt input is refused."""
    from gci_amd import hostio, synth
    from gci_amd._lib import GciError
    from gci_amd.formats import bam
    many = tuple(("contig_with_a_long_name_%04d" % i, 30_000 + i) for i in range(3000))   # header > 64 KiB
    rs = synth.simulate_reads((("a", 300_000), ("b", 90_000)) if kind == "hifi" else (("a", 200_000),), 8, kind, seed=seed)
    stream, offs = synth.to_bam_stream(rs)
    stream = stream.copy()
    first = bam.parse_header(stream).first_record
    # two records that contradict their block_size: negative l_seq, and SEQ + QUAL longer than the record
    stream[int(offs[3]) + 20:int(offs[3]) + 24] = np.frombuffer(np.int32(-7).tobytes(), dtype=np.uint8)
    stream[int(offs[9]) + 20:int(offs[9]) + 24] = np.frombuffer(np.int32(1 << 28).tobytes(), dtype=np.uint8)
    if kind == "hifi":                                                   # and a long header in front
        hdr = np.frombuffer(bam.encode_header(["a", "b"] + [n for n, _ in many], [300_000, 90_000] + [l for _, l in many]),
                            dtype=np.uint8)
        assert hdr.shape[0] > 2 * 65_536
        offs = offs - np.uint64(first) + np.uint64(hdr.shape[0])
        stream = np.concatenate([hdr, stream[first:]])
        first = int(hdr.shape[0])
    p = str(tmp_path / "h.bam")
    bam.write_bam_stream(p, stream, level=1, threads=4)
    raw = np.fromfile(p, dtype=np.uint8)
    want, want_offs = heads_expected(stream, offs, first)
    for group in (0, 1 << 20, 70_000, 65_536, 1):
        with hostio.bam_heads(raw, threads=4, group_bytes=group, check_crc=True) as h:
            assert h.first_record == first
            assert np.array_equal(h.offsets, want_offs), group
            assert h.stream.tobytes() == want, group
    assert len(want) - first < (stream.shape[0] - first) // (20 if kind == "hifi" else 2)
    with pytest.raises(GciError):
        hostio.bam_heads(raw[:-41], threads=3)                             # truncated last member
    cut = str(tmp_path / "cut.bam")
    bam.write_bam_stream(cut, stream[:-9], level=1, threads=2)             # truncated last record
    with pytest.raises(GciError):
        hostio.bam_heads(np.fromfile(cut, dtype=np.uint8), threads=3)
    bad = raw.copy(); bad[len(bad) // 2] ^= 0xFF
    with pytest.raises(GciError):
        hostio.bam_heads(bad, threads=3, check_crc=True)
    short = stream.copy()
    short[int(offs[5]):int(offs[5]) + 4] = np.frombuffer(np.int32(31).tobytes(), dtype=np.uint8)   # block_size < 32
    bam.write_bam_stream(cut, short, level=1, threads=2)
    with pytest.raises(GciError):
        hostio.bam_heads(np.fromfile(cut, dtype=np.uint8), threads=3)
    with pytest.raises(GciError):
        hostio.bam_heads(np.zeros(0, dtype=np.uint8))                      # no header at all


def test_paf_line_error_replay_gives_pythons_exception(tmp_path):
    """pipeline._paf_line_exception (error path only): the exception GCI.py:217-229 dies with on the first offending line, in the
    order the reference touches the columns -- type and message are Python's own."""
    from gci_amd import pipeline
    good = "q\t100\t0\t50\t+\tt0\t1000\t0\t50\t50\t50\t60\n"
    cases = [("q\t100\t0\t50\t+\tt0\t1000\t0\t50\tfifty\t50\t60\n", ValueError, "invalid literal for int() with base 10: 'fifty'"),
             ("q\t100\t0\n", IndexError, "list index out of range"),
             ("q\t100\t0\t50\t+\tt0\t1000\t0\n", IndexError, "list index out of range"),
             ("q\tx\t0\t50\t+\tt0\t1000\n", ValueError, "invalid literal for int() with base 10: 'x'"),      # int(paf[1]) before paf[7]
             ("q\t100\t0\t50\t+\tt0\t1000\t0\t50\t50\t0\t60\n", ZeroDivisionError, "division by zero"),
             ("\n", IndexError, "list index out of range"),
             ("q\tx\t0\t50\t+\tother\t1000\n", None, None)]                                                  # not a selected contig: skipped unread
    for k, (bad, typ, msg) in enumerate(cases):
        p = tmp_path / ("e%d.paf" % k)
        p.write_text(good + bad + good)
        exc = pipeline._paf_line_exception([str(p)], ["t0"])
        if typ is None:
            assert exc is None
        else:
            assert type(exc) is typ and str(exc) == msg, (k, exc)
    # files in order: the first file's bad line wins
    exc = pipeline._paf_line_exception([str(tmp_path / "e1.paf"), str(tmp_path / "e0.paf")], ["t0"])
    assert type(exc) is IndexError


def test_threaded_member_table_equals_the_serial_walk(tmp_path, monkeypatch):
    """gci_bgzf_table_build: ranges walked by threads and stitched give the table of the serial BSIZE chain -- also when member DATA
    is full of chains of well-formed BGZF members (stored blocks holding BGZF bytes: every range then starts on a false member and
    the stitching has to notice), and a damaged header is reported as by the serial walk."""
    import zlib
    from gci_amd import hostio
    from gci_amd.formats import bgzf
    rng = np.random.default_rng(3)

    def member(payload: bytes, level: int) -> bytes:
        c = zlib.compressobj(level, zlib.DEFLATED, -15)
        body = c.compress(payload) + c.flush()
        size = 18 + len(body) + 8
        return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + (size - 1).to_bytes(2, "little") + body +
                zlib.crc32(payload).to_bytes(4, "little") + len(payload).to_bytes(4, "little"))

    plain = b"".join(member(rng.integers(0, 4, int(rng.integers(1, 60000)), dtype=np.uint8).tobytes(), 1) for _ in range(300))
    fake = bgzf.BGZF_EOF * 2000                                   # 2000 empty members back to back: 56 000 bytes of "headers"
    nested = b"".join(member(fake[:int(rng.integers(28, 56000))], 0) for _ in range(120))     # level 0: stored, the bytes as they are
    for raw_b in (plain, nested, plain + nested + plain + bgzf.BGZF_EOF):
        raw = np.frombuffer(raw_b, dtype=np.uint8)
        monkeypatch.delenv("GCI_BGZF_RANGE", raising=False)
        want_pos, want_isz = hostio.bgzf_blocks(raw, threads=1)
        assert int(want_pos[-1]) == raw.shape[0] and want_isz.shape[0] == want_pos.shape[0] - 1
        for rng_bytes, threads in ((4096, 8), (70000, 3), (1 << 20, 16)):
            monkeypatch.setenv("GCI_BGZF_RANGE", str(rng_bytes))
            pos, isz = hostio.bgzf_blocks(raw, threads=threads)
            assert np.array_equal(pos, want_pos) and np.array_equal(isz, want_isz), (rng_bytes, threads)
            # the table of the beginning of the string: the members that start in front of `limit`, whatever the ranges cut
            for limit in (1, raw.shape[0] // 3, raw.shape[0] - 1):
                ppos, pisz = hostio.bgzf_blocks(raw, threads=threads, limit=limit)
                k = int(np.searchsorted(want_pos[:-1], limit, side="left"))
                assert np.array_equal(ppos, want_pos[:k + 1]) and np.array_equal(pisz, want_isz[:k]), ("prefix", rng_bytes, threads, limit)
            # the same table read through a descriptor (gci_bgzf_table_build_fd: pread instead of the mapping)
            (tmp_path / "t.bgzf").write_bytes(raw_b)
            pos, isz = hostio.bgzf_blocks_file(str(tmp_path / "t.bgzf"), threads=threads)
            assert np.array_equal(pos, want_pos) and np.array_equal(isz, want_isz), ("fd", rng_bytes, threads)
    # a member whose extra field holds another subfield in front of BC (RFC 1952 allows it; htslib never writes it)
    def member_x(payload: bytes) -> bytes:
        c = zlib.compressobj(1, zlib.DEFLATED, -15)
        body = c.compress(payload) + c.flush()
        size = 12 + 10 + len(body) + 8
        return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x0a\x00XY\x00\x00BC\x02\x00" + (size - 1).to_bytes(2, "little") + body +
                zlib.crc32(payload).to_bytes(4, "little") + len(payload).to_bytes(4, "little"))
    odd = b"".join(member_x(b"ACGT" * int(rng.integers(1, 9000))) for _ in range(40)) + bgzf.BGZF_EOF
    (tmp_path / "odd.bgzf").write_bytes(odd)
    monkeypatch.setenv("GCI_BGZF_RANGE", "4096")
    want = hostio.bgzf_blocks(np.frombuffer(odd, dtype=np.uint8), threads=1)
    assert want[1].shape[0] == 41
    for threads in (1, 4):
        got = hostio.bgzf_blocks_file(str(tmp_path / "odd.bgzf"), threads=threads)
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    # damage: a header byte in the middle -> the same refusal
    bad = bytearray(plain)
    k = int(hostio.bgzf_blocks(np.frombuffer(plain, dtype=np.uint8), threads=1)[0][150])
    bad[k + 1] = 0
    monkeypatch.setenv("GCI_BGZF_RANGE", "4096")
    (tmp_path / "bad.bgzf").write_bytes(bytes(bad))
    (tmp_path / "cut.bgzf").write_bytes(plain[:-5])
    for threads in (1, 8):
        with pytest.raises(Exception):
            hostio.bgzf_blocks(np.frombuffer(bytes(bad), dtype=np.uint8), threads=threads)
        with pytest.raises(Exception):
            hostio.bgzf_blocks_file(str(tmp_path / "bad.bgzf"), threads=threads)
        with pytest.raises(Exception):
            hostio.bgzf_blocks_file(str(tmp_path / "cut.bgzf"), threads=threads)
