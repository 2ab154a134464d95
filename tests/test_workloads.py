"""CPU checks of the genome-scale test infrastructure: the oracle's heads-stream mode and its exact restriction of the
join to chosen contigs (used by bench.py and tests/test_gpu_genome.py at BASELINE configs[2] size), and the
determinism of the workload generator."""
import numpy as np
import pytest

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


def test_bgzf_file_from_a_heads_stream_round_trips(tmp_path):
    """workloads.write_bgzf_from_heads (what bench.py feeds the command line at genome size): the file inflates to records whose
    heads are exactly the heads stream it was made from, SEQ / QUAL bytes of the right lengths in between, every BGZF member cut
    at a record boundary the way htslib's writer cuts (bgzf_flush_try), the same bytes for any number of worker processes."""
    from bam_util import heads_expected
    from gci_amd.formats import bam, bgzf
    rs = synth.simulate_reads((("a", 900_000), ("b", 120_000)), 8, "hifi", seed=5)
    heads, hoffs = synth.to_bam_stream(rs, heads=True)
    paths = []
    for procs in (1, 3):
        p = str(tmp_path / ("t%d.bam" % procs))
        info = workloads.write_bgzf_from_heads(p, heads, hoffs, seed=3, procs=procs, part_records=97)
        paths.append(p)
    a, b = open(paths[0], "rb").read(), open(paths[1], "rb").read()
    assert a == b and len(a) == info["bytes"] and a.endswith(bgzf.BGZF_EOF)
    stream, hdr, offs = bam.read_bam(paths[1], threads=2)
    assert stream.shape[0] == info["inflated_bytes"] and len(offs) == len(rs)
    hb, ho = heads_expected(stream, offs, hdr.first_record)
    assert hb == heads.tobytes() and np.array_equal(ho, hoffs)
    blocks = bgzf.scan_blocks(a)
    assert len(blocks) - 1 == info["members"]                              # (+ the EOF member)
    ends = np.cumsum([x[2] for x in blocks])
    rec_ends = set(np.concatenate([offs[1:], [stream.shape[0]]]).tolist()) | {int(hdr.first_record)}
    assert all(int(e) in rec_ends for e in ends) and max(x[2] for x in blocks) <= 0xFF00
    assert 1.8 < stream.shape[0] / len(a) < 3.5                            # realistic entropy: not the 100 : 1 of constant fill


def test_bgzf_member_size_walks_the_extra_subfields():
    """A BGZF member whose BC sub-field is not the first extra sub-field (valid gzip, valid BGZF): its size is found all the same."""
    import struct
    import zlib
    from gci_amd.formats import bgzf
    payload = b"hello hello hello"
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    body = c.compress(payload) + c.flush()
    extra = b"XY" + struct.pack("<H", 3) + b"abc" + b"BC" + struct.pack("<H", 2)
    total = 12 + len(extra) + 2 + len(body) + 8
    member = struct.pack("<BBBBIBBH", 0x1F, 0x8B, 8, 4, 0, 0, 0xFF, len(extra) + 2) + extra + struct.pack("<H", total - 1) + body + \
        struct.pack("<II", zlib.crc32(payload), len(payload))
    assert len(member) == total
    raw = np.frombuffer(member + bgzf.BGZF_EOF, dtype=np.uint8)
    assert bgzf.member_size(raw, 0) == total and bgzf.member_size(raw, total) == len(bgzf.BGZF_EOF)
    assert [b[:2] for b in bgzf.scan_blocks(bytes(raw))] == [(0, total), (total, 28)]
    with pytest.raises(bgzf.BGZFError):
        bgzf.member_size(raw, 3)


def test_phase_log_is_off_unless_asked_for(tmp_path):
    from gci_amd import phases
    phases.stop()
    with phases.wall("x"):
        pass
    phases.note("k", 1)
    assert not phases.on()
    phases.start()
    with phases.wall("a"):
        with phases.wall("b"):
            pass
    with phases.wall("a"):
        pass
    phases.add("n", 2)
    phases.add("n", 3)
    rep = phases.report(str(tmp_path / "p.json"))
    phases.stop()
    ages = {k: v for k, v in rep["notes"].items() if k.startswith("process_age_s_")}       # (how old the process was: two notes of the log's own)
    assert set(ages) == {"process_age_s_when_the_phase_clock_started", "process_age_s_at_the_report"}
    assert 0 <= ages["process_age_s_when_the_phase_clock_started"] <= ages["process_age_s_at_the_report"]
    assert {k: v for k, v in rep["notes"].items() if k not in ages} == {"n": 5}
    assert set(rep["wall_s"]) == {"a", "b"} and rep["gpu_s"] == {} and rep["total_s"] >= rep["wall_s"]["a"]
    import json
    assert json.load(open(tmp_path / "p.json"))["notes"] == rep["notes"]


def test_paf_written_at_a_given_size_keeps_its_twelve_columns(tmp_path):
    """workloads.write_paf_at_size: the PAF text with a cg:Z: tag behind every line so that the file has the size asked for (the
    reference's published run read a 3.6 GB and a 48 GB PAF): the same lines in the same order, their first twelve columns untouched,
    one tag per line, the size within a line's worth of the target -- and the oracle's PAF filter sees the same file."""
    from gci_amd import workloads
    rs = synth.simulate_reads((("a", 400_000), ("b", 90_000)), 12, "ont", seed=5)
    text = synth.to_paf_text(rs, seed=9)
    p = str(tmp_path / "big.paf")
    target = 40 * int(text.shape[0])
    n = workloads.write_paf_at_size(p, text, target, chunk_lines=37)
    out = open(p, "rb").read()
    assert n == len(out) and abs(len(out) - target) < 2 * target // max(1, out.count(b"\n")) + 64
    a, b = text.tobytes().split(b"\n")[:-1], out.split(b"\n")[:-1]
    assert len(a) == len(b) > 100
    for x, y in zip(a, b):
        assert y.startswith(x + b"\tcg:Z:") and y.split(b"\t")[:12] == x.split(b"\t")[:12]
    small = str(tmp_path / "small.paf")
    assert workloads.write_paf_at_size(small, text, 10) == text.shape[0] and open(small, "rb").read() == text.tobytes()
