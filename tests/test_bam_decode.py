"""R1 at the pysam boundary is "parity unpinned" by the reference, so it is anchored
here on hand-assembled BAM records: bytes written out literally, expected values derived by hand
from the SAM/BAM specification and htslib's documented behaviour."""
import struct

import numpy as np
import pytest

from gci_amd.formats import bam, bgzf


def rec_bytes(ref_id, pos, name, mapq, flag, cigar_words, l_seq, aux, n_cigar=None):
    nm = name + b"\x00"
    n_cigar = len(cigar_words) if n_cigar is None else n_cigar
    body = struct.pack("<iiBBHHHiiii", ref_id, pos, len(nm), mapq, 4680, n_cigar, flag, l_seq, -1, -1, 0)
    body += nm + b"".join(struct.pack("<I", w) for w in cigar_words) + b"\x11" * ((l_seq + 1) // 2) + b"\xff" * l_seq + aux
    return struct.pack("<i", len(body)) + body


def op(length, code):
    return (length << 4) | "MIDNSHP=X".index(code)


def stream_of(records, refs=(("chr1", 100000), ("chr2", 5000))):
    hdr = bam.encode_header([r for r, _ in refs], [l for _, l in refs])
    data = hdr + b"".join(records)
    s = np.frombuffer(data, dtype=np.uint8).copy()
    h = bam.parse_header(s)
    return s, bam.record_offsets(s, h.first_record), h


HAND = [
    # (description, record, expected (pass, hq, start, end, qlen) or error name)
    ("plain 100M NM:C:3", rec_bytes(0, 1000, b"r1", 60, 0, [op(100, "M")], 100, b"NMC\x03"), (1, 1, 1000, 1100, 100)),
    ("=/X/I/D mix, NM last after Z and B tags",
     rec_bytes(0, 2000, b"r2", 40, 16, [op(50, "="), op(1, "X"), op(2, "I"), op(47, "="), op(3, "D"), op(10, "=")], 110,
               b"MDZ50A47^ACG10\x00" + b"mlBC\x03\x00\x00\x00\x01\x02\x03" + b"NMS\x06\x00"),
     (1, 0, 2000, 2111, 110)),
    ("10 % soft clip passes exactly (S/(M+I+S) = 10/100 <= 0.1)",
     rec_bytes(0, 10, b"r3", 60, 0, [op(10, "S"), op(90, "M")], 100, b"NMi\x00\x00\x00\x00"), (1, 1, 10, 100, 100)),
    ("11 % soft clip fails", rec_bytes(0, 10, b"r4", 60, 0, [op(11, "S"), op(89, "M")], 100, b"NMC\x00"), (0, 0, 0, 0, 0)),
    ("identity 0.9 exactly passes: 100M NM=10", rec_bytes(0, 10, b"r5", 60, 0, [op(100, "M")], 100, b"NMC\x0a"),
     (1, 1, 10, 110, 100)),
    ("identity below 0.9 fails: 100M NM=11", rec_bytes(0, 10, b"r6", 60, 0, [op(100, "M")], 100, b"NMC\x0b"), (0, 0, 0, 0, 0)),
    ("hard clips and N: H ignored, N consumes reference",
     rec_bytes(1, 100, b"r7", 55, 0, [op(20, "H"), op(30, "M"), op(500, "N"), op(30, "M"), op(5, "H")], 60, b"NMC\x00"),
     (1, 1, 100, 660, 60)),
    ("secondary", rec_bytes(0, 10, b"r8", 60, 0x100, [op(100, "M")], 0, b"NMC\x00"), (0, 0, 0, 0, 0)),
    ("supplementary", rec_bytes(0, 10, b"r9", 60, 0x800, [op(100, "M")], 100, b"NMC\x00"), (0, 0, 0, 0, 0)),
    ("unmapped but placed", rec_bytes(0, 10, b"r10", 0, 0x4, [], 100, b""), (0, 0, 0, 0, 0)),
    ("MAPQ 29 < 30", rec_bytes(0, 10, b"r11", 29, 0, [op(100, "M")], 100, b"NMC\x00"), (0, 0, 0, 0, 0)),
    ("MAPQ 30 passes, not high quality", rec_bytes(0, 10, b"r12", 30, 0, [op(100, "M")], 100, b"NMC\x00"),
     (1, 0, 10, 110, 100)),
    ("long CIGAR parked in CG:B,I (placeholder 100S100N)",
     rec_bytes(0, 500, b"r13", 60, 0, [op(100, "S"), op(100, "N")], 100,
               b"NMC\x02" + b"CGBI" + struct.pack("<i", 3) + struct.pack("<3I", op(60, "M"), op(2, "D"), op(40, "M"))),
     (1, 1, 500, 602, 100)),
    ("placeholder-looking CIGAR without CG tag is taken literally: all clipped -> clip test fails",
     rec_bytes(0, 500, b"r14", 60, 0, [op(100, "S"), op(100, "N")], 100, b"NMC\x00"), (0, 0, 0, 0, 0)),
    ("negative NM type c", rec_bytes(0, 10, b"r15", 60, 0, [op(100, "M")], 100, b"NMc\xff"), (1, 1, 10, 110, 100)),
    ("unselected contig", rec_bytes(1, 10, b"r16", 60, 0, [op(100, "M")], 100, b"NMC\x00"), None),
]


def test_hand_assembled_records(oracle):
    recs = [r for _, r, _ in HAND]
    s, offs, h = stream_of(recs)
    # select both contigs except for the last case, checked separately
    a = oracle.bam_filter_arrays(s, offs, np.array([0, 1], dtype=np.int32), 30, 50, 0.1, 0.9)
    for i, (desc, _, want) in enumerate(HAND):
        if want is None:
            continue
        got = (int(a["passed"][i]), int(a["hq"][i]), int(a["start"][i]), int(a["end"][i]), int(a["qlen"][i]))
        assert got == want, desc
    b = oracle.bam_filter_arrays(s, offs, np.array([0, -1], dtype=np.int32), 30, 50, 0.1, 0.9)
    assert b["passed"][len(HAND) - 1] == 0 and b["passed"][6] == 0
    names = oracle.read_names(s, a["name_off"], a["name_len"])
    assert names[:3] == ["r1", "r2", "r3"]


def test_python_twin_agrees_with_c(oracle):
    s, offs, h = stream_of([r for _, r, _ in HAND])
    a = oracle.bam_filter_arrays(s, offs, np.array([0, 1], dtype=np.int32), 30, 50, 0.1, 0.9)
    for i, o in enumerate(offs):
        rec = bam.decode_record(s, o)
        r = oracle.bam_filter_record_py(rec, h.references, list(h.references), 30, 50, 0.1, 0.9)
        assert (r is not None) == bool(a["passed"][i]), HAND[i][0]
        if r is not None:
            assert r[1] == (h.references[rec.ref_id], int(a["start"][i]), int(a["end"][i]), int(a["qlen"][i]))
            assert r[2] == bool(a["hq"][i])


def test_reference_errors_are_reported(oracle):
    s, offs, _ = stream_of([rec_bytes(0, 10, b"ok", 60, 0, [op(100, "M")], 100, b"NMC\x00"),
                            rec_bytes(0, 10, b"nonm", 60, 0, [op(100, "M")], 100, b"ASi\x00\x00\x00\x00")])
    with pytest.raises(oracle.OracleRecordError) as e:
        oracle.bam_filter_arrays(s, offs, np.array([0, 1], dtype=np.int32), 30, 50, 0.1, 0.9)
    assert e.value.status == oracle.E_NO_NM and e.value.rec == 1
    s, offs, _ = stream_of([rec_bytes(0, 10, b"zd", 60, 0, [op(100, "H")], 0, b"NMC\x00")])
    with pytest.raises(oracle.OracleRecordError) as e:
        oracle.bam_filter_arrays(s, offs, np.array([0, 1], dtype=np.int32), 30, 50, 0.1, 0.9)
    assert e.value.status == oracle.E_ZERO_DIV


def test_bgzf_and_container_round_trip(tmp_path):
    recs = [r for _, r, _ in HAND]
    p = str(tmp_path / "h.bam")
    bam.write_bam(p, ["chr1", "chr2"], [100000, 5000], recs, level=6)
    s, h, offs = bam.read_bam(p, threads=2)
    assert h.references == ("chr1", "chr2") and h.lengths == (100000, 5000) and len(offs) == len(recs)
    assert bam.read_header(p).references == h.references
    raw = open(p, "rb").read()
    assert raw.endswith(bgzf.BGZF_EOF)
    big = np.random.default_rng(0).integers(0, 255, 300_000).astype(np.uint8).tobytes()
    assert bgzf.decompress(bgzf.compress(big, threads=3), threads=3, check_crc=True).tobytes() == big
    r = bam.decode_record(s, offs[12])
    assert len(r.cigar) == 3 and "CG" not in r.aux and r.n_cigar_field == 2


def test_bai_written_and_read_back(tmp_path):
    """write_bai / read_bai (SAM spec 5.2): per reference the virtual offsets of its first record and of the end of its
    last one -- what the contig-sharded ingestion uses to inflate only a rank's share of a file."""
    from gci_amd import synth
    from gci_amd.formats import bgzf
    contigs = (("a", 300_000), ("nothing_here", 40_000), ("b", 100_000), ("c", 5_000))
    rs = synth.simulate_reads((("a", 300_000), ("b", 100_000), ("c", 5_000)), 15, "hifi", seed=4)
    rs.ref_id = np.where(rs.ref_id >= 1, rs.ref_id + 1, rs.ref_id).astype(np.int32)        # contig 1 has no records
    rs.contigs = contigs
    p = str(tmp_path / "x.bam")
    synth.write_bam_file(p, rs, level=6, threads=2)
    idx = bam.read_bai(p + ".bai")
    raw = open(p, "rb").read()
    stream = bgzf.decompress(raw)
    hdr = bam.parse_header(stream)
    offs = bam.record_offsets(stream, hdr.first_record)
    blocks = bgzf.scan_blocks(raw)
    ustart = np.cumsum([0] + [b[2] for b in blocks])
    cpos = [b[0] for b in blocks]
    to_u = lambda v: int(ustart[cpos.index(v >> 16)]) + (v & 0xFFFF)        # noqa: E731
    ref = np.array([int(stream[o + 4:o + 8].view(np.int32)[0]) for o in offs.tolist()])
    assert len(idx) == 4 and idx[1] is None
    for r in (0, 2, 3):
        ii = np.flatnonzero(ref == r)
        assert to_u(idx[r][0]) == offs[ii[0]]
        assert to_u(idx[r][1]) == (offs[ii[-1] + 1] if ii[-1] + 1 < len(offs) else stream.shape[0])
    assert bam.read_bai(str(tmp_path / "missing.bai")) is None
    bam.write_bai(str(tmp_path / "empty.bai"), 3)
    assert bam.read_bai(str(tmp_path / "empty.bai")) == [None, None, None]
