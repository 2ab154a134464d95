"""Record pages (include/gci_hip.h, gci_bam_pages_*): the device converter against the plain-Python statement of the format
(tests/pages_ref.py), byte for byte, on inputs that reach every record kind -- inline, CIGAR in the blob, whole record in
the blob, malformed -- and every page size; empty input."""
import numpy as np
import pytest
import torch

from gci_amd import synth
from gci_amd.formats import bam
import pages_ref
from bam_util import heads_expected

pytestmark = pytest.mark.gpu


def _check(engine, stream, offs, has_seq, page_bytes):
    want, n_pages, blob_off = pages_ref.build_pages(stream, offs, has_seq, page_bytes)
    pg = engine.bam_pages(engine.to_device(stream), engine.to_device(np.asarray(offs, dtype=np.uint64)), has_seq, page_bytes)
    assert (pg.n_pages, pg.blob_off, pg.n_rec) == (n_pages, blob_off, len(offs))
    got = pg.buf.cpu().numpy()
    assert got.shape[0] == want.shape[0]
    if not np.array_equal(got, want):
        bad = int(np.flatnonzero(got != want)[0])
        raise AssertionError("pages differ at byte %d (page %d + %d; blob at %d)" % (bad, bad // page_bytes, bad % page_bytes, blob_off))
    return pg


@pytest.mark.parametrize("kind,page_bytes", [("hifi", 8192), ("hifi", 24576), ("ont", 16384), ("ont", 32768)])
def test_pages_equal_the_python_statement(engine, kind, page_bytes):
    rs = synth.simulate_reads((("a", 300_000), ("b", 60_000)), 12 if kind == "hifi" else 40, kind, seed=5, long_cigar_frac=0.02 if kind == "ont" else 0.0)
    stream, offs = synth.to_bam_stream(rs)
    pg = _check(engine, stream, offs, True, page_bytes)
    assert pg.n_pages >= (2 if kind == "hifi" else 1)
    h_bytes, h_offs = heads_expected(stream, offs, bam.parse_header(stream).first_record)
    _check(engine, np.frombuffer(h_bytes, dtype=np.uint8), h_offs, False, page_bytes)


def test_pages_of_odd_records(engine):
    """Names of every length, CIGARs around the inline limit, kilobytes of tags, truncated and inconsistent records."""
    rng = np.random.default_rng(3)
    recs = []
    for i in range(600):
        n_ops = int(rng.choice([0, 1, 3, 70, 150, 200, 230, 260, 700]))
        ops = [(int(rng.choice([0, 7, 8, 1, 2, 4])), int(rng.integers(1, 300))) for _ in range(n_ops)]
        qlen = sum(l for o, l in ops if (bam.QUERY_CONSUMING >> o) & 1)
        name = bytes(rng.integers(33, 127, int(rng.choice([1, 11, 12, 27, 28, 43, 44, 100, 254]))).astype(np.uint8)).decode()
        tags = [("NM", "C", 3)]
        if rng.random() < 0.3:
            tags.append(("XZ", "Z", "t" * int(rng.choice([1, 100, 600, 900, 990, 1100, 3000]))))
        if rng.random() < 0.1:
            tags.append(("XB", "B:I", list(range(int(rng.integers(0, 400))))))
        recs.append(bam.encode_record(int(rng.integers(0, 2)), int(rng.integers(0, 10_000)), name, 60, 0, ops, qlen, bam.encode_aux(tags)))
    hdr = bam.encode_header(["x", "y"], [100_000, 100_000])
    stream = np.frombuffer(hdr + b"".join(recs), dtype=np.uint8).copy()
    offs = bam.record_offsets(stream, bam.parse_header(stream).first_record)
    # damage: a block_size beyond the stream, one below 32, a negative l_seq, an l_seq that swallows the aux block
    s2 = stream.copy()
    for k, (field, val) in enumerate(((0, 1 << 30), (0, 8), (20, -5), (20, 1 << 20))):
        o = int(offs[10 + 50 * k])
        s2[o + field:o + field + 4] = np.array([val], dtype="<i4").view(np.uint8)
    cut = int(offs[-1]) + 20                                               # the last record: fewer than 36 bytes of it
    for pb in (8192, 12288, 24576):
        _check(engine, stream, offs, True, pb)
        _check(engine, s2[:cut], offs, True, pb)
    pg = engine.bam_pages(engine.to_device(stream[:16]), torch.zeros(0, dtype=torch.int64, device=engine.device), True)
    assert (pg.n_pages, pg.n_rec) == (0, 0)
    recs_out, _ = engine.bam_filter_pages(pg, engine.to_device(np.zeros(2, dtype=np.int32)), 30, 50, 0.1, 0.9)
    assert recs_out.shape[0] == 0
