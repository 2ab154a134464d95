"""N1 on the GPU (k_inflate.hip): gci_bgzf_inflate_device against zlib on members of every DEFLATE flavour (stored, fixed and
dynamic Huffman codes, several blocks per member, codes longer than the primary tables, matches at the maximum distance),
its length / CRC checks, and gci_bam_record_offsets_device against the serial block_size walk on whole streams and on
chunks that end inside a record."""
import struct
import zlib

import numpy as np
import pytest

from gci_amd import hostio, synth
from gci_amd._lib import GciError, GCI_E_MALFORMED
from gci_amd.formats import bam as bamfmt
from gci_amd.formats import bgzf

pytestmark = pytest.mark.gpu


def member(payload: bytes, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, mem_level=8, corrupt_crc=False) -> bytes:
    c = zlib.compressobj(level, zlib.DEFLATED, -15, mem_level, strategy)
    body = c.compress(payload) + c.flush()
    bsize = 12 + 6 + len(body) + 8 - 1
    assert bsize <= 0xFFFF
    head = struct.pack("<BBBBIBBHBBHH", 0x1F, 0x8B, 8, 4, 0, 0, 0xFF, 6, 66, 67, 2, bsize)
    crc = (zlib.crc32(payload) ^ (1 if corrupt_crc else 0)) & 0xFFFFFFFF
    return head + body + struct.pack("<II", crc, len(payload))


def inflate_gpu(engine, raw: bytes, check_crc=True):
    buf = np.frombuffer(raw, dtype=np.uint8)
    pos, isz = hostio.bgzf_blocks(buf)
    return engine.bgzf_inflate(buf, pos, isz, check_crc=check_crc).cpu().numpy().tobytes()


def test_members_of_every_flavour(engine):
    rng = np.random.default_rng(41)
    qual = synth._hifi_qual_lut()[rng.integers(0, 256, 60000, dtype=np.uint8)].tobytes()
    seq = synth._SEQ_LUT[rng.integers(0, 16, 30000, dtype=np.uint8)].tobytes()
    text = (b"chr1\t12345\tACGTTGCA" * 4000)[:65000]
    skew = bytes(rng.choice(np.arange(256, dtype=np.uint8), size=64000, p=np.r_[0.5, 0.25, np.full(254, 0.25 / 254)]))   # long codes
    far = rng.integers(0, 256, 32768, dtype=np.uint8).tobytes()
    far = far + far[:258] + b"x" + far[1:700]                                    # matches at distance 32768 / 32767
    payloads = [b"", b"a", b"ab" * 3, qual, seq + qual[:30000], text, skew, far, bytes(65280), rng.integers(0, 256, 65280, dtype=np.uint8).tobytes()]
    members, want = [], []
    for p in payloads:
        for kw in (dict(level=1), dict(level=6), dict(level=9), dict(level=0), dict(level=6, strategy=zlib.Z_FIXED),
                   dict(level=6, strategy=zlib.Z_HUFFMAN_ONLY), dict(level=9, mem_level=1)):      # mem_level 1: many small blocks
            try:
                members.append(member(p, **kw))
            except AssertionError:
                continue                                                    # does not fit a BGZF member at this setting
            want.append(p)
    raw = b"".join(members) + bgzf.BGZF_EOF
    assert inflate_gpu(engine, raw) == b"".join(want)
    assert inflate_gpu(engine, raw, check_crc=False) == b"".join(want)
    assert len(members) > 50


def test_the_wave_decoder_takes_the_members_and_hands_back_what_it_cannot_list(engine):
    """Who decoded what (gci_bgzf_inflate_last_stats): members of a BAM-like mix -- stored, fixed, dynamic, several blocks, long
    codes -- are the wave decoder's (k_inflate_wave.hip); a member whose symbols are one or two bits long (more symbols per piece
    than a lane lists) goes to the lane decoder -- and both give zlib's bytes."""
    rng = np.random.default_rng(43)
    qual = synth._hifi_qual_lut()[rng.integers(0, 256, 60000, dtype=np.uint8)].tobytes()
    seq = synth._SEQ_LUT[rng.integers(0, 16, 30000, dtype=np.uint8)].tobytes()
    skew = bytes(rng.choice(np.arange(256, dtype=np.uint8), size=64000, p=np.r_[0.5, 0.25, np.full(254, 0.25 / 254)]))
    usual = [member(qual, level=1), member(seq + qual[:30000], level=6), member(skew, level=6), member(qual[:20000], level=0),
             member(qual[:30000], level=6, strategy=zlib.Z_FIXED), member(seq + qual[:30000], level=9, mem_level=1)]
    want = [qual, seq + qual[:30000], skew, qual[:20000], qual[:30000], seq + qual[:30000]]
    raw = b"".join(usual) + bgzf.BGZF_EOF
    assert inflate_gpu(engine, raw) == b"".join(want)
    st = engine.inflate_stats()
    assert st["decoded"] == len(usual) + 1 and st["not tried"] == 0, st
    two_symbols = bytes(rng.choice(np.array([65, 66], dtype=np.uint8), size=60000))         # Huffman only: one bit per symbol
    raw = member(two_symbols, level=6, strategy=zlib.Z_HUFFMAN_ONLY) + member(qual, level=1) + bgzf.BGZF_EOF
    assert inflate_gpu(engine, raw) == two_symbols + qual
    st = engine.inflate_stats()
    assert st["decoded"] == 2 and sum(v for k, v in st.items() if k not in ("decoded", "not tried")) == 1, st


def test_many_small_batches_on_two_streams(tmp_path):
    """The host side of the wave inflate with its knobs turned down: batches of 64 members alternating between two streams (scratch of
    their own), the copy kernel's workgroups looping over a batch, members handed out by a counter per batch -- 1 500 members of every
    flavour and size in a process of its own (the knobs are read once per process), equal to zlib's bytes; and the same with one stream
    and the lane decoder alone."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "run.py"
    script.write_text(
        "import sys, zlib, struct, numpy as np\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "from test_gpu_inflate import member\n"
        "from gci_amd import hostio, synth\n"
        "from gci_amd.device import Engine\n"
        "from gci_amd.formats import bgzf\n"
        "rng = np.random.default_rng(7)\n"
        "lut = synth._hifi_qual_lut()\n"
        "members, want = [], []\n"
        "for i in range(1500):\n"
        "    n = int(rng.choice([0, 1, 17, 300, 5000, 30000, 65000]))\n"
        "    kind = i %% 4\n"
        "    p = (lut[rng.integers(0, 256, n, dtype=np.uint8)].tobytes() if kind == 0 else rng.integers(0, 256, n, dtype=np.uint8).tobytes() if kind == 1\n"
        "         else (b'chr7\\t1234\\tACGT' * (n // 14 + 1))[:n] if kind == 2 else bytes(n))\n"
        "    kw = [dict(level=1), dict(level=6), dict(level=0), dict(level=6, strategy=zlib.Z_FIXED), dict(level=9, mem_level=1)][i %% 5]\n"
        "    try:\n"
        "        members.append(member(p, **kw)); want.append(p)\n"
        "    except AssertionError:\n"
        "        pass\n"
        "raw = np.frombuffer(b''.join(members) + bgzf.BGZF_EOF, dtype=np.uint8)\n"
        "pos, isz = hostio.bgzf_blocks(raw)\n"
        "e = Engine(0)\n"
        "got = e.bgzf_inflate(raw, pos, isz).cpu().numpy().tobytes()\n"
        "assert got == b''.join(want), 'inflate differs'\n"
        "print('members', len(members), e.inflate_stats())\n" % (root, os.path.join(root, "tests")))
    for env in (dict(GCI_INFLATE_BATCH="64", GCI_INFLATE_STREAMS="2"), dict(GCI_INFLATE_BATCH="200", GCI_INFLATE_STREAMS="1", GCI_INFLATE_COPY_GRID="members"),
                dict(GCI_INFLATE="lane")):
        r = subprocess.run([sys.executable, str(script)], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (env, r.stderr[-2000:])
        assert "members" in r.stdout
        if "GCI_INFLATE" not in env:
            assert "'decoded': 0" not in r.stdout and "'not tried': 0" in r.stdout, r.stdout


def test_match_shapes(engine):
    """Every way a match is copied: periods 1 .. 7 and distances 8 .. 40 against lengths 3 .. 258, at the start of a
    member, in its middle and ending exactly at its end (where the 8-byte stores must not run over into the next member)."""
    rng = np.random.default_rng(5)
    payloads = []
    for dist in list(range(1, 41)) + [63, 64, 65, 257, 258, 259, 4096, 32768]:
        seed = rng.integers(0, 256, dist, dtype=np.uint8).tobytes()
        for length in (3, 4, 7, 8, 9, 15, 16, 17, 31, 32, 33, 257, 258):
            rep = (seed * (length // dist + 2))[:length]
            lead = rng.integers(0, 256, int(rng.integers(0, 24)), dtype=np.uint8).tobytes()
            payloads.append(lead + seed + rep)                                   # the member ends with the match
            payloads.append(seed + rep + lead)
    members = [member(p, level=9) for p in payloads] + [member(p, level=6, strategy=zlib.Z_FIXED) for p in payloads[::7]]
    want = payloads + payloads[::7]
    raw = b"".join(members) + bgzf.BGZF_EOF
    assert inflate_gpu(engine, raw) == b"".join(want)


def test_damaged_payloads(engine):
    """One flipped bit (or a cut) in the compressed bytes of one member of a file: the call reports that member whenever
    zlib cannot reproduce the payload from the damaged bytes -- and never runs away (a stream of empty blocks ends at the
    payload's end)."""
    rng = np.random.default_rng(8)
    text = (b"m64011_190830_220126/%d/ccs\t" * 700) % tuple(range(700))
    base = [rng.integers(0, 64, 5000, dtype=np.uint8).tobytes(), text[:20000], bytes(3000), synth._hifi_qual_lut()[rng.integers(0, 256, 9000, dtype=np.uint8)].tobytes()]
    good = [member(p, level=int(rng.integers(1, 10))) for p in base * 2]
    reported = 0
    for trial in range(60):
        j = int(rng.integers(0, len(good)))
        m = bytearray(good[j])
        body = slice(18, len(m) - 8)
        if trial % 6 == 5:                                                       # the payload replaced by empty stored blocks
            fill = (b"\x00\x00\x00\xff\xff" * (len(m) // 5 + 1))[:body.stop - body.start]
            m[body] = fill
        else:
            k = int(rng.integers(body.start, body.stop))
            m[k] ^= 1 << int(rng.integers(0, 8))
        try:
            ok = zlib.decompress(bytes(m[body]), -15) == (base * 2)[j]
        except zlib.error:
            ok = False
        raw = b"".join(good[:j]) + bytes(m) + b"".join(good[j + 1:]) + bgzf.BGZF_EOF
        buf = np.frombuffer(raw, dtype=np.uint8)
        pos, isz = hostio.bgzf_blocks(buf)
        if ok:
            assert engine.bgzf_inflate(buf, pos, isz).cpu().numpy().tobytes() == b"".join(base * 2)
        else:
            with pytest.raises(GciError) as e:
                engine.bgzf_inflate(buf, pos, isz)
            assert e.value.status == GCI_E_MALFORMED and e.value.rec == j
            reported += 1
    assert reported > 40


def test_bad_members_are_reported(engine):
    good = member(b"hello world" * 500)
    bad_crc = member(b"hello world" * 500, corrupt_crc=True)
    for k, raw in enumerate((good + bad_crc + good, good + good[:-9] + b"\x00" + good[-8:] + good)):
        buf = np.frombuffer(raw + bgzf.BGZF_EOF, dtype=np.uint8)
        pos, isz = hostio.bgzf_blocks(buf)
        with pytest.raises(GciError) as e:
            engine.bgzf_inflate(buf, pos, isz)
        assert e.value.status == GCI_E_MALFORMED and e.value.rec == 1
    # a wrong CRC passes when the check is off (the payload is intact)
    buf = np.frombuffer(good + bad_crc + bgzf.BGZF_EOF, dtype=np.uint8)
    pos, isz = hostio.bgzf_blocks(buf)
    assert engine.bgzf_inflate(buf, pos, isz, check_crc=False).cpu().numpy().tobytes() == b"hello world" * 1000


@pytest.mark.parametrize("kind,cov,seq_qual", [("hifi", 20, "random"), ("ont", 12, "const")])
def test_bam_file_inflate_and_record_walk(engine, tmp_path, kind, cov, seq_qual):
    contigs = (("a", 900_000), ("b", 300_000))
    rs = synth.simulate_reads(contigs, cov, kind, seed=77, **({"long_cigar_frac": 0.02} if kind == "ont" else {}))
    stream, offs = synth.to_bam_stream(rs, seq_qual=seq_qual, seed=3)
    path = str(tmp_path / "x.bam")
    bamfmt.write_bam_stream(path, stream, level=6 if kind == "hifi" else 1, threads=4)
    raw = np.fromfile(path, dtype=np.uint8)
    pos, isz = hostio.bgzf_blocks(raw)
    d = engine.bgzf_inflate(raw, pos, isz)
    assert np.array_equal(d.cpu().numpy(), stream)
    hdr = bamfmt.parse_header(stream)
    got, used, ok = engine.bam_record_offsets(d, hdr.first_record, len(hdr.references))
    assert ok and used == stream.shape[0]
    assert np.array_equal(got.cpu().numpy().view(np.uint64), offs)
    # a stream that does not start at a 16-byte address (the candidate test reads aligned blocks)
    for skip in (1, 5, 15):
        got, used, ok = engine.bam_record_offsets(d[skip:], hdr.first_record - skip, len(hdr.references))
        assert ok and used == stream.shape[0] - skip
        assert np.array_equal(got.cpu().numpy().view(np.uint64), offs - np.uint64(skip))
    # a chunk that ends inside a record: the complete records, and where the partial one begins
    cut = int(offs[len(offs) // 2]) + 50
    got, used, ok = engine.bam_record_offsets(d[:cut], hdr.first_record, len(hdr.references))
    assert ok and used == int(offs[len(offs) // 2])
    assert np.array_equal(got.cpu().numpy().view(np.uint64), offs[:len(offs) // 2])
    # ... and one that ends inside a record's first 36 bytes
    cut = int(offs[10]) + 20
    got, used, ok = engine.bam_record_offsets(d[:cut], hdr.first_record, len(hdr.references))
    assert ok and used == int(offs[10]) and got.shape[0] == 10
    # a record the strict format test rejects (l_seq < 0) breaks the chain: reported, not guessed around
    broken = stream.copy()
    broken[int(offs[5]) + 20:int(offs[5]) + 24] = np.frombuffer(np.int32(-7).tobytes(), dtype=np.uint8)
    got, used, ok = engine.bam_record_offsets(engine.to_device(broken), hdr.first_record, len(hdr.references))
    assert not ok and used == int(offs[4])
