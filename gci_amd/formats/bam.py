"""BAM container: header, record-offset table, record encoder, field decoder.

Written from the SAM/BAM specification; the reference touches
this layer only through pysam (/root/reference/GCI.py:150-168, 201-208, 963-976).

The *inflated* stream layout is what the device consumes:

    "BAM\\1" | l_text i32 | text | n_ref i32 | {l_name i32 | name\\0 | l_ref i32} x n_ref
    then records:  block_size i32 | refID i32 | pos i32 | l_read_name u8 | mapq u8 |
                   bin u16 | n_cigar_op u16 | flag u16 | l_seq i32 | next_refID i32 |
                   next_pos i32 | tlen i32 | read_name\\0 | cigar u32 x n | seq | qual | aux

``record_offsets`` walks the ``block_size`` chain once (a pointer chase, the one serial
step of the decode) and returns the byte offset of every record's ``block_size`` word;
everything after that is per-record parallel and happens on the GPU (K1 ``bam_filter``).
"""
from __future__ import annotations

import struct
from dataclasses import dataclass, field
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np

from . import bgzf

CIGAR_OPS = "MIDNSHP=XB"
OP = {c: i for i, c in enumerate(CIGAR_OPS)}
REF_CONSUMING = (1 << 0) | (1 << 2) | (1 << 3) | (1 << 7) | (1 << 8)   # M D N = X
QUERY_CONSUMING = (1 << 0) | (1 << 1) | (1 << 4) | (1 << 7) | (1 << 8)  # M I S = X

FLAG_UNMAPPED = 0x4
FLAG_REVERSE = 0x10
FLAG_SECONDARY = 0x100
FLAG_SUPPLEMENTARY = 0x800

CORE_BYTES = 36  # block_size + 32-byte fixed core


class BAMError(ValueError):
    pass


@dataclass
class BamHeader:
    text: str
    references: Tuple[str, ...]
    lengths: Tuple[int, ...]
    first_record: int  # byte offset of the first record in the inflated stream


def encode_header(references: Sequence[str], lengths: Sequence[int], text: Optional[str] = None) -> bytes:
    if text is None:
        text = "@HD\tVN:1.6\tSO:coordinate\n" + "".join(
            "@SQ\tSN:%s\tLN:%d\n" % (n, l) for n, l in zip(references, lengths))
    tb = text.encode()
    out = [b"BAM\x01", struct.pack("<i", len(tb)), tb, struct.pack("<i", len(references))]
    for n, l in zip(references, lengths):
        nb = n.encode() + b"\x00"
        out += [struct.pack("<i", len(nb)), nb, struct.pack("<i", l)]
    return b"".join(out)


def parse_header(stream) -> BamHeader:
    buf = memoryview(stream)
    if bytes(buf[:4]) != b"BAM\x01":
        raise BAMError("missing BAM magic")
    l_text = struct.unpack_from("<i", buf, 4)[0]
    text = bytes(buf[8:8 + l_text]).split(b"\x00", 1)[0].decode(errors="replace")
    p = 8 + l_text
    n_ref = struct.unpack_from("<i", buf, p)[0]
    p += 4
    names, lens = [], []
    for _ in range(n_ref):
        l_name = struct.unpack_from("<i", buf, p)[0]
        p += 4
        names.append(bytes(buf[p:p + l_name]).split(b"\x00", 1)[0].decode())
        p += l_name
        lens.append(struct.unpack_from("<i", buf, p)[0])
        p += 4
    return BamHeader(text, tuple(names), tuple(lens), p)


def record_offsets(stream: np.ndarray, first_record: int) -> np.ndarray:
    """Offsets (uint64) of each record's block_size word.  Serial pointer chase."""
    n = int(stream.shape[0])
    offs: List[int] = []
    mv = memoryview(stream)
    p = first_record
    unpack = struct.Struct("<i").unpack_from
    while p < n:
        if p + 4 > n:
            raise BAMError("truncated record length at byte %d" % p)
        bs = unpack(mv, p)[0]
        if bs < 32 or p + 4 + bs > n:
            raise BAMError("bad block_size %d at byte %d" % (bs, p))
        offs.append(p)
        p += 4 + bs
    return np.asarray(offs, dtype=np.uint64)


def reg2bin(beg: int, end: int) -> int:
    end -= 1
    if beg >> 14 == end >> 14:
        return ((1 << 15) - 1) // 7 + (beg >> 14)
    if beg >> 17 == end >> 17:
        return ((1 << 12) - 1) // 7 + (beg >> 17)
    if beg >> 20 == end >> 20:
        return ((1 << 9) - 1) // 7 + (beg >> 20)
    if beg >> 23 == end >> 23:
        return ((1 << 6) - 1) // 7 + (beg >> 23)
    if beg >> 26 == end >> 26:
        return ((1 << 3) - 1) // 7 + (beg >> 26)
    return 0


def encode_aux(tags: Sequence[Tuple[str, str, object]]) -> bytes:
    """tags: (two-letter tag, type char, value).  Types: A c C s S i I f Z H, and
    'B:<sub>' with a sequence value."""
    out = []
    for tag, typ, val in tags:
        t = tag.encode()
        if typ == "A":
            out.append(t + b"A" + (val.encode() if isinstance(val, str) else bytes([val])))
        elif typ in "cCsSiIf":
            out.append(t + typ.encode() + struct.pack("<" + {"c": "b", "C": "B", "s": "h", "S": "H",
                                                              "i": "i", "I": "I", "f": "f"}[typ], val))
        elif typ in "ZH":
            out.append(t + typ.encode() + str(val).encode() + b"\x00")
        elif typ.startswith("B:"):
            sub = typ[2]
            fmt = {"c": "b", "C": "B", "s": "h", "S": "H", "i": "i", "I": "I", "f": "f"}[sub]
            arr = list(val)
            out.append(t + b"B" + sub.encode() + struct.pack("<i", len(arr)) +
                       struct.pack("<%d%s" % (len(arr), fmt), *arr))
        else:
            raise BAMError("unknown aux type %r" % typ)
    return b"".join(out)


def encode_record(ref_id: int, pos: int, name: str, mapq: int, flag: int,
                  cigar: Sequence[Tuple[int, int]], l_seq: int, aux: bytes = b"",
                  seq_fill: int = 0x11, qual_fill: int = 0xFF,
                  n_cigar_field: Optional[int] = None, next_ref: int = -1, next_pos: int = -1,
                  tlen: int = 0) -> bytes:
    """One record incl. its block_size prefix.  ``cigar`` = [(op_code, length)].
    ``l_seq`` 0 encodes SEQ '*'.  More than 65535 ops are stored the spec's way: a
    ``<l_seq>S<ref_len>N`` placeholder plus a ``CG:B,I`` tag appended to ``aux``."""
    nb = name.encode() + b"\x00"
    if len(nb) > 255:
        raise BAMError("read name too long")
    ops = [(l << 4) | o for o, l in cigar]
    rlen = sum(l for o, l in cigar if (REF_CONSUMING >> o) & 1)
    if len(ops) > 65535:
        aux = aux + encode_aux([("CG", "B:I", ops)])
        ops = [(l_seq << 4) | OP["S"], (rlen << 4) | OP["N"]]
    n_cig = len(ops) if n_cigar_field is None else n_cigar_field
    end = pos + (rlen if rlen > 0 else 1)
    core = struct.pack("<iiBBHHHiiii", ref_id, pos, len(nb), mapq, reg2bin(max(pos, 0), max(end, 1)),
                       n_cig, flag, l_seq, next_ref, next_pos, tlen)
    body = core + nb + struct.pack("<%dI" % len(ops), *ops) + bytes([seq_fill]) * ((l_seq + 1) // 2) + \
        bytes([qual_fill]) * l_seq + aux
    return struct.pack("<i", len(body)) + body


def write_bam(path: str, references: Sequence[str], lengths: Sequence[int], records: Iterable[bytes],
              level: int = 1, threads: int = 1, header_text: Optional[str] = None) -> None:
    payload = encode_header(references, lengths, header_text) + b"".join(records)
    bgzf.write_file(path, payload, level=level, threads=threads)


def write_bam_stream(path: str, stream: np.ndarray, level: int = 1, threads: int = 1) -> None:
    """BGZF-wrap an already assembled inflated stream (header + records)."""
    bgzf.write_file(path, stream.tobytes() if isinstance(stream, np.ndarray) else bytes(stream),
                    level=level, threads=threads)


def read_bam(path: str, threads: int = 1):
    """-> (inflated stream uint8[n], BamHeader, record offsets uint64[R])."""
    stream = bgzf.read_file(path, threads=threads)
    hdr = parse_header(stream)
    offs = record_offsets(stream, hdr.first_record)
    return stream, hdr, offs


def read_header(path: str) -> BamHeader:
    """Header only: inflate BGZF members one at a time until the reference table is complete."""
    import zlib
    data = bytearray()
    with open(path, "rb") as f:
        while True:
            head = f.read(18)
            if len(head) < 18:
                raise BAMError("truncated BAM header in %s" % path)
            if head[0] != 0x1F or head[1] != 0x8B or head[12:16] != b"BC\x02\x00":
                raise BAMError("%s is not a BGZF file" % path)
            bsize = head[16] | (head[17] << 8)
            rest = f.read(bsize + 1 - 18)
            data += zlib.decompress(rest[:-8], -15)
            try:
                return parse_header(bytes(data))
            except (struct.error, IndexError, UnicodeDecodeError):
                continue


# ----------------------------------------------------------------------------------------------
# Plain-Python field decoder: used by the pysam stand-in (tools/ref_shim) and by the numpy
# oracle's small-case path.  It makes no filtering decisions.
# ----------------------------------------------------------------------------------------------

@dataclass
class RecordView:
    ref_id: int
    pos: int
    mapq: int
    flag: int
    l_seq: int
    name: str
    cigar: List[Tuple[int, int]]
    aux: Dict[str, Tuple[str, object]] = field(default_factory=dict)
    n_cigar_field: int = 0


_AUX_SIZE = {"A": 1, "c": 1, "C": 1, "s": 2, "S": 2, "i": 4, "I": 4, "f": 4}
_AUX_FMT = {"c": "b", "C": "B", "s": "h", "S": "H", "i": "i", "I": "I", "f": "f"}


def parse_aux(buf: bytes) -> List[Tuple[str, str, object]]:
    out = []
    p, n = 0, len(buf)
    while p + 3 <= n:
        tag = buf[p:p + 2].decode(errors="replace")
        typ = chr(buf[p + 2])
        p += 3
        if typ == "A":
            out.append((tag, typ, chr(buf[p])))
            p += 1
        elif typ in _AUX_FMT:
            out.append((tag, typ, struct.unpack_from("<" + _AUX_FMT[typ], buf, p)[0]))
            p += _AUX_SIZE[typ]
        elif typ in "ZH":
            e = buf.index(b"\x00", p)
            out.append((tag, typ, buf[p:e].decode(errors="replace")))
            p = e + 1
        elif typ == "B":
            sub = chr(buf[p])
            cnt = struct.unpack_from("<i", buf, p + 1)[0]
            p += 5
            out.append((tag, "B:" + sub, list(struct.unpack_from("<%d%s" % (cnt, _AUX_FMT[sub]), buf, p))))
            p += cnt * _AUX_SIZE[sub]
        else:
            raise BAMError("unknown aux type %r" % typ)
    return out


def decode_record(stream: np.ndarray, off: int, restore_long_cigar: bool = True) -> RecordView:
    mv = memoryview(stream)
    off = int(off)
    (bs, ref_id, pos, l_rn, mapq, _bin, n_cig, flag, l_seq, _nr, _np, _tl) = struct.unpack_from(
        "<iiiBBHHHiiii", mv, off)
    p = off + CORE_BYTES
    name = bytes(mv[p:p + l_rn]).split(b"\x00", 1)[0].decode(errors="replace")
    p += l_rn
    raw_ops = struct.unpack_from("<%dI" % n_cig, mv, p)
    p += 4 * n_cig
    p += (l_seq + 1) // 2 + l_seq
    aux_list = parse_aux(bytes(mv[p:off + 4 + bs]))
    cigar = [(v & 0xF, v >> 4) for v in raw_ops]
    aux: Dict[str, Tuple[str, object]] = {}
    for tag, typ, val in aux_list:
        aux.setdefault(tag, (typ, val))  # first occurrence wins, as bam_aux_get does
    # htslib restores a >65535-op CIGAR from the CG tag when op0 == <l_seq>S
    if (restore_long_cigar and n_cig > 0 and ref_id >= 0 and pos >= 0 and cigar[0] == (OP["S"], l_seq)
            and "CG" in aux and aux["CG"][0] in ("B:I", "B:i")):
        real = aux["CG"][1]
        if n_cig <= len(real) < (1 << 29):
            cigar = [(int(v) & 0xF, (int(v) & 0xFFFFFFFF) >> 4) for v in real]
            del aux["CG"]
    return RecordView(ref_id, pos, mapq, flag, l_seq, name, cigar, aux, n_cig)


# ----------------------------------------------------------------------------------------------
# Minimal BAI writer: only so generated files look conventional.  Nothing in this package reads
# the index: "all fetch() chunks of a contig" == "all records with that refID" (SURVEY.md 8a-R1).
# ----------------------------------------------------------------------------------------------

BAI_PSEUDO_BIN = 37450        # samtools' per-reference metadata bin: chunk 0 = (ref_beg, ref_end) virtual offsets


def write_bai(path: str, n_ref: int, bam_path: Optional[str] = None) -> None:
    """`<bam>.bai` for a coordinate-sorted BAM written by write_bam_stream (SAM spec 5.2).  Per reference with records:
    bin 0 holding one chunk [first record, end of its last record), the pseudo-bin 37450 samtools writes (chunk 0 = the
    same range, chunk 1 = mapped / unmapped counts) and the 16 kb linear index.  Virtual offset = compressed offset of
    the BGZF member << 16 | offset inside its payload.  Without bam_path: an index without entries."""
    if bam_path is None:
        with open(path, "wb") as f:
            f.write(b"BAI\x01" + struct.pack("<i", n_ref))
            for _ in range(n_ref):
                f.write(struct.pack("<ii", 0, 0))
        return
    with open(bam_path, "rb") as f:
        raw = f.read()
    blocks = bgzf.scan_blocks(raw)
    stream = bgzf.decompress(raw)
    hdr = parse_header(stream)
    offs = record_offsets(stream, hdr.first_record)
    ustart = np.cumsum([0] + [b[2] for b in blocks])              # inflated offset of every member
    cstart = np.asarray([b[0] for b in blocks] + [len(raw)], dtype=np.int64)

    def voffset(u: int) -> int:
        k = int(np.searchsorted(ustart, u, side="right")) - 1
        while k + 1 < len(blocks) and blocks[k][2] == 0:
            k += 1
        if k >= len(blocks):
            k = len(blocks) - 1
        return (int(cstart[k]) << 16) | (u - int(ustart[k]))

    o = offs.astype(np.int64)
    ref = stream[(o[:, None] + np.arange(4, 8)[None, :])].copy().view("<i4").reshape(-1) if o.shape[0] else np.zeros(0, np.int32)
    pos = stream[(o[:, None] + np.arange(8, 12)[None, :])].copy().view("<i4").reshape(-1) if o.shape[0] else np.zeros(0, np.int32)
    flag = stream[(o[:, None] + np.arange(18, 20)[None, :])].copy().view("<u2").reshape(-1) if o.shape[0] else np.zeros(0, np.uint16)
    ends = np.concatenate([o[1:], [stream.shape[0]]]) if o.shape[0] else np.zeros(0, np.int64)
    with open(path, "wb") as f:
        f.write(b"BAI\x01" + struct.pack("<i", n_ref))
        for r in range(n_ref):
            idx = np.flatnonzero(ref == r)
            if idx.shape[0] == 0:
                f.write(struct.pack("<ii", 0, 0))
                continue
            beg, end = voffset(int(o[idx[0]])), voffset(int(ends[idx[-1]]))
            n_un = int(((flag[idx] & 4) != 0).sum())
            f.write(struct.pack("<i", 2))
            f.write(struct.pack("<Ii", 0, 1) + struct.pack("<QQ", beg, end))
            f.write(struct.pack("<Ii", BAI_PSEUDO_BIN, 2) + struct.pack("<QQ", beg, end) + struct.pack("<QQ", idx.shape[0] - n_un, n_un))
            n_win = int(hdr.lengths[r] + 16383) // 16384
            first = np.full(n_win, -1, dtype=np.int64)
            w = np.clip(pos[idx].astype(np.int64), 0, None) // 16384
            w = np.minimum(w, n_win - 1)
            order = np.arange(idx.shape[0])
            np.minimum.at(first, w, order)                          # (records are sorted: the first record of a window)
            f.write(struct.pack("<i", n_win))
            last = 0
            for k in range(n_win):
                if first[k] >= 0 and first[k] < idx.shape[0]:
                    last = voffset(int(o[idx[int(first[k])]]))
                f.write(struct.pack("<Q", last))
        f.write(struct.pack("<Q", 0))                               # n_no_coor


def read_bai(path: str) -> Optional[List[Optional[Tuple[int, int]]]]:
    """-> per reference (virtual offset of its first record, virtual offset behind its last record) or None when the
    index lists nothing for it; None when there is no usable index file.  The range is the pseudo-bin's (37450) when
    present, else the hull of all chunks."""
    try:
        with open(path, "rb") as f:
            raw = f.read()
    except OSError:
        return None
    if raw[:4] != b"BAI\x01":
        return None
    p = 4
    try:
        (n_ref,) = struct.unpack_from("<i", raw, p)
        p += 4
        out: List[Optional[Tuple[int, int]]] = []
        for _ in range(n_ref):
            (n_bin,) = struct.unpack_from("<i", raw, p)
            p += 4
            lo, hi, meta = None, None, None
            for _b in range(n_bin):
                b, n_chunk = struct.unpack_from("<Ii", raw, p)
                p += 8
                chunks = struct.unpack_from("<%dQ" % (2 * n_chunk), raw, p)
                p += 16 * n_chunk
                if b == BAI_PSEUDO_BIN:
                    if n_chunk >= 1:
                        meta = (chunks[0], chunks[1])
                    continue
                for k in range(n_chunk):
                    lo = chunks[2 * k] if lo is None else min(lo, chunks[2 * k])
                    hi = chunks[2 * k + 1] if hi is None else max(hi, chunks[2 * k + 1])
            (n_intv,) = struct.unpack_from("<i", raw, p)
            p += 4 + 8 * n_intv
            out.append(meta if meta is not None else ((lo, hi) if lo is not None else None))
        return out
    except struct.error:
        return None
