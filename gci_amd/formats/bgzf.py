"""BGZF container (the blocked-gzip framing BAM files use), written from the SAM/BAM
specification -- there is no htslib in this image.

A BGZF file is a sequence of gzip members, each at most 64 KiB, each carrying a ``BC``
extra sub-field whose value is ``BSIZE = total member bytes - 1``; the file ends with a
fixed 28-byte empty member.  The reference reaches this layer only through
``pysam.AlignmentFile`` (/root/reference/GCI.py:150-151, 201, 963, 974).

Host side only: inflate is serial within a block, parallel across blocks.  zlib releases
the GIL, so a thread pool scales with cores.
"""
from __future__ import annotations

import struct
import zlib
from concurrent.futures import ThreadPoolExecutor
from typing import List, Tuple

import numpy as np

BGZF_EOF = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")
MAX_BLOCK_PAYLOAD = 0xFF00  # htslib's choice: leaves room for incompressible data


class BGZFError(ValueError):
    pass


def _member(payload: bytes, level: int) -> bytes:
    comp = zlib.compressobj(level, zlib.DEFLATED, -15)
    body = comp.compress(payload) + comp.flush()
    bsize = 12 + 6 + len(body) + 8 - 1
    if bsize > 0xFFFF:  # incompressible: store
        comp = zlib.compressobj(0, zlib.DEFLATED, -15)
        body = comp.compress(payload) + comp.flush()
        bsize = 12 + 6 + len(body) + 8 - 1
    head = struct.pack("<BBBBIBBHBBHH", 0x1F, 0x8B, 8, 4, 0, 0, 0xFF, 6, 66, 67, 2, bsize)
    tail = struct.pack("<II", zlib.crc32(payload) & 0xFFFFFFFF, len(payload))
    return head + body + tail


def compress(data: bytes, level: int = 1, threads: int = 1) -> bytes:
    """Whole-buffer BGZF encode (blocks of MAX_BLOCK_PAYLOAD bytes + EOF marker)."""
    view = memoryview(data)
    chunks = [bytes(view[i:i + MAX_BLOCK_PAYLOAD]) for i in range(0, len(view), MAX_BLOCK_PAYLOAD)]
    if threads > 1 and len(chunks) > 1:
        with ThreadPoolExecutor(threads) as ex:
            members = list(ex.map(lambda c: _member(c, level), chunks))
    else:
        members = [_member(c, level) for c in chunks]
    return b"".join(members) + BGZF_EOF


def member_size(raw, pos: int) -> int:
    """Total bytes of the BGZF member that starts at `pos` (its BSIZE + 1), found by walking the extra sub-fields: BC need not be
    the first one."""
    n = len(raw)
    if n - pos < 18 or raw[pos] != 0x1F or raw[pos + 1] != 0x8B or raw[pos + 2] != 8 or not (raw[pos + 3] & 4):
        raise BGZFError("not a BGZF member at byte %d" % pos)
    xlen = int(raw[pos + 10]) | (int(raw[pos + 11]) << 8)
    p, end = pos + 12, min(n, pos + 12 + xlen)
    while p + 4 <= end:
        slen = int(raw[p + 2]) | (int(raw[p + 3]) << 8)
        if raw[p] == 66 and raw[p + 1] == 67 and slen == 2 and p + 6 <= end:
            return (int(raw[p + 4]) | (int(raw[p + 5]) << 8)) + 1
        p += 4 + slen
    raise BGZFError("gzip member without BC sub-field at byte %d" % pos)


def scan_blocks(raw: bytes) -> List[Tuple[int, int, int]]:
    """Return [(member_offset, member_size, isize)] by walking the BSIZE chain."""
    out = []
    pos, n = 0, len(raw)
    while pos < n:
        if n - pos < 18:
            raise BGZFError("truncated BGZF header at byte %d" % pos)
        if raw[pos] != 0x1F or raw[pos + 1] != 0x8B or raw[pos + 2] != 8 or not (raw[pos + 3] & 4):
            raise BGZFError("not a BGZF member at byte %d" % pos)
        xlen = raw[pos + 10] | (raw[pos + 11] << 8)
        p, end, bsize = pos + 12, pos + 12 + xlen, -1
        while p + 4 <= end:
            si1, si2 = raw[p], raw[p + 1]
            slen = raw[p + 2] | (raw[p + 3] << 8)
            if si1 == 66 and si2 == 67 and slen == 2:
                bsize = raw[p + 4] | (raw[p + 5] << 8)
            p += 4 + slen
        if bsize < 0:
            raise BGZFError("gzip member without BC sub-field at byte %d" % pos)
        size = bsize + 1
        if pos + size > n:
            raise BGZFError("truncated BGZF member at byte %d" % pos)
        isize = struct.unpack_from("<I", raw, pos + size - 4)[0]
        out.append((pos, size, isize))
        pos += size
    return out


def decompress(raw: bytes, threads: int = 1, check_crc: bool = False) -> np.ndarray:
    """Inflate a whole BGZF byte string into one contiguous uint8 array."""
    blocks = scan_blocks(raw)
    isz = np.fromiter((b[2] for b in blocks), dtype=np.int64, count=len(blocks))
    offs = np.zeros(len(blocks) + 1, dtype=np.int64)
    np.cumsum(isz, out=offs[1:])
    out = np.empty(int(offs[-1]), dtype=np.uint8)
    mv = memoryview(raw)

    def work(i: int) -> None:
        pos, size, isize = blocks[i]
        if isize == 0:
            return
        xlen = raw[pos + 10] | (raw[pos + 11] << 8)
        data = zlib.decompress(mv[pos + 12 + xlen: pos + size - 8], -15, isize)
        if len(data) != isize:
            raise BGZFError("ISIZE mismatch in member at byte %d" % pos)
        if check_crc:
            crc = struct.unpack_from("<I", raw, pos + size - 8)[0]
            if zlib.crc32(data) & 0xFFFFFFFF != crc:
                raise BGZFError("CRC mismatch in member at byte %d" % pos)
        out[offs[i]:offs[i + 1]] = np.frombuffer(data, dtype=np.uint8)

    if threads > 1 and len(blocks) > 1:
        # one task per contiguous run of blocks: zlib releases the GIL, the per-task overhead does not
        n_tasks = min(len(blocks), threads * 4)
        step = -(-len(blocks) // n_tasks)

        def span(k: int) -> None:
            for i in range(k * step, min(len(blocks), (k + 1) * step)):
                work(i)

        with ThreadPoolExecutor(threads) as ex:
            list(ex.map(span, range(n_tasks)))
    else:
        for i in range(len(blocks)):
            work(i)
    return out


def read_file(path: str, threads: int = 1) -> np.ndarray:
    with open(path, "rb") as f:
        raw = f.read()
    return decompress(raw, threads=threads)


def write_file(path: str, data: bytes, level: int = 1, threads: int = 1) -> None:
    with open(path, "wb") as f:
        f.write(compress(data, level=level, threads=threads))
