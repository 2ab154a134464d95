"""`.depth.gz`, BED3 and region-BED I/O.

Grammar of the depth file (/root/reference/GCI.py:110-117; consumers
utility/GCI_score.py:25-37, utility/convert_samtools_depth.py:11-20):

    ( '>' contig '\\n' ( decimal '\\n' ) ^ contig_length ) *      contigs in header order

The reference emits one gzip member per (contig, thread-chunk) and its compressed bytes
carry mtime and file names, so only the *decompressed* stream is comparable
(SURVEY.md F5).  Any multi-member gzip whose concatenated payload equals that text is a
valid output; members here are compressed in parallel at a fast level.
"""
from __future__ import annotations

import gzip
import hashlib
import zlib
from concurrent.futures import ThreadPoolExecutor
from typing import Dict, Iterable, Iterator, Tuple

import numpy as np

MEMBER_BYTES = 8 << 20


def _gzip_member(data, level: int) -> bytes:
    c = zlib.compressobj(level, zlib.DEFLATED, 31)
    return c.compress(data) + c.flush()


def write_depth_gz(path: str, pieces: Iterable[Tuple[str, memoryview]], level: int = 1, threads: int = 1) -> None:
    """pieces: (contig name, text bytes of that contig's depth lines) in header order."""
    with open(path, "wb") as f, ThreadPoolExecutor(max(1, threads)) as ex:
        for name, text in pieces:
            mv = memoryview(text)
            chunks = [(">%s\n" % name).encode()]
            chunks += [mv[i:i + MEMBER_BYTES] for i in range(0, len(mv), MEMBER_BYTES)]
            for member in ex.map(lambda c: _gzip_member(c, level), chunks):
                f.write(member)


def iter_depth_text(path: str, block: int = 1 << 24) -> Iterator[bytes]:
    with gzip.open(path, "rb") as f:
        while True:
            b = f.read(block)
            if not b:
                return
            yield b


def sha256_of_text(path: str) -> str:
    h = hashlib.sha256()
    for b in iter_depth_text(path):
        h.update(b)
    return h.hexdigest()


def read_depth_gz(path: str) -> Dict[str, np.ndarray]:
    """Parse into {contig: int64 array}.  Vectorised: one pass over the inflated bytes."""
    data = np.frombuffer(b"".join(iter_depth_text(path)), dtype=np.uint8)
    out: Dict[str, np.ndarray] = {}
    if data.size == 0:
        return out
    nl = np.flatnonzero(data == 10)
    starts = np.concatenate(([0], nl[:-1] + 1)) if nl.size else np.zeros(0, dtype=np.int64)
    is_hdr = data[starts] == ord(">")
    hdr_idx = np.flatnonzero(is_hdr)
    # numeric value of every line (garbage for header lines, skipped below)
    digit = data.astype(np.int64) - 48
    lens = nl - starts
    maxlen = int(lens[~is_hdr].max()) if (~is_hdr).any() else 0
    vals = np.zeros(starts.shape[0], dtype=np.int64)
    for k in range(maxlen):
        sel = (lens > k) & ~is_hdr
        vals[sel] = vals[sel] * 10 + digit[starts[sel] + k]
    bounds = list(hdr_idx) + [starts.shape[0]]
    for a, b in zip(bounds[:-1], bounds[1:]):
        name = bytes(data[starts[a] + 1:nl[a]]).decode()
        out[name] = vals[a + 1:b].copy()
    return out
