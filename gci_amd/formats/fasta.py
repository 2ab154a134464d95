"""FASTA reading for the two things the hot path needs from the assembly:

* the record ids, in file order (``SeqIO.parse(...).id`` at /root/reference/GCI.py:939-941);
* the N/n runs of every record as 0-based half-open intervals in *sequence* coordinates
  (``re.compile(r'(?i)N+').finditer(str(record.seq))`` at /root/reference/GCI.py:29-35).

Sequence coordinates follow Bio.SeqIO's FASTA reader: the id is the title up to the first
whitespace, line ends are stripped and blanks / carriage returns inside the sequence are
removed before positions are counted.  Done with numpy over the raw bytes, one record at
a time, so a 3 Gb assembly never becomes a Python string.
"""
from __future__ import annotations

from typing import Dict, Iterator, List, Tuple

import numpy as np

_DROP = np.zeros(256, dtype=bool)
for _c in b"\n\r ":
    _DROP[_c] = True
_IS_N = np.zeros(256, dtype=bool)
_IS_N[ord("N")] = _IS_N[ord("n")] = True


def _records(buf: np.ndarray) -> Iterator[Tuple[str, np.ndarray]]:
    """Yield (id, body-bytes view) per record."""
    n = buf.shape[0]
    if n == 0:
        return
    nl = np.flatnonzero(buf == 10)
    line_starts = np.concatenate(([0], nl + 1))
    line_starts = line_starts[line_starts < n]
    hdr_lines = line_starts[buf[line_starts] == ord(">")]
    for k, hs in enumerate(hdr_lines):
        he_idx = np.searchsorted(nl, hs)
        he = int(nl[he_idx]) if he_idx < nl.shape[0] else n
        title = bytes(buf[hs + 1:he]).decode(errors="replace").rstrip()
        parts = title.split(None, 1)
        rid = parts[0] if parts else ""
        body_end = int(hdr_lines[k + 1]) if k + 1 < hdr_lines.shape[0] else n
        yield rid, buf[min(he + 1, n):body_end]


def load(path: str) -> np.ndarray:
    return np.fromfile(path, dtype=np.uint8)


def record_ids(path: str) -> List[str]:
    return [rid for rid, _ in _records(load(path))]


def record_lengths(path: str) -> Dict[str, int]:
    return {rid: int(np.count_nonzero(~_DROP[body])) for rid, body in _records(load(path))}


def n_runs(path: str) -> Tuple[List[str], Dict[str, List[Tuple[int, int]]]]:
    """-> (record ids in file order, {id: [(start, end), ...]} only for ids that have runs).

    Insertion order of the dict = first appearance in the file, as the reference's
    ``Ns_bed`` dict has it; a repeated id accumulates into the same list."""
    ids: List[str] = []
    runs: Dict[str, List[Tuple[int, int]]] = {}
    for rid, body in _records(load(path)):
        ids.append(rid)
        keep = ~_DROP[body]
        seq = body[keep] if not keep.all() else body
        isn = _IS_N[seq]
        if not isn.any():
            continue
        d = np.diff(isn.astype(np.int8), prepend=np.int8(0), append=np.int8(0))
        starts = np.flatnonzero(d == 1)
        ends = np.flatnonzero(d == -1)
        runs.setdefault(rid, []).extend((int(a), int(b)) for a, b in zip(starts, ends))
    return ids, runs


def write(path: str, records: List[Tuple[str, bytes]], width: int = 60) -> None:
    with open(path, "wb") as f:
        for rid, seq in records:
            f.write(b">" + rid.encode() + b"\n")
            for i in range(0, len(seq), width):
                f.write(seq[i:i + width] + b"\n")
