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
    """The file's bytes.  A large file (a 3 GB assembly) is read by eight threads, each its share straight into the array
    (np.fromfile alone: 0.55 s of the command line at genome size)."""
    import os
    n = os.path.getsize(path)
    if n < (64 << 20):
        return np.fromfile(path, dtype=np.uint8)
    from concurrent.futures import ThreadPoolExecutor
    buf = np.empty(n, dtype=np.uint8)
    parts = 8
    step = -(-n // parts)

    def read(k):
        a, b = k * step, min(n, (k + 1) * step)
        with open(path, "rb", buffering=0) as f:
            f.seek(a)
            mv = memoryview(buf)[a:b]
            got = 0
            while got < b - a:
                r = f.readinto(mv[got:])
                if not r:
                    raise IOError("short read of %s" % path)
                got += r

    with ThreadPoolExecutor(parts) as ex:
        list(ex.map(read, range(parts)))
    return buf


def record_ids(path: str) -> List[str]:
    return [rid for rid, _ in _records(load(path))]


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


def record_spans(buf: np.ndarray) -> List[Tuple[str, int, int]]:
    """[(id, body begin, body end)] byte offsets of every record, in file order: the body runs from behind the title
    line to the next title line (or the end of the file)."""
    n = int(buf.shape[0])
    gt = np.flatnonzero(buf == ord(">"))
    starts = [int(p) for p in gt if p == 0 or buf[p - 1] == 10]
    out = []
    for k, hs in enumerate(starts):
        stop = starts[k + 1] if k + 1 < len(starts) else n
        nl = np.flatnonzero(buf[hs:stop] == 10)
        he = hs + int(nl[0]) if nl.shape[0] else stop
        parts = bytes(buf[hs + 1:he]).decode(errors="replace").rstrip().split(None, 1)
        out.append((parts[0] if parts else "", min(he + 1, stop), stop))
    return out


_INDEXED = None           # (key, bytes of the file, record spans) of the file read last: the command line asks for the
                          # record ids first (GCI.py:939-941) and for the N runs afterwards (GCI.py:29-35)


def indexed(path: str):
    """-> (file bytes, [(id, body begin, body end)]), title lines found by the native helper (gci_fasta_titles); the
    result for the last file is kept until n_runs_device() has used it."""
    global _INDEXED
    import os
    from .. import hostio
    st = os.stat(path)
    key = (os.path.realpath(path), st.st_size, st.st_mtime_ns)
    if _INDEXED is not None and _INDEXED[0] == key:
        return _INDEXED[1], _INDEXED[2]
    # a large assembly is looked at through a mapping of the file (the title lines are found by threads that fault the page
    # cache's pages in sixteen at a time; the bytes go to the device through the pinned ring straight from it) instead of being
    # read into 3 GB of fresh memory first (0.34 - 0.55 s of the command line at genome size); GCI_FASTA_LOAD=read: as before
    if st.st_size >= (64 << 20) and os.environ.get("GCI_FASTA_LOAD", "mmap") == "mmap":
        buf = np.memmap(path, dtype=np.uint8, mode="r")
    else:
        buf = load(path)
    n = int(buf.shape[0])
    starts = [int(p) for p in hostio.fasta_titles(buf)]
    spans = []
    for k, hs in enumerate(starts):
        stop = starts[k + 1] if k + 1 < len(starts) else n
        he, step = -1, 4096
        while he < 0:                                       # the end of the title line, looked for in growing windows
            nl = np.flatnonzero(buf[hs:min(hs + step, stop)] == 10)
            if nl.shape[0]:
                he = hs + int(nl[0])
            elif hs + step >= stop:
                he = stop
            step *= 16
        parts = bytes(buf[hs + 1:he]).decode(errors="replace").rstrip().split(None, 1)
        spans.append((parts[0] if parts else "", min(he + 1, stop), stop))
    _INDEXED = (key, buf, spans)
    return buf, spans


def record_ids_indexed(path: str) -> List[str]:
    return [rid for rid, _, _ in indexed(path)[1]]


def n_runs_device(engine, path: str) -> Tuple[List[str], Dict[str, List[Tuple[int, int]]]]:
    """n_runs() with the scan on the GPU (gci_fasta_n_scan): the file's bytes are uploaded as they are; the device
    returns the byte offsets where runs of N / n begin and end and how many bytes of every 4096-byte tile count as
    sequence; the handful of offsets is turned into sequence coordinates here."""
    global _INDEXED
    buf, spans = indexed(path)
    _INDEXED = None
    ids = [rid for rid, _, _ in spans]
    if not spans:
        return ids, {}
    bodies = np.array([(b, e) for _, b, e in spans], dtype=np.int64)
    kept, keys = engine.fasta_n_scan(buf, bodies)
    before_tile = np.concatenate(([0], np.cumsum(kept, dtype=np.int64)))

    def coord(off: int, r: int) -> int:
        """sequence bytes of the file in front of byte `off` (which lies in, or at the end of, the body of record r)"""
        t0 = (off // 4096) * 4096
        c = int(before_tile[off // 4096])
        lo = t0
        # the part of the tile in front of `off`: only bytes inside record bodies count
        q = r
        while q >= 0 and bodies[q, 1] > lo:
            a, b = max(int(bodies[q, 0]), lo), min(int(bodies[q, 1]), off)
            if b > a:
                c += int(np.count_nonzero(~_DROP[buf[a:b]]))
            q -= 1
        return c

    runs: Dict[str, List[Tuple[int, int]]] = {}
    rec_of = np.searchsorted(bodies[:, 0], (keys >> np.uint64(1)).astype(np.int64), side="right") - 1
    # keys are sorted by byte offset, hence grouped by record: one cut per record (not one mask over all keys per record),
    # and only the records that hold a run are visited
    cut = np.searchsorted(rec_of, np.arange(len(spans) + 1))
    for r in np.unique(rec_of).tolist():
        rid, b, e = spans[r]
        mine = keys[cut[r]:cut[r + 1]]
        base = coord(b, r)
        pos = [coord(int(k >> np.uint64(1)), r) - base for k in mine]
        if len(pos) & 1:                                   # the run reaches the end of the record
            pos.append(coord(e, r) - base)
        runs.setdefault(rid, []).extend((pos[i], pos[i + 1]) for i in range(0, len(pos), 2))
    return ids, runs


def write(path: str, records: List[Tuple[str, bytes]], width: int = 60) -> None:
    with open(path, "wb") as f:
        for rid, seq in records:
            f.write(b">" + rid.encode() + b"\n")
            for i in range(0, len(seq), width):
                f.write(seq[i:i + width] + b"\n")
