"""Native host-side container I/O (gci_amd/csrc/host_io.cpp through the C ABI): parallel BGZF inflate,
the BAM record-offset chase and parallel gzip framing.  The pure-Python twins in gci_amd/formats/ stay as the
readable reference of the formats (and are what the golden generator and the pysam stand-in use); these are
what the product path calls (tests/test_host_logic.py checks the two agree)."""
from __future__ import annotations

import ctypes
import os
from typing import Optional, Tuple

import numpy as np

from . import _lib
from ._lib import GciError


def _chk(st: int, what: str) -> None:
    if st != 0:
        raise GciError(st, "%s: %s" % (what, _lib.load().gci_strerror(st).decode()))


def default_threads() -> int:
    """Host threads for the native helpers: the CPUs this process may really use -- the affinity mask and, inside a
    container, the cgroup CPU quota (twice the quota: the helpers block on memory) -- at most 64."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, 2 * -(-int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(64, n))


def pick_threads(requested: int = 1) -> int:
    """`-t` is a hint (results never depend on it): the default of 1 means "all this process may use", a larger value
    is honoured up to that."""
    d = default_threads()
    return d if int(requested) <= 1 else min(int(requested), d)


def bgzf_inflate(raw, threads: int = 0, out: np.ndarray = None, check_crc: bool = False) -> np.ndarray:
    """Inflate a whole BGZF byte string (bytes / uint8 array) into one uint8 array."""
    lib = _lib.load()
    buf = np.frombuffer(raw, dtype=np.uint8) if not isinstance(raw, np.ndarray) else raw
    nb, total = ctypes.c_uint64(0), ctypes.c_uint64(0)
    _chk(lib.gci_bgzf_scan(buf.ctypes.data_as(ctypes.c_void_p), buf.shape[0], ctypes.byref(nb), ctypes.byref(total)),
         "gci_bgzf_scan")
    if out is None or out.shape[0] < total.value:
        out = np.empty(total.value, dtype=np.uint8)
    _chk(lib.gci_bgzf_inflate(buf.ctypes.data_as(ctypes.c_void_p), buf.shape[0], out.ctypes.data_as(ctypes.c_void_p),
                              out.shape[0], int(threads or default_threads()), int(check_crc)), "gci_bgzf_inflate")
    return out[:total.value]


def read_bgzf_file(path: str, threads: int = 0) -> np.ndarray:
    return bgzf_inflate(np.fromfile(path, dtype=np.uint8), threads=threads)


def bam_record_offsets(stream: np.ndarray) -> Tuple[np.ndarray, int]:
    """-> (uint64 offsets of every record's block_size word, byte offset of the first record)."""
    lib = _lib.load()
    stream = np.ascontiguousarray(stream, dtype=np.uint8)
    n, first = ctypes.c_uint64(0), ctypes.c_uint64(0)
    p = stream.ctypes.data_as(ctypes.c_void_p)
    _chk(lib.gci_bam_record_offsets(p, stream.shape[0], None, 0, ctypes.byref(n), ctypes.byref(first)),
         "gci_bam_record_offsets")
    offs = np.empty(n.value, dtype=np.uint64)
    _chk(lib.gci_bam_record_offsets(p, stream.shape[0], offs.ctypes.data_as(ctypes.c_void_p), offs.shape[0],
                                    ctypes.byref(n), ctypes.byref(first)), "gci_bam_record_offsets")
    return offs, int(first.value)


def gzip_members(text, threads: int = 0, chunk: int = 8 << 20, level: int = 1) -> bytes:
    """Multi-member gzip of `text` (bytes-like / uint8 array), members compressed in parallel."""
    lib = _lib.load()
    buf = np.frombuffer(text, dtype=np.uint8) if not isinstance(text, np.ndarray) else np.ascontiguousarray(text)
    if buf.shape[0] == 0:
        return b""
    cap = int(lib.gci_gzip_bound(buf.shape[0], chunk))
    out = np.empty(cap, dtype=np.uint8)
    n = ctypes.c_uint64(0)
    _chk(lib.gci_gzip_members(buf.ctypes.data_as(ctypes.c_void_p), buf.shape[0], chunk, level,
                              int(threads or default_threads()), out.ctypes.data_as(ctypes.c_void_p), cap,
                              ctypes.byref(n)), "gci_gzip_members")
    return out[:n.value].tobytes()


def bgzf_blocks(raw: np.ndarray, threads: Optional[int] = None, limit: Optional[int] = None) -> Tuple[np.ndarray, np.ndarray]:
    """Member table of a BGZF byte array: (uint64 byte offsets with one trailing entry = len(raw), uint64 ISIZEs).  One pass over
    the member headers by `threads` host threads (default: what the container may use; gci_bgzf_table_build).
    limit: only the members that start in front of that byte (the trailing offset = where the last of them ends;
    gci_bgzf_table_build_prefix)."""
    lib = _lib.load()
    p = raw.ctypes.data_as(ctypes.c_void_p)
    h = ctypes.c_void_p(None)
    th = int(threads if threads else default_threads())
    if limit is not None and limit < raw.shape[0]:
        _chk(lib.gci_bgzf_table_build_prefix(p, raw.shape[0], int(limit), th, ctypes.byref(h)), "gci_bgzf_table_build_prefix")
    else:
        _chk(lib.gci_bgzf_table_build(p, raw.shape[0], th, ctypes.byref(h)), "gci_bgzf_table_build")
    try:
        n = int(lib.gci_bgzf_table_count(h))
        pos = np.empty(n + 1, dtype=np.uint64)
        isz = np.empty(max(n, 1), dtype=np.uint64)
        _chk(lib.gci_bgzf_table_export(h, pos.ctypes.data_as(ctypes.c_void_p), isz.ctypes.data_as(ctypes.c_void_p)), "gci_bgzf_table_export")
    finally:
        lib.gci_bgzf_table_free(h)
    return pos, isz[:n]


def bgzf_blocks_file(path: str, threads: Optional[int] = None) -> Tuple[np.ndarray, np.ndarray]:
    """bgzf_blocks() of a file, read with pread() through a descriptor instead of a mapping (gci_bgzf_table_build_fd): no page
    fault per member, nothing of the file enters the address space."""
    import os
    lib = _lib.load()
    h = ctypes.c_void_p(None)
    fd = os.open(path, os.O_RDONLY)
    try:
        size = os.fstat(fd).st_size
        _chk(lib.gci_bgzf_table_build_fd(fd, size, int(threads if threads else default_threads()), ctypes.byref(h)), "gci_bgzf_table_build_fd")
    finally:
        os.close(fd)
    try:
        n = int(lib.gci_bgzf_table_count(h))
        pos = np.empty(n + 1, dtype=np.uint64)
        isz = np.empty(max(n, 1), dtype=np.uint64)
        _chk(lib.gci_bgzf_table_export(h, pos.ctypes.data_as(ctypes.c_void_p), isz.ctypes.data_as(ctypes.c_void_p)), "gci_bgzf_table_export")
    finally:
        lib.gci_bgzf_table_free(h)
    return pos, isz[:n]


def bam_chunk_offsets(buf: np.ndarray, start: int = 0) -> Tuple[np.ndarray, int]:
    """Record offsets of one chunk of an inflated stream -> (uint64 offsets, bytes consumed); buf[consumed:] is the
    partial record to carry into the next chunk."""
    lib = _lib.load()
    n, used = ctypes.c_uint64(0), ctypes.c_uint64(0)
    p = buf.ctypes.data_as(ctypes.c_void_p)
    _chk(lib.gci_bam_chunk_offsets(p, buf.shape[0], int(start), None, 0, ctypes.byref(n), ctypes.byref(used)),
         "gci_bam_chunk_offsets")
    offs = np.empty(n.value, dtype=np.uint64)
    _chk(lib.gci_bam_chunk_offsets(p, buf.shape[0], int(start), offs.ctypes.data_as(ctypes.c_void_p), offs.shape[0],
                                   ctypes.byref(n), ctypes.byref(used)), "gci_bam_chunk_offsets")
    return offs, int(used.value)


def fasta_titles(buf: np.ndarray, threads: int = 0) -> np.ndarray:
    """Byte offsets (uint64) of every '>' that begins a line of a FASTA byte array, in file order."""
    lib = _lib.load()
    n = ctypes.c_uint64(0)
    p = buf.ctypes.data_as(ctypes.c_void_p)
    th = int(threads or default_threads())
    pos = np.empty(1 << 16, dtype=np.uint64)
    st = lib.gci_fasta_titles(p, buf.shape[0], th, pos.ctypes.data_as(ctypes.c_void_p), pos.shape[0], ctypes.byref(n))
    if st == _lib.GCI_E_CAPACITY:                                # more records than guessed: n holds the count
        pos = np.empty(n.value, dtype=np.uint64)
        st = lib.gci_fasta_titles(p, buf.shape[0], th, pos.ctypes.data_as(ctypes.c_void_p), pos.shape[0], ctypes.byref(n))
    _chk(st, "gci_fasta_titles")
    return pos[:n.value].copy()


class BamHeads:
    """A BAM file as its heads stream (gci_bam_heads): the BAM header followed by every record without SEQ / QUAL, and
    the offset of every record.  `stream` and `offsets` are views into native memory, valid until close()."""

    def __init__(self, handle, stream: np.ndarray, offsets: np.ndarray, first_record: int):
        self._h, self.stream, self.offsets, self.first_record = handle, stream, offsets, first_record

    def close(self) -> None:
        if self._h is not None:
            self.stream = self.offsets = None
            _lib.load().gci_bam_heads_free(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        self.close()


def bam_heads(raw: np.ndarray, threads: int = 0, group_bytes: int = 0, check_crc: bool = False) -> BamHeads:
    """BGZF bytes of a BAM file -> BamHeads, inflate / record walk / compaction pipelined in native threads."""
    lib = _lib.load()
    raw = np.frombuffer(raw, dtype=np.uint8) if not isinstance(raw, np.ndarray) else raw
    h = ctypes.c_void_p(None)
    _chk(lib.gci_bam_heads(raw.ctypes.data_as(ctypes.c_void_p), raw.shape[0], int(threads or default_threads()),
                           int(group_bytes), int(check_crc), ctypes.byref(h)), "gci_bam_heads")
    nb, nr = int(lib.gci_bam_heads_bytes(h)), int(lib.gci_bam_heads_count(h))
    stream = np.ctypeslib.as_array(ctypes.cast(lib.gci_bam_heads_stream(h), ctypes.POINTER(ctypes.c_uint8)), shape=(nb,))
    if nr:
        offs = np.ctypeslib.as_array(ctypes.cast(lib.gci_bam_heads_offsets(h), ctypes.POINTER(ctypes.c_uint64)), shape=(nr,))
    else:
        offs = np.zeros(0, dtype=np.uint64)
    return BamHeads(h, stream, offs, int(lib.gci_bam_heads_first(h)))


def paf_filter(paths, targets, map_qual: int, mq_cutoff: int, iden_percent: float, threads: int = 0):
    """The PAF filter of filter() (GCI.py:211-254) in native code (gci_paf_filter): per PAF file, in command-line
    order, -> (records uint8 [n, 32] = gci_rec, names uint8 blob, int64 [n + 1] name offsets): one entry per query seen
    so far, in first-appearance order; the HQ flag carries the reference's high_qual set."""
    lib = _lib.load()
    bufs = [np.fromfile(p, dtype=np.uint8) for p in paths]
    n = len(bufs)
    ptrs = (ctypes.c_void_p * max(n, 1))(*[b.ctypes.data if b.shape[0] else None for b in bufs])
    sizes = (ctypes.c_uint64 * max(n, 1))(*[int(b.shape[0]) for b in bufs])
    tnames = [t.encode() for t in targets]
    tarr = (ctypes.c_char_p * max(len(tnames), 1))(*tnames)
    handle, line = ctypes.c_void_p(None), ctypes.c_uint64(0)
    st = lib.gci_paf_filter(ptrs, sizes, n, tarr, len(tnames), int(map_qual), int(mq_cutoff), float(iden_percent),
                            int(threads or default_threads()), ctypes.byref(handle), ctypes.byref(line))
    if st != 0:
        raise GciError(st, "gci_paf_filter: %s (line %d)" % (lib.gci_strerror(st).decode(), line.value))
    try:
        out = []
        for f in range(n):
            cnt, nb = int(lib.gci_paf_count(handle, f)), int(lib.gci_paf_name_bytes(handle, f))
            recs = np.zeros((cnt, 32), dtype=np.uint8)
            names = np.zeros(max(nb, 1), dtype=np.uint8)
            off = np.zeros(cnt + 1, dtype=np.uint64)
            _chk(lib.gci_paf_export(handle, f, recs.ctypes.data_as(ctypes.c_void_p), names.ctypes.data_as(ctypes.c_void_p),
                                    off.ctypes.data_as(ctypes.c_void_p)), "gci_paf_export")
            out.append((recs, names, off.astype(np.int64)))
        return out
    finally:
        lib.gci_paf_free(handle)
