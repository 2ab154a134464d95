"""libgci_cpu.so: the function seams of include/gci_hip.h on host memory and host threads (gci_amd/csrc/cpu/gci_cpu.cpp),
bound with ctypes over NumPy arrays.

What it is for: the CPU baseline behind the same C-ABI (bench.py: `cpu_baseline.kind = "libgci_cpu"`, every core of the
host) and running the seam tests without a GPU (tests/test_cpu_seams.py holds it against the oracle).  It is NOT a fall-back
of the product: gci_amd.pipeline / gci_amd.cli / GCI.py never import this module -- without an MI355X they refuse.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from ctypes import POINTER, c_char_p, c_double, c_int, c_int32, c_int64, c_size_t, c_uint32, c_uint64, c_void_p
from typing import List, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(_HERE, "csrc", "cpu", "gci_cpu.cpp")
LIB_PATH = os.path.join(_HERE, "csrc", "libgci_cpu.so")

REC_DTYPE = np.dtype([("name_hash", "<u8"), ("contig", "<i4"), ("start", "<i4"), ("end", "<i4"), ("qlen", "<i4"),
                      ("rec_idx", "<u4"), ("mapq", "u1"), ("flags", "u1"), ("name_len", "<u2")])
IVL_DTYPE = np.dtype([("contig", "<i4"), ("start", "<i4"), ("end", "<i4"), ("pad", "<i4")])
GCI_TILE = 4096


class CpuError(RuntimeError):
    def __init__(self, status: int, msg: str, rec: int = -1):
        super().__init__(msg)
        self.status, self.rec = status, rec


class _JoinFile(ctypes.Structure):
    _fields_ = [("d_recs", c_void_p), ("n_recs", c_uint32), ("name_delta", c_uint32), ("d_name_base", c_void_p), ("d_name_off", c_void_p)]


class _Window(ctypes.Structure):
    _fields_ = [("begin", c_int64), ("end", c_int64)]


# the seam set: the same names and argument lists as in gci_amd/_lib.py (include/gci_hip.h), + gci_cpu_option
EXPORTS = [
    ("gci_abi_version", c_int, []),
    ("gci_ctx_create", c_int, [c_int, c_void_p, c_int, POINTER(c_void_p)]),
    ("gci_ctx_destroy", c_int, [c_void_p]),
    ("gci_sync", c_int, [c_void_p]),
    ("gci_strerror", c_char_p, [c_int]),
    ("gci_last_error", c_char_p, [c_void_p]),
    ("gci_malloc", c_int, [c_void_p, c_size_t, POINTER(c_void_p)]),
    ("gci_free", c_int, [c_void_p, c_void_p]),
    ("gci_memcpy_h2d", c_int, [c_void_p, c_void_p, c_void_p, c_size_t]),
    ("gci_memcpy_d2h", c_int, [c_void_p, c_void_p, c_void_p, c_size_t]),
    ("gci_memset", c_int, [c_void_p, c_void_p, c_int, c_size_t]),
    ("gci_layout_set", c_int, [c_void_p, c_int32, c_void_p]),
    ("gci_layout_total", c_int64, [c_void_p]),
    ("gci_layout_offsets", c_int, [c_void_p, c_void_p]),
    ("gci_name_hash", c_uint64, [c_void_p, c_uint32]),
    ("gci_decode_status", c_int, [c_uint64, POINTER(c_uint32)]),
    ("gci_bam_filter", c_int, [c_void_p, c_void_p, c_uint64, c_void_p, c_uint32, c_void_p, c_int32, c_int, c_int, c_double, c_double,
                               c_uint32, c_void_p, c_void_p]),
    ("gci_name_join", c_int, [c_void_p, c_void_p, c_int, c_double, c_void_p, c_void_p, c_uint32, c_void_p, c_void_p]),
    ("gci_depth_build", c_int, [c_void_p, c_void_p, c_void_p, c_uint32, c_int, c_void_p]),
    ("gci_gap_mask", c_int, [c_void_p, c_void_p, c_void_p, c_uint32]),
    ("gci_max2", c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    ("gci_issue_scan", c_int, [c_void_p, c_void_p, c_double, c_double, c_int, c_void_p, c_uint32, c_void_p]),
    ("gci_issue_scan_windows", c_int, [c_void_p, c_void_p, c_void_p, c_uint32, c_double, c_double, c_void_p, c_uint32, c_void_p]),
    ("gci_depth_text_size", c_int, [c_void_p, c_void_p, c_void_p]),
    ("gci_depth_text_write", c_int, [c_void_p, c_void_p, c_void_p, c_uint64]),
    ("gci_depth_sum", c_int, [c_void_p, c_void_p, c_void_p]),
    ("gci_range_sums", c_int, [c_void_p, c_void_p, c_void_p, c_uint64, c_void_p]),
    ("gci_cpu_option", c_int, [c_void_p, c_char_p, c_int]),
    ("gci_depth_deflate_from_build", c_int, [c_void_p, c_void_p]),
    ("gci_depth_deflate_size", c_int, [c_void_p] * 4 + [c_uint32] + [c_void_p] * 4),
    ("gci_depth_deflate_write", c_int, [c_void_p] * 4 + [c_uint32] + [c_void_p] * 5 + [c_uint64]),
]


def needs_build() -> bool:
    return not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < max(
        os.path.getmtime(p) for p in (SRC, os.path.join(_HERE, "csrc", "gci_common.h"), os.path.join(_HERE, "..", "include", "gci_hip.h")))


def build(force: bool = False) -> str:
    """g++ over the one source file: include/gci_hip.h a second time."""
    if force or needs_build():
        subprocess.run(["g++", "-O3", "-std=c++17", "-shared", "-fPIC", "-pthread", "-fno-fast-math", "-ffp-contract=off", "-Wall", "-o", LIB_PATH, SRC], check=True)
    return LIB_PATH


_LIB = None


def load():
    global _LIB
    if _LIB is None:
        if needs_build():
            build()
        lib = ctypes.CDLL(LIB_PATH)
        for name, res, args in EXPORTS:
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _LIB = lib
    return _LIB


def _p(a: Optional[np.ndarray]):
    return None if a is None else ctypes.c_void_p(a.ctypes.data)


class CpuEngine:
    """The seam calls over NumPy arrays (host memory is this library's "device" memory)."""

    def __init__(self, threads: Optional[int] = None):
        self.lib = load()
        h = c_void_p()
        self._chk(self.lib.gci_ctx_create(0, None, 0, ctypes.byref(h)), "gci_ctx_create")
        self.ctx = h
        if threads:
            self.lib.gci_cpu_option(self.ctx, b"threads", int(threads))
        self.lengths: List[int] = []
        self.offsets = np.zeros(0, dtype=np.int64)
        self.total = 0

    def close(self):
        if self.ctx:
            self.lib.gci_ctx_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def threads(self) -> int:
        return int(self.lib.gci_cpu_option(self.ctx, b"threads", 0))

    def heads(self, on: bool):
        """gci_bam_filter over a heads stream (records without SEQ / QUAL) from now on."""
        self.lib.gci_cpu_option(self.ctx, b"heads", 1 if on else 0)

    def _chk(self, st: int, what: str):
        if st != 0:
            raise CpuError(st, "%s: %s" % (what, self.lib.gci_strerror(st).decode()))

    def _status(self, word: np.ndarray, what: str):
        rec = c_uint32()
        st = self.lib.gci_decode_status(int(word[0]), ctypes.byref(rec))
        if st != 0:
            raise CpuError(st, "%s: %s at record %d" % (what, self.lib.gci_strerror(st).decode(), rec.value), rec.value)

    def set_layout(self, lengths: Sequence[int]):
        self.lengths = [int(l) for l in lengths]
        a = np.asarray(self.lengths, dtype=np.int64)
        self._chk(self.lib.gci_layout_set(self.ctx, len(self.lengths), _p(a)), "gci_layout_set")
        self.total = int(self.lib.gci_layout_total(self.ctx))
        self.offsets = np.zeros(len(self.lengths), dtype=np.int64)
        self._chk(self.lib.gci_layout_offsets(self.ctx, _p(self.offsets)), "gci_layout_offsets")

    def new_track(self) -> np.ndarray:
        return np.zeros(self.total, dtype=np.int32)

    def contig(self, track: np.ndarray, c: int) -> np.ndarray:
        o = int(self.offsets[c])
        return track[o:o + self.lengths[c]]

    # ---- R1
    def bam_filter(self, stream: np.ndarray, offs: np.ndarray, ref_sel: np.ndarray, mq: int, cut: int, cp: float, ip: float,
                   rec_idx_base: int = 0, check: bool = True) -> np.ndarray:
        stream = np.ascontiguousarray(stream, dtype=np.uint8)
        offs = np.ascontiguousarray(offs, dtype=np.uint64)
        ref_sel = np.ascontiguousarray(ref_sel, dtype=np.int32)
        out = np.zeros(offs.shape[0], dtype=REC_DTYPE)
        self.last_status = np.zeros(1, dtype=np.uint64)
        self._chk(self.lib.gci_bam_filter(self.ctx, _p(stream), stream.shape[0], _p(offs), offs.shape[0], _p(ref_sel), ref_sel.shape[0], mq, cut, cp, ip,
                                          rec_idx_base, _p(out), _p(self.last_status)), "gci_bam_filter")
        if check:
            self._status(self.last_status, "gci_bam_filter")
        return out

    # ---- R5
    def name_join(self, files: Sequence[Tuple[np.ndarray, np.ndarray, np.ndarray, int]], ovlp: float,
                  contig_map: Optional[np.ndarray] = None) -> np.ndarray:
        """files: (records, name base bytes, name offsets by record position, name_delta) in reference order -> intervals."""
        arr = (_JoinFile * len(files))()
        keep = []
        cap = 0
        for k, (recs, base, off, delta) in enumerate(files):
            recs = np.ascontiguousarray(recs, dtype=REC_DTYPE)
            base = np.ascontiguousarray(base, dtype=np.uint8)
            off = np.ascontiguousarray(off, dtype=np.uint64)
            keep += [recs, base, off]
            arr[k] = _JoinFile(recs.ctypes.data, recs.shape[0], int(delta), base.ctypes.data, off.ctypes.data)
            cap += int(recs.shape[0])
        out = np.zeros(max(cap, 1), dtype=IVL_DTYPE)
        n = np.zeros(1, dtype=np.uint32)
        st = np.zeros(1, dtype=np.uint64)
        cm = None if contig_map is None else np.ascontiguousarray(contig_map, dtype=np.int32)
        self._chk(self.lib.gci_name_join(self.ctx, arr, len(files), ovlp, _p(cm), _p(out), out.shape[0], _p(n), _p(st)), "gci_name_join")
        self._status(st, "gci_name_join")
        return out[:int(n[0])]

    # ---- R6, R8, R9
    def depth_build(self, ivl: np.ndarray, flank: int, track: Optional[np.ndarray] = None) -> np.ndarray:
        ivl = np.ascontiguousarray(ivl, dtype=IVL_DTYPE)
        track = self.new_track() if track is None else track
        self._chk(self.lib.gci_depth_build(self.ctx, _p(ivl), None, ivl.shape[0], flank, _p(track)), "gci_depth_build")
        return track

    def gap_mask(self, track: np.ndarray, gaps: Sequence[Tuple[int, int, int]]):
        g = np.zeros(len(gaps), dtype=IVL_DTYPE)
        for k, (c, a, b) in enumerate(gaps):
            g[k] = (c, a, b, 0)
        self._chk(self.lib.gci_gap_mask(self.ctx, _p(track), _p(g), g.shape[0]), "gci_gap_mask")

    def max2(self, a: np.ndarray, b: np.ndarray) -> np.ndarray:
        out = self.new_track()
        self._chk(self.lib.gci_max2(self.ctx, _p(a), _p(b), _p(out)), "gci_max2")
        return out

    # ---- R10
    def issue_keys(self, track: np.ndarray, lo: float, hi: float, flank: int, windows: Optional[Sequence[Tuple[int, int]]] = None) -> np.ndarray:
        cap = 1 << 12
        while True:
            keys = np.zeros(cap, dtype=np.uint64)
            n = np.zeros(1, dtype=np.uint32)
            if windows is None:
                self._chk(self.lib.gci_issue_scan(self.ctx, _p(track), lo, hi, flank, _p(keys), cap, _p(n)), "gci_issue_scan")
            else:
                w = (_Window * len(windows))(*[_Window(int(a), int(b)) for a, b in windows])
                self._chk(self.lib.gci_issue_scan_windows(self.ctx, _p(track), w, len(windows), lo, hi, _p(keys), cap, _p(n)), "gci_issue_scan_windows")
            if int(n[0]) <= cap:
                return np.sort(keys[:int(n[0])])
            cap = int(n[0])

    def issue_runs(self, track: np.ndarray, lo: float, hi: float, flank: int, n_windows: Optional[int] = None,
                   windows: Optional[Sequence[Tuple[int, int]]] = None) -> List[List[Tuple[int, int]]]:
        """Sorted keys -> per window the runs (start, end) relative to the window's beginning."""
        keys = self.issue_keys(track, lo, hi, flank, windows)
        nw = len(self.lengths) if windows is None else len(windows)
        runs: List[List[Tuple[int, int]]] = [[] for _ in range(nw)]
        for k in range(0, keys.shape[0], 2):
            a, b = int(keys[k]), int(keys[k + 1])
            assert (a >> 33) == (b >> 33) and not (a & 1) and (b & 1)
            runs[a >> 33].append(((a >> 1) & 0xFFFFFFFF, (b >> 1) & 0xFFFFFFFF))
        return runs

    # ---- R7, R15
    def depth_text(self, track: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
        off = np.zeros(len(self.lengths) + 1, dtype=np.uint64)
        self._chk(self.lib.gci_depth_text_size(self.ctx, _p(track), _p(off)), "gci_depth_text_size")
        out = np.zeros(max(int(off[-1]), 1), dtype=np.uint8)
        self._chk(self.lib.gci_depth_text_write(self.ctx, _p(track), _p(out), out.shape[0]), "gci_depth_text_write")
        return out[:int(off[-1])], off

    MEMBER_BASES = 64 * 4096

    def depth_deflate(self, track: np.ndarray) -> List[bytes]:
        """-> per contig the bytes of the gzip members whose payload is its depth lines (device.Engine.depth_deflate's twin)."""
        elem, cnt, first = [], [], [0]
        for off, length in zip(self.offsets.tolist(), self.lengths):
            for g in range(0, int(length), self.MEMBER_BASES):
                elem.append(int(off) + g)
                cnt.append(min(self.MEMBER_BASES, int(length) - g))
            first.append(len(elem))
        nm = len(elem)
        if nm == 0:
            return [b"" for _ in self.lengths]
        d_elem, d_cnt = np.asarray(elem, dtype=np.uint64), np.asarray(cnt, dtype=np.uint32)
        tile_bytes = np.zeros(nm * 64, dtype=np.uint32)
        mb, crc, isz = (np.zeros(nm, dtype=np.uint32) for _ in range(3))
        self._chk(self.lib.gci_depth_deflate_size(self.ctx, _p(track), _p(d_elem), _p(d_cnt), nm, _p(tile_bytes), _p(mb), _p(crc), _p(isz)),
                  "gci_depth_deflate_size")
        offs = np.zeros(nm + 1, dtype=np.uint64)
        np.cumsum(mb.astype(np.uint64), out=offs[1:])
        out = np.zeros(int(offs[nm]), dtype=np.uint8)
        self._chk(self.lib.gci_depth_deflate_write(self.ctx, _p(track), _p(d_elem), _p(d_cnt), nm, _p(tile_bytes), _p(crc), _p(isz), _p(offs),
                                                   _p(out), out.shape[0]), "gci_depth_deflate_write")
        return [out[int(offs[first[c]]):int(offs[first[c + 1]])].tobytes() for c in range(len(self.lengths))]

    def depth_sum(self, track: np.ndarray) -> np.ndarray:
        s = np.zeros(len(self.lengths), dtype=np.int64)
        self._chk(self.lib.gci_depth_sum(self.ctx, _p(track), _p(s)), "gci_depth_sum")
        return s

    def range_sums(self, track: np.ndarray, ranges: np.ndarray) -> np.ndarray:
        r = np.ascontiguousarray(ranges, dtype=np.int64).reshape(-1, 2)
        s = np.zeros(r.shape[0], dtype=np.int64)
        self._chk(self.lib.gci_range_sums(self.ctx, _p(track), _p(r), r.shape[0], _p(s)), "gci_range_sums")
        return s
