"""Device side of the path: one `Engine` per GPU wrapping a gci_ctx (include/gci_hip.h).

HBM buffers, streams and events come from a provider (gci_amd/hbm.py): the library's own `gci_dev_*` exports for the single-GPU
command line (no `import torch` in that process), or torch for contig-sharded runs and the test fixtures -- plumbing either way:
every per-record and per-base operation is a hand-written gfx950 kernel reached through the C ABI.  Nothing here computes on
the CPU; a missing library or GPU raises.
"""
from __future__ import annotations

import ctypes
import os
import threading
from concurrent.futures import ThreadPoolExecutor
import warnings
from dataclasses import dataclass
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import _lib, hbm
from ._lib import BuildOpts, GciError, JoinFile, Window

REC_DTYPE = np.dtype([("name_hash", "<u8"), ("contig", "<i4"), ("start", "<i4"), ("end", "<i4"), ("qlen", "<i4"),
                      ("rec_idx", "<u4"), ("mapq", "u1"), ("flags", "u1"), ("name_len", "<u2")])
assert REC_DTYPE.itemsize == 32
IVL_DTYPE = np.dtype([("contig", "<i4"), ("start", "<i4"), ("end", "<i4"), ("pad", "<i4")])

_M64 = (1 << 64) - 1
Buffer = Any                  # a device buffer of the engine's provider: hbm.Buf or Buffer


def name_hash_np(names: Sequence[bytes]) -> np.ndarray:
    """Vectorised twin of gci_name_hash (gci_amd/csrc/gci_common.h) for host-built records."""
    n = len(names)
    lens = np.fromiter((len(x) for x in names), dtype=np.int64, count=n)
    width = int(((lens.max() if n else 0) + 7) // 8 * 8) or 8
    buf = np.zeros((n, width), dtype=np.uint8)
    for i, x in enumerate(names):
        buf[i, :len(x)] = np.frombuffer(x, dtype=np.uint8)
    words = buf.view("<u8")

    def mix(x):
        x = x ^ (x >> np.uint64(30))
        x = x * np.uint64(0xbf58476d1ce4e5b9)
        x = x ^ (x >> np.uint64(27))
        x = x * np.uint64(0x94d049bb133111eb)
        return x ^ (x >> np.uint64(31))

    acc = np.zeros(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        for k in range(width // 8):
            use = (k * 8) < lens
            key = (0x9E3779B97F4A7C15 * (k + 1)) & _M64
            mul = np.uint64(((key >> 1) ^ 0xbf58476d1ce4e5b9) | 1)
            acc = acc + np.where(use, (words[:, k] ^ np.uint64(key)) * mul, np.uint64(0))
        return mix(acc ^ (lens.astype(np.uint64) * np.uint64(0xD6E8FEB86659FD93)))


def _forget_pages(raw, lo: int, hi: int) -> None:
    """The pages [lo, hi) of a memory-mapped input file have been read for the last time BY THE CPU: drop their entries from this
    process's page table (madvise DONTNEED; the page cache keeps the data), so that the final unmap of a 77 GB file does not
    tear down 19 M entries at once under the address-space lock (every hipMalloc and every new mapping of the process waits
    for that: the join's scratch behind the second whole-genome file was measured at 3 s of wall time for 1.5 ms of kernels).
    ONLY for ranges the GPU driver never saw: a pageable host-to-device copy registers the user pages with the driver, and a
    madvise / munmap over registered pages goes through its MMU notifier, which evicts the process's GPU queues while it
    invalidates -- measured: every kernel of the run 1.4 - 8 x slower, the command line at 1/4 genome 9.2 s instead of 5.2 s
    (profiles/r04j_forget_pages_ab.txt).  pipeline._RunUploads therefore stages the file's bytes through pinned memory itself."""
    import mmap as _mmap
    mm = getattr(raw, "_mmap", None)
    if mm is None or hi <= lo or not hasattr(mm, "madvise") or os.environ.get("GCI_FORGET_PAGES", "1") == "0":
        return
    a = lo // _mmap.PAGESIZE * _mmap.PAGESIZE
    try:
        mm.madvise(_mmap.MADV_DONTNEED, a, hi - a)
    except (OSError, ValueError, AttributeError):
        pass


class _Staging:
    """A ring of pinned host buffers through which the bytes of a memory-mapped file travel to the device: host threads copy a
    piece of the mapping into a slot (parallel memcpy out of the page cache: the page faults are theirs, not the copy engine's),
    the slot leaves by DMA on the copy stream, and is reused once that copy's event has passed.  One per engine, made once."""
    SLOT = int(os.environ.get("GCI_STAGING_SLOT_MB", "64")) << 20    # (page-locking costs ~0.3 s per GB on these hosts: a ring of 256 MB, not of 1.2 GB)
    SLOTS = int(os.environ.get("GCI_STAGING_SLOTS", "4"))
    # (copying threads: 6, 12 and 24 bring a tmpfs file to the device at 46 - 47 GB/s inside the command line, 48 at 57 GB/s -- the
    #  link's own rate -- on the 256-thread hosts of the pool: profiles/r06e_staging_threads.txt; a quarter of the host's threads, at most 48)
    # As one rank of N on a node (LOCAL_WORLD_SIZE, set by `GCI.py --gpus N` and by torch.distributed.run) a ring takes its share of that.
    _RANKS = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1") or 1)) if "RANK" in os.environ else 1
    THREADS = int(os.environ.get("GCI_STAGING_THREADS", str(min(48, max(4, (os.cpu_count() or 24) // (4 * _RANKS))))))

    def __init__(self, engine):
        # the loop over the slots is the library's (staging.cpp: gci_stage_send)
        self.engine = engine
        self._fds = {}
        h = ctypes.c_void_p()
        engine._chk(engine.lib.gci_stage_create(engine.ctx, self.SLOT, self.SLOTS, self.THREADS, ctypes.byref(h)), "gci_stage_create")
        self.native = h

    def close(self):
        """The ring's threads, pinned slots and events (gci_stage_free), and the descriptors the pread path kept open."""
        if self.native is not None:
            self.engine.lib.gci_stage_free(self.native)
            self.native = None
        for fd in self._fds.values():
            try:
                os.close(fd)
            except OSError:
                pass
        self._fds.clear()

    def send(self, raw, p0: int, p1: int, dst: Buffer, stream, urgent: bool = True) -> None:
        """raw[p0:p1] -> dst[:p1 - p0] (device), enqueued on `stream`; returns when the last piece is enqueued.  A sender that is not
        urgent (the assembly, whose N runs nobody waits for) lets the urgent ones (the runs of a BAM file: the device inflates them
        as they arrive) go first, piece by piece."""
        if self.native is not None:
            if p1 <= p0:
                return
            path = getattr(raw, "filename", None)
            if path is not None and os.environ.get("GCI_STAGING_READ", "mmap") == "pread":
                # the file's bytes by pread() into the slots (one descriptor per file, kept): no page faults on the mapping
                fd = self._fds.get(path)
                if fd is None:
                    fd = self._fds[path] = os.open(path, os.O_RDONLY)
                # where raw[0] lies in the file: a SLICE of a memmap keeps its parent's .offset (numpy: m[50:].offset == 0), so the
                # position comes from the addresses -- first byte of this array minus first byte of the mapping's own array
                whole = raw
                while isinstance(getattr(whole, "base", None), np.ndarray):
                    whole = whole.base
                base = int(getattr(whole, "offset", 0)) + (raw.ctypes.data - whole.ctypes.data)
                self.engine._chk(self.engine.lib.gci_stage_send_fd(self.engine.ctx, self.native, fd, base + p0, p1 - p0, ctypes.c_void_p(dst.data_ptr()),
                                                                   ctypes.c_void_p(stream.cuda_stream), 1 if urgent else 0), "gci_stage_send_fd")
                return
            src = raw.ctypes.data + p0                      # (a numpy array / memmap of uint8: its bytes as they lie)
            forget = 1 if (getattr(raw, "_mmap", None) is not None and os.environ.get("GCI_FORGET_PAGES", "1") != "0") else 0
            self.engine._chk(self.engine.lib.gci_stage_send(self.engine.ctx, self.native, ctypes.c_void_p(src), p1 - p0, ctypes.c_void_p(dst.data_ptr()),
                                                            ctypes.c_void_p(stream.cuda_stream), forget, 1 if urgent else 0), "gci_stage_send")
            return
        raise GciError(_lib.GCI_E_INVALID, "the staging ring of this engine has been closed")


@dataclass
class JoinInput:
    """One file of the join: compact records on the device + where its name bytes are."""
    recs: Buffer            # uint8 [n, 32]
    name_base: Buffer       # uint8 blob (BAM: the inflated stream)
    name_off: Buffer        # int64 [*], indexed by rec_idx
    name_delta: int               # 36 for BAM records, 0 for a names blob


@dataclass
class Pages:
    """Record pages of one alignment file on the device (gci_bam_pages_write)."""
    buf: Buffer             # uint8: pages | blob | 16 zero bytes
    n_pages: int
    page_bytes: int
    blob_off: int
    n_rec: int


class Engine:
    """A gci_ctx bound to a stream of its provider (hbm.py) on `device`."""

    def __init__(self, device: int = 0, stream: Optional[Any] = None, backend: Optional[str] = None):
        """`stream`: the provider's stream this context enqueues on (default: the device's current stream).  `backend`: "native"
        (the library's own gci_dev_* exports: no torch in the process) or "torch"; default hbm.provider()'s rule."""
        self.lib = _lib.load()
        self.T = T = hbm.provider_of(stream) if (stream is not None and backend is None) else hbm.provider(backend)
        if not T.is_available():
            raise GciError(_lib.GCI_E_HIP, "no MI355X visible: the HIP path has no CPU fallback")
        self.device = T.device(device)
        T.set_device(self.device)
        self.stream = stream if stream is not None else T.current_stream(self.device)
        from . import phases
        phases.set_provider(T)
        h = ctypes.c_void_p()
        st = self.lib.gci_ctx_create(device, ctypes.c_void_p(self.stream.cuda_stream), 0, ctypes.byref(h))
        if st != 0:
            raise GciError(st, "gci_ctx_create: %s" % self.lib.gci_strerror(st).decode())
        self.ctx = h
        from . import HW_QUEUES_OK
        if HW_QUEUES_OK:                          # (the runtime has hardware queues to spare: the inflate may take a second stream)
            self.lib.gci_bgzf_inflate_streams(h, 2)
        self.lengths: List[int] = []
        self.offsets: List[int] = []
        self.total = 0
        self._status = self.T.zeros(1, self.T.int64, self.device)
        self._count = self.T.zeros(1, self.T.int32, self.device)
        self._members_host = None                 # pinned staging buffer of depth_deflate()
        self._staging = None
        self._copy_stream = None
        self._lock = threading.Lock()
        m = os.environ.get("GCI_JOIN", "")
        self.join_mode = 1 if m.startswith("c") else 2 if m.startswith("p") else 0    # what gci_ctx_create read

    def close(self) -> None:
        if getattr(self, "ctx", None):
            st, self._staging = getattr(self, "_staging", None), None
            if st is not None and getattr(st, "engine", None) is self:     # (a side engine borrows the main engine's ring)
                st.close()                                # before the context it was made with goes
            self.lib.gci_ctx_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------------------------------
    def _chk(self, st: int, what: str) -> None:
        if st != 0:
            detail = self.lib.gci_last_error(self.ctx).decode() if st == _lib.GCI_E_HIP else ""
            raise GciError(st, "%s: %s %s" % (what, self.lib.gci_strerror(st).decode(), detail))

    def sync(self) -> None:
        self._chk(self.lib.gci_sync(self.ctx), "gci_sync")

    @staticmethod
    def _p(t: Optional[Buffer]):
        return ctypes.c_void_p(t.data_ptr()) if t is not None else None

    def to_device(self, a: np.ndarray) -> Buffer:
        return self.T.from_numpy(a, self.device)

    def _host_src(self, a) -> Any:
        """A host array as the source of a buffer's copy_(): numpy for the native provider, a tensor over the same memory for torch."""
        a = np.asarray(a)
        if self.T.name == "torch":
            return self.T.t.from_numpy(a)
        return a

    def upload_staged(self, a: np.ndarray) -> Buffer:
        """A large host array (the 3 GB assembly) to the device through the ring of pinned buffers (`_Staging`) instead of one
        pageable copy: host threads fill a slot while the one before it crosses PCIe.  Ordered before what this engine's stream
        does next."""
        a = np.ascontiguousarray(a).reshape(-1).view(np.uint8)
        n = int(a.shape[0])
        staging, copy = self.staging(), self.copy_stream()
        dst = self.T.empty(max(n, 1), self.T.uint8, self.device)
        copy.wait_stream(self.stream)
        staging.send(a, 0, n, dst, copy, urgent=os.environ.get("GCI_FASTA_URGENT", "0") == "1")
        self.stream.wait_stream(copy)
        return dst[:n]

    def staging(self) -> "_Staging":
        """The engine's ring of pinned buffers (made on first use; a helper thread may be the first)."""
        with self._lock:
            if self._staging is None:
                self._staging = _Staging(self)
            return self._staging

    def copy_stream(self) -> Any:
        with self._lock:
            if self._copy_stream is None:
                self._copy_stream = self.T.Stream(self.device)
            return self._copy_stream

    # ---- per-kernel HIP-event timing (library side, on the ctx stream) -------------------------
    def profile_enable(self, mask: int) -> None:
        self._chk(self.lib.gci_profile_enable(self.ctx, int(mask)), "gci_profile_enable")

    def profile_read(self, reset: bool = True) -> Dict[str, Tuple[float, int]]:
        """-> {kernel name: (total ms, launches)} accumulated since the last reset."""
        out = {}
        for k in range(_lib.PROF_COUNT):
            ms, n = ctypes.c_double(0), ctypes.c_uint64(0)
            self._chk(self.lib.gci_profile_read(self.ctx, k, ctypes.byref(ms), ctypes.byref(n), int(reset)),
                      "gci_profile_read")
            if n.value:
                out[self.lib.gci_profile_name(k).decode()] = (ms.value, int(n.value))
        return out

    # ---- layout ------------------------------------------------------------------------------
    def set_layout(self, lengths: Sequence[int]) -> List[int]:
        arr = np.asarray(lengths, dtype=np.int64)
        self._chk(self.lib.gci_layout_set(self.ctx, arr.shape[0], arr.ctypes.data_as(ctypes.c_void_p)), "gci_layout_set")
        self.lengths = [int(x) for x in arr]
        offs = np.zeros(arr.shape[0], dtype=np.int64)
        self._chk(self.lib.gci_layout_offsets(self.ctx, offs.ctypes.data_as(ctypes.c_void_p)), "gci_layout_offsets")
        self.offsets = [int(x) for x in offs]
        self.total = int(self.lib.gci_layout_total(self.ctx))
        return self.offsets

    def new_track(self) -> Buffer:
        return self.T.empty(max(self.total, 1), self.T.int32, self.device)

    # ---- R1 ----------------------------------------------------------------------------------
    def bam_filter(self, d_bam: Buffer, d_rec_off: Buffer, d_ref_sel: Buffer, map_qual: int,
                   mq_cutoff: int, clip_percent: float, iden_percent: float, out: Optional[Buffer] = None,
                   check: bool = True, rec_idx_base: int = 0) -> Buffer:
        """K1 over the whole inflated stream (the round-1 / 2 kernel, GCI_K1=stream; the product runs bam_pages +
        bam_filter_pages)."""
        n = int(d_rec_off.shape[0])
        if out is None:
            out = self.T.empty((max(n, 1), 32), self.T.uint8, self.device)
        st = self.lib.gci_bam_filter(self.ctx, self._p(d_bam), int(d_bam.shape[0]), self._p(d_rec_off), n, self._p(d_ref_sel),
                int(d_ref_sel.shape[0]), int(map_qual), int(mq_cutoff), float(clip_percent), float(iden_percent),
                int(rec_idx_base), self._p(out), self._p(self._status))
        self._chk(st, "gci_bam_filter")
        if check:
            self.check_status("gci_bam_filter")
        return out[:n]

    # ---- R1 over record pages (include/gci_hip.h: gci_bam_pages_*, gci_bam_filter_pages) -------------------------------
    def bam_pages(self, d_stream: Buffer, d_rec_off: Buffer, has_seq: bool, page_bytes: int = 0) -> "Pages":
        """The records of an inflated BAM stream (has_seq) or of a heads stream laid out as record pages: the bytes read_sam
        looks at, 16-byte aligned, no offset table -- what the record filter is fastest on."""
        page_bytes = int(page_bytes or os.environ.get("GCI_PAGE_BYTES", 0) or _lib.PAGE_BYTES_DEFAULT)
        n = int(d_rec_off.shape[0])
        h = (ctypes.c_uint64 * 3)()
        self._chk(self.lib.gci_bam_pages_size(self.ctx, self._p(d_stream), int(d_stream.shape[0]), self._p(d_rec_off), n, int(has_seq),
                                              page_bytes, h), "gci_bam_pages_size")
        buf = self.T.empty(int(h[1]), self.T.uint8, self.device)
        self._chk(self.lib.gci_bam_pages_write(self.ctx, self._p(d_stream), int(d_stream.shape[0]), self._p(d_rec_off), n, int(has_seq),
                                               self._p(buf), int(h[1])), "gci_bam_pages_write")
        return Pages(buf, int(h[0]), page_bytes, int(h[2]), n)

    def bam_filter_pages(self, pages: "Pages", d_ref_sel: Buffer, map_qual: int, mq_cutoff: int, clip_percent: float,
                         iden_percent: float, out: Optional[Buffer] = None, name_off: Optional[Buffer] = None,
                         check: bool = True, rec_idx_base: int = 0) -> Tuple[Buffer, Buffer]:
        """-> (records uint8 [n, 32], name offsets int64 [n] into pages.buf)."""
        n = pages.n_rec
        if out is None:
            out = self.T.empty((max(n, 1), 32), self.T.uint8, self.device)
        if name_off is None:
            name_off = self.T.empty(max(n, 1), self.T.int64, self.device)
        st = self.lib.gci_bam_filter_pages(self.ctx, self._p(pages.buf), int(pages.buf.shape[0]), pages.page_bytes, pages.n_pages, n,
                                           self._p(d_ref_sel), int(d_ref_sel.shape[0]), int(map_qual), int(mq_cutoff),
                                           float(clip_percent), float(iden_percent), int(rec_idx_base), self._p(out),
                                           self._p(name_off), self._p(self._status))
        self._chk(st, "gci_bam_filter_pages")
        if check:
            self.check_status("gci_bam_filter_pages")
        return out[:n], name_off[:n]

    def check_status(self, what: str) -> None:
        w = int(self._status.item()) & _M64
        rec = ctypes.c_uint32(0)
        st = self.lib.gci_decode_status(w, ctypes.byref(rec))
        if st != 0:
            raise GciError(st, "%s: %s (record %d)" % (what, self.lib.gci_strerror(st).decode(), rec.value),
                           rec=int(rec.value))

    # ---- R5 ----------------------------------------------------------------------------------
    def set_join_mode(self, mode: str = "auto") -> None:
        """'auto' (by size), 'classic' (open-addressing table in HBM) or 'partition' (radix partition + tables in LDS)."""
        self.join_mode = {"auto": 0, "classic": 1, "partition": 2}[mode]
        self._chk(self.lib.gci_join_mode(self.ctx, self.join_mode), "gci_join_mode")

    def _join_files(self, files: Sequence[JoinInput]):
        arr = (JoinFile * len(files))()
        for i, f in enumerate(files):
            arr[i].d_recs = f.recs.data_ptr()
            arr[i].n_recs = int(f.recs.shape[0])
            arr[i].name_delta = int(f.name_delta)
            arr[i].d_name_base = f.name_base.data_ptr()
            arr[i].d_name_off = f.name_off.data_ptr()
        return arr

    def name_join(self, files: Sequence[JoinInput], ovlp_percent: float, contig_map: Optional[Buffer] = None,
                  out: Optional[Buffer] = None, count: Optional[Buffer] = None, check: bool = True,
                  count_flank: Optional[int] = None, fallback: bool = True,
                  status: Optional[Buffer] = None) -> Tuple[Buffer, Buffer]:
        """-> (intervals int32 [cap, 4], device count).  With check=True the count is read back,
        capacity is grown if needed and the record-level status is raised.

        count_flank = the flank of the depth build that follows: gci_name_join_count then does that build's counting
        pass inside the join (pass counted=True to depth_build_fused / depth_build)."""
        total = sum(int(f.recs.shape[0]) for f in files)
        if out is None:
            out = self.T.empty((max(total, 1), 4), self.T.int32, self.device)
        if count is None:
            count = self.T.zeros(1, self.T.int32, self.device)
        arr = self._join_files(files)
        st_word = self._status if status is None else status          # status: the caller reads the word later (check=False)
        while True:
            if count_flank is None:
                st = self.lib.gci_name_join(self.ctx, arr, len(files), float(ovlp_percent), self._p(contig_map),
                                            self._p(out), int(out.shape[0]), self._p(count), self._p(st_word))
            else:
                st = self.lib.gci_name_join_count(self.ctx, arr, len(files), float(ovlp_percent), self._p(contig_map),
                                                  self._p(out), int(out.shape[0]), self._p(count), self._p(st_word),
                                                  int(count_flank))
            self._chk(st, "gci_name_join")
            if not check:
                return out, count
            try:
                self.check_status("gci_name_join")
            except GciError as e:
                # the partitioned join (large inputs) found more distinct names in one hash bucket than its LDS table
                # holds -- only adversarial names get there: redo on the classic global table
                if e.status != _lib.GCI_E_CAPACITY or not fallback:
                    raise
                self._chk(self.lib.gci_join_mode(self.ctx, 1), "gci_join_mode")
                try:
                    return self.name_join(files, ovlp_percent, contig_map, out, count, True, count_flank, False)
                finally:
                    self._chk(self.lib.gci_join_mode(self.ctx, self.join_mode), "gci_join_mode")
            n = int(count.item())
            if n <= out.shape[0]:
                return out, count
            out = self.T.empty((n, 4), self.T.int32, self.device)

    # ---- R7 / N2: gzip members of the depth text, written on the device ----------------------------------------
    MEMBER_BASES = 64 * 4096

    def depth_deflate(self, track: Buffer, from_build: bool = False) -> List[bytes]:
        """-> per contig of the layout, the bytes of the gzip members whose payload is that contig's depth lines
        (f'{depth}\\n' per base, GCI.py:115-117; no '>' line), as memoryviews of a pinned buffer the engine reuses: valid until
        the next call.  gci_depth_deflate_size / _write.  from_build: the caller states that `track` is exactly what the last
        depth_build_fused(want_runs=True) of THIS engine wrote and that nothing has written it since (no other engine, no torch
        operation): the members are then encoded from the run lists that build kept and the track is not read
        (gci_depth_deflate_from_build; the lists serve one call)."""
        elem, cnt, first = [], [], [0]
        for off, length in zip(self.offsets, self.lengths):
            for g in range(0, int(length), self.MEMBER_BASES):
                elem.append(int(off) + g)
                cnt.append(min(self.MEMBER_BASES, int(length) - g))
            first.append(len(elem))
        nm = len(elem)
        if nm == 0:
            return [b"" for _ in self.lengths]
        d_elem = self.to_device(np.asarray(elem, dtype=np.uint64))
        d_cnt = self.to_device(np.asarray(cnt, dtype=np.uint32))
        dev = self.device
        tile_bytes = self.T.empty(nm * 64, self.T.int32, dev)
        mb, crc, isz = (self.T.empty(nm, self.T.int32, dev) for _ in range(3))
        if from_build:
            self._chk(self.lib.gci_depth_deflate_from_build(self.ctx, self._p(track)), "gci_depth_deflate_from_build")
        self._chk(self.lib.gci_depth_deflate_size(self.ctx, self._p(track), self._p(d_elem), self._p(d_cnt), nm, self._p(tile_bytes),
                                                  self._p(mb), self._p(crc), self._p(isz)), "gci_depth_deflate_size")
        # member offsets on the device (no round trip through the host for them); one sync for the total
        d_off64 = self.T.scan_u32_u64(mb)
        total = int(d_off64[nm].item())
        out = self.T.empty(total, self.T.uint8, dev)
        self._chk(self.lib.gci_depth_deflate_write(self.ctx, self._p(track), self._p(d_elem), self._p(d_cnt), nm, self._p(tile_bytes),
                                                   self._p(crc), self._p(isz), self._p(d_off64), self._p(out), total),
                  "gci_depth_deflate_write")
        # the members leave through a pinned staging buffer of the engine (a pageable destination halves the D2H rate and
        # .tobytes() per contig copied everything once more): the caller gets views of it, valid until the next call
        if self._members_host is None or int(self._members_host.shape[0]) < total:
            self._members_host = self.T.pinned(max(total + (total >> 3), 1 << 20))
        host = self._members_host[:total]
        host.copy_(out, non_blocking=True)
        offs = d_off64.cpu().numpy()
        self.sync()
        blob = host.numpy()
        return [memoryview(blob[int(offs[first[c]]):int(offs[first[c + 1]])]) for c in range(len(self.lengths))]

    def hash_bucket(self, recs: Buffer, n_parts: int, part_cap: int, out: Buffer,
                    next_out: Optional[Buffer] = None) -> None:
        """out: int64 [n_parts * (part_cap + 1)]; word 0 of each bucket = its count.  next_out: the array the caller
        alternates with `out` (its count words are zeroed for the next call; `out`'s must be zero on entry)."""
        self._chk(self.lib.gci_hash_bucket(self.ctx, self._p(recs), int(recs.shape[0]), int(n_parts), int(part_cap),
                                           self._p(out), self._p(next_out)), "gci_hash_bucket")

    def hash_conflicts(self, buckets: Buffer, n_parts: int, part_cap: int, n_conflicts: Buffer) -> None:
        """Adds to n_conflicts (int32 [1])."""
        self._chk(self.lib.gci_hash_conflicts(self.ctx, self._p(buckets), int(n_parts), int(part_cap),
                                              self._p(n_conflicts)), "gci_hash_conflicts")

    # ---- multi-GPU: buckets of the name-hash-sharded join (gci_route_*; the collectives are shard.ShardedJoin's) ----------
    def route_records(self, f: JoinInput, n_parts: int, cap: int, out_recs: Buffer, out_names: Buffer,
                      name_slot: int, status: Buffer) -> None:
        """out_recs uint8 [n_parts * (cap + 1), 32], out_names uint8 [n_parts * cap * name_slot], status int64 [1]."""
        self._chk(self.lib.gci_route_records(self.ctx, self._join_files([f]), int(n_parts), int(cap), self._p(out_recs),
                                             self._p(out_names), int(name_slot), self._p(status)), "gci_route_records")

    def route_seal_records(self, recs: Buffer, n_parts: int, cap: int, status: Buffer) -> None:
        self._chk(self.lib.gci_route_seal_records(self.ctx, self._p(recs), int(n_parts), int(cap), self._p(status)),
                  "gci_route_seal_records")

    def route_intervals(self, ivl: Buffer, count: Buffer, owner: Buffer, n_parts: int, cap: int,
                        out: Buffer, status: Buffer) -> None:
        """owner int32 [n_contigs]: rank of every contig; out int32 [n_parts * (cap + 1), 4]."""
        self._chk(self.lib.gci_route_intervals(self.ctx, self._p(ivl), self._p(count), int(ivl.shape[0]), self._p(owner),
                                               int(owner.shape[0]), int(n_parts), int(cap), self._p(out), self._p(status)),
                  "gci_route_intervals")

    def route_seal_intervals(self, ivl: Buffer, n_parts: int, cap: int, cmap: Buffer, status: Buffer) -> None:
        self._chk(self.lib.gci_route_seal_intervals(self.ctx, self._p(ivl), int(n_parts), int(cap), self._p(cmap),
                                                    int(cmap.shape[0]), self._p(status)), "gci_route_seal_intervals")

    def pack_names(self, f: JoinInput) -> Tuple[Buffer, Buffer]:
        n = int(f.recs.shape[0])
        off = self.T.zeros(n + 1, self.T.int64, self.device)
        arr = self._join_files([f])
        self._chk(self.lib.gci_pack_names(self.ctx, arr, None, 0, self._p(off)), "gci_pack_names(size)")
        total = int(off[n].item())
        blob = self.T.empty(max(total, 1), self.T.uint8, self.device)
        self._chk(self.lib.gci_pack_names(self.ctx, arr, self._p(blob), total, self._p(off)), "gci_pack_names")
        return blob[:total], off

    # ---- N1: BGZF inflate + record walk on the device (k_inflate.hip) ---------------------------------------------------
    def upload_padded(self, raw: np.ndarray) -> Buffer:
        """The bytes of a BGZF file on the device with 16 readable bytes behind them (gci_bgzf_inflate_device takes its
        input as whole aligned 16-byte blocks)."""
        n_raw = int(raw.shape[0])
        with self.T.stream(self.stream), warnings.catch_warnings():      # (may be called from a helper thread)
            warnings.simplefilter("ignore", UserWarning)                     # a read-only memmap is only read
            d_raw = self.T.empty(n_raw + 16, self.T.uint8, self.device)
            d_raw[n_raw:].zero_()
            d_raw[:n_raw].copy_(self._host_src(raw))
        return d_raw

    def bgzf_inflate(self, raw: Optional[np.ndarray], pos: np.ndarray, isize: np.ndarray, check_crc: bool = True,
                     prefix: Optional[Buffer] = None, d_raw: Optional[Buffer] = None) -> Buffer:
        """raw: the bytes of a BGZF file (or of a run of its members) -- or d_raw, the same already on the device
        (upload_padded); pos (uint64, n + 1) / isize (uint64, n): its member table (hostio.bgzf_blocks), pos relative to
        raw.  -> the inflated bytes on the device, behind the bytes of `prefix` (the partial record a previous run of
        members ended in).  Raises GciError(GCI_E_MALFORMED, rec = member) on a bad member / CRC."""
        n = int(isize.shape[0])
        off = np.zeros(n + 1, dtype=np.uint64)
        np.cumsum(isize, out=off[1:])
        total = int(off[n])
        n_pre = int(prefix.shape[0]) if prefix is not None else 0
        if d_raw is None:
            d_raw = self.upload_padded(raw)
        d_pos, d_off = self.to_device(np.ascontiguousarray(pos[:n + 1], dtype=np.uint64)), self.to_device(off)
        # (sizes in steps of 256 MiB: the runs of a file differ by a few MB, and a request a little larger than the block the run before
        #  gave back is a new device allocation -- 44 ms beside a copy in flight -- where the same step finds that block again)
        want = max(n_pre + total, 1)
        out = self.T.empty(((want + (1 << 28) - 1) >> 28) << 28 if want > (1 << 28) else want, self.T.uint8, self.device)
        if n_pre:
            out[:n_pre].copy_(prefix)
        self._chk(self.lib.gci_bgzf_inflate_device(self.ctx, self._p(d_raw), self._p(d_pos), self._p(d_off), n, ctypes.c_void_p(out.data_ptr() + n_pre), total,
                                                   int(check_crc), self._p(self._status)), "gci_bgzf_inflate_device")
        self.check_status("gci_bgzf_inflate_device")
        return out[:n_pre + total]

    def bgzf_inflate_ahead(self, d_raw: Buffer, pos: np.ndarray, isize: np.ndarray, headroom: int, check_crc: bool = True
                           ) -> Tuple[Buffer, Buffer, int]:
        """bgzf_inflate() for a run of members of a file that goes through the device run by run, ENQUEUED and not waited for: the
        inflated bytes go `headroom` bytes into a fresh buffer (in front of them the caller later puts the partial record the run
        before ended in -- which it knows only when that run has been walked, while this run is already being inflated) and the
        status word into a buffer of its own, which the caller checks (check_status_word) once it has made its stream wait for this
        one.  -> (buffer of headroom + total bytes, status word, total)."""
        n = int(isize.shape[0])
        off = np.zeros(n + 1, dtype=np.uint64)
        np.cumsum(isize, out=off[1:])
        total = int(off[n])
        d_pos, d_off = self.to_device(np.ascontiguousarray(pos[:n + 1], dtype=np.uint64)), self.to_device(off)
        want = max(int(headroom) + total, 1)
        out = self.T.empty(((want + (1 << 28) - 1) >> 28) << 28 if want > (1 << 28) else want, self.T.uint8, self.device)
        status = self.T.empty(1, self.T.int64, self.device)
        self._chk(self.lib.gci_bgzf_inflate_device(self.ctx, self._p(d_raw), self._p(d_pos), self._p(d_off), n, ctypes.c_void_p(out.data_ptr() + int(headroom)),
                                                   total, int(check_crc), self._p(status)), "gci_bgzf_inflate_device")
        return out[:int(headroom) + total], status, total

    def check_status_word(self, status: Buffer, what: str) -> None:
        """check_status() over a status word of the caller's (read on the calling thread's current stream)."""
        w = int(status.item()) & _M64
        rec = ctypes.c_uint32(0)
        st = self.lib.gci_decode_status(w, ctypes.byref(rec))
        if st != 0:
            raise GciError(st, "%s: %s (record %d)" % (what, self.lib.gci_strerror(st).decode(), rec.value), rec=int(rec.value))

    def inflate_stats(self) -> dict:
        """How the members of the last bgzf_inflate fared with the wave decoder (gci_bgzf_inflate_last_stats; synchronises)."""
        c = (ctypes.c_uint32 * 32)()
        self._chk(self.lib.gci_bgzf_inflate_last_stats(self.ctx, c), "gci_bgzf_inflate_last_stats")
        names = ("decoded", "header", "no meeting point", "false end of block", "undecodable", "length", "lanes") + tuple("cause %d" % k for k in range(7, 31)) + ("not tried",)
        return {k: int(v) for k, v in zip(names, c) if v or k in ("decoded", "not tried")}

    def inflate_round(self) -> int:
        """Members gci_bgzf_inflate_device decodes at a time on this device (0: unknown)."""
        return int(self.lib.gci_bgzf_inflate_round(self.ctx))

    def start_upload(self, raw: np.ndarray, parts: int = 2):
        """Begin uploading the bytes of a BGZF file in `parts` pieces on a copy stream of its own (a helper thread: the
        copies come from pageable memory and block their caller) -> a handle for bgzf_inflate_uploaded, which starts
        inflating the members of a piece as soon as that piece has arrived."""
        from concurrent.futures import ThreadPoolExecutor
        n_raw = int(raw.shape[0])
        cuts = sorted({min(n_raw, (n_raw * (k + 1) // parts + 15) & ~15) for k in range(parts)} | {n_raw})
        d_raw = self.T.empty(n_raw + 16, self.T.uint8, self.device)
        copy_stream = self.copy_stream()                         # (creating a stream costs milliseconds: one per engine)
        copy_stream.wait_stream(self.stream)                     # (the allocation may recycle memory still in use on the main stream)
        events = [self.T.Event() for _ in cuts]
        queued = [threading.Event() for _ in cuts]               # (a CUDA event that was never recorded counts as complete)

        def run():
            with self.T.stream(copy_stream), warnings.catch_warnings():
                warnings.simplefilter("ignore", UserWarning)     # a read-only memmap is only read
                d_raw[n_raw:].zero_()
                lo = 0
                for k, hi in enumerate(cuts):
                    if hi > lo:
                        d_raw[lo:hi].copy_(self._host_src(raw[lo:hi]))
                    events[k].record(copy_stream)
                    queued[k].set()
                    lo = hi
            return True

        pool = ThreadPoolExecutor(1)
        return dict(d_raw=d_raw, cuts=cuts, events=events, queued=queued, future=pool.submit(run), pool=pool, stream=copy_stream)

    def bgzf_inflate_uploaded(self, up, pos: np.ndarray, isize: np.ndarray, check_crc: bool = True) -> Buffer:
        """bgzf_inflate over a file whose upload start_upload began: one launch per uploaded piece, over the members that lie
        wholly inside what has arrived (and the 16 bytes the decoder may read behind a member)."""
        n = int(isize.shape[0])
        off = np.zeros(n + 1, dtype=np.uint64)
        np.cumsum(isize, out=off[1:])
        total = int(off[n])
        d_pos, d_off = self.to_device(np.ascontiguousarray(pos[:n + 1], dtype=np.uint64)), self.to_device(off)
        out = self.T.empty(max(total, 1), self.T.uint8, self.device)
        ends = np.asarray(pos[1:n + 1], dtype=np.uint64)
        n_raw = int(up["d_raw"].shape[0]) - 16
        status = self.T.zeros(len(up["cuts"]), self.T.int64, self.device)
        m_lo, bases = 0, []
        try:
            for k, (cut, ev) in enumerate(zip(up["cuts"], up["events"])):
                m_hi = n if cut >= n_raw else int(np.searchsorted(ends, np.uint64(max(0, cut - 16)), side="right"))
                m_hi = max(m_hi, m_lo)
                while not up["queued"][k].wait(0.05):
                    if up["future"].done():
                        up["future"].result()                    # the upload failed: its exception, not a hang
                        break
                self.stream.wait_event(ev)
                bases.append(m_lo)
                if m_hi > m_lo:
                    self._chk(self.lib.gci_bgzf_inflate_device(self.ctx, self._p(up["d_raw"]), ctypes.c_void_p(d_pos.data_ptr() + 8 * m_lo),
                                                               ctypes.c_void_p(d_off.data_ptr() + 8 * m_lo), m_hi - m_lo, self._p(out), total,
                                                               int(check_crc), ctypes.c_void_p(status.data_ptr() + 8 * k)), "gci_bgzf_inflate_device")
                else:
                    status[k] = -1
                m_lo = m_hi
            up["future"].result()
        finally:
            up["pool"].shutdown()
        for k, w in enumerate(status.cpu().numpy().view(np.uint64).tolist()):
            rec = ctypes.c_uint32(0)
            st = self.lib.gci_decode_status(int(w) & _M64, ctypes.byref(rec))
            if st != 0:
                raise GciError(st, "gci_bgzf_inflate_device: %s (record %d)" % (self.lib.gci_strerror(st).decode(), rec.value + bases[k]),
                               rec=int(rec.value) + bases[k])
        self.stream.wait_stream(up["stream"])
        return out[:total]

    def bam_record_offsets(self, d_stream: Buffer, first_record: int, n_ref: int) -> Tuple[Buffer, int, bool]:
        """-> (int64 offsets of every record on the device, bytes consumed, chain_ok).  chain_ok False: a record the strict
        format test rejects sits on the chain -- the caller walks the chain on the host instead."""
        n = int(d_stream.shape[0])
        res = self.T.zeros(3, self.T.int64, self.device)
        cap = max(1024, n // 256)
        while True:
            offs = self.T.empty(cap, self.T.int64, self.device)
            self._chk(self.lib.gci_bam_record_offsets_device(self.ctx, self._p(d_stream), n, int(first_record), int(n_ref), self._p(offs), cap,
                                                             self._p(res)), "gci_bam_record_offsets_device")
            n_rec, used, broken = (int(x) for x in res.cpu().tolist())
            if n_rec <= cap:
                return offs[:n_rec], used, broken == 0
            cap = n_rec

    # ---- R3 / N4: the PAF path of filter() on the device (k_paf.hip) -----------------------------------------------
    def paf_filter(self, paths: Sequence[str], targets: Sequence[str], map_qual: int, mq_cutoff: int, iden_percent: float
                   ) -> List[JoinInput]:
        """gci_paf_filter_device over the PAF files `paths` (command-line order): -> one JoinInput per file, holding a
        compact record per query seen so far (files < i included: the reference never resets its block table) and
        pointing at the names inside the uploaded text.  Raises GciError with .rec = the 1-based line the reference
        would have raised on (GCI_E_MALFORMED: IndexError / ValueError, GCI_E_ZERO_DIV: ZeroDivisionError)."""
        sizes = [int(os.path.getsize(p)) for p in paths]
        ends = np.cumsum(sizes, dtype=np.uint64) if sizes else np.zeros(0, np.uint64)
        total = int(ends[-1]) if sizes else 0
        if total < int(os.environ.get("GCI_PAF_STAGE_MIN", str(256 << 20))):          # (small files: read, put together, one copy)
            bufs = [np.fromfile(p, dtype=np.uint8) for p in paths]
            text = np.concatenate(bufs) if bufs and total else np.zeros(1, np.uint8)
            return self.paf_filter_text(self.to_device(text), ends, targets, map_qual, mq_cutoff, iden_percent)
        # Files of GBs (the reference's published CHM13 run reads a 3.6 GB HiFi and a 48 GB ONT PAF, README.md:321): their bytes go
        # from the mapped files through the ring of pinned slots straight to their places in ONE device buffer -- round 5 read every
        # file into host memory, concatenated the arrays and handed the result to a pageable copy: three passes over 52 GB of host
        # memory and 104 GB of it held, before the first byte moved
        d_text = self.T.empty(total, self.T.uint8, self.device)
        staging, copy = self.staging(), self.copy_stream()
        copy.wait_stream(self.stream)
        at = 0
        for p, n in zip(paths, sizes):
            if n:
                raw = np.memmap(p, dtype=np.uint8, mode="r")
                staging.send(raw, 0, n, d_text[at:at + n], copy)
                del raw
            at += n
        self.stream.wait_stream(copy)
        return self.paf_filter_text(d_text, ends, targets, map_qual, mq_cutoff, iden_percent)

    def paf_filter_text(self, d_text: Buffer, ends: np.ndarray, targets: Sequence[str], map_qual: int, mq_cutoff: int,
                        iden_percent: float) -> List[JoinInput]:
        """paf_filter() over PAF text that is on the device already: the bytes of all files back to back, ends[i] = end offset
        of file i."""
        ends = np.ascontiguousarray(ends, dtype=np.uint64)
        bufs = list(range(int(ends.shape[0])))
        tnames = [t.encode() for t in targets]
        tarr = (ctypes.c_char_p * max(len(tnames), 1))(*tnames)
        handle, line = ctypes.c_void_p(None), ctypes.c_uint64(0)
        st = self.lib.gci_paf_filter_device(self.ctx, self._p(d_text), ends.ctypes.data_as(ctypes.c_void_p), len(bufs), tarr,
                                            len(tnames), int(map_qual), int(mq_cutoff), float(iden_percent), ctypes.byref(handle),
                                            ctypes.byref(line))
        if st != 0:
            detail = self.lib.gci_last_error(self.ctx).decode() if st == _lib.GCI_E_HIP else ""
            raise GciError(st, "gci_paf_filter_device: %s (line %d) %s" % (self.lib.gci_strerror(st).decode(), line.value, detail),
                           rec=int(line.value))
        try:
            return self._paf_dev_inputs(handle, len(bufs), d_text)
        finally:
            self.lib.gci_paf_dev_free(handle)

    # ---- the PAF filter in two halves (runs that shard a PAF file by byte range: shard.paf_by_byte_range) ----
    PAF_HIT_BYTES = 80

    def _paf_dev_inputs(self, handle, n_files: int, d_names: Buffer) -> List[JoinInput]:
        out = []
        for f in range(n_files):
            n = int(self.lib.gci_paf_dev_count(handle, f))
            recs = self.T.empty((max(n, 1), 32), self.T.uint8, self.device)
            off = self.T.empty(max(n, 1), self.T.int64, self.device)
            self._chk(self.lib.gci_paf_dev_export(handle, f, self._p(recs), self._p(off)), "gci_paf_dev_export")
            out.append(JoinInput(recs[:n], d_names, off[:n], 0))
        self.sync()
        return out

    def paf_hits_text(self, d_text: Buffer, ends: np.ndarray, targets: Sequence[str], map_qual: int, mq_cutoff: int,
                      iden_percent: float) -> List[Buffer]:
        """Stage A (gci_paf_hits_device) over the byte ranges in d_text (ends[i] = end of file i's range): per file the lines
        that pass, as a uint8 tensor [n, 80] of gci_paf_hit in line order (qn_off points into d_text)."""
        ends = np.ascontiguousarray(ends, dtype=np.uint64)
        n_files = int(ends.shape[0])
        tnames = [t.encode() for t in targets]
        tarr = (ctypes.c_char_p * max(len(tnames), 1))(*tnames)
        handle, line = ctypes.c_void_p(None), ctypes.c_uint64(0)
        st = self.lib.gci_paf_hits_device(self.ctx, self._p(d_text), ends.ctypes.data_as(ctypes.c_void_p), n_files, tarr, len(tnames),
                                          int(map_qual), int(mq_cutoff), float(iden_percent), ctypes.byref(handle), ctypes.byref(line))
        if st != 0:
            raise GciError(st, "gci_paf_hits_device: %s (line %d of the range)" % (self.lib.gci_strerror(st).decode(), line.value),
                           rec=int(line.value))
        try:
            out = []
            for f in range(n_files):
                n = int(self.lib.gci_paf_hits_count(handle, f))
                h = self.T.empty((max(n, 1), self.PAF_HIT_BYTES), self.T.uint8, self.device)
                self._chk(self.lib.gci_paf_hits_export(handle, f, self._p(h)), "gci_paf_hits_export")
                out.append(h[:n])
            self.sync()
            return out
        finally:
            self.lib.gci_paf_hits_free(handle)

    def route_hits(self, hits: Buffer, d_name_base: Buffer, n_parts: int, cap: int, out_hits: Buffer,
                   out_names: Buffer, name_slot: int, status: Buffer) -> None:
        """gci_route_hits: hits [n, 80] -> out_hits [n_parts * (cap + 1), 80] (slot 0 of a bucket: header, qhash = count), their
        query names into out_names [n_parts * cap * name_slot]."""
        n = int(hits.shape[0])
        self._chk(self.lib.gci_route_hits(self.ctx, self._p(hits) if n else None, n, self._p(d_name_base), int(n_parts), int(cap),
                                          self._p(out_hits), self._p(out_names), int(name_slot), self._p(status)), "gci_route_hits")

    def paf_score_hits(self, d_names: Buffer, d_hits: Buffer, upto: Sequence[int], targets: Sequence[str]) -> List[JoinInput]:
        """Stage B (gci_paf_score_device): d_hits [total, 80] = the hits of the queries this rank owns, file after file
        (upto[f] = first hit of file f, upto[-1] = total), qn_off relative to d_names -> one JoinInput per file."""
        n_files = len(upto) - 1
        h_upto = np.ascontiguousarray(upto, dtype=np.uint32)
        tnames = [t.encode() for t in targets]
        tarr = (ctypes.c_char_p * max(len(tnames), 1))(*tnames)
        handle = ctypes.c_void_p(None)
        st = self.lib.gci_paf_score_device(self.ctx, self._p(d_names), self._p(d_hits) if int(h_upto[-1]) else None,
                                           h_upto.ctypes.data_as(ctypes.c_void_p), n_files, tarr, len(tnames), ctypes.byref(handle))
        if st != 0:
            raise GciError(st, "gci_paf_score_device: %s" % self.lib.gci_strerror(st).decode(), rec=0)
        try:
            return self._paf_dev_inputs(handle, n_files, d_names)
        finally:
            self.lib.gci_paf_dev_free(handle)

    # ---- R6 / R8 / R9 / R15 --------------------------------------------------------------------
    def depth_build(self, ivl: Buffer, count: Optional[Buffer], flank: int, track: Buffer,
                    max_n: Optional[int] = None) -> Buffer:
        n = int(ivl.shape[0]) if max_n is None else int(max_n)
        st = self.lib.gci_depth_build(self.ctx, self._p(ivl) if n else None, self._p(count), n, int(flank),
                                      self._p(track))
        self._chk(st, "gci_depth_build")
        return track

    def depth_build_fused(self, ivl: Buffer, count: Optional[Buffer], flank: int, track: Buffer,
                          want_text: bool = True, want_sums: bool = False,
                          issue: Optional[Tuple[float, float, int]] = None, max_n: Optional[int] = None,
                          counted: bool = False, key_cap: int = 1 << 16, want_runs: bool = False):
        """Depth build that also returns what the reference derives from the fresh depths, computed in
        the same pass (no re-read of the track): decimal text, per-contig sums and -- only valid when
        no gap mask follows -- the raw issue runs for (lo, hi, flank).  want_runs: the library keeps the constant-depth runs of
        every tile as the build wrote them, and a depth_deflate() of this track -- before anything else writes it -- takes them
        from there instead of reading the track (gci_build_opts.want_runs).

        -> dict(text=uint8 tensor | None, text_off=int64 ndarray | None, sums=ndarray | None,
                runs=list of per-contig arrays | None)"""
        n = int(ivl.shape[0]) if max_n is None else int(max_n)
        nc = len(self.lengths)
        o = BuildOpts()
        o.flank = int(flank)
        o.counted = 1 if counted else 0
        o.want_text = 1 if want_text else 0
        o.want_runs = 1 if want_runs else 0
        text_off = self.T.zeros(nc + 1, self.T.int64, self.device) if want_text else None
        # want_sums: True = per-contig sums returned as a host array; a device tensor (int64 [n_contigs]) = written there, nothing copied
        sums_dev = want_sums if self.T.is_buffer(want_sums) else None
        want_sums = bool(sums_dev is not None or want_sums is True)
        sums = sums_dev if sums_dev is not None else (self.T.zeros(max(nc, 1), self.T.int64, self.device) if want_sums else None)
        o.d_contig_text_off = text_off.data_ptr() if want_text else None
        o.d_sums = sums.data_ptr() if want_sums else None
        cap = int(key_cap)                                 # grown (and the first pass repeated) when more run boundaries turn up
        keys = None
        while True:
            if issue is not None:
                keys = self.T.empty(cap, self.T.int64, self.device)
                o.d_n_keys, o.d_keys, o.key_cap = self._count.data_ptr(), keys.data_ptr(), cap
                o.lo, o.hi, o.issue_flank = float(issue[0]), float(issue[1]), int(issue[2])
            self._chk(self.lib.gci_depth_build_begin(self.ctx, self._p(ivl) if n else None, self._p(count), n,
                                                     ctypes.byref(o)), "gci_depth_build_begin")
            if issue is None:
                break
            nk = int(self._count.item())
            if nk <= cap:
                break
            cap = nk
            o.counted = 0                                  # the join's counts are used up: the second begin counts again
        out = dict(text=None, text_off=None, sums=None, runs=None)
        text = None
        if want_text:
            h = text_off.cpu().numpy()
            total = int(h[nc])
            text = self.T.empty(max(total, 1), self.T.uint8, self.device)
            out["text_off"] = h
        self._chk(self.lib.gci_depth_build_finish(self.ctx, self._p(track), self._p(text), int(text.shape[0]) if want_text else 0),
                  "gci_depth_build_finish")
        if want_text:
            out["text"] = text[:int(out["text_off"][nc])]
        if want_sums:
            out["sums"] = sums if sums_dev is not None else sums.cpu().numpy()[:nc]
        if issue is not None:
            out["runs"] = self._keys_to_runs(keys[:nk].cpu().numpy().view(np.uint64), nc)
        return out

    def gap_mask(self, track: Buffer, gaps: Buffer) -> Buffer:
        n = int(gaps.shape[0])
        if n:
            self._chk(self.lib.gci_gap_mask(self.ctx, self._p(track), self._p(gaps), n), "gci_gap_mask")
        return track

    def max2(self, a: Buffer, b: Buffer, out: Optional[Buffer] = None) -> Buffer:
        if out is None:
            out = self.new_track()
        self._chk(self.lib.gci_max2(self.ctx, self._p(a), self._p(b), self._p(out)), "gci_max2")
        return out

    def two_type_tail(self, a: Buffer, b: Buffer, gaps: Optional[np.ndarray], lo: float, hi: float, flank: int,
                      out: Optional[Buffer] = None, keys: Optional[Buffer] = None, n_keys: Optional[Buffer] = None,
                      read: bool = True, sums: Optional[Buffer] = None):
        """gci_two_type_tail: N-run masks of both tracks (in place; gaps = int32 [n, 4] rows (contig, start, end, 0) on the HOST, or
        None), their maximum and the issue runs of all three in one pass.  -> (maximum track, [runs of a, of b, of the maximum]);
        read=False: (maximum track, keys int64 [3, cap] on the device, n_keys int32 [3]) with nothing copied to the host.
        sums: int64 [3, n_contigs] on the device, filled with the per-contig sums of depth of the three tracks."""
        if out is None:
            out = self.new_track()
        g = np.ascontiguousarray(gaps, dtype=np.int32).reshape(-1, 4) if gaps is not None and len(gaps) else None
        cap = int(keys.shape[1]) if keys is not None else 1 << 16
        while True:
            if keys is None:
                keys = self.T.empty((3, cap), self.T.int64, self.device)
            if n_keys is None:
                n_keys = self.T.zeros(3, self.T.int32, self.device)
            self._chk(self.lib.gci_two_type_tail(self.ctx, self._p(a), self._p(b), self._p(out),
                                                 ctypes.c_void_p(g.ctypes.data) if g is not None else None, 0 if g is None else int(g.shape[0]),
                                                 float(lo), float(hi), int(flank), self._p(keys), cap, self._p(n_keys), self._p(sums)),
                      "gci_two_type_tail")
            if not read:
                return out, keys, n_keys
            n = n_keys.cpu().numpy()
            if int(n.max()) <= cap:
                break
            cap, keys = int(n.max()), None                  # (masking again changes nothing; the maximum is recomputed)
        hk = keys.cpu().numpy().view(np.uint64)
        return out, [self._keys_to_runs(hk[x, :int(n[x])], len(self.lengths)) for x in range(3)]

    def depth_sum(self, track: Buffer) -> np.ndarray:
        sums = self.T.zeros(max(len(self.lengths), 1), self.T.int64, self.device)
        self._chk(self.lib.gci_depth_sum(self.ctx, self._p(track), self._p(sums)), "gci_depth_sum")
        return sums.cpu().numpy()[:len(self.lengths)]

    def fasta_n_scan(self, text: np.ndarray, bodies: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
        """gci_fasta_n_scan over the bytes of a FASTA file: bodies = int64 [n_records, 2] byte ranges of the record
        bodies -> (uint32 kept-byte count per 4096-byte tile, sorted uint64 keys (offset << 1 | is_end))."""
        n = int(text.shape[0])
        d_text = self.upload_staged(text) if n >= (256 << 20) else self.to_device(text)
        d_body = self.to_device(np.ascontiguousarray(bodies, dtype=np.int64).reshape(-1, 2))
        tiles = (n + 4095) // 4096
        kept = self.T.zeros(max(tiles, 1), self.T.int32, self.device)
        cap = 1 << 16
        while True:
            keys = self.T.empty(cap, self.T.int64, self.device)
            self._chk(self.lib.gci_fasta_n_scan(self.ctx, self._p(d_text), n, self._p(d_body), int(bodies.shape[0]), self._p(kept),
                                                self._p(keys), cap, self._p(self._count)), "gci_fasta_n_scan")
            nk = int(self._count.item())
            if nk <= cap:
                break
            cap = nk
        return kept.cpu().numpy().view(np.uint32)[:tiles], np.sort(keys[:nk].cpu().numpy().view(np.uint64))

    def range_sums(self, track: Buffer, ranges: np.ndarray) -> np.ndarray:
        """Sum of the depths in each [begin, end) of track element indices (int64 [n, 2]) -> int64 [n]."""
        ranges = np.ascontiguousarray(ranges, dtype=np.int64).reshape(-1, 2)
        n = int(ranges.shape[0])
        if n == 0:
            return np.zeros(0, dtype=np.int64)
        d_r = self.to_device(ranges)
        sums = self.T.empty(n, self.T.int64, self.device)
        self._chk(self.lib.gci_range_sums(self.ctx, self._p(track), self._p(d_r), n, self._p(sums)), "gci_range_sums")
        return sums.cpu().numpy()

    # ---- R10 ---------------------------------------------------------------------------------
    @staticmethod
    def _keys_to_runs(keys: np.ndarray, n_windows: int) -> List[np.ndarray]:
        """Sorted boundary keys -> per window int64 [k, 2] of (rel start, rel end)."""
        keys = np.sort(keys.astype(np.uint64))
        win = (keys >> np.uint64(33)).astype(np.int64)
        rel = ((keys >> np.uint64(1)) & np.uint64(0xFFFFFFFF)).astype(np.int64)
        is_end = (keys & np.uint64(1)).astype(bool)
        out = []
        bounds = np.searchsorted(win, np.arange(n_windows + 1))
        for w in range(n_windows):
            a, b = bounds[w], bounds[w + 1]
            r, e = rel[a:b], is_end[a:b]
            if ((b - a) & 1) or e[0::2].any() or not e[1::2].all():
                raise GciError(_lib.GCI_E_INVALID, "issue scan produced unpaired run boundaries")
            out.append(np.stack([r[0::2], r[1::2]], axis=1) if b > a else np.zeros((0, 2), dtype=np.int64))
        return out

    def _scan(self, call, n_windows: int) -> List[np.ndarray]:
        cap = 1 << 16
        while True:
            keys = self.T.empty(cap, self.T.int64, self.device)
            call(keys, cap)
            n = int(self._count.item())
            if n <= cap:
                return self._keys_to_runs(keys[:n].cpu().numpy().view(np.uint64), n_windows)
            cap = n

    def issue_scan(self, track: Buffer, lo: float, hi: float, flank: int) -> List[np.ndarray]:
        """Raw maximal runs of lo < depth <= hi inside [flank, L - flank) of every contig, as
        positions relative to the window start (add `flank` for contig coordinates)."""
        def call(keys, cap):
            self._chk(self.lib.gci_issue_scan(self.ctx, self._p(track), float(lo), float(hi), int(flank),
                                              self._p(keys), cap, self._p(self._count)), "gci_issue_scan")
        return self._scan(call, len(self.lengths))

    def issue_scan_windows(self, track: Buffer, windows: Sequence[Tuple[int, int]], lo: float, hi: float
                           ) -> List[np.ndarray]:
        arr = (Window * max(len(windows), 1))()
        for i, (a, b) in enumerate(windows):
            arr[i].begin, arr[i].end = int(a), int(b)

        def call(keys, cap):
            self._chk(self.lib.gci_issue_scan_windows(self.ctx, self._p(track), arr, len(windows), float(lo), float(hi),
                                                      self._p(keys), cap, self._p(self._count)),
                      "gci_issue_scan_windows")
        return self._scan(call, len(windows))

    # ---- R7 ----------------------------------------------------------------------------------
    def depth_text(self, track: Buffer, out: Optional[Buffer] = None) -> Tuple[Buffer, np.ndarray]:
        """-> (uint8 text of all contigs back to back, int64 [n_contigs + 1] byte offsets)."""
        n = len(self.lengths)
        offs = self.T.zeros(n + 1, self.T.int64, self.device)
        self._chk(self.lib.gci_depth_text_size(self.ctx, self._p(track), self._p(offs)), "gci_depth_text_size")
        h = offs.cpu().numpy()
        total = int(h[n])
        if out is None or out.shape[0] < total:
            out = self.T.empty(max(total, 1), self.T.uint8, self.device)
        self._chk(self.lib.gci_depth_text_write(self.ctx, self._p(track), self._p(out), total), "gci_depth_text_write")
        return out[:total], h
