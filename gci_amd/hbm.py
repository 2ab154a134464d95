"""HBM buffers, streams and events of the host side -- two interchangeable providers behind one small interface.

`native()`  : this module's own, over the `gci_dev_*` exports of libgci_hip.so (include/gci_hip.h, k_hbm.hip) -- a device buffer is
              a pointer + shape + dtype, the allocator a stream-ordered cache over gci_dev_malloc.  What the single-GPU command
              line runs on: no `import torch` (half a second of interpreter time and a few hundred MB of libraries for what is, on
              this path, a malloc and a memcpy), no allocator teardown at exit.
`torch()`   : the same interface over torch tensors / streams -- what a contig-sharded run (`--gpus N`: torch.distributed over
              RCCL wants tensors) and the test-suite's fixtures use.

The kernels do not care: every export takes plain pointers.  `provider()` picks: GCI_HBM=native|torch, else torch when the process
has imported it already (a test, an embedding application, a rank under torch.distributed.run), else native.

The interface (both providers): dtypes `uint8 int16 int32 int64`; `empty / zeros(shape, dtype, device)`, `from_numpy(a, device)`,
`cat(bufs)`, `pinned(nbytes)`; `Stream(device)`, `Event(enable_timing)`, `stream(s)` (context manager: the calling thread's
current stream), `current_stream(device)`, `synchronize()`; `add_i64`, `rec_flags_and`, `scan_u32_u64` (the three element-wise
helpers); `is_buffer(x)`.  A buffer offers what the host code uses of a tensor: `shape`, `data_ptr()`, slices along the first
axis, `cpu().numpy()`, `item()`, `clone()`, `zero_()`, `copy_()`, `view(dtype)`, `record_stream()`.
"""
from __future__ import annotations

import ctypes
import os
import sys
import threading
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from ._lib import GciError


# =====================================================================================================================
# native provider
# =====================================================================================================================

class DType:
    __slots__ = ("name", "np", "itemsize")

    def __init__(self, name: str, npdt):
        self.name, self.np = name, np.dtype(npdt)
        self.itemsize = self.np.itemsize

    def __repr__(self):
        return "hbm." + self.name


uint8, int16, int32, int64 = DType("uint8", np.uint8), DType("int16", np.int16), DType("int32", np.int32), DType("int64", np.int64)
_BY_NP = {d.np: d for d in (uint8, int16, int32, int64)}
_BY_NP[np.dtype(np.uint32)] = int32          # (as the torch provider: unsigned words travel as their signed twins)
_BY_NP[np.dtype(np.uint64)] = int64
_BY_NP[np.dtype(np.uint16)] = int16
_BY_NP[np.dtype(np.int8)] = uint8


class Device:
    __slots__ = ("index", "type")

    def __init__(self, index: int = 0):
        self.index, self.type = int(index), "cuda"

    def __repr__(self):
        return "hbm.device(%d)" % self.index

    def __eq__(self, other):
        return isinstance(other, Device) and other.index == self.index

    def __hash__(self):
        return hash(("hbm", self.index))


def _dev_index(device) -> int:
    if device is None:
        return 0
    return int(getattr(device, "index", device) or 0)


def _chk(st: int, what: str) -> None:
    if st != 0:
        lib = _lib.load()
        raise GciError(st, "%s: %s %s" % (what, lib.gci_strerror(st).decode(), lib.gci_dev_last_error().decode()))


class Event:
    """hipEvent_t.  Made on first record (a never-recorded event counts as complete, as torch's does)."""
    __slots__ = ("handle", "device", "timing")

    def __init__(self, enable_timing: bool = False, device=0):
        self.handle, self.device, self.timing = None, _dev_index(device), bool(enable_timing)

    def record(self, stream: Optional["Stream"] = None) -> None:
        stream = stream if stream is not None else current_stream(self.device)
        self.device = stream.device
        if self.handle is None:
            h = ctypes.c_void_p()
            _chk(_lib.load().gci_dev_event_create(self.device, int(self.timing), ctypes.byref(h)), "gci_dev_event_create")
            self.handle = h.value
        _chk(_lib.load().gci_dev_event_record(self.device, ctypes.c_void_p(self.handle), ctypes.c_void_p(stream.handle)), "gci_dev_event_record")

    def synchronize(self) -> None:
        if self.handle is not None:
            _chk(_lib.load().gci_dev_event_sync(self.device, ctypes.c_void_p(self.handle)), "gci_dev_event_sync")

    def elapsed_time(self, end: "Event") -> float:
        ms = ctypes.c_double(0)
        _chk(_lib.load().gci_dev_event_elapsed_ms(self.device, ctypes.c_void_p(self.handle), ctypes.c_void_p(end.handle), ctypes.byref(ms)),
             "gci_dev_event_elapsed_ms")
        return float(ms.value)

    def __del__(self):
        h, self.handle = self.handle, None
        if h is not None:
            try:
                _lib.load().gci_dev_event_destroy(self.device, ctypes.c_void_p(h))
            except Exception:                              # noqa: BLE001  (interpreter shutdown)
                pass


class Stream:
    """A non-blocking hipStream_t.  Lives as long as the process (a run makes three or four)."""
    __slots__ = ("handle", "device")

    def __init__(self, device=0, _handle: Optional[int] = None):
        self.device = _dev_index(device)
        if _handle is None:
            h = ctypes.c_void_p()
            _chk(_lib.load().gci_dev_stream_create(self.device, ctypes.byref(h)), "gci_dev_stream_create")
            _handle = h.value
        self.handle = _handle

    @property
    def cuda_stream(self) -> int:          # (the name torch gives the raw handle: what gci_ctx_create / gci_stage_send take)
        return self.handle or 0

    def wait_event(self, ev: Event) -> None:
        if ev.handle is not None:
            _chk(_lib.load().gci_dev_stream_wait_event(self.device, ctypes.c_void_p(self.handle), ctypes.c_void_p(ev.handle)),
                 "gci_dev_stream_wait_event")

    def wait_stream(self, other: "Stream") -> None:
        if other is self or other.handle == self.handle:
            return
        ev = Event(device=self.device)
        ev.record(other)
        self.wait_event(ev)

    def synchronize(self) -> None:
        _chk(_lib.load().gci_dev_stream_sync(self.device, ctypes.c_void_p(self.handle)), "gci_dev_stream_sync")


_TLS = threading.local()
_DEFAULT_STREAMS: Dict[int, Stream] = {}
_GLOBAL_LOCK = threading.Lock()


def default_stream(device=0) -> Stream:
    d = _dev_index(device)
    with _GLOBAL_LOCK:
        s = _DEFAULT_STREAMS.get(d)
        if s is None:
            s = _DEFAULT_STREAMS[d] = Stream(d)
        return s


def current_stream(device=0) -> Stream:
    d = _dev_index(device)
    cur = getattr(_TLS, "cur", None)
    if cur:
        s = cur.get(d)
        if s is not None:
            return s
    return default_stream(d)


class stream:                              # noqa: N801  (torch.cuda.stream's name)
    """with hbm.stream(s): allocations, copies and fills of the calling thread go to `s`."""

    def __init__(self, s: Stream):
        self.s, self.prev = s, None

    def __enter__(self):
        cur = getattr(_TLS, "cur", None)
        if cur is None:
            cur = _TLS.cur = {}
        self.prev = cur.get(self.s.device)
        cur[self.s.device] = self.s
        return self.s

    def __exit__(self, *exc):
        if self.prev is None:
            _TLS.cur.pop(self.s.device, None)
        else:
            _TLS.cur[self.s.device] = self.prev
        return False


def synchronize(device=0) -> None:
    _chk(_lib.load().gci_dev_sync(_dev_index(device)), "gci_dev_sync")


def is_available() -> bool:
    n = ctypes.c_int(0)
    try:
        _lib.load().gci_dev_count(ctypes.byref(n))
    except Exception:                                      # noqa: BLE001
        return False
    return n.value > 0


# ---- the allocator ---------------------------------------------------------------------------------------------------

def _round_size(n: int) -> int:
    """Sizes are classes -- multiples of 512 B below 1 MiB, of 1/16 of their power of two above (<= 6 % over) -- so that the
    buffers a run asks for again and again (a run's inflated bytes, its pages, the scratch of a step) find the block the last
    one gave back."""
    n = max(int(n), 1)
    if n < (1 << 20):
        return (n + 511) & ~511
    step = 1 << (n.bit_length() - 5)
    return (n + step - 1) // step * step


class _Pool:
    """Cached device blocks of one device, per stream: a block given back by a buffer that lived in stream S's order is handed to
    the next request made under S without waiting for anything -- whatever still reads or writes it was enqueued on S before
    (torch's caching allocator works the same way, and the ingestion's double buffers lean on it).  Uses on other streams are
    told with Buf.record_stream(): the block then carries an event per such stream, and its next owner's stream waits for them
    on the device."""

    def __init__(self, device: int):
        self.device = device
        self.lock = threading.Lock()
        self.free: Dict[int, List[Tuple[int, int, tuple]]] = {}      # stream handle -> [(size, ptr, events)]
        self.held = 0                                                # bytes of device memory this pool owns (in use + cached)

    def take(self, nbytes: int, s: Stream) -> Tuple[int, int]:
        size = _round_size(nbytes)
        best = None
        with self.lock:
            lst = self.free.get(s.handle)
            if lst:
                limit = size + (size >> 2) if size >= (1 << 20) else size
                for k, (sz, _, _) in enumerate(lst):
                    if size <= sz <= limit and (best is None or sz < lst[best][0]):
                        best = k
                if best is not None:
                    sz, ptr, events = lst.pop(best)
        if best is not None:
            for ev in events:
                s.wait_event(ev)
            return ptr, sz
        lib = _lib.load()
        p = ctypes.c_void_p()
        st = lib.gci_dev_malloc(self.device, size, ctypes.byref(p))
        if st == _lib.GCI_E_NOMEM:
            self.release_cached()
            st = lib.gci_dev_malloc(self.device, size, ctypes.byref(p))
        _chk(st, "gci_dev_malloc(%d bytes)" % size)
        with self.lock:
            self.held += size
        return p.value, size

    def give_back(self, ptr: int, size: int, stream_handle: int, events: tuple) -> None:
        with self.lock:
            self.free.setdefault(stream_handle, []).append((size, ptr, events))

    def release_cached(self) -> None:
        """Everything cached goes back to the driver (the device is synchronised first: a cached block may still be in use by
        work that was enqueued before it was given back)."""
        lib = _lib.load()
        with self.lock:
            blocks = [b for lst in self.free.values() for b in lst]
            self.free.clear()
        if blocks:
            lib.gci_dev_sync(self.device)
            for size, ptr, _ in blocks:
                lib.gci_dev_free(self.device, ctypes.c_void_p(ptr))
            with self.lock:
                self.held -= sum(b[0] for b in blocks)


_POOLS: Dict[int, _Pool] = {}


def _pool(device: int) -> _Pool:
    with _GLOBAL_LOCK:
        p = _POOLS.get(device)
        if p is None:
            p = _POOLS[device] = _Pool(device)
        return p


def empty_cache(device=0) -> None:
    _pool(_dev_index(device)).release_cached()


def memory_held(device=0) -> int:
    return _pool(_dev_index(device)).held


class _Block:
    """One allocation: goes back to its pool when the last buffer that views it dies."""
    __slots__ = ("ptr", "size", "device", "stream_handle", "extra")

    def __init__(self, ptr, size, device, stream_handle):
        self.ptr, self.size, self.device, self.stream_handle = ptr, size, device, stream_handle
        self.extra: Optional[List[Stream]] = None

    def __del__(self):
        try:
            events = ()
            if self.extra:
                evs = []
                for s in self.extra:
                    ev = Event(device=self.device)
                    ev.record(s)
                    evs.append(ev)
                events = tuple(evs)
            _POOLS[self.device].give_back(self.ptr, self.size, self.stream_handle, events)
        except Exception:                                  # noqa: BLE001  (interpreter shutdown: the process's memory goes with it)
            pass


class HostArray(np.ndarray):
    """What Buf.cpu() returns: a numpy array that also answers .numpy() (the host code says `.cpu().numpy()` to either provider)."""

    def numpy(self):
        return self.view(np.ndarray)

    def cpu(self):
        return self


def _prod(shape) -> int:
    n = 1
    for x in shape:
        n *= int(x)
    return n


class Buf:
    """A contiguous array in HBM: pointer, shape, dtype.  Slices along the first axis are views."""
    __slots__ = ("_blk", "ptr", "shape", "dtype", "device")

    def __init__(self, blk, ptr: int, shape: Tuple[int, ...], dtype: DType, device: Device):
        self._blk, self.ptr, self.shape, self.dtype, self.device = blk, ptr, tuple(int(x) for x in shape), dtype, device

    # -- what the ctypes calls need
    def data_ptr(self) -> int:
        return self.ptr

    def numel(self) -> int:
        return _prod(self.shape)

    @property
    def nbytes(self) -> int:
        return _prod(self.shape) * self.dtype.itemsize

    def __len__(self) -> int:
        return self.shape[0]

    def dim(self) -> int:
        return len(self.shape)

    def __repr__(self):
        return "hbm.Buf(shape=%s, dtype=%s, ptr=0x%x)" % (self.shape, self.dtype.name, self.ptr)

    # -- views
    def _row_bytes(self) -> int:
        return _prod(self.shape[1:]) * self.dtype.itemsize

    def __getitem__(self, key) -> "Buf":
        if isinstance(key, slice):
            lo, hi, step = key.indices(self.shape[0])
            if step != 1:
                raise NotImplementedError("hbm.Buf: strided slices")
            return Buf(self._blk, self.ptr + lo * self._row_bytes(), (max(hi - lo, 0),) + self.shape[1:], self.dtype, self.device)
        if isinstance(key, (int, np.integer)):
            k = int(key)
            if k < 0:
                k += self.shape[0]
            if not 0 <= k < self.shape[0]:
                raise IndexError(key)
            return Buf(self._blk, self.ptr + k * self._row_bytes(), self.shape[1:], self.dtype, self.device)
        raise NotImplementedError("hbm.Buf: only slices and integers along the first axis (got %r)" % (key,))

    def __setitem__(self, key, value) -> None:
        dst = self[key]
        if isinstance(value, (Buf, np.ndarray, PinnedBuf)):
            dst.copy_(value)
        elif isinstance(value, (int, np.integer)):
            v = int(value)
            if v == 0 or self.dtype is uint8 or v == -1:
                dst._fill(v & 0xFF)
            else:
                raise NotImplementedError("hbm.Buf: fill with %r" % (value,))
        else:
            raise TypeError(type(value))

    def view(self, dtype: DType) -> "Buf":
        if dtype is self.dtype:
            return self
        last = self.shape[-1] * self.dtype.itemsize
        if last % dtype.itemsize:
            raise ValueError("hbm.Buf.view: the last axis does not divide")
        return Buf(self._blk, self.ptr, self.shape[:-1] + (last // dtype.itemsize,), dtype, self.device)

    def reshape(self, *shape) -> "Buf":
        if len(shape) == 1 and isinstance(shape[0], (tuple, list)):
            shape = tuple(shape[0])
        n = self.numel()
        if -1 in shape:
            known = _prod([x for x in shape if x != -1])
            shape = tuple(n // max(known, 1) if x == -1 else x for x in shape)
        if _prod(shape) != n:
            raise ValueError("hbm.Buf.reshape: %s -> %s" % (self.shape, shape))
        return Buf(self._blk, self.ptr, shape, self.dtype, self.device)

    def contiguous(self) -> "Buf":
        return self

    # -- work (on the calling thread's current stream)
    def _fill(self, byte: int) -> None:
        s = current_stream(self.device.index)
        _chk(_lib.load().gci_dev_memset_async(self.device.index, ctypes.c_void_p(self.ptr), int(byte), self.nbytes, ctypes.c_void_p(s.handle)),
             "gci_dev_memset_async")

    def zero_(self) -> "Buf":
        self._fill(0)
        return self

    def copy_(self, src, non_blocking: bool = False) -> "Buf":
        lib, d = _lib.load(), self.device.index
        s = current_stream(d)
        if isinstance(src, Buf):
            if src.nbytes != self.nbytes:
                raise ValueError("hbm.Buf.copy_: %d bytes into %d" % (src.nbytes, self.nbytes))
            _chk(lib.gci_dev_memcpy_async(d, ctypes.c_void_p(self.ptr), ctypes.c_void_p(src.ptr), self.nbytes, 3, ctypes.c_void_p(s.handle)),
                 "gci_dev_memcpy_async(d2d)")
            return self
        if isinstance(src, PinnedBuf):
            if src.nbytes != self.nbytes:
                raise ValueError("hbm.Buf.copy_: %d bytes into %d" % (src.nbytes, self.nbytes))
            _chk(lib.gci_dev_memcpy_async(d, ctypes.c_void_p(self.ptr), ctypes.c_void_p(src.ptr), self.nbytes, 1, ctypes.c_void_p(s.handle)),
                 "gci_dev_memcpy_async(h2d)")
            if not non_blocking:
                s.synchronize()
            return self
        a = np.ascontiguousarray(src)
        if a.nbytes != self.nbytes:
            raise ValueError("hbm.Buf.copy_: %d bytes into %d" % (a.nbytes, self.nbytes))
        if a.nbytes:
            _chk(lib.gci_dev_memcpy_async(d, ctypes.c_void_p(self.ptr), ctypes.c_void_p(a.ctypes.data), a.nbytes, 1, ctypes.c_void_p(s.handle)),
                 "gci_dev_memcpy_async(h2d)")
            s.synchronize()                  # (pageable memory: the array may die with this call)
        return self

    def clone(self) -> "Buf":
        out = empty(self.shape, self.dtype, self.device)
        if self.nbytes:
            out.copy_(self)
        return out

    def cpu(self) -> HostArray:
        out = np.empty(self.shape, dtype=self.dtype.np)
        if self.nbytes:
            d = self.device.index
            s = current_stream(d)
            _chk(_lib.load().gci_dev_memcpy_async(d, ctypes.c_void_p(out.ctypes.data), ctypes.c_void_p(self.ptr), self.nbytes, 2,
                                                  ctypes.c_void_p(s.handle)), "gci_dev_memcpy_async(d2h)")
            s.synchronize()
        return out.view(HostArray)

    def item(self):
        if self.numel() != 1:
            raise ValueError("hbm.Buf.item: %d elements" % self.numel())
        return self.cpu().numpy().reshape(-1)[0].item()

    def tolist(self):
        return self.cpu().numpy().tolist()

    def record_stream(self, s: Stream) -> None:
        """The buffer is (also) used by work on `s`: its block is not handed out again before that work is through."""
        blk = self._blk
        if blk is not None and s.handle != blk.stream_handle:
            if blk.extra is None:
                blk.extra = []
            if all(x.handle != s.handle for x in blk.extra):
                blk.extra.append(s)


class PinnedBuf:
    """Page-locked host bytes (gci_dev_host_alloc): the staging end of an asynchronous copy."""
    __slots__ = ("_own", "ptr", "nbytes", "device", "_arr")

    def __init__(self, own, ptr: int, nbytes: int, device: int):
        self._own, self.ptr, self.nbytes, self.device = own, ptr, int(nbytes), device
        self._arr = None

    @property
    def shape(self):
        return (self.nbytes,)

    def numpy(self) -> np.ndarray:
        if self._arr is None:
            self._arr = np.ctypeslib.as_array((ctypes.c_uint8 * max(self.nbytes, 1)).from_address(self.ptr))[:self.nbytes]
        return self._arr

    def __getitem__(self, key) -> "PinnedBuf":
        lo, hi, step = key.indices(self.nbytes)
        if step != 1:
            raise NotImplementedError
        return PinnedBuf(self._own, self.ptr + lo, max(hi - lo, 0), self.device)

    def copy_(self, src: Buf, non_blocking: bool = False) -> "PinnedBuf":
        if src.nbytes != self.nbytes:
            raise ValueError("hbm.PinnedBuf.copy_: %d bytes into %d" % (src.nbytes, self.nbytes))
        s = current_stream(self.device)
        if self.nbytes:
            _chk(_lib.load().gci_dev_memcpy_async(self.device, ctypes.c_void_p(self.ptr), ctypes.c_void_p(src.ptr), self.nbytes, 2,
                                                  ctypes.c_void_p(s.handle)), "gci_dev_memcpy_async(d2h)")
        if not non_blocking:
            s.synchronize()
        return self


class _PinnedOwner:
    __slots__ = ("ptr", "device")

    def __init__(self, ptr, device):
        self.ptr, self.device = ptr, device

    def __del__(self):
        try:
            _lib.load().gci_dev_host_free(self.device, ctypes.c_void_p(self.ptr))
        except Exception:                                  # noqa: BLE001
            pass


def pinned(nbytes: int, device=0) -> PinnedBuf:
    d = _dev_index(device)
    p = ctypes.c_void_p()
    _chk(_lib.load().gci_dev_host_alloc(d, int(nbytes), ctypes.byref(p)), "gci_dev_host_alloc")
    return PinnedBuf(_PinnedOwner(p.value, d), p.value, int(nbytes), d)


def _shape(shape) -> Tuple[int, ...]:
    if isinstance(shape, (int, np.integer)):
        return (int(shape),)
    return tuple(int(x) for x in shape)


def empty(shape, dtype: DType = uint8, device=None) -> Buf:
    dev = device if isinstance(device, Device) else Device(_dev_index(device))
    shape = _shape(shape)
    s = current_stream(dev.index)
    ptr, size = _pool(dev.index).take(_prod(shape) * dtype.itemsize, s)
    return Buf(_Block(ptr, size, dev.index, s.handle), ptr, shape, dtype, dev)


def zeros(shape, dtype: DType = uint8, device=None) -> Buf:
    b = empty(shape, dtype, device)
    if b.nbytes:
        b._fill(0)
    return b


def from_numpy(a: np.ndarray, device=None) -> Buf:
    a = np.ascontiguousarray(a)
    dt = _BY_NP.get(a.dtype)
    if dt is None:
        raise TypeError("hbm.from_numpy: dtype %s" % a.dtype)
    b = empty(a.shape, dt, device)
    if a.nbytes:
        b.copy_(a.view(dt.np) if a.dtype != dt.np else a)
    return b


def cat(bufs: Sequence[Buf]) -> Buf:
    bufs = list(bufs)
    first = bufs[0]
    n = sum(b.shape[0] for b in bufs)
    out = empty((n,) + first.shape[1:], first.dtype, first.device)
    at = 0
    for b in bufs:
        if b.shape[1:] != first.shape[1:] or b.dtype is not first.dtype:
            raise ValueError("hbm.cat: mismatched parts")
        if b.shape[0]:
            out[at:at + b.shape[0]].copy_(b)
        at += b.shape[0]
    return out


def add_i64(b: Buf, delta: int) -> Buf:
    """-> a new int64 buffer: b + delta."""
    out = empty(b.shape, int64, b.device)
    s = current_stream(b.device.index)
    _chk(_lib.load().gci_dev_i64_add(b.device.index, ctypes.c_void_p(b.ptr), b.numel(), int(delta), ctypes.c_void_p(out.ptr), ctypes.c_void_p(s.handle)),
         "gci_dev_i64_add")
    return out


def rec_flags_and(recs: Buf, mask: int) -> None:
    """gci_rec.flags &= mask for every record of a uint8 [n, 32] buffer, in place."""
    s = current_stream(recs.device.index)
    _chk(_lib.load().gci_dev_rec_flags_and(recs.device.index, ctypes.c_void_p(recs.ptr), recs.shape[0], int(mask), ctypes.c_void_p(s.handle)),
         "gci_dev_rec_flags_and")


def scan_u32_u64(b: Buf) -> Buf:
    """-> int64 [n + 1]: 0, b[0], b[0] + b[1], ... (the entries of b read as uint32)."""
    n = b.numel()
    out = empty(n + 1, int64, b.device)
    s = current_stream(b.device.index)
    _chk(_lib.load().gci_dev_u32_scan_u64(b.device.index, ctypes.c_void_p(b.ptr), n, ctypes.c_void_p(out.ptr), ctypes.c_void_p(s.handle)),
         "gci_dev_u32_scan_u64")
    return out


def is_buffer(x) -> bool:
    return isinstance(x, Buf)


def device(index: int = 0) -> Device:
    return Device(index)


def set_device(index) -> None:
    pass                                      # (every gci_dev_* call names its device)


name = "native"


# =====================================================================================================================
# torch provider: the same interface over torch (contig-sharded runs, the test fixtures)
# =====================================================================================================================

class _Torch:
    name = "torch"

    def __init__(self):
        import torch
        self.t = torch
        self.uint8, self.int16, self.int32, self.int64 = torch.uint8, torch.int16, torch.int32, torch.int64
        self.Stream = lambda device=None: torch.cuda.Stream(device=device)
        self.Event = lambda enable_timing=False, device=None: torch.cuda.Event(enable_timing=enable_timing)
        self.stream = torch.cuda.stream
        self.cat = torch.cat

    def device(self, index: int = 0):
        return self.t.device("cuda", index)

    def set_device(self, dev) -> None:
        self.t.cuda.set_device(dev)

    def is_available(self) -> bool:
        return bool(self.t.cuda.is_available())

    def current_stream(self, device=None):
        return self.t.cuda.current_stream(device)

    def synchronize(self, device=None) -> None:
        self.t.cuda.synchronize()

    def empty(self, shape, dtype=None, device=None):
        return self.t.empty(shape, dtype=dtype or self.t.uint8, device=device)

    def zeros(self, shape, dtype=None, device=None):
        return self.t.zeros(shape, dtype=dtype or self.t.uint8, device=device)

    def from_numpy(self, a: np.ndarray, device=None):
        a = np.ascontiguousarray(a)
        if not a.flags.writeable:
            a = a.copy()
        if a.dtype == np.uint64:
            a = a.view(np.int64)
        elif a.dtype == np.uint32:
            a = a.view(np.int32)
        elif a.dtype == np.uint16:
            a = a.view(np.int16)
        return self.t.from_numpy(a).to(device)

    def pinned(self, nbytes: int, device=None):
        return self.t.empty(int(nbytes), dtype=self.t.uint8).pin_memory()

    def add_i64(self, b, delta: int):
        return b + int(delta)

    def rec_flags_and(self, recs, mask: int) -> None:
        recs[:, 29] &= int(mask)

    def scan_u32_u64(self, b):
        t = self.t
        out = t.zeros(int(b.shape[0]) + 1, dtype=t.int64, device=b.device)
        t.cumsum(b.view(t.int32).to(t.int64) & 0xFFFFFFFF, 0, out=out[1:])
        return out

    def is_buffer(self, x) -> bool:
        return isinstance(x, self.t.Tensor)

    def empty_cache(self, device=None) -> None:
        self.t.cuda.empty_cache()


_TORCH: Optional[_Torch] = None


def torch_provider() -> _Torch:
    global _TORCH
    if _TORCH is None:
        _TORCH = _Torch()
    return _TORCH


def native():
    return sys.modules[__name__]


def provider(kind: Optional[str] = None):
    """The provider an Engine is made with: `kind` ("native" / "torch"), else GCI_HBM, else torch if the process has imported it
    already, else native."""
    kind = kind or os.environ.get("GCI_HBM") or ("torch" if "torch" in sys.modules else "native")
    if kind == "torch":
        return torch_provider()
    if kind == "native":
        return native()
    raise ValueError("GCI_HBM must be 'native' or 'torch'")


def provider_of(buf):
    """The provider a buffer (or stream, or event) came from."""
    return native() if isinstance(buf, (Buf, PinnedBuf, Stream, Event)) else torch_provider()
