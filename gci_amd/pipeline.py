"""Host-side mirror of the reference's hot-path functions, driving the HIP kernels.

Same names, argument meaning, outputs and error behaviour as /root/reference/GCI.py so that a
maintainer (and the parity tests) can read one against the other:

    get_Ns_ref            GCI.py:18-46      filter                  GCI.py:172-312
    merge_gaps_depths     GCI.py:315-329    merge_two_type_depth    GCI.py:332-353
    collapse_depth_range  GCI.py:356-390    merge_depth             GCI.py:393-419
    compute_index         GCI.py:522-657    GCI                     GCI.py:897-1028

What differs is where the work happens: `depths` is a `DepthTracks` object whose per-base
data lives in HBM as one int32 buffer for all contigs; records are decoded and filtered, names
joined, depth built, masked, merged, scanned and rendered to text by gfx950 kernels
(gci_amd/csrc/gci_hip.hip).  The host parses containers (BGZF, PAF, FASTA), does the tiny
interval algebra (gci_amd/score.py) and writes files.
"""
from __future__ import annotations

import ctypes
import os
import threading
import warnings
import sys
from concurrent.futures import ThreadPoolExecutor
from typing import Dict, List, Optional, Sequence, Set, Tuple

import numpy as np

from . import hbm, phases, score
from .device import Buffer, Engine, JoinInput, REC_DTYPE, name_hash_np
from . import _lib
from ._lib import GciError, REC_HQ, REC_PASS
from .formats import bam as bamfmt
from .formats import bgzf
from .formats import fasta

_ENGINE: Optional[Engine] = None
SHARD = None              # gci_amd.shard.Context of a multi-GPU run (set by cli.main), None for a single process


_ENGINE_LOCK = threading.Lock()


def default_engine() -> Engine:
    global _ENGINE
    with _ENGINE_LOCK:                                    # (the helper thread of prefetch_member_tables may be the first to ask)
        if _ENGINE is None:
            # (a contig-sharded run exchanges tensors through torch.distributed: its buffers are torch's; a single process holds its
            # own -- hbm.py -- unless GCI_HBM or an already imported torch says otherwise)
            _ENGINE = Engine(SHARD.device_index if SHARD is not None else 0, backend="torch" if SHARD is not None else None)
        return _ENGINE


def note_device_memory() -> None:
    """Into the phase log: what the process took from the driver (the arena's slabs: include/gci_hip.h gci_dev_arena_info) and what of it
    is handed out at the end of the run -- a driver allocation costs by the GB on this chip, so this is a number to keep small."""
    if _ENGINE is None or not phases.on():
        return
    try:
        r, u, n = ctypes.c_uint64(0), ctypes.c_uint64(0), ctypes.c_uint32(0)
        _ENGINE.lib.gci_dev_arena_info(_ENGINE.device.index or 0, ctypes.byref(r), ctypes.byref(u), ctypes.byref(n))
        phases.note("device_memory", {"provider": _ENGINE.T.name, "arena_slab_bytes": int(r.value), "arena_bytes_handed_out": int(u.value),
                                      "arena_slabs": int(n.value), "pool_bytes_held": int(hbm.memory_held(_ENGINE.device.index or 0)) if _ENGINE.T.name == "native" else None})
    except Exception:                                      # noqa: BLE001  (a note, not a result)
        pass


def _slice_bound(v: int, L: int) -> int:
    if v < 0:
        return max(v + L, 0)
    return min(v, L)


class DepthTracks:
    """The reference's `depths` dict (contig -> per-base array), resident in HBM."""

    def __init__(self, engine: Engine, targets_length: Dict[str, int], track: Buffer):
        self.engine = engine
        self.targets_length = dict(targets_length)
        self.targets = list(targets_length.keys())
        self.lengths = [int(targets_length[t]) for t in self.targets]
        self.track = track
        self.all_targets = list(self.targets)      # contig-sharded run: every selected contig in header order (this rank holds `targets`)
        # by-products of the fused build, valid until the track is modified (gap mask)
        self._fresh_runs = None        # ((lo, hi, flank), per-contig raw runs)
        self._fresh_sums = None
        # N runs whose mask is still to be applied (merge_gaps_depths(lazy=True): a two-read-type run masks both tracks in the one
        # pass that also merges them, merge_two_type_depth) -- applied by whatever looks at the track first
        self._pending_gaps: Optional[np.ndarray] = None
        self._masked_with: Optional[bytes] = None      # the rows of the mask this track already carries

    def invalidate(self) -> None:
        self._fresh_runs = None
        self._fresh_sums = None

    def _bind_layout(self) -> None:
        if self.engine.lengths != self.lengths:
            self.engine.set_layout(self.lengths)

    def _bind(self) -> None:
        self._bind_layout()
        if self._pending_gaps is not None:
            rows, self._pending_gaps = self._pending_gaps, None
            self.engine.gap_mask(self.track, self.engine.to_device(rows))
            self._masked_with = rows.tobytes()
            self.invalidate()

    def keys(self):
        return self.targets_length.keys()

    def __contains__(self, t) -> bool:
        return t in self.targets_length

    def __len__(self) -> int:
        return len(self.targets)

    def __getitem__(self, target: str) -> np.ndarray:
        """Host copy of one contig (int64, as the reference's arrays are)."""
        self._bind()
        c = self.targets.index(target)
        o = self.engine.offsets[c]
        return self.track[o:o + self.lengths[c]].cpu().numpy().astype(np.int64)

    def to_host(self) -> Dict[str, np.ndarray]:
        return {t: self[t] for t in self.targets}

    def sums(self) -> np.ndarray:
        if self._fresh_sums is not None:
            return self._fresh_sums
        self._bind()
        return self.engine.depth_sum(self.track)

    def mean(self) -> float:
        """np.mean over the concatenation of all contigs (GCI.py:862-868): integers, so exact.  In a contig-sharded run
        the numerator and denominator are ONE integer all-reduce over the ranks (RCCL over xGMI)."""
        total, bases = int(self.sums().sum()), sum(self.lengths)
        if _sharded():
            total, bases = SHARD.all_reduce_sum([total, bases])
        return float(total) / float(bases)


# ==============================================================================================
# gaps
# ==============================================================================================

def _is_root() -> bool:
    return SHARD is None or SHARD.root


def _sharded() -> bool:
    """A contig-sharded run: more than one rank -- or ONE rank made to take the sharded path (GCI_FORCE_SHARDED=1 under
    torch.distributed.run --nproc-per-node 1: every collective of the multi-GPU path, RCCL included, on a single GPU)."""
    return SHARD is not None and (SHARD.world > 1 or SHARD.forced)


def refuse_overwrite(path: str, force) -> None:
    """The reference's guard in front of every output file (`-f` / `--force`).  In a contig-sharded run rank 0 looks -- another
    rank could find what rank 0 has written in the meantime -- and all ranks leave together."""
    hit = os.path.exists(path) and force == False  # noqa: E712
    if _sharded():
        hit = SHARD.all_reduce_sum([1 if (hit and _is_root()) else 0])[0] > 0
    if hit:
        sys.exit(f'ERROR!!! The file "{path}" exists\nPlease use "-f" or "--force" to rewrite')


def get_Ns_ref(reference=None, prefix="GCI", directory=".", force=False):
    if _INGEST_AHEAD:                                                     # (a helper thread owns the default context just now)
        side = _side_engine()
        with side.T.stream(side.stream):
            _, ns_bed = fasta.n_runs_device(side, reference)
    else:
        _, ns_bed = fasta.n_runs_device(default_engine(), reference)      # N4: the scan itself runs on the GPU
    if len(ns_bed) > 0:
        path = f"{directory}/{prefix}.gaps.bed"
        refuse_overwrite(path, force)
        if _is_root():
            with open(path, "w") as f:
                for target, segments in ns_bed.items():
                    for a, b in segments:
                        f.write(f"{target}\t{a}\t{b}\n")
        return ns_bed, path
    return None, None


def merge_gaps_depths(depths: DepthTracks = None, Ns_bed=None, lazy: bool = False) -> DepthTracks:
    """lazy: the mask is noted and applied by whatever reads the track first -- in a two-read-type run that is the one pass of
    merge_two_type_depth, which masks both tracks, merges them and finds the issue runs of all three."""
    if Ns_bed is not None:
        rows = [(depths.targets.index(t), a, b, 0) for t, segs in Ns_bed.items() if t in depths for a, b in segs]
        if rows:
            arr = np.asarray(rows, dtype=np.int32).reshape(-1, 4)
            if depths._pending_gaps is None and depths._masked_with == arr.tobytes():
                return depths                         # masked with exactly these runs already (the merged track of two masked ones)
            depths._bind()
            if lazy:
                depths._pending_gaps = arr
                depths.invalidate()
                return depths
            depths.engine.gap_mask(depths.track, depths.engine.to_device(arr))
            depths._masked_with = arr.tobytes()
            depths.invalidate()
    return depths


# ==============================================================================================
# PAF path (host; SURVEY.md R3 -- GPU tokeniser is a "next" row)
# ==============================================================================================

def _merge_span(pairs: List[Tuple[int, int]]) -> Tuple[int, int, int]:
    """Union of closed-touching blocks: (covered length, start, end of the longest merged block,
    the leftmost one on ties)."""
    pairs = sorted(pairs)
    covered = 0
    best = (-1, 0, 0)
    lo, hi = pairs[0]
    for a, b in pairs[1:] + [(None, None)]:
        if a is not None and hi >= a:
            hi = max(hi, b)
            continue
        covered += hi - lo
        if hi - lo > best[0]:
            best = (hi - lo, lo, hi)
        if a is not None:
            lo, hi = a, b
    return covered, best[1], best[2]


def paf_filter(paf_files: Sequence[str], targets: Sequence[str], map_qual: int, mq_cutoff: int, iden_percent: float
               ) -> Tuple[List[Dict[str, Tuple[str, int, int, int]]], Set[str]]:
    """GCI.py:211-254 as dicts, from the native filter (hostio.paf_filter / gci_paf_filter): what the reference's
    paf_lines and high_qual hold.  The product path (filter()) uploads the native records directly."""
    from . import hostio
    per_file, high_qual = [], set()
    for recs, names, off in hostio.paf_filter(paf_files, targets, map_qual, mq_cutoff, iden_percent):
        r = recs.reshape(-1).view(REC_DTYPE)
        d = {}
        for i in range(r.shape[0]):
            q = bytes(names[int(off[i]):int(off[i + 1])]).decode()
            d[q] = (targets[int(r["contig"][i])], int(r["start"][i]), int(r["end"][i]), int(r["qlen"][i]))
            if int(r["flags"][i]) & REC_HQ:
                high_qual.add(q)
        per_file.append(d)
    return per_file, high_qual


def _paf_join_input(engine: Engine, d: Dict[str, Tuple[str, int, int, int]], high_qual: Set[str],
                    tindex: Dict[str, int]) -> JoinInput:
    names = [q.encode() for q in d.keys()]
    n = len(names)
    recs = np.zeros(n, dtype=REC_DTYPE)
    if n:
        recs["name_hash"] = name_hash_np(names)
        vals = list(d.values())
        recs["contig"] = [tindex[v[0]] for v in vals]
        for fld, k in (("start", 1), ("end", 2), ("qlen", 3)):
            col = np.asarray([v[k] for v in vals], dtype=np.int64)
            if (col > 0x7FFFFFFF).any() or (col < -0x80000000).any():
                raise GciError(-1, "PAF coordinate does not fit int32")
            recs[fld] = col
        recs["rec_idx"] = np.arange(n)
        recs["flags"] = [REC_PASS | (REC_HQ if q in high_qual else 0) for q in d.keys()]
        recs["name_len"] = [len(x) for x in names]
    lens = np.fromiter((len(x) for x in names), dtype=np.int64, count=n)
    off = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(lens, out=off[1:])
    blob = np.frombuffer(b"".join(names) or b"\x00", dtype=np.uint8)
    return JoinInput(engine.to_device(recs.view(np.uint8).reshape(n, 32) if n else np.zeros((0, 32), np.uint8)),
                     engine.to_device(blob), engine.to_device(off), 0)


# Where the PAF path of filter() runs: "gpu" (default, k_paf.hip) or "host" (native threads, gci_paf_filter).
PAF_FILTER = os.environ.get("GCI_PAF", "gpu")

# Every BGZF member's CRC-32 is verified while inflating, as htslib does (a corrupted block that still inflates to the
# right size must not pass silently); GCI_BGZF_CRC=0 skips the check.
BGZF_CRC = os.environ.get("GCI_BGZF_CRC", "1") != "0"

# ingest = "gpu" keeps the whole inflated stream of a file on the device: only files up to this size go that way
GPU_INFLATE_MAX = int(os.environ.get("GCI_GPU_INFLATE_MAX", str(32 << 30)))

# A BAM whose inflated stream exceeds this many bytes is streamed through the GPU chunk by chunk (K1 per chunk,
# compact records + packed names kept, SEQ/QUAL bytes dropped): real 40x whole-genome BAMs inflate to hundreds of GB.
BAM_CHUNK_BYTES = int(os.environ.get("GCI_BAM_CHUNK_BYTES", str(4 << 30)))     # (8 GiB until round 6: the same wall time, 10 GB more device memory)


# What the record filter runs over.  "pages" (default): the records are laid out as RECORD PAGES on the device first
# (gci_bam_pages_*: the ~400 bytes of a record read_sam looks at, 16-byte aligned, no offset table) and filtered there
# (gci_bam_filter_pages) -- the kernel bench.py times; the inflated stream (SEQ / QUAL: 98 % of it) is released as soon as its
# pages exist, the names stay addressable inside the pages.  "stream": gci_bam_filter[_heads] over the stream itself.
K1_MODE = os.environ.get("GCI_K1", "pages")


def _filter_stream(engine: Engine, d_stream: Buffer, d_off: Buffer, has_seq: bool, ref_sel: Buffer, filt,
                   rec_idx_base: int = 0) -> JoinInput:
    """K1 over the records of an inflated BAM stream (has_seq) or a heads stream -> join input (records + where their names are)."""
    map_qual, mq_cutoff, clip_percent, iden_percent = filt
    if K1_MODE == "pages" or not has_seq:                 # (a heads stream is only ever read through its pages)
        with phases.gpu("record pages"):
            pages = engine.bam_pages(d_stream, d_off, has_seq)
        with phases.gpu("record filter (K1)"):
            recs, noff = engine.bam_filter_pages(pages, ref_sel, map_qual, mq_cutoff, clip_percent, iden_percent, rec_idx_base=rec_idx_base)
        ji = JoinInput(recs, pages.buf, noff, 0)
        ji.blob_bytes = int(pages.buf.shape[0]) - pages.blob_off - 16      # CIGARs / oversize records behind the pages (ONT: GBs)
        return ji
    recs = engine.bam_filter(d_stream, d_off, ref_sel, map_qual, mq_cutoff, clip_percent, iden_percent, rec_idx_base=rec_idx_base)
    return JoinInput(recs, d_stream, d_off, 36)


def _keep_part(engine: Engine, ji: JoinInput) -> JoinInput:
    """What is kept of one run of a file that goes through the device run by run: the records and their names -- the pages
    as they are (they ARE the compact form), or, of a whole stream, the names packed (gci_pack_names)."""
    if ji.name_delta == 0 and getattr(ji, "blob_bytes", 0) == 0:
        return JoinInput(ji.recs.clone(), ji.name_base, ji.name_off.clone(), 0)
    # pages with a blob behind them (an ONT run: ~12 KB of CIGAR words per record, tens of GB per file) or a whole stream: only
    # the names are needed from here on -- packed, the rest is released with the run
    names, noff = engine.pack_names(ji)
    recs = ji.recs.clone()
    if ji.name_delta == 0:
        engine.T.rec_flags_and(recs, 3)                   # packed names no longer lie at 16 k + 4: not GCI_REC_NAME16
    return JoinInput(recs, names.clone(), noff[:-1].clone(), 0)


def _concat_parts(engine: Engine, parts: List[JoinInput]) -> JoinInput:
    dev, T = engine.device, engine.T
    if not parts:
        return JoinInput(T.zeros((0, 32), T.uint8, dev), T.zeros(1, T.uint8, dev), T.zeros(1, T.int64, dev), 0)
    if len(parts) == 1:
        return parts[0]
    offs, base = [], 0
    for p in parts:
        offs.append(T.add_i64(p.name_off, base))
        base += (int(p.name_base.shape[0]) + 15) // 16 * 16            # (pages keep their 16-byte phase: GCI_REC_NAME16)
    names = T.zeros(max(base, 1), T.uint8, dev)
    at = 0
    for p in parts:
        names[at:at + int(p.name_base.shape[0])] = p.name_base
        at += (int(p.name_base.shape[0]) + 15) // 16 * 16
    return JoinInput(T.cat([p.recs for p in parts]), names, T.cat(offs), 0)


def _runs_of_members(engine: Engine, isz: np.ndarray, chunk_bytes: int) -> List[Tuple[int, int]]:
    """Runs of members of at most chunk_bytes inflated -- and a whole number of the device's decode rounds each: every member takes
    about as long as every other, so 65 536 members (4 GiB) on a device that decodes 28 672 at a time cost three rounds for the
    work of 2.3 (bench.py's ingest of 64.5 GB: 4.1 s; with whole rounds per run: see DESIGN.md section 5).  Greedy, cut by cut
    on the running sum (a loop over the 3.4 M members of a whole-genome file is a quarter of a second of interpreter)."""
    n = int(isz.shape[0])
    cs = np.concatenate([np.zeros(1, dtype=np.int64), np.cumsum(isz.astype(np.int64))])
    total = int(cs[-1])
    rnd = engine.inflate_round()
    per_run = 0
    if rnd > 0:
        per_run = max(1, int(chunk_bytes // max(1, total // max(1, n))) // rnd) * rnd
    groups, a = [], 0
    while a < n:
        i = int(np.searchsorted(cs, cs[a] + chunk_bytes, side="right")) - 1      # the most members that stay within chunk_bytes
        if per_run:
            i = min(i, a + per_run)
        i = min(n, max(i, a + 1))
        groups.append((a, i))
        a = i
    return groups or [(0, 0)]


class _Members:
    """The member table of a BGZF file as far as it is known -- the whole of it, or the table of the BEGINNING of the file with more
    still being made (`rest`: a future of (offsets, ISIZEs, the future behind that or None)) -- and the runs of members the
    ingestion goes through (`group(k)`).  Runs are cut from what is known (all but the last, possibly partial, one while more is
    to come), so that the first run can be on the device before the walk through the 3.4 M member headers of a whole-genome file
    (0.7 s of page faults) is over; asking for a run behind the known ones waits for the next piece."""

    def __init__(self, engine: Engine, chunk_bytes: int, pos: np.ndarray, isz: np.ndarray, rest=None):
        self.engine, self.chunk_bytes = engine, int(chunk_bytes)
        self.pos, self.isz, self._rest = pos, isz, None
        self.lock = threading.Lock()
        self.groups: List[Tuple[int, int]] = []
        self._extend(pos, isz, rest)

    def _extend(self, pos, isz, rest) -> None:
        done = self.groups[-1][1] if self.groups else 0
        if done < int(isz.shape[0]):
            more = [(a + done, b + done) for a, b in _runs_of_members(self.engine, isz[done:], self.chunk_bytes)]
            self.groups += more[:-1] if rest is not None else more
        self.pos, self.isz, self._rest = pos, isz, rest

    def group(self, k: int) -> Optional[Tuple[int, int]]:
        """Run k as (first member, one past its last), None behind the last run; self.pos / self.isz cover it afterwards."""
        # (The wait for the next piece IS under the lock, and the main thread, asking for run k while the uploader of run k + 1 waits
        #  here for a piece, stands behind it -- 0.25 s at the head of a whole-genome file.  Waiting outside the lock, with pieces of 1,
        #  2 and 4 run sizes, was built and measured on the round's last day and gained nothing: 5.08 s against 5.10 for the command
        #  line at genome size -- the first inflate, now beside the walk through the next piece and the assembly's upload, took what the
        #  wait had taken.  DESIGN.md section 8 "The head of the ingestion".)
        with self.lock:
            while k >= len(self.groups) and self._rest is not None:
                self._extend(*self._rest.result())
            return self.groups[k] if k < len(self.groups) else None

    def is_last(self, k: int) -> bool:
        """Run k is known to be the last one (False while more of the table is to come)."""
        with self.lock:
            return self._rest is None and k + 1 >= len(self.groups)

    def run_bytes(self) -> List[int]:
        """File bytes of every run known so far."""
        with self.lock:
            return [int(self.pos[hi]) - int(self.pos[lo]) for lo, hi in self.groups]

    def whole(self) -> Tuple[np.ndarray, np.ndarray]:
        with self.lock:
            while self._rest is not None:
                self._extend(*self._rest.result())
            return self.pos, self.isz

    def lazy(self) -> bool:
        return self._rest is not None


class _RunUploads:
    """Uploads of a large BGZF file run by run of members, one run AHEAD of its consumer: two device buffers taken in turns, the
    bytes staged through pinned memory by host threads (`_Staging`) and copied on a stream of their own from a helper thread, so
    that run k + 1 crosses PCIe while the device inflates, walks and filters run k."""

    def __init__(self, engine: Engine, raw, members: _Members):
        from concurrent.futures import ThreadPoolExecutor
        self.engine, self.raw, self.m = engine, raw, members
        self.bufs = [None, None]
        self.copy = engine.copy_stream()
        self.freed = [None, None]                            # main-stream event behind the last kernel that read the buffer
        self.staged = isinstance(raw, np.memmap)               # (a mapped file goes through the ring of pinned slots; an array in memory as it is)
        self.staging = engine.staging() if self.staged else None
        self.last_sent = threading.Event()                   # the last run's bytes are all enqueued: the ring is the next file's
        self.pool = ThreadPoolExecutor(1)
        self.pending = {}
        self._start(0)

    def _buffer(self, k: int, need: int):
        """The device buffer of run k (>= need bytes): one of two, made -- or made larger -- when a run asks for more than is there
        (all the runs known by then are looked at, so that a file is served by two allocations, three when its first run was cut
        from the beginning of the table)."""
        i = k & 1
        if self.bufs[i] is None or int(self.bufs[i].shape[0]) < need:
            want = max([need] + [n + 16 for n in self.m.run_bytes()])
            self.bufs[i] = None
            # from the COPY stream's pool: a block recycled there was last used in that stream's order, so the upload need not wait
            # for whatever the main stream is busy with (the file in front is still being inflated when the next file's first
            # run leaves); the main stream's use of it is told to the allocator instead
            T = self.engine.T
            with T.stream(self.copy):
                self.bufs[i] = T.empty(want, T.uint8, self.engine.device)
            self.bufs[i].record_stream(self.engine.stream)
        return self.bufs[i]

    def _start(self, k: int):
        freed = self.freed[k & 1]

        def run():
            g = self.m.group(k)                              # (a run behind the first may wait here for the rest of the table)
            if g is None:
                self.last_sent.set()
                return None
            lo, hi = g
            p0, p1 = int(self.m.pos[lo]), int(self.m.pos[hi])
            n = p1 - p0
            buf = self._buffer(k, n + 16)
            t_send = phases.now()
            T = self.engine.T
            with T.stream(self.copy), warnings.catch_warnings():
                warnings.simplefilter("ignore", UserWarning)     # a read-only memmap is only read
                if freed is not None:
                    self.copy.wait_event(freed)
                if self.staged:
                    self.staging.send(self.raw, p0, p1, buf, self.copy)
                else:
                    buf[:n].copy_(self.engine._host_src(self.raw[p0:p1]))
                buf[n:n + 16].zero_()
                ev = T.Event()
                ev.record(self.copy)
            phases.trace("upload", k, t_send, phases.now(), n)
            if self.m.is_last(k):
                self.last_sent.set()
            return buf[:n + 16], ev

        self.pending[k] = self.pool.submit(run)

    def take(self, k: int):
        """-> run k's bytes on the device (+ 16 zero bytes), ordered before whatever the main stream does next; None: no run k."""
        got = self.pending.pop(k).result()
        if got is None:
            return None
        d_raw, ev = got
        self.engine.stream.wait_event(ev)
        self._start(k + 1)                                   # into the other buffer: its last reader (run k - 1) is enqueued
        return d_raw

    def release(self, k: int):
        """Everything that reads run k's buffer has been enqueued on the main stream: the run after next may overwrite it."""
        ev = self.engine.T.Event()
        ev.record(self.engine.stream)
        self.freed[k & 1] = ev

    def close(self):
        for f in list(self.pending.values()):
            try:
                f.result()
            except Exception:                             # noqa: BLE001
                pass
        self.pending.clear()
        self.pool.shutdown(wait=True)
        self.last_sent.set()


def _bam_join_input_gpu(engine: Engine, path: str, raw, members: _Members, ref_sel_for, filt, upload=None,
                        uploads: Optional[_RunUploads] = None) -> Optional[JoinInput]:
    """ingest = "gpu".  A file whose inflated stream fits GCI_GPU_INFLATE_MAX stays on the device whole (the join reads
    the names inside it); a larger one goes through run by run of members (at most chunk_bytes inflated each): inflate,
    record walk, K1, and only the 32-byte records and the packed names are kept -- the partial record a run ends in is
    put in front of the next one.  None: the parallel record walk lost the chain (the caller takes the host path)."""
    map_qual, mq_cutoff, clip_percent, iden_percent = filt
    hdr = bamfmt.read_header(path)
    ref_sel = ref_sel_for(hdr)
    n_ref = len(hdr.references)
    if not members.lazy() and int(members.isz.sum()) <= GPU_INFLATE_MAX:
        pos, isz = members.pos, members.isz
        total = int(isz.sum())
        if uploads is not None:
            uploads.close()
            uploads = None
        with phases.gpu("bgzf_inflate + crc"):
            if upload is not None:
                d_bam = engine.bgzf_inflate_uploaded(upload, pos, isz, check_crc=BGZF_CRC)
                upload = None
            else:
                d_bam = engine.bgzf_inflate(raw, pos, isz, check_crc=BGZF_CRC)
        with phases.gpu("record walk"):
            d_off, used, ok = engine.bam_record_offsets(d_bam, hdr.first_record, n_ref)
        if not ok:
            return None
        if used != total:
            raise bamfmt.BAMError("truncated BAM: %d trailing bytes do not form a record" % (total - used))
        return _filter_stream(engine, d_bam, d_off, True, ref_sel, filt)
    if upload is not None:                                  # larger than it looked: uploaded run by run instead
        upload["future"].result()
        upload["pool"].shutdown()
        upload = None
    # the bytes of run k + 1 travel while run k is inflated and filtered (the first run may be on its way already:
    # prefetch_member_tables); and (round 6) run k + 1 is INFLATED while run k is walked, paged and filtered: the inflate stays on this
    # engine's stream, everything behind it runs on a second context with a stream of its own (_walk_engine), two output buffers in
    # turns.  The partial record run k ends in is known only when run k has been walked -- by then run k + 1 is being inflated --, so
    # every run is inflated INGEST_HEADROOM bytes into its buffer and the carried bytes are put in front of it afterwards.
    ahead = uploads if uploads is not None else _RunUploads(engine, raw, members)
    T = engine.T
    walk = _walk_engine(engine) if INGEST_OVERLAP else engine
    parts: List[JoinInput] = []
    carry, start, n_done = None, hdr.first_record, 0

    def launch(k: int):
        """Run k's inflate on the engine's stream, as soon as its bytes are on the device; None behind the last run."""
        t_take = phases.now()
        with phases.wall("  wait for the upload of a run (host blocked)"):
            d_raw = ahead.take(k)
        if d_raw is None:
            return None
        t_got = phases.now()
        lo, hi = members.group(k)
        pos, isz = members.pos, members.isz
        p0 = int(pos[lo])
        with phases.gpu("bgzf_inflate + crc"):
            buf, status, total = engine.bgzf_inflate_ahead(d_raw, pos[lo:hi + 1] - np.uint64(p0), isz[lo:hi], INGEST_HEADROOM, check_crc=BGZF_CRC)
        ahead.release(k)
        ev = T.Event()
        ev.record(engine.stream)
        if walk is not engine:
            buf.record_stream(walk.stream)                   # (read over there: not handed out again before that is through)
            status.record_stream(walk.stream)
        phases.trace("run", k, t_take, t_got, phases.now())
        return dict(buf=buf, status=status, lo=lo, ev=ev)

    try:
        k = 0
        cur = launch(0)
        while cur is not None:
            nxt = launch(k + 1)                              # enqueued BEFORE run k is walked: the two overlap on the device
            with T.stream(walk.stream):
                walk.stream.wait_event(cur["ev"])
                try:
                    walk.check_status_word(cur["status"], "gci_bgzf_inflate_device")
                except GciError as e:
                    if e.rec >= 0:
                        e.rec += cur["lo"]
                    raise
                buf = cur["buf"]
                n_carry = int(carry.shape[0]) if carry is not None else 0
                if n_carry > INGEST_HEADROOM:                 # (a record of more than the headroom: put together in a buffer of its own)
                    whole = T.empty(n_carry + int(buf.shape[0]) - INGEST_HEADROOM, T.uint8, engine.device)
                    whole[:n_carry].copy_(carry)
                    whole[n_carry:].copy_(buf[INGEST_HEADROOM:])
                    d_buf = whole
                else:
                    if n_carry:
                        buf[INGEST_HEADROOM - n_carry:INGEST_HEADROOM].copy_(carry)
                    d_buf = buf[INGEST_HEADROOM - n_carry:]
                del buf
                cur = None
                if int(d_buf.shape[0]) <= start:              # still inside the header
                    carry, start = None, start - int(d_buf.shape[0])
                else:
                    with phases.gpu("record walk"):
                        d_off, used, ok = walk.bam_record_offsets(d_buf, start, n_ref)
                    if not ok:
                        return None
                    carry, start = (d_buf[used:].clone() if used < int(d_buf.shape[0]) else None), 0
                    if int(d_off.shape[0]):
                        try:
                            ji = _filter_stream(walk, d_buf, d_off, True, ref_sel, filt, rec_idx_base=n_done)
                        except GciError as e:
                            if e.rec >= 0:
                                e.rec += n_done
                            raise
                        kept = _keep_part(walk, ji)
                        if walk is not engine:                # (made in the walk stream's order, joined in the engine's)
                            for t in (kept.recs, kept.name_base, kept.name_off):
                                t.record_stream(engine.stream)
                        parts.append(kept)
                        n_done += int(d_off.shape[0])
                        del ji
                    del d_off
                del d_buf
            phases.trace("run_done", k, phases.now())
            cur, nxt = nxt, None
            k += 1
    finally:
        ahead.close()
        if walk is not engine:
            engine.stream.wait_stream(walk.stream)           # (what the join reads was written over there)
    if carry is not None:
        raise bamfmt.BAMError("truncated BAM: %d trailing bytes do not form a record" % int(carry.shape[0]))
    return _concat_parts(engine, parts)


# A large file's runs: run k + 1 is inflated while run k is walked, paged and filtered on a second context (GCI_INGEST_OVERLAP=0: one
# after the other on one stream, as in rounds 3 - 5); the bytes kept free in front of every run's inflated bytes for the record the run
# before it ended in (a longer one -- an ONT read of megabases -- is put together in a buffer of its own).
INGEST_OVERLAP = os.environ.get("GCI_INGEST_OVERLAP", "1") != "0"
INGEST_HEADROOM = int(os.environ.get("GCI_INGEST_HEADROOM", str(8 << 20)))
_WALK_ENGINES: Dict[int, Engine] = {}


def _walk_engine(engine: Engine) -> Engine:
    """The context that walks, pages and filters a run while `engine` inflates the next one: a stream and scratch of its own (a gci_ctx is
    one stream and is not shared between streams), the same provider and device."""
    with _ENGINE_LOCK:
        w = _WALK_ENGINES.get(id(engine))
        if w is None:
            w = _WALK_ENGINES[id(engine)] = Engine(engine.device.index or 0, stream=engine.T.Stream(engine.device))
        return w


_DROP_QUEUE = None


def _drop_later(*objs) -> None:
    """Give the last reference of `objs` to a helper thread.  Unmapping a 1.5 GB file mapping whose pages were all touched takes
    14 - 18 ms on the GPU boxes (tools/exp_cli_teardown.py) -- a tenth of the command line's time at chr19, spent by the thread that
    should be launching kernels."""
    global _DROP_QUEUE
    if _DROP_QUEUE is None:
        import queue
        import threading
        _DROP_QUEUE = queue.SimpleQueue()

        def run(q):
            while True:
                q.get()                      # (taken and dropped: the last reference dies here)

        threading.Thread(target=run, args=(_DROP_QUEUE,), daemon=True).start()
    _DROP_QUEUE.put(objs)


_TABLES: Dict[str, object] = {}        # path -> (memory map of the file, future of (member offsets, ISIZEs, first-run uploader or
                                       # None)): prefetch_member_tables


def prefetch_member_tables(paths: Sequence[str]) -> None:
    """Start on the BAM files of a run before anything needs them, on a helper thread, file after file: the BGZF member table
    (host threads over the mapping of the file; GCI_BGZF_TABLE=pread reads the headers through a descriptor instead -- no page
    fault per member, but slower on the boxes measured) and, for a file that will go through the device run by run, the upload
    of its FIRST run.  The first file's table is made in pieces (_Members): its first run leaves as soon as the table of the
    file's beginning is there; a later file's as soon as the file in front of it has put its last run on the copy stream.  The
    command line calls this as soon as it knows its inputs, so what used to sit in front of a file's first byte on the device --
    0.7 s of table and a third of a second of upload per file at genome size -- happens beside the assembly's N scan and beside
    the inflate of the file in front.  bam_join_input() picks the results up; an unreadable or damaged file raises there (a
    damaged member behind the table's first piece: where the run that holds it is asked for)."""
    from concurrent.futures import Future
    from . import hostio
    if os.environ.get("GCI_BAM_INGEST", "gpu") != "gpu" or not paths or _sharded():
        return                                            # (a contig-sharded run reads only its contigs' members, through the index)
    todo = []
    for path in paths:
        if path in _TABLES:
            continue
        try:
            if not os.path.getsize(path):
                continue
            raw = np.memmap(path, dtype=np.uint8, mode="r")
        except OSError:
            continue                                      # (bam_join_input meets the same error itself)
        fut = Future()
        _TABLES[path] = (raw, fut)
        todo.append((path, raw, fut))
    # (the first file's table is wanted as soon as the assembly has been scanned; the later ones have the seconds the file before
    # them takes on the device, and their threads would compete with the ones that stage that file's bytes: a quarter as many)
    many = hostio.default_threads()
    by_fd = os.environ.get("GCI_BGZF_TABLE", "mmap") == "pread"     # (measured at genome size on tmpfs: pread 1.7 - 2.2 s, the mapping 1.2 - 1.4 s)
    first_run = os.environ.get("GCI_FIRST_RUN_AHEAD", "1") != "0"

    def table(path, raw, threads, limit=None, known=None):
        """The member table up to byte `limit` (None: all of the file); known = the table of a beginning of the file: only what lies
        behind it is walked."""
        with phases.wall("bgzf_member_table (ahead, on a helper thread)"):
            if by_fd and limit is None and known is None:
                return hostio.bgzf_blocks_file(path, threads=threads)
            begin = int(known[0][-1]) if known is not None else 0
            if begin >= int(raw.shape[0]):
                return known
            pos, isz = hostio.bgzf_blocks(np.asarray(raw[begin:]), threads=threads, limit=None if limit is None else limit - begin)
            if known is None:
                return pos, isz
            return np.concatenate([known[0][:-1], pos + np.uint64(begin)]), np.concatenate([known[1], isz])

    # The tables of the files BEHIND the first are walked beside the first one's (a thread each), not one after the other: with the
    # uploads at 40 - 57 GB/s a 77 GB file is through the device in under two seconds, and the table of the file behind it -- 1.2 s of page
    # faults over 3.4 M member headers, started only when the first file's own 1.2 s were over -- was not there yet (round 6, one run
    # of the command line at genome size: 1.3 s of waiting for it, 5.7 s instead of 4.4).
    later_pool = ThreadPoolExecutor(max(1, min(4, len(todo) - 1))) if len(todo) > 1 else None
    later = {k: later_pool.submit(table, path, raw, max(2, many // 4)) for k, (path, raw, _) in enumerate(todo) if k > 0} if later_pool else {}

    def chain():
        from concurrent.futures import Future
        before = None                                     # the uploader of the file in front
        for k, (path, raw, fut) in enumerate(todo):
            if _QUIESCE.is_set():                         # (the run is over: nobody will ask for this file's table)
                if not fut.done():
                    fut.set_exception(RuntimeError("the run ended before %s was reached" % path))
                continue
            try:
                threads = max(2, many // 2) if k == 0 else max(2, many // 4)
                n_raw = int(raw.shape[0])
                if k == 0 and first_run and n_raw > GPU_INFLATE_MAX + (GPU_INFLATE_MAX >> 6):
                    # (more bytes than a whole-file ingestion may inflate to: run by run for certain.)  The table of the beginning of
                    # the file, the first run on its way, and only then the walk through the rest of the file's members
                    engine = default_engine()
                    # (5/8 of a run's inflated size in file bytes holds a whole first run at the usual 2.4 - 4 : 1; four times a
                    # run's size holds eight more: a second of inflate, which covers the walk through the rest)
                    limits = [x for x in (BAM_CHUNK_BYTES * 5 // 8, BAM_CHUNK_BYTES * 4) if x < n_raw]
                    rests = [Future() for _ in limits]
                    first = table(path, raw, threads, limit=limits[0]) if limits else table(path, raw, threads)
                    members = _Members(engine, BAM_CHUNK_BYTES, first[0], first[1], rests[0] if rests else None)
                    before = _RunUploads(engine, raw, members)
                    fut.set_result((members, before))
                    nxt = first
                    for j, rest in enumerate(rests):
                        try:
                            nxt = table(path, raw, threads, limit=limits[j + 1] if j + 1 < len(limits) else None, known=nxt)
                            rest.set_result((nxt[0], nxt[1], rests[j + 1] if j + 1 < len(rests) else None))
                        except BaseException as e:        # noqa: BLE001  (raised where the run behind the known ones is asked for)
                            for r in rests[j:]:
                                r.set_exception(e)
                            break
                    continue
                pos, isz = later.pop(k).result() if k in later else table(path, raw, threads)
                if first_run and n_raw > GPU_INFLATE_MAX // 8 and int(isz.sum()) > GPU_INFLATE_MAX:
                    engine = default_engine()
                    members = _Members(engine, BAM_CHUNK_BYTES, pos, isz)
                    if before is not None:
                        while not before.last_sent.wait(0.05):            # (the ring of pinned buffers is the file's in front until then)
                            if _QUIESCE.is_set():
                                break
                    before = _RunUploads(engine, raw, members)
                    fut.set_result((members, before))
                else:
                    fut.set_result(((pos, isz), None))
            except BaseException as e:                    # noqa: BLE001  (handed to the thread that asks for the table)
                if not fut.done():
                    fut.set_exception(e)

    def chain_and_pool():
        try:
            chain()
        finally:
            if later_pool is not None:
                later_pool.shutdown(wait=True)

    th = threading.Thread(target=chain_and_pool, daemon=True)
    _CHAINS.append(th)
    th.start()


_CHAINS: List[threading.Thread] = []    # the helper threads of prefetch_member_tables (quiesce_ahead joins them)
_QUIESCE = threading.Event()             # set: a chain stops in front of its next file
_INGEST_AHEAD: Dict[str, tuple] = {}   # path -> (key, future of its JoinInput): start_ingest_ahead
_SIDE_ENGINE = None


def _targets_of(first, chrs_list) -> Dict[str, int]:
    return {r: l for r, l in zip(first.references, first.lengths) if (len(chrs_list) == 0 or r in chrs_list)}


def start_ingest_ahead(bam_files: Sequence[str], chrs_list, filt: Tuple[int, int, float, float], threads: int) -> None:
    """The ingestion of the run's FIRST BAM file -- upload, inflate, record walk, pages, K1: bam_join_input() as filter() will
    call it -- started on a helper thread before the command line turns to the assembly's N runs: at genome size the device
    inflates for four seconds per file and used to idle through the 0.6 s the assembly takes to cross PCIe and be scanned.
    The helper is the ONLY user of the default engine's context until filter() has taken its result (filter() asks for it
    before it touches the context; the N scan meanwhile runs on a context and stream of its own: get_Ns_ref).  Whatever the
    ingestion raises is raised by filter() where bam_join_input() would have raised it."""
    from concurrent.futures import Future
    if (os.environ.get("GCI_INGEST_AHEAD", "1") == "0" or os.environ.get("GCI_BAM_INGEST", "gpu") != "gpu" or _sharded()
            or not bam_files or bam_files[0] in _INGEST_AHEAD):
        return
    path = bam_files[0]
    try:
        targets = list(_targets_of(bamfmt.read_header(path), chrs_list).keys())
    except Exception:                                     # noqa: BLE001  (filter() meets the same error itself)
        return
    fut = Future()
    engine = default_engine()
    key = (id(engine), tuple(targets), tuple(filt), threads)

    def run():
        try:
            fut.set_result(bam_join_input(engine, path, targets, filt, threads))
        except BaseException as e:                        # noqa: BLE001
            fut.set_exception(e)

    _INGEST_AHEAD[path] = (key, fut)
    threading.Thread(target=run, daemon=True).start()


def quiesce_ahead() -> None:
    """Nothing started ahead is left running or waiting: ingestions not taken by a filter() are waited for (their helper threads
    own the default context until they are done), tables and first-run uploads nobody asked for are dropped."""
    for path in list(_INGEST_AHEAD):
        _, fut = _INGEST_AHEAD.pop(path)
        fut.exception()                                   # (waits; the outcome is nobody's any more)
    # the threads that make tables and first uploads ahead: told to stop in front of their next file, their uploaders (a later
    # file's waits for the ring until the one in front has sent its last run) closed, and waited for -- so that the interpreter never
    # finalises under a thread that is inside HIP or torch (an early sys.exit would otherwise race it)
    _QUIESCE.set()
    for path in list(_TABLES):
        _drop_ahead(_TABLES.pop(path))
    for th in list(_CHAINS):
        if th is not threading.current_thread():
            th.join(timeout=30.0)
    del _CHAINS[:]
    _QUIESCE.clear()


def _side_engine() -> Engine:
    """A second context on a stream of its own (sharing the default engine's ring of pinned buffers): what the main thread works
    with while a helper thread owns the default one (start_ingest_ahead)."""
    global _SIDE_ENGINE
    if _SIDE_ENGINE is None:
        main = default_engine()
        _SIDE_ENGINE = Engine(main.device.index, stream=main.T.Stream(main.device))
        _SIDE_ENGINE._staging = main.staging()
    return _SIDE_ENGINE


def _drop_ahead(ahead) -> None:
    """A prefetched table nobody will use: its uploader (if one was started) is closed once it exists."""
    def done(fut):
        try:
            up = fut.result()[1]
            if up is not None:
                up.close()
        except BaseException:                             # noqa: BLE001
            pass
    ahead[1].add_done_callback(done)


def bam_join_input(engine: Engine, path: str, targets: Sequence[str], filt: Tuple[int, int, float, float],
                   threads: int = 1, chunk_bytes: Optional[int] = None, ingest: Optional[str] = None) -> JoinInput:
    """K1 over one BAM file -> the file's join input (compact records + where their names are).

    ingest = "gpu" (default; GCI_BAM_INGEST overrides): the file's bytes are uploaded as they are and inflated on the
    device, the record offsets come from the parallel walk (_bam_join_input_gpu); a stream that walk cannot follow
    takes the next path.
    ingest = "heads": the native host pipeline (gci_bam_heads) inflates the file group by group and keeps every record
    without its SEQ / QUAL bytes; only that heads stream (about 400 B of a 27 KB HiFi record) is uploaded and filtered
    through its record pages; names stay addressable inside those.
    ingest = "full" (or an explicit chunk_bytes): the whole stream, inflated on the host, goes to the device.  Small
    files: one upload.  Large files: groups of BGZF members are inflated into a host buffer (the next group on a
    background thread while the GPU works on the current one), the partial record at the end of a group is carried
    over, K1 runs per chunk and only the 32-byte records and the packed names (gci_pack_names) are kept on the device."""
    from concurrent.futures import ThreadPoolExecutor
    from . import hostio
    ingest = ingest or ("full" if chunk_bytes else os.environ.get("GCI_BAM_INGEST", "gpu"))
    if ingest not in ("heads", "full", "gpu"):
        raise ValueError("ingest must be 'heads', 'full' or 'gpu'")
    chunk_bytes = int(chunk_bytes or BAM_CHUNK_BYTES)
    nthreads = hostio.pick_threads(threads)
    ahead = _TABLES.pop(path, None)
    if ahead is not None and ingest != "gpu":
        _drop_ahead(ahead)
        ahead = None
    raw = ahead[0] if ahead is not None else (np.memmap(path, dtype=np.uint8, mode="r") if os.path.getsize(path) else np.zeros(0, np.uint8))
    map_qual, mq_cutoff, clip_percent, iden_percent = filt

    def ref_sel_for(hdr):
        for t in targets:
            if t not in hdr.references:
                raise ValueError(f"invalid contig `{t}`")          # what pysam's fetch() raises
        tindex = {t: i for i, t in enumerate(targets)}
        return engine.to_device(np.asarray([tindex.get(r, -1) for r in hdr.references], dtype=np.int32))

    if ingest == "gpu":
        # N1 on the device: the file's bytes are uploaded as they are, the BGZF members are inflated there
        # (gci_bgzf_inflate_device, CRC verified), the record offsets come from a parallel walk
        # (gci_bam_record_offsets_device) and K1 runs over the inflated stream.  A stream the parallel walk cannot follow
        # takes the heads path below.
        # (the member table is made on the host while the file's bytes are on their way to the device; a file that may not
        # fit -- BGZF deflates BAM 2.4 - 4 : 1 -- is uploaded run by run instead)
        upload = None
        if 0 < raw.shape[0] <= GPU_INFLATE_MAX // 8:
            upload = engine.start_upload(raw, parts=2 if raw.shape[0] >= (256 << 20) else 1)
        members = uploads = None
        try:
            with phases.wall("bgzf_member_table"):
                if ahead is not None:
                    members, uploads = ahead[1].result()
                else:
                    members = hostio.bgzf_blocks(np.asarray(raw))
        except BaseException:
            if upload is not None:
                upload["pool"].shutdown()
            raise
        if isinstance(members, _Members) and (members.engine is not engine or members.chunk_bytes != chunk_bytes):
            if uploads is not None:
                uploads.close()                               # (made for another engine or other runs: not this call's)
            members, uploads = members.whole(), None
        if not isinstance(members, _Members):
            members = _Members(engine, chunk_bytes, members[0], members[1])
        if members.lazy() or int(members.isz.sum()) > 0:
            with phases.wall("bam_ingest (upload | inflate + crc | record walk | pages | filter, overlapped)"):
                try:
                    ji = _bam_join_input_gpu(engine, path, raw, members, ref_sel_for, filt, upload, uploads)
                finally:
                    if uploads is not None:
                        uploads.close()                       # (idempotent: a failure in front of the run loop leaves it open)
                if phases.on():
                    engine.T.synchronize()
            if phases.on():
                pos, isz = members.whole()
                phases.add("bgzf_bytes", int(raw.shape[0]))
                phases.add("bgzf_members", int(isz.shape[0]))
                phases.add("inflated_bytes", int(isz.sum()))
            if ji is not None:
                _drop_later(raw)                              # (the unmapping, off this thread)
                del raw
                return ji
        elif upload is not None:
            upload["pool"].shutdown()
        ingest = "heads"
    if ingest == "heads":
        try:
            heads = hostio.bam_heads(np.asarray(raw), threads=nthreads, check_crc=BGZF_CRC)
        except GciError as e:
            if e.status != _lib.GCI_E_NOMEM:
                raise
            heads = None              # the address-space reservation was refused (strict overcommit): whole-stream ingestion
        if heads is not None:
            with heads:
                hdr = bamfmt.parse_header(heads.stream)
                d_bam, d_off = engine.to_device(heads.stream), engine.to_device(heads.offsets)
            return _filter_stream(engine, d_bam, d_off, False, ref_sel_for(hdr), filt)

    pos, isz = hostio.bgzf_blocks(np.asarray(raw))
    if int(isz.sum()) <= chunk_bytes:
        stream = hostio.bgzf_inflate(np.asarray(raw), threads=nthreads, check_crc=BGZF_CRC)
        hdr = bamfmt.parse_header(stream)
        offs, _ = hostio.bam_record_offsets(stream)
        d_bam, d_off = engine.to_device(stream), engine.to_device(offs)
        return _filter_stream(engine, d_bam, d_off, True, ref_sel_for(hdr), filt)

    # ---- streamed ------------------------------------------------------------------------------------------
    groups, a, acc = [], 0, 0
    for i, sz in enumerate(isz.tolist()):
        if acc and acc + sz > chunk_bytes:
            groups.append((a, i))
            a, acc = i, 0
        acc += sz
    groups.append((a, len(isz)))

    def inflate(g):
        lo, hi = g
        return hostio.bgzf_inflate(np.asarray(raw[int(pos[lo]):int(pos[hi])]), threads=nthreads, check_crc=BGZF_CRC)

    parts: List[JoinInput] = []
    carry = np.zeros(0, dtype=np.uint8)
    n_done, hdr, ref_sel = 0, None, None
    with ThreadPoolExecutor(1) as ex:
        nxt = ex.submit(inflate, groups[0])
        for k in range(len(groups)):
            data = nxt.result()
            if k + 1 < len(groups):
                nxt = ex.submit(inflate, groups[k + 1])
            buf = np.concatenate([carry, data]) if carry.shape[0] else data
            start = 0
            if hdr is None:
                try:
                    hdr = bamfmt.parse_header(buf)
                except Exception:                              # header longer than one chunk: keep accumulating
                    carry = buf
                    continue
                start, ref_sel = hdr.first_record, ref_sel_for(hdr)
            offs, used = hostio.bam_chunk_offsets(buf, start)
            carry = buf[used:].copy()
            if offs.shape[0] == 0:
                continue
            d_buf, d_off = engine.to_device(buf[:used]), engine.to_device(offs)
            try:
                ji = _filter_stream(engine, d_buf, d_off, True, ref_sel, filt, rec_idx_base=n_done)
            except GciError as e:
                if e.rec >= 0:
                    e.rec += n_done
                raise
            parts.append(_keep_part(engine, ji))
            n_done += int(offs.shape[0])
            del d_buf, d_off, ji
    if carry.shape[0]:
        raise bamfmt.BAMError("truncated BAM: %d trailing bytes do not form a record" % carry.shape[0])
    if hdr is None:
        raise bamfmt.BAMError("no BAM header in %s" % path)
    return _concat_parts(engine, parts)


def filter(paf_files=[], bam_files=[], prefix="GCI", map_qual=30, mq_cutoff=50, iden_percent=0.9,  # noqa: A001
           clip_percent=0.1, ovlp_percent=0.9, flank_len=15, directory=".", force=False, log_reads_type="",
           chrs_list=[], threads=1, engine: Optional[Engine] = None, write=True, issue_hint=None):
    """Filter the PAF and BAM file(s), build the per-base depth in HBM and write
    `{prefix}.depth.gz`.  Returns (depths, targets_length) like the reference.

    `issue_hint` = (leftmost, rightmost, flank_len) of the collapse_depth_range() call that will follow:
    its run boundaries are then detected in the same pass that builds the depth (used as long as no gap
    mask modifies the track first)."""
    engine = engine or default_engine()
    if write:
        refuse_overwrite(f"{directory}/{prefix}.depth.gz", force)
    print(f"Filtering {log_reads_type} alignment files ...")
    if _sharded():
        return _filter_sharded(paf_files, bam_files, prefix, map_qual, mq_cutoff, iden_percent, clip_percent, ovlp_percent,
                               flank_len, directory, log_reads_type, chrs_list, threads, engine, write, issue_hint)

    first = bamfmt.read_header(bam_files[0])
    targets_length = _targets_of(first, chrs_list)
    targets = list(targets_length.keys())
    tindex = {t: i for i, t in enumerate(targets)}
    filt = (map_qual, mq_cutoff, clip_percent, iden_percent)
    # The depth track is asked for NOW, while the first file is still being ingested on its helper thread (this thread would only
    # wait for it): a driver allocation costs by the GB on this chip -- the kernel driver clears VRAM it does not know to be clean,
    # 35 - 45 ms per GB (tools/hwtests/reserve_timing.py) -- and the 12.5 GB of a human genome's track, asked for behind the last
    # file's last byte, were half a second of the command line with nothing beside them.
    early_track = None
    total_elems = sum((int(targets_length[t]) + _lib.GCI_TILE - 1) // _lib.GCI_TILE * _lib.GCI_TILE for t in targets)
    if total_elems >= (1 << 26) and os.environ.get("GCI_EARLY_TRACK", "1") != "0":
        early_track = engine.T.empty(max(total_elems, 1), engine.T.int32, engine.device)
    # the first file may have been started on already (start_ingest_ahead): its helper thread owns this context until it is done
    first_ahead = None
    for path in [p for p in _INGEST_AHEAD if p != bam_files[0]] + ([bam_files[0]] if bam_files[0] in _INGEST_AHEAD else []):
        key, fut = _INGEST_AHEAD.pop(path)
        try:
            res = ("ok", fut.result())
        except BaseException as e:                        # noqa: BLE001  (raised below, where bam_join_input() would have)
            res = ("err", e)
        if path == bam_files[0] and key == (id(engine), tuple(targets), filt, threads):
            first_ahead = res
    engine.set_layout([targets_length[t] for t in targets])

    inputs: List[JoinInput] = []
    high_qual: Set[str] = set()
    if len(paf_files) != 0:
        try:
            if PAF_FILTER == "host":                          # the native host filter (host_io.cpp), kept as a switch
                from . import hostio
                native = hostio.paf_filter(paf_files, targets, map_qual, mq_cutoff, iden_percent, threads=hostio.pick_threads(threads))
                inputs += [JoinInput(engine.to_device(r), engine.to_device(nm), engine.to_device(off), 0) for r, nm, off in native]
            else:                                             # K2: tokeniser, grouping and scoring on the GPU
                inputs += engine.paf_filter(paf_files, targets, map_qual, mq_cutoff, iden_percent)
                if bam_files and os.environ.get("GCI_PAF_POOL_KEEP_GB") is None:
                    # the PAF stage is over: its pooled scratch (raw device memory, outside torch's allocator) back to the driver before
                    # the BAM ingestion, the join and the depth build ask for theirs (a harness that repeats the stage says KEEP)
                    engine.lib.gci_paf_pool_release(engine.ctx)
        except GciError as e:
            e.paf_replay = (paf_files, targets)               # a bad LINE: Python's own exception for it, as the reference dies
            _reraise_like_reference(e)
    for k, path in enumerate(bam_files):
        try:
            if k == 0 and first_ahead is not None:
                if first_ahead[0] == "err":
                    raise first_ahead[1]
                inputs.append(first_ahead[1])
                first_ahead = None
            else:
                inputs.append(bam_join_input(engine, path, targets, filt, threads))
        except GciError as e:
            _reraise_like_reference(e)
    try:
        with phases.wall("name_join"), phases.gpu("name join"):
            ivl, count = engine.name_join(inputs, ovlp_percent, count_flank=flank_len)     # + the build's counting pass
    except GciError as e:
        _reraise_like_reference(e)
    track = early_track if (early_track is not None and int(early_track.shape[0]) == max(engine.total, 1)) else engine.new_track()
    with phases.wall("depth_build"), phases.gpu("depth build"):
        fused = engine.depth_build_fused(ivl, count, flank_len, track, want_text=False, want_sums=True,
                                         issue=issue_hint, counted=True, want_runs=bool(write))
    depths = DepthTracks(engine, targets_length, track)
    depths._fresh_sums = fused["sums"]
    if issue_hint is not None:
        depths._fresh_runs = (tuple(float(x) for x in issue_hint[:2]) + (int(issue_hint[2]),), fused["runs"])

    print(f"Filtering {log_reads_type} alignment files done!!!")
    if write:
        print(f'Writing depths into "{directory}/{prefix}.depth.gz" ...')
        with phases.wall("write_depth_gz (deflate on the device, D2H, file)"):
            _write_depth_members(directory, prefix, depths, from_build=True)         # (the build just above kept its run lists)
        print("Writing depths done!!!\n\n")
    return depths, targets_length


# ==============================================================================================
# contig-sharded run (one process per GPU)
# ==============================================================================================

def bam_records_of_contigs(engine: Engine, path: str, targets: Sequence[str], own: Sequence[str], filt, threads: int = 1) -> JoinInput:
    """K1 over the records of the contigs `own` of one BAM file -- the part of the reference's fan-out over contigs
    (GCI.py:257-270) that falls to this rank.  gci_rec.contig is the index in `targets` (all selected contigs).

    With an index next to the file (`<bam>.bai`; the reference needs one for pysam's fetch) only the BGZF members
    that hold those records are read and inflated: the pseudo-bin samtools writes per reference (or the hull of its
    chunks) gives the virtual offsets of the contig's first record and of the end of its last one; that run of members
    is inflated and walked on the device (GCI_BAM_INGEST=gpu, the default) or on host threads.  Without an index the
    whole file is ingested and K1 drops the records of the other contigs."""
    from . import hostio
    map_qual, mq_cutoff, clip_percent, iden_percent = filt
    hdr = bamfmt.read_header(path)
    for t in targets:
        if t not in hdr.references:
            raise ValueError(f"invalid contig `{t}`")              # what pysam's fetch() raises
    tindex = {t: i for i, t in enumerate(targets)}
    own_set = set(own)
    ref_sel = engine.to_device(np.asarray([tindex[r] if r in own_set else -1 for r in hdr.references], dtype=np.int32))
    index = bamfmt.read_bai(path + ".bai")
    dev = engine.device
    nthreads = hostio.pick_threads(threads)
    if index is None or len(index) != len(hdr.references):
        raw = np.fromfile(path, dtype=np.uint8)
        with hostio.bam_heads(raw, threads=nthreads, check_crc=BGZF_CRC) as heads:
            d_bam, d_off = engine.to_device(heads.stream), engine.to_device(heads.offsets)
        return _filter_stream(engine, d_bam, d_off, False, ref_sel, filt)
    raw = np.memmap(path, dtype=np.uint8, mode="r")
    n_raw = int(raw.shape[0])
    on_device = os.environ.get("GCI_BAM_INGEST", "gpu") == "gpu"
    parts: List[JoinInput] = []
    n_done = 0
    for r, name in enumerate(hdr.references):
        if name not in own_set or index[r] is None:
            continue
        beg, end = index[r]
        # the member table of THIS contig's run only: from the member of its first record (beg >> 16) to the member its range
        # ends in (end >> 16), hopping BSIZE from header to header -- a rank never touches the members of the other ranks'
        # contigs (round 2 scanned the whole file on every rank: N x the file's pages through the page cache)
        c_beg, c_end = beg >> 16, end >> 16
        if not (0 <= c_beg <= c_end and c_end + 18 <= n_raw):
            raise bamfmt.BAMError("index of %s does not match its BGZF members" % path)
        try:
            run_end = c_end + bgzf.member_size(raw, c_end)                           # (the BC sub-field wherever it stands)
        except bgzf.BGZFError as e:
            raise bamfmt.BAMError("index of %s does not match its BGZF members" % path) from e
        try:
            pos, isz = hostio.bgzf_blocks(np.asarray(raw[c_beg:min(run_end, n_raw)]))
        except Exception as e:                                                    # noqa: BLE001
            raise bamfmt.BAMError("index of %s does not match its BGZF members" % path) from e
        pos = pos + np.uint64(c_beg)
        cpos = pos[:isz.shape[0]].astype(np.int64)
        a, b = 0, int(isz.shape[0]) - 1
        if b < 0 or cpos[a] != c_beg or cpos[b] != c_end:
            raise bamfmt.BAMError("index of %s does not match its BGZF members" % path)
        last = b if (end & 0xFFFF) else b - 1                                    # (a range ending at offset 0 of a member stops before it)
        if last < a:
            continue
        stop = int(isz[a:b].sum()) + (end & 0xFFFF)                              # end of the contig's last record inside the run
        d_buf = d_off = None
        if on_device and int(isz[a:last + 1].sum()) <= GPU_INFLATE_MAX:
            # the run's members inflated and walked on the device (N1, as in bam_join_input)
            p0 = int(pos[a])
            try:
                d_run = engine.bgzf_inflate(raw[p0:int(pos[last + 1])], pos[a:last + 2] - np.uint64(p0), isz[a:last + 1], check_crc=BGZF_CRC)
            except GciError as e:
                if e.rec >= 0:
                    e.rec += a
                raise
            d_off, used, ok = engine.bam_record_offsets(d_run[:stop], beg & 0xFFFF, len(hdr.references))
            if ok and used == stop:
                d_buf = d_run[:stop]
            del d_run
        if d_buf is None:
            buf = hostio.bgzf_inflate(np.asarray(raw[int(pos[a]):int(pos[last + 1])]), threads=nthreads, check_crc=BGZF_CRC)
            offs, _ = hostio.bam_chunk_offsets(buf[:stop], beg & 0xFFFF)
            d_buf, d_off = engine.to_device(buf[:stop]), engine.to_device(offs)
        if d_off.shape[0] == 0:
            continue
        try:
            ji = _filter_stream(engine, d_buf, d_off, True, ref_sel, filt, rec_idx_base=n_done)
        except GciError as e:
            if e.rec >= 0:
                e.rec += n_done
            e.contig = tindex[name]                 # (the ranks agree on the error of the earliest contig: _agree_on_error)
            raise
        parts.append(_keep_part(engine, ji))
        n_done += int(d_off.shape[0])
        del d_buf, d_off, ji
    return _concat_parts(engine, parts)


def _replicate(engine: Engine, ji: JoinInput) -> JoinInput:
    """One file's compact records + names of every rank, on every rank (all-gather; RCCL over xGMI).  Rank r's records
    keep their order and sit behind those of ranks < r: a contig's records come from one rank, so "the last record of a
    name wins" (contig order first, then file order; GCI.py:269) is decided as in a single process."""
    from . import shard
    n = int(ji.recs.shape[0])
    if n:
        blob, off = engine.pack_names(ji)
    else:
        blob, off = engine.T.zeros(0, engine.T.uint8, engine.device), engine.T.zeros(1, engine.T.int64, engine.device)
    ex = shard.RecordExchange(n, int(blob.shape[0]), engine.device, via_host=SHARD.backend != "nccl")
    ex.send_recs[:n] = ji.recs
    ex.send_recs[:n, 29] &= 3                          # the names travel packed: no longer GCI_REC_NAME16
    ex.send_names[:blob.shape[0]] = blob
    ex.send_off[:n + 1] = off
    g = ex.gather()
    return JoinInput(g.recs, g.names, g.name_index, 0)


def _own_names_only(engine: Engine, ji: JoinInput, world: int, rank: int) -> JoinInput:
    """Records every rank holds alike (PAF files are filtered whole on every rank): keep those whose NAME this rank owns --
    (hash >> 33) % world, the rule of gci_route_records -- so that they meet the routed BAM records of the same names."""
    if int(ji.recs.shape[0]) == 0:
        return ji
    import torch                                       # (a contig-sharded run: the torch provider)
    h = ji.recs[:, 0:8].contiguous().view(torch.int64).reshape(-1)
    dest = ((h >> 33) & 0x7FFFFFFF) % world
    recs = ji.recs.clone()
    recs[dest != rank, 29] = 0                       # gci_rec.flags
    return JoinInput(recs, ji.name_base, ji.name_off, ji.name_delta)


def _agree_on_error(err: Optional[GciError], contig: int = 1 << 30) -> None:
    """A contig-sharded run fails as ONE run: every rank learns whether any rank's record filter / join raised, and all of them
    raise the error of the earliest contig (the one the reference's loop over contigs would have met first) -- a rank that
    went on alone would wait in the next collective for ever."""
    code = 0 if err is None else -int(err.status)
    key = (min(int(contig), (1 << 30)) << 8 | code) if code else (1 << 40)
    best = -SHARD.all_reduce_max([-key])[0]
    if best >= (1 << 40):
        return
    _reraise_like_reference(err if (err is not None and key == best) else GciError(-(best & 0xFF), "record filter / join failed on another rank"))


def _filter_sharded(paf_files, bam_files, prefix, map_qual, mq_cutoff, iden_percent, clip_percent, ovlp_percent, flank_len,
                    directory, log_reads_type, chrs_list, threads, engine: Engine, write, issue_hint):
    """filter() of a contig-sharded run.  Contigs are dealt to the ranks (longest first onto the least loaded); depth
    build, gap mask, two-type max, issue scan and depth text are contig-local.  The read-name join is not -- a read
    aligned to contigs of two ranks must be dropped, a name repeated across contigs keeps its last record
    (GCI.py:269, 296-297) -- but it is independent per NAME: every rank filters the records of ITS contigs (K1), routes them
    by name hash to the rank that owns the name (two all-to-alls per file), joins the names it owns and routes the surviving
    intervals to the owners of their contigs (shard.ShardedJoin; round 2 replicated every record on every rank).
    PAF files are sharded by byte range (shard.paf_by_byte_range: every rank tokenises the lines of its range, the hits travel to
    the rank that owns their query name and are scored there); GCI_PAF_SHARDING=whole keeps round 2's way (every rank filters the
    whole files and keeps the names it owns), which is also what runs when a line raises."""
    from . import shard
    first = bamfmt.read_header(bam_files[0])
    pairs = [(r, l) for r, l in zip(first.references, first.lengths) if (len(chrs_list) == 0 or r in chrs_list)]
    targets_length = {r: l for r, l in pairs}
    targets = list(targets_length.keys())
    mine = SHARD.assign([targets_length[t] for t in targets])
    if SHARD.world > len(targets):                  # (every rank sees this: rank 0 says it)
        sys.exit(f"ERROR!!! {SHARD.world} GPUs for {len(targets)} contig(s): use at most one GPU per contig")
    local_tl = {targets[c]: targets_length[targets[c]] for c in mine}
    engine.set_layout(list(local_tl.values()))
    filt = (map_qual, mq_cutoff, clip_percent, iden_percent)
    paf_inputs: List[JoinInput] = []
    local: List[JoinInput] = []
    err, err_contig = None, 1 << 30
    try:
        if len(paf_files) != 0:
            # every rank tokenises 1 / world of the files' bytes; the hits are scored on the rank that owns their query name
            # (None: a line the reference raises on -- then every rank reads the whole files, and the exception comes out exact)
            by_range = None
            if PAF_SHARDING == "range":
                by_range = shard.paf_by_byte_range(engine, paf_files, targets, map_qual, mq_cutoff, iden_percent, SHARD.world, SHARD.rank,
                                                   SHARD.all_reduce_max, engine.device, via_host=SHARD.backend != "nccl")
            try:
                paf_inputs = by_range if by_range is not None else [
                    _own_names_only(engine, ji, SHARD.world, SHARD.rank)
                    for ji in engine.paf_filter(paf_files, targets, map_qual, mq_cutoff, iden_percent)]
            except GciError as e:
                e.paf_replay = (paf_files, targets)
                raise
        for path in bam_files:
            local.append(bam_records_of_contigs(engine, path, targets, list(local_tl), filt, threads))
    except GciError as e:
        err, err_contig = e, getattr(e, "contig", 0)
    _agree_on_error(err, err_contig)
    # routed name slots: as long as the longest query name of the run (a multiple of 16, the same on every rank)
    import torch                                       # (a contig-sharded run: the torch provider)
    longest = max([int(ji.recs[:, 30:32].contiguous().view(torch.int16).max().item()) if int(ji.recs.shape[0]) else 0 for ji in local] + [1])
    slot = (SHARD.all_reduce_max([longest])[0] + 15) // 16 * 16
    sj = shard.ShardedJoin(engine, [int(ji.recs.shape[0]) for ji in local], SHARD.owner, engine.device,
                           via_host=SHARD.backend != "nccl", name_slot=max(16, slot),
                           extra_records=sum(int(ji.recs.shape[0]) for ji in paf_inputs))
    classic = False
    try:
        while True:
            inputs = paf_inputs + sj.exchange_files(local)          # (ONE all-to-all for the records and names of every file)
            ivl, n_slots = sj.join(inputs, ovlp_percent)
            err = None
            try:
                def decode(w, what):
                    rec = ctypes.c_uint32(0)
                    st = engine.lib.gci_decode_status(w, ctypes.byref(rec))
                    if st != 0:
                        e = GciError(st, "%s: %s" % (what, engine.lib.gci_strerror(st).decode()), rec=int(rec.value))
                        e.what = what
                        raise e
                sj.check(decode)
            except GciError as e:
                err = e
            cap = err is not None and err.status == _lib.GCI_E_CAPACITY
            # GCI_E_CAPACITY has two senders.  The partitioned JOIN refuses what its 32-byte entries / LDS tables cannot hold
            # (adversarial names): every rank then joins on the classic table, once -- larger buckets would change nothing.
            # A ROUTE / SEAL step reports a bucket that overflowed (names hash unevenly): larger buckets, again.
            from_join = cap and getattr(err, "what", "") == "gci_name_join"
            refuse, grow = SHARD.all_reduce_max([1 if from_join else 0, 1 if (cap and not from_join) else 0])
            if grow:
                sj.grow()
                continue
            if refuse and not classic:
                classic = True
                engine._chk(engine.lib.gci_join_mode(engine.ctx, 1), "gci_join_mode")
                continue
            _agree_on_error(err)
            break
    finally:
        if classic:
            engine._chk(engine.lib.gci_join_mode(engine.ctx, engine.join_mode), "gci_join_mode")
    track = engine.new_track()
    fused = engine.depth_build_fused(ivl, None, flank_len, track, want_text=False, want_sums=True, issue=issue_hint, counted=False)
    depths = DepthTracks(engine, local_tl, track)
    depths.all_targets = targets
    depths._fresh_sums = fused["sums"]
    if issue_hint is not None:
        depths._fresh_runs = (tuple(float(x) for x in issue_hint[:2]) + (int(issue_hint[2]),), fused["runs"])
    print(f"Filtering {log_reads_type} alignment files done!!!")
    if write:
        print(f'Writing depths into "{directory}/{prefix}.depth.gz" ...')
        _write_depth_members(directory, prefix, depths)
        print("Writing depths done!!!\n\n")
    return depths, targets_length


def _paf_line_exception(paf_files: Sequence[str], targets: Sequence[str]) -> Optional[BaseException]:
    """ERROR PATH ONLY.  The device PAF filter has reported a line the reference raises on (GCI_E_MALFORMED / GCI_E_ZERO_DIV with a
    line number); the reference would die with Python's own exception for that line -- IndexError for a missing column,
    ValueError with int()'s message for a bad number, ZeroDivisionError for a zero alignment length -- and which one depends on
    the order GCI.py:217-229 touches the columns in.  This replays exactly those statements over the files in order and returns
    the first exception they raise (None if no line raises).  It computes nothing that is used: no dict, no score."""
    tset = set(targets)
    for path in paf_files:
        with open(path, "r") as f:
            for line in f:
                try:
                    paf = line.strip().split("\t")
                    if paf[5] in tset:
                        int(paf[1]), int(paf[2]), int(paf[3]), int(paf[7]), int(paf[8])
                        num_match_res, len_aln = int(paf[9]), int(paf[10])
                        int(paf[11])
                        num_match_res / len_aln
                except (IndexError, ValueError, ZeroDivisionError) as exc:
                    return exc
    return None


def _reraise_like_reference(e: GciError):
    from . import _lib
    replay = getattr(e, "paf_replay", None)
    if replay is not None and e.status in (_lib.GCI_E_MALFORMED, _lib.GCI_E_ZERO_DIV) and getattr(e, "rec", 0) > 0:
        exc = _paf_line_exception(*replay)
        if exc is not None:
            raise exc from e
    if e.status == _lib.GCI_E_NO_NM:
        raise KeyError("tag 'NM' not present") from e
    if e.status == _lib.GCI_E_ZERO_DIV:
        raise ZeroDivisionError("division by zero") from e
    raise e


# Contig-sharded runs: "range" (default) = PAF files by byte range, "whole" = every rank the whole files.
PAF_SHARDING = os.environ.get("GCI_PAF_SHARDING", "range")

# `{prefix}.depth.gz` is written by the device: gzip members straight from the track or from the run lists the build kept
# (gci_depth_deflate_*: no text buffer, a few MB cross PCIe).  (Rounds 1 - 5 kept the older way -- text rendered on the device, gzip on
# host threads -- behind GCI_DEPTH_GZ=host; the text seam itself, gci_depth_text_*, stays in the library and in bench.py's step.)


def write_depth(directory=".", prefix="GCI", depths: DepthTracks = None, threads=1) -> None:
    """`{directory}/{prefix}.depth.gz`: '>contig' line then one decimal per line (GCI.py:99-143), as a multi-member
    gzip (any gzip whose payload equals the reference's text is a valid .depth.gz)."""
    depths._bind()
    _write_depth_members(directory, prefix, depths)


def _write_depth_members(directory, prefix, depths: DepthTracks, from_build: bool = False) -> None:
    """'>contig' member (host), then the contig's lines as the members the device wrote.  Contig-sharded run: every rank
    deflates the contigs it owns, rank 0 gathers the members and writes them in header order.  from_build: only filter() says
    so, for the track its own build has just written (Engine.depth_deflate)."""
    from . import hostio
    blobs = depths.engine.depth_deflate(depths.track, from_build=from_build)               # views of the engine's pinned staging buffer
    items = list(zip(depths.targets, depths.lengths, blobs))
    if _sharded():
        # to rank 0 only (the members of a genome are GBs), as bytes: a sized gather of one uint8 tensor per rank -- which contigs a
        # rank owns every rank knows (SHARD.owner), so names and lengths need not travel
        order = {t: i for i, t in enumerate(depths.all_targets)}
        parts = SHARD.gather_bytes_to_root([np.frombuffer(bytes(b), dtype=np.uint8) if not isinstance(b, np.ndarray) else b for _, _, b in items])
        if not SHARD.root:
            return
        all_t = depths.all_targets
        items = []
        for r, blobs_r in enumerate(parts):
            owned = [c for c, o in enumerate(SHARD.owner) if o == r]
            assert len(owned) == len(blobs_r), (r, len(owned), len(blobs_r))
            # (a contig of length 0 has no members: its empty blob stands for "L == 0" below)
            items += [(all_t[c], int(b.shape[0]), b.tobytes()) for c, b in zip(owned, blobs_r)]
        items.sort(key=lambda x: order[x[0]])
    path = f"{directory}/{prefix}.depth.gz"
    if os.path.exists(path):
        os.remove(path)
    layout = {}
    with open(path, "wb") as f:
        for t, L, blob in items:
            if L == 0:
                continue              # the reference's chunk loop never runs for an empty contig: not even the '>' line
            at = f.tell()
            f.write(hostio.gzip_members((">%s\n" % t).encode(), threads=1))
            f.write(blob)
            layout[t] = (at, f.tell())
    phases.note("depth_gz_layout:" + path, layout)    # (a harness that checks single contigs of a genome-size file reads this)


def merge_two_type_depth(hifi_depths: DepthTracks = None, nano_depths: DepthTracks = None, prefix="GCI_two_type",
                         directory=".", force=False, threads=1, write=True, issue_hint=None) -> DepthTracks:
    """issue_hint = (leftmost, rightmost, flank_len) of the collapse_depth_range() calls that follow (as for filter()): the merge
    is then ONE pass over the two tracks that also applies their pending N-run masks (merge_gaps_depths(lazy=True)) and finds the
    issue runs of the HiFi, the ONT and the merged track (gci_two_type_tail) -- 12 bytes per base instead of 24."""
    print("Merging HiFi and ONT depth file ...")
    if write:
        refuse_overwrite(f"{directory}/{prefix}.depth.gz", force)
    if nano_depths.targets != hifi_depths.targets or nano_depths.lengths != hifi_depths.lengths:
        # the reference indexes nano by the HiFi dict's keys; GCI() has already checked both headers agree
        raise KeyError("HiFi and ONT depth tracks cover different contigs")
    engine = hifi_depths.engine
    pend = [d._pending_gaps for d in (hifi_depths, nano_depths)]
    same_pending = (pend[0] is None and pend[1] is None) or (pend[0] is not None and pend[1] is not None and
                                                             pend[0].tobytes() == pend[1].tobytes())
    if issue_hint is not None and nano_depths.engine is engine and same_pending:
        hifi_depths._bind_layout()
        lo, hi, fl = float(issue_hint[0]), float(issue_hint[1]), int(issue_hint[2])
        d_sums = engine.T.zeros((3, max(len(hifi_depths.targets), 1)), engine.T.int64, engine.device)
        two, runs = engine.two_type_tail(hifi_depths.track, nano_depths.track, pend[0], lo, hi, fl, sums=d_sums)
        merged = DepthTracks(engine, hifi_depths.targets_length, two)
        h_sums = d_sums.cpu().numpy()[:, :len(hifi_depths.targets)]
        for k, (d, r) in enumerate(zip((hifi_depths, nano_depths, merged), runs)):
            if pend[0] is not None:
                d._pending_gaps, d._masked_with = None, pend[0].tobytes()
            d._fresh_sums = h_sums[k].copy()                     # (the numerator of the mean depth of a `-p` run: no pass of its own)
            d._fresh_runs = ((lo, hi, fl), r)
    else:
        hifi_depths._bind()
        nano_depths._bind()
        merged = DepthTracks(engine, hifi_depths.targets_length, engine.max2(hifi_depths.track, nano_depths.track))
    merged.all_targets = hifi_depths.all_targets
    if write:
        write_depth(directory, prefix, merged, threads)
    print("Merging HiFi and ONT depth file done!!!\n\n")
    return merged


# ==============================================================================================
# issue scan
# ==============================================================================================

def _issues_from_runs(runs: np.ndarray, n_slice: int, chr_len: int, flank_len: int, start_pos: int
                      ) -> List[Tuple[int, int]]:
    """Apply the reference's closing rules (GCI.py:380-388) to raw maximal runs given relative to
    the scanned slice: a run that ends at an out-of-range base at relative index i is reported
    only if i > flank_len; a run reaching the last scanned base is reported iff that base has
    relative index chr_len - 2 * flank_len - 1."""
    out: List[Tuple[int, int]] = []
    for rs, re_ in runs.tolist():
        if re_ == n_slice:
            if n_slice - 1 != chr_len - 2 * flank_len - 1:
                continue
        elif not (re_ > flank_len):
            continue
        out.append((rs + flank_len + start_pos, re_ + flank_len + start_pos))
    return out


def collapse_depth_range(depths: DepthTracks = None, leftmost=-1, rightmost=0, flank_len=15, start_pos=0
                         ) -> Dict[str, List[Tuple[int, int]]]:
    depths._bind()
    key = (float(leftmost), float(rightmost), int(flank_len))
    if depths._fresh_runs is not None and depths._fresh_runs[0] == key:
        runs = depths._fresh_runs[1]
    else:
        runs = depths.engine.issue_scan(depths.track, leftmost, rightmost, flank_len)
    out = {}
    for c, t in enumerate(depths.targets):
        L = depths.lengths[c]
        a, b = _slice_bound(flank_len, L), _slice_bound(L - flank_len, L)
        out[t] = _issues_from_runs(runs[c], max(0, b - a), L, flank_len, start_pos)
    return out


def collapse_regions(depths: DepthTracks, regions: Sequence[Tuple[str, int, int]], leftmost, rightmost
                     ) -> List[List[Tuple[int, int]]]:
    depths._bind()
    wins, meta = [], []
    for target, start, end in regions:
        c = depths.targets.index(target)
        L = depths.lengths[c]
        a, b = _slice_bound(start, L), _slice_bound(end, L)
        b = max(a, b)
        o = depths.engine.offsets[c]
        wins.append((o + a, o + b))
        meta.append((b - a, start))
    runs = depths.engine.issue_scan_windows(depths.track, wins, leftmost, rightmost)
    return [_issues_from_runs(r, n, n, 0, sp) for r, (n, sp) in zip(runs, meta)]


def merge_depth(depths: DepthTracks = None, prefix="GCI", threshold=0, flank_len=15, directory=".", force=False,
                log_reads_type=""):
    print(f"Getting {log_reads_type} issues bed file detected by GCI ...")
    path = f"{directory}/{prefix}.{threshold}.depth.bed"
    refuse_overwrite(path, force)
    merged = collapse_depth_range(depths, -1, threshold, flank_len, 0)
    if _sharded():                      # every rank scanned its contigs; the lists (<= 10^4 items) travel as objects
        union = {}
        for part in SHARD.gather_objects(merged):
            union.update(part)
        merged = {t: union[t] for t in depths.all_targets}          # header order
    if _is_root():
        with open(path, "w") as f:
            for target, segments in merged.items():
                for s, e in segments:
                    f.write(f"{target}\t{s}\t{e}\n")
    print(f"Getting {log_reads_type} issues bed file done!!!\n\n")
    return merged


# ==============================================================================================
# N3: the `-p` numeric front-end
# ==============================================================================================

def sliding_window_average_depth_many(depths: DepthTracks, items: Sequence[Tuple[str, int, Optional[int]]], window_size=50000,
                                      max_depth=None):
    """sliding_window_average_depth(depths[target][start:end], window_size, max_depth, start, target) of the reference
    (GCI.py:660-705) for every (target, start, end) of `items` over a track in HBM, in TWO device calls for all of them:
    -> [(positions in Mb: list of float, values: float64 array, the reference's warning line or None)].

    The reference walks the bases one by one and restarts its window at every zero-depth base.  Which bases emit a value follows
    from the zero runs and the window size alone, so the device finds the zero runs of all items (ONE gci_issue_scan_windows)
    and sums all their windows (ONE gci_range_sums); the few thousand windows per contig are assembled here."""
    depths._bind()
    geo = []
    for target, start, end in items:
        c = depths.targets.index(target)
        L = depths.lengths[c]
        a = _slice_bound(start, L)
        b = max(a, _slice_bound(L if end is None else end, L))
        n = b - a
        w, warn = window_size, None
        if n < window_size:
            warn = (f'Warning!!! The length ({n}) of plotting region ({target}:{start}-{start + n}) is less than the window size '
                    f'({window_size}), and therefore the window size will be 1 bp')
            w = 1
        geo.append((depths.engine.offsets[c] + a, n, w, warn, start))
    live = [k for k, g in enumerate(geo) if g[1] > 0]
    zeros = depths.engine.issue_scan_windows(depths.track, [(geo[k][0], geo[k][0] + geo[k][1]) for k in live], -1, 0) if live else []
    plans, ranges = {}, []
    for k, z in zip(live, zeros):
        o, n, w, _, _ = geo[k]
        zero = z.reshape(-1, 2)                                  # runs of depth 0, relative to the item's first base
        edges = np.concatenate([[0], zero.reshape(-1), [n]]).astype(np.int64)       # the non-zero runs in between
        p, q = edges[0::2], edges[1::2]
        keep = q > p
        p, q = p[keep], q[keep]
        full = (q - p) // w                                      # whole windows per run, then one partial flush
        rem = (q - p) - full * w
        # window k of run r: [p + k w, p + (k + 1) w); emitted at its last base
        run_of = np.repeat(np.arange(p.shape[0]), full)
        kk = np.arange(int(full.sum()), dtype=np.int64) - np.repeat(np.cumsum(full) - full, full)
        wb = p[run_of] + kk * w
        has_rem = rem > 0
        begins = np.concatenate([wb, (q - rem)[has_rem]])
        ends = np.concatenate([wb + w, q[has_rem]])
        plans[k] = (zero, begins, ends, len(ranges), begins.shape[0])
        ranges.append(np.stack([begins + o, ends + o], axis=1))
    sums = depths.engine.range_sums(depths.track, np.concatenate(ranges)) if ranges else np.zeros(0, dtype=np.int64)
    at = np.concatenate([[0], np.cumsum([r.shape[0] for r in ranges])]).astype(np.int64) if ranges else np.zeros(1, np.int64)
    out = []
    for k, (o, n, w, warn, start) in enumerate(geo):
        if n == 0:
            out.append(([], np.array([], dtype=np.float64), warn))
            continue
        zero, begins, ends, slot, cnt = plans[k]
        s = sums[int(at[slot]):int(at[slot]) + cnt]
        means = s.astype(np.float64) / (ends - begins).astype(np.float64)   # int / int, both < 2^53: as Python's true division
        means = np.where(means > max_depth, np.float64(max_depth), means)
        zlen = zero[:, 1] - zero[:, 0]
        zidx = np.repeat(zero[:, 0] - (np.cumsum(zlen) - zlen), zlen) + np.arange(int(zlen.sum()), dtype=np.int64)
        idx = np.concatenate([ends - 1, zidx])
        val = np.concatenate([means, np.zeros(zidx.shape[0], dtype=np.float64)])
        order = np.argsort(idx, kind="stable")
        idx, val = idx[order], val[order]
        out.append((((idx + start) / 1e6).tolist(), val, warn))
    return out


def sliding_window_average_depth(depths: DepthTracks, target: str, window_size=50000, max_depth=None, start=0, end=None):
    """One (target, start, end): -> (positions in Mb: list of float, values: float64 array); the reference's warning for a region
    shorter than the window goes to stderr (GCI.py:675-677)."""
    pos, val, warn = sliding_window_average_depth_many(depths, [(target, start, end)], window_size, max_depth)[0]
    if warn:
        print(warn, file=sys.stderr)
    return pos, val


def pre_plot_base(depths_list: Sequence[DepthTracks], max_depths: Sequence[float], window_size=50000, start=0,
                  region: Optional[Tuple[str, int, int]] = None):
    """pre_plot_base of the reference (GCI.py:708-739).  region = (target, start, end) is the reference's
    `{target: depths[start:end]}` input of the regions loop (GCI.py:887-892); None = every contig, whole.  Per track all its
    contigs go through the device together (two calls); the warnings come out in the reference's order (target by target)."""
    targets = [region[0]] if region else depths_list[0].targets
    items = [(t, region[1] if region else start, region[2] if region else None) for t in targets]
    per_track = [sliding_window_average_depth_many(tr, items, window_size, max_depths[i]) for i, tr in enumerate(depths_list)]
    averaged = [{} for _ in depths_list]
    maxima = [[] for _ in depths_list]
    for k, target in enumerate(targets):
        for i in range(len(depths_list)):
            pos, val, warn = per_track[i][k]
            if warn:
                print(warn, file=sys.stderr)
            averaged[i][target] = (pos, val)
            maxima[i].append(max(val))
    if _sharded() and region is None:            # the axis limits span every contig: the few maxima of all ranks
        maxima = [[x for part in parts for x in part] for parts in zip(*SHARD.gather_objects(maxima))]
    y_max = max(maxima[0]) + 10
    y_min = 0 if len(depths_list) == 1 else max(maxima[1]) + 10
    return averaged, y_min / (y_max + y_min), y_min, y_max


# ==============================================================================================
# score files
# ==============================================================================================

complement_merged_depth = score.complement_merged_depth
compute_n50 = score.compute_n50
merge_merged_depth_bed = score.merge_merged_depth_bed


def compute_index(targets_length={}, prefix="GCI", directory=".", force=False, merged_depths_bed_list=[],
                  type_list=[], flank_len=15, dist_percent=0.005, regions_bed={}, depths_list=[], threshold=0,
                  chrs_list=[]):
    gci_path = f"{directory}/{prefix}.gci"
    refuse_overwrite(gci_path, force)
    if _is_root():
        with open(gci_path, "w"):
            pass
    reg_path = f"{directory}/{prefix}.regions.gci"
    if len(regions_bed) > 0:
        refuse_overwrite(reg_path, force)
    # progress lines are printed where the work happens (same lines, same order as GCI.py:553-607)
    print("Computing Theoretical minimum N50 and contigs number ...")
    rows, exp_n50, exp_ctg = score.expected_table(targets_length, chrs_list)
    print("Computing Theoretical minimum N50 and contigs number done!!!")
    for t, bed in zip(type_list, merged_depths_bed_list):
        print(f"Computing Curated N50 and contigs number for {t} ...")
        obs_n50, obs_ctg = score.curated_table(bed, targets_length, flank_len, dist_percent, rows[-1])
        print(f"Computing Curated N50 and contigs number for {t} done!!!")
        print(f"Writing results to {gci_path} ...")
        if _is_root():
            with open(gci_path, "a") as f:
                f.write(score.section_text(t, rows, exp_n50, exp_ctg, obs_n50, obs_ctg))
        print(f"Writing results to {gci_path} done!!!\n\n")
    if len(regions_bed) > 0:
        print("Computing GCI scores for regions ...")
        flat = [(t, s, e) for t, segs in regions_bed.items() for s, e in segs]
        mine = [r for r in flat if r[0] in depths_list[0]]                      # the regions on contigs this rank holds
        per_track = [collapse_regions(d, mine, -1, threshold) for d in depths_list]
        lookup = {(i, r): per_track[i][k] for i in range(len(depths_list)) for k, r in enumerate(mine)}
        if _sharded():
            merged_lookup = {}
            for part in SHARD.gather_objects(lookup):
                merged_lookup.update(part)
            lookup = merged_lookup
        if _is_root():
            text = score.regions_text(regions_bed, type_list, len(depths_list),
                                      lambda i, t, s, e: lookup[(i, (t, s, e))], dist_percent,
                                      warn=lambda m: print(m, file=sys.stderr))
            with open(reg_path, "w") as f:
                f.write(text)
        print("Computing GCI scores for regions done!!!\n\n")
