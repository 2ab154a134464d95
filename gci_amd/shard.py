"""Multi-GPU sharding of the path: one process per GPU, contigs sharded, ONE real
exchange step.

* Depth build, gap mask, two-type max, issue scan, text and per-contig scoring are independent
  per contig -> each rank owns a set of contigs (longest-processing-time packing).
* The read-name join is NOT contig-local (a read aligned to different contigs in two files must
  be dropped, GCI.py:296-297; a repeated name keeps only its last record, GCI.py:269) but it is
  independent per NAME: every rank filters the records of its contigs (K1), routes them by name hash
  to the rank that owns the name (one RCCL all-to-all of 32-byte records and one of names per file),
  joins the names it owns, and routes the surviving 16-byte intervals to the owners of their contigs
  (`ShardedJoin`).  `RecordExchange` (all-gather, every rank joins everything) is the round-2 way, kept
  for names too long for a routed slot.
* Genome-wide totals (sum of depth, bases) are one integer all-reduce: exact in any order.

Everything here is device-agnostic torch (`nccl` == RCCL on the GPUs, `gloo` in the CPU tests).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist


class Context:
    """One contig-sharded run of the command line: this process is rank `rank` of `world`, one process per GPU
    (python -m torch.distributed.run --nproc-per-node N GCI.py ..., or GCI.py --gpus N which re-launches itself that way).
    backend "nccl" = RCCL over xGMI; "gloo" (GCI_DIST_BACKEND=gloo) stages collectives through host memory, which is how
    the two-rank tests run on a single GPU."""

    def __init__(self):
        import os
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.backend = os.environ.get("GCI_DIST_BACKEND", "nccl")
        # the GPU of this rank: LOCAL_RANK, or GCI_DIST_DEVICE (all ranks on one device: tests)
        self.device_index = int(os.environ.get("GCI_DIST_DEVICE", str(self.local_rank)))
        # one rank made to take the sharded path (every collective runs, over a world of one: the RCCL self-test on one GPU)
        self.forced = os.environ.get("GCI_FORCE_SHARDED", "0") == "1"
        self.owner: List[int] = []

    @property
    def root(self) -> bool:
        return self.rank == 0

    def init(self) -> None:
        import os
        import sys
        if dist.is_initialized():
            return
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        if os.environ.get("NCCL_DEBUG", "").upper() in ("VERSION", "INFO", "WARN"):
            os.environ.pop("NCCL_DEBUG")                  # RCCL logs to stdout: keep the reference's transcript clean
        if self.backend == "gloo":
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")     # single node: the container's hostname may not resolve
        torch.cuda.set_device(self.device_index)
        kw = dict(device_id=torch.device("cuda", self.device_index)) if self.backend == "nccl" else {}
        if self.backend == "gloo":
            # gloo reports its connections on file descriptor 1 when they are first made: the transcript of the run stays
            # the reference's, so that chatter is sent to stderr (and the connections are made here, by a barrier)
            sys.stdout.flush()
            saved = os.dup(1)
            os.dup2(2, 1)
            try:
                dist.init_process_group(self.backend, rank=self.rank, world_size=self.world, **kw)
                dist.barrier()
            finally:
                os.dup2(saved, 1)
                os.close(saved)
        else:
            dist.init_process_group(self.backend, rank=self.rank, world_size=self.world, **kw)

    def assign(self, lengths: Sequence[int]) -> List[int]:
        """Longest-processing-time packing of the selected contigs; -> indices of the contigs this rank owns."""
        self.owner = lpt_assign(lengths, self.world)
        return [c for c, o in enumerate(self.owner) if o == self.rank]

    def gather_objects(self, obj) -> list:
        out: List[Optional[object]] = [None] * self.world
        dist.all_gather_object(out, obj)
        return out

    def gather_to_root(self, obj) -> Optional[list]:
        """obj of every rank on rank 0 (in rank order); None elsewhere.  For what only rank 0 writes (the .depth.gz members of
        a genome are GBs: every rank holding every rank's would be world x that)."""
        out: Optional[List[Optional[object]]] = [None] * self.world if self.root else None
        dist.gather_object(obj, out, dst=0)
        return out

    def gather_bytes_to_root(self, blobs: Sequence) -> Optional[List[List[np.ndarray]]]:
        """Per rank a list of byte strings (the .depth.gz members of the contigs it owns) -> on rank 0, in rank order, every rank's
        list as uint8 arrays; None elsewhere.  Nothing is pickled: the counts and the sizes travel as int64 tensors (all-gathers of
        fixed shape), the bytes as ONE gather of a uint8 tensor padded to the largest rank's total."""
        dev = torch.device("cuda", self.device_index) if self.backend == "nccl" else torch.device("cpu")
        arrs = [np.frombuffer(b, dtype=np.uint8) if not isinstance(b, np.ndarray) else b.reshape(-1).view(np.uint8) for b in blobs]
        count = torch.tensor([len(arrs)], dtype=torch.int64, device=dev)
        counts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(self.world)]
        dist.all_gather(counts, count)
        counts = [int(c.item()) for c in counts]
        width = max(counts + [1])
        mine = torch.zeros(width, dtype=torch.int64, device=dev)
        if arrs:
            mine[:len(arrs)] = torch.tensor([int(a.shape[0]) for a in arrs], dtype=torch.int64)
        sizes = [torch.zeros(width, dtype=torch.int64, device=dev) for _ in range(self.world)]
        dist.all_gather(sizes, mine)
        sizes = [s.cpu().numpy()[:c] for s, c in zip(sizes, counts)]
        total = max([int(s.sum()) for s in sizes] + [1])
        buf = torch.zeros(total, dtype=torch.uint8, device=dev)
        at = 0
        for a in arrs:
            buf[at:at + a.shape[0]] = torch.from_numpy(np.ascontiguousarray(a)).to(dev)
            at += int(a.shape[0])
        got = [torch.zeros(total, dtype=torch.uint8, device=dev) for _ in range(self.world)] if self.root else None
        dist.gather(buf, got, dst=0)
        if not self.root:
            return None
        out = []
        for g, sz in zip(got, sizes):
            h = g.cpu().numpy()
            cuts = np.concatenate([[0], np.cumsum(sz)]).astype(np.int64)
            out.append([h[cuts[k]:cuts[k + 1]] for k in range(len(sz))])
        return out

    def all_reduce_max(self, values: Sequence[int]) -> List[int]:
        dev = torch.device("cuda", self.device_index) if self.backend == "nccl" else torch.device("cpu")
        t = torch.tensor([int(v) for v in values], dtype=torch.int64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return [int(x) for x in t.cpu().tolist()]

    def all_reduce_sum(self, values: Sequence[int]) -> List[int]:
        dev = torch.device("cuda", self.device_index) if self.backend == "nccl" else torch.device("cpu")
        t = torch.tensor([int(v) for v in values], dtype=torch.int64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return [int(x) for x in t.cpu().tolist()]


def lpt_assign(lengths: Sequence[int], world: int) -> List[int]:
    """Owner rank of each contig: longest first onto the least loaded rank (ties -> lowest rank)."""
    load = [0] * world
    owner = [0] * len(lengths)
    for c in sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i)):
        r = min(range(world), key=lambda k: (load[k], k))
        owner[c] = r
        load[r] += int(lengths[c])
    return owner


def contig_map_for(owner: Sequence[int], rank: int) -> Tuple[np.ndarray, List[int]]:
    """-> (int32 map global contig -> local track index or -1, list of owned global contigs in order)."""
    mine = [c for c, o in enumerate(owner) if o == rank]
    m = np.full(len(owner), -1, dtype=np.int32)
    for k, c in enumerate(mine):
        m[c] = k
    return m, mine


def record_slices(n_rec: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous, near-equal [lo, hi) slices of a file's records, one per rank."""
    step = -(-n_rec // world) if n_rec else 0
    return [(min(n_rec, r * step), min(n_rec, (r + 1) * step)) for r in range(world)]


@dataclass
class Gathered:
    recs: torch.Tensor          # uint8 [world * max_n, 32]; rows past a rank's count have flags == 0
    names: torch.Tensor         # uint8 [world * name_cap]
    name_index: torch.Tensor    # int64 [world * max_n]: byte offset of record (r * max_n + i)'s name in `names`
    max_n: int


class RecordExchange:
    """All-gather of one file's compact records + packed names with fixed (padded) shapes, so the
    collective sizes are identical on every rank and every step.

    Rank r's record i travels as global index r * max_n + i; K1 is called with
    rec_idx_base = r * max_n so `gci_rec.rec_idx` already is that global index."""

    def __init__(self, n_local: int, name_bytes_local: int, device: torch.device, group=None, via_host: bool = False):
        """via_host: the collectives run on host copies (gloo backend with device tensors)."""
        self.group = group
        self.via_host = via_host
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.device = device
        sizes = torch.tensor([n_local, name_bytes_local], dtype=torch.int64, device="cpu" if via_host else device)
        allsz = [torch.zeros_like(sizes) for _ in range(self.world)]
        dist.all_gather(allsz, sizes, group=group)
        self.counts = [int(s[0].item()) for s in allsz]
        self.max_n = max(1, max(self.counts))
        self.name_cap = max(16, max(int(s[1].item()) for s in allsz))
        self.rec_idx_base = self.rank * self.max_n
        self.send_recs = torch.zeros((self.max_n, 32), dtype=torch.uint8, device=device)
        self.send_names = torch.zeros(self.name_cap, dtype=torch.uint8, device=device)
        self.send_off = torch.zeros(self.max_n + 1, dtype=torch.int64, device=device)
        self.g_recs = torch.zeros((self.world * self.max_n, 32), dtype=torch.uint8, device=device)
        self.g_names = torch.zeros(self.world * self.name_cap, dtype=torch.uint8, device=device)
        self.g_off = torch.zeros(self.world * (self.max_n + 1), dtype=torch.int64, device=device)
        self._chunk_base = (torch.arange(self.world, device=device, dtype=torch.int64) * self.name_cap
                            ).repeat_interleave(self.max_n + 1)

    def gather(self) -> Gathered:
        """Call after filling send_recs[:n], send_names and send_off[:n + 1]."""
        if self.via_host:
            for dst, src in ((self.g_recs, self.send_recs), (self.g_names, self.send_names), (self.g_off, self.send_off)):
                h = src.cpu()
                parts = [torch.empty_like(h) for _ in range(self.world)]
                dist.all_gather(parts, h, group=self.group)
                dst.copy_(torch.cat(parts).to(dst.device))
        else:
            dist.all_gather_into_tensor(self.g_recs, self.send_recs, group=self.group)
            dist.all_gather_into_tensor(self.g_names, self.send_names, group=self.group)
            dist.all_gather_into_tensor(self.g_off, self.send_off, group=self.group)
        goff = self.g_off + self._chunk_base
        idx = goff.view(self.world, self.max_n + 1)[:, :self.max_n].reshape(-1).contiguous()
        return Gathered(self.g_recs, self.g_names, idx, self.max_n)


class NameCheck:
    """Exact, constant-per-rank test "does any query name occur on more than one rank?" for contig-sharded runs
    (gci_hash_bucket / gci_hash_conflicts in include/gci_hip.h): every rank sends the 64-bit hash of each passing
    record to rank (hash >> 33) % world with ONE all-to-all of fixed-size buckets (word 0 = count); the receiver
    counts hashes that arrive from two different ranks into a local device counter.  The caller sums that counter
    over ranks with whatever integer all-reduce it already does (the genome-wide totals) -- zero => every rank's
    local join equals the global join (equal names have equal hashes).  Non-zero (a shared name, a 56-bit hash
    collision or a bucket overflow) => run the replicated join over gathered records + names instead.

    `bucket_fn(recs, n_parts, cap, out)` and `conflict_fn(buckets, n_parts, cap, n_conflicts)` are the two device
    operations (Engine.hash_bucket / Engine.hash_conflicts on the GPU)."""

    def __init__(self, n_local: int, device: torch.device, bucket_fn, conflict_fn, group=None, alternate: bool = False,
                 via_host: bool = False):
        """alternate: two send arrays used in turn, bucket_fn(recs, n_parts, cap, out, next_out) zeroes the count
        words of the one the next call fills (Engine.hash_bucket: no clearing launch per step).
        via_host: the all-to-all runs on host copies (gloo backend with device tensors: two-rank tests on one GPU)."""
        self.group = group
        self.via_host = via_host
        self.world = dist.get_world_size(group)
        self.bucket_fn, self.conflict_fn = bucket_fn, conflict_fn
        n = torch.tensor([n_local], dtype=torch.int64, device="cpu" if via_host else device)
        alln = [torch.zeros_like(n) for _ in range(self.world)]
        dist.all_gather(alln, n, group=group)
        max_n = max(int(x.item()) for x in alln)
        self.cap = -(-max_n * 13 // (10 * self.world)) + 1024       # 1.3 x the mean bucket + slack
        self.send = torch.zeros(self.world * (self.cap + 1), dtype=torch.int64, device=device)
        self.send_alt = torch.zeros_like(self.send) if alternate else None
        self.recv = torch.zeros_like(self.send)
        self.n_conf = torch.zeros(1, dtype=torch.int32, device=device)   # local, cumulative until reset()

    def reset(self) -> None:
        self.n_conf.zero_()

    def enqueue(self, recs: torch.Tensor) -> torch.Tensor:
        """Asynchronous: adds this step's LOCAL conflict count to self.n_conf (device) and returns it.  A caller
        may go on speculatively with the local join and read the (all-reduced) verdict later."""
        if self.send_alt is not None:
            self.bucket_fn(recs, self.world, self.cap, self.send, self.send_alt)
        else:
            self.bucket_fn(recs, self.world, self.cap, self.send)
        self._all_to_all()
        if self.send_alt is not None:
            self.send, self.send_alt = self.send_alt, self.send
        self.conflict_fn(self.recv, self.world, self.cap, self.n_conf)
        return self.n_conf

    def _all_to_all(self) -> None:
        if not self.via_host:
            dist.all_to_all_single(self.recv, self.send, group=self.group)
            return
        h_send = self.send.cpu()
        h_recv = torch.empty_like(h_send)
        dist.all_to_all_single(h_recv, h_send, group=self.group)
        self.recv.copy_(h_recv)

    def enqueue_files(self, recs_per_file: Sequence[torch.Tensor]) -> torch.Tensor:
        """enqueue() for a rank that holds records of several input files: the hashes of all of them go into the same
        buckets (a name in two files of ONE rank is not a conflict: same source) and travel in ONE all-to-all.  Needs
        alternate=True (the bucket counters then accumulate across the calls of one step)."""
        if self.send_alt is None:
            raise ValueError("NameCheck.enqueue_files needs alternate=True")
        for recs in recs_per_file:
            self.bucket_fn(recs, self.world, self.cap, self.send, self.send_alt)
        self._all_to_all()
        self.send, self.send_alt = self.send_alt, self.send
        self.conflict_fn(self.recv, self.world, self.cap, self.n_conf)
        return self.n_conf

    def conflicts(self, recs: torch.Tensor) -> int:
        """Synchronous, global verdict for one record set."""
        self.reset()
        t = self.enqueue(recs).clone()
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return int(t.item())


def allreduce_totals(sum_depth: int, n_bases: int, device: torch.device, group=None) -> Tuple[int, int]:
    """Genome-wide (sum of depth, bases): the numerator / denominator of the global mean depth
    (GCI.py:862-868).  Integers, so the result is independent of the reduction order."""
    t = torch.tensor([int(sum_depth), int(n_bases)], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return int(t[0].item()), int(t[1].item())


def gather_interval_lists(local: List[Tuple[int, int, int]], group=None) -> List[Tuple[int, int, int]]:
    """(global contig, start, end) issue intervals of every rank, on every rank, sorted by
    (contig, start): what rank 0 needs to write the BED / .gci in header order.  <= 10^4 items."""
    out: List[Optional[list]] = [None] * dist.get_world_size(group)
    dist.all_gather_object(out, list(local), group=group)
    return sorted(x for part in out for x in part)


class ShardedJoin:
    """The name-hash-sharded join of a contig-sharded run (gci_route_* in include/gci_hip.h, k_shard.hip).

    Per step, with F input files:  F x [route the rank's passing records by (name hash >> 33) % world into record buckets and name
    slots]  ->  ONE all-to-all of all of them (exchange_files; exchange_file, the per-file form with two all-to-alls each, stays for
    callers that take the files one at a time)  ->  F x seal  ->  gci_name_join over what arrived (the names this rank owns,
    from every contig)  ->  route the intervals by the owner of their contig -> all-to-all -> seal (global contig -> index in
    this rank's track layout).  Bucket capacities are fixed at construction (1.3 x the even share + slack): the collectives
    have one shape for every step and nothing is sized on the host per step; an overflow or a name longer than a routed
    slot sets `status` (GCI_E_CAPACITY) and the caller grows the buckets (`grow()`) or takes the replicated join.

    `ops` provides the device operations: route_records, route_seal_records, route_intervals, route_seal_intervals and
    name_join (an `Engine`; the CPU tests pass numpy stand-ins)."""

    def __init__(self, ops, n_local: Sequence[int], owner: Sequence[int], device: torch.device, group=None,
                 via_host: bool = False, slack: float = 1.3, name_slot: int = 48, extra_records: int = 0):
        """name_slot: bytes of a routed name slot, a multiple of 16 and at least the longest query name of the run (the same on
        every rank).  extra_records: rows of the join inputs that do NOT come through exchange_file (PAF records already on the
        rank that owns their name): the join can emit an interval for each of them (GCI.py:279-280, 298-299)."""
        self.ops, self.group, self.via_host, self.device = ops, group, via_host, device
        self.ROUTE_NAME = int(name_slot)
        self.extra_records = int(extra_records)
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.n_files = len(n_local)
        n = torch.tensor([int(x) for x in n_local], dtype=torch.int64, device="cpu" if via_host else device)
        alln = [torch.zeros_like(n) for _ in range(self.world)]
        dist.all_gather(alln, n, group=group)
        per_file_max = [max(int(a[f].item()) for a in alln) for f in range(self.n_files)]
        per_file_sum = [sum(int(a[f].item()) for a in alln) for f in range(self.n_files)]
        self.owner_np = np.asarray(owner, dtype=np.int32)
        cmap, self.mine = contig_map_for(owner, self.rank)
        self.owner = torch.from_numpy(self.owner_np.copy()).to(device)
        self.cmap = torch.from_numpy(cmap).to(device)
        self._alloc([int(m * slack / self.world) + 1024 for m in per_file_max],
                    int(max(per_file_sum + [1]) * slack * 1.25 / (self.world * self.world)) + 1024)

    def _alloc(self, rec_caps: Sequence[int], ivl_cap: int) -> None:
        dev, W = self.device, self.world
        self.rec_cap = [int(c) for c in rec_caps]
        self.ivl_cap = int(ivl_cap)
        z = lambda *shape, dt=torch.uint8: torch.zeros(shape, dtype=dt, device=dev)            # noqa: E731
        self.send_recs = [z(W * (c + 1), 32) for c in self.rec_cap]
        self.recv_recs = [z(W * (c + 1), 32) for c in self.rec_cap]
        self.send_names = [z(W * c * self.ROUTE_NAME) for c in self.rec_cap]
        self.recv_names = [z(W * c * self.ROUTE_NAME) for c in self.rec_cap]
        self.name_off = []
        for c in self.rec_cap:
            i = torch.arange(W * (c + 1), dtype=torch.int64, device=dev)
            d, k = i // (c + 1), i % (c + 1) - 1
            self.name_off.append(((d * c + k.clamp(min=0)) * self.ROUTE_NAME).contiguous())
        # what the local join emits (global contig indices): at most one interval per distinct name, i.e. <= every record that can
        # arrive (the routed BAM buckets) + every PAF record the caller passes to join() beside them (`extra_records`)
        total = sum(W * (c + 1) for c in self.rec_cap) + self.extra_records
        self.ivl = z(max(total, 1), 4, dt=torch.int32)
        self.count = z(1, dt=torch.int32)
        self.send_ivl = z(W * (self.ivl_cap + 1), 4, dt=torch.int32)
        self.recv_ivl = z(W * (self.ivl_cap + 1), 4, dt=torch.int32)
        self.status = torch.full((2 * self.n_files + 3,), -1, dtype=torch.int64, device=dev)   # route / seal per file, join, route, seal

    def grow(self, factor: float = 2.0) -> None:
        self.send_all = self.recv_all = None
        self._alloc([int(c * factor) for c in self.rec_cap], int(self.ivl_cap * factor))

    def _all_to_all(self, recv: torch.Tensor, send: torch.Tensor) -> None:
        all_to_all_bytes(recv, send, self.group, self.via_host)

    def exchange_file(self, f: int, ji) -> "object":
        """One file's records of this rank -> the records (of every rank) whose names this rank owns, as a join input."""
        from .device import JoinInput
        cap, W = self.rec_cap[f], self.world
        self.ops.route_records(ji, W, cap, self.send_recs[f], self.send_names[f], self.ROUTE_NAME, self.status[2 * f:2 * f + 1])
        self._all_to_all(self.recv_recs[f], self.send_recs[f])
        self._all_to_all(self.recv_names[f], self.send_names[f])
        self.ops.route_seal_records(self.recv_recs[f], W, cap, self.status[2 * f + 1:2 * f + 2])
        return JoinInput(self.recv_recs[f], self.recv_names[f], self.name_off[f], 0)

    def exchange_files(self, jis: Sequence["object"]) -> List["object"]:
        """exchange_file() for ALL files of the step in ONE collective (round 6: a step was 2 F + 1 all-to-alls + 1 all-reduce
        around ~1 ms of kernels per rank at 8 GPUs -- latency, not bytes; it is 3 collectives now whatever F is).  Every file's
        record buckets and name slots are routed into their own arrays as before (the kernels keep their contiguous bucket
        layout), packed side by side into one send buffer -- per destination rank [records of file 0 | ... | names of file 0 | ...],
        2 F strided copies --, exchanged, and unpacked into the per-file arrays the seal step and the join read."""
        from .device import JoinInput
        W, F = self.world, self.n_files
        if len(jis) != F:
            raise ValueError("ShardedJoin.exchange_files: %d inputs for %d files" % (len(jis), F))
        if getattr(self, "send_all", None) is None or int(self.send_all.shape[1]) != self._per_dest():
            self.send_all = torch.zeros((W, self._per_dest()), dtype=torch.uint8, device=self.device)
            self.recv_all = torch.zeros_like(self.send_all)
        at = 0
        spans = []
        for f, ji in enumerate(jis):
            cap = self.rec_cap[f]
            self.ops.route_records(ji, W, cap, self.send_recs[f], self.send_names[f], self.ROUTE_NAME, self.status[2 * f:2 * f + 1])
            nr, nn = (cap + 1) * 32, cap * self.ROUTE_NAME
            self.send_all[:, at:at + nr] = self.send_recs[f].view(W, nr)
            self.send_all[:, at + nr:at + nr + nn] = self.send_names[f].view(W, nn)
            spans.append((at, nr, nn))
            at += nr + nn
        self._all_to_all(self.recv_all, self.send_all)
        out = []
        for f, (a, nr, nn) in enumerate(spans):
            cap = self.rec_cap[f]
            self.recv_recs[f].view(W, nr).copy_(self.recv_all[:, a:a + nr])
            self.recv_names[f].view(W, nn).copy_(self.recv_all[:, a + nr:a + nr + nn])
            self.ops.route_seal_records(self.recv_recs[f], W, cap, self.status[2 * f + 1:2 * f + 2])
            out.append(JoinInput(self.recv_recs[f], self.recv_names[f], self.name_off[f], 0))
        return out

    def _per_dest(self) -> int:
        return sum((c + 1) * 32 + c * self.ROUTE_NAME for c in self.rec_cap)

    COLLECTIVES_PER_STEP = 3          # exchange_files (records + names of every file), the intervals, the all-reduce of the sums

    def join(self, inputs: Sequence["object"], ovlp_percent: float) -> Tuple[torch.Tensor, int]:
        """inputs: what exchange_file returned for every file, in the reference's file order (a PAF input whose records already
        sit on the rank that owns their name is passed as it is).  -> (intervals int32 [W * (cap + 1), 4] on the contigs of THIS
        rank, contig = index in its track layout or -1 for an empty slot; their number of slots)."""
        F, W = self.n_files, self.world
        rows = sum(int(i.recs.shape[0]) for i in inputs)
        if rows > int(self.ivl.shape[0]):                 # (inputs the constructor was not told about: never emit past the array)
            self.ivl = torch.zeros((rows, 4), dtype=torch.int32, device=self.device)
        self.ops.name_join(inputs, ovlp_percent, None, self.ivl, self.count, False, None, status=self.status[2 * F:2 * F + 1])
        self.ops.route_intervals(self.ivl, self.count, self.owner, W, self.ivl_cap, self.send_ivl, self.status[2 * F + 1:2 * F + 2])
        self._all_to_all(self.recv_ivl, self.send_ivl)
        self.ops.route_seal_intervals(self.recv_ivl, W, self.ivl_cap, self.cmap, self.status[2 * F + 2:2 * F + 3])
        return self.recv_ivl, int(self.recv_ivl.shape[0])

    def bytes_per_step(self) -> int:
        """What this rank sends to the OTHER ranks per step (fixed bucket shapes)."""
        W = self.world
        per = sum((c + 1) * 32 + c * self.ROUTE_NAME for c in self.rec_cap) + (self.ivl_cap + 1) * 16
        return per * (W - 1)

    def check(self, decode) -> None:
        """Raises through `decode(word, what)` for the first status word that reports something (read after a sync)."""
        what = []
        for f in range(self.n_files):
            what += ["gci_route_records[%d]" % f, "gci_route_seal_records[%d]" % f]
        what += ["gci_name_join", "gci_route_intervals", "gci_route_seal_intervals"]
        for w, name in zip(self.status.cpu().numpy().view(np.uint64).tolist(), what):
            decode(w, name)
        n = int(self.count.cpu().reshape(-1)[0])          # the join counts what it could not store as well: never a silent loss
        if n > int(self.ivl.shape[0]):
            raise RuntimeError("ShardedJoin: the join emitted %d intervals into %d rows" % (n, int(self.ivl.shape[0])))



# ---- PAF files sharded by byte range (the PAF half of filter(), /root/reference/GCI.py:211-254) ------------------------------

def line_start_at_or_after(raw: np.ndarray, pos: int) -> int:
    """The first position >= pos of the byte array `raw` at which a line starts, as Python's universal newlines cut lines (what
    `for line in f` at GCI.py:217 sees; k_paf.hip: line_starts_at): position 0, behind '\\n', or behind a '\\r' that no '\\n'
    follows.  len(raw) when no line starts at or behind pos."""
    n = int(raw.shape[0])
    if pos <= 0:
        return 0
    p = int(pos)
    while p < n:
        hi = min(n, p + (1 << 16))
        prev = np.asarray(raw[p - 1:hi - 1])
        cur = np.asarray(raw[p:hi])
        hit = np.flatnonzero((prev == 10) | ((prev == 13) & (cur != 10)))
        if hit.shape[0]:
            return p + int(hit[0])
        p = hi
    return n


def byte_range_of_rank(raw: np.ndarray, rank: int, world: int) -> Tuple[int, int]:
    """Rank `rank`'s share of a text file: the lines that START in [size * rank / world, size * (rank + 1) / world)."""
    n = int(raw.shape[0])
    lo = line_start_at_or_after(raw, n * rank // world)
    hi = n if rank + 1 == world else line_start_at_or_after(raw, n * (rank + 1) // world)
    return lo, max(lo, hi)


def all_to_all_bytes(recv: torch.Tensor, send: torch.Tensor, group=None, via_host: bool = False) -> None:
    """Equal-split all-to-all of two device tensors of the same shape (via_host: staged through host memory, gloo)."""
    if not via_host:
        dist.all_to_all_single(recv.view(-1), send.view(-1), group=group)
        return
    h = send.reshape(-1).cpu()
    r = torch.empty_like(h)
    dist.all_to_all_single(r, h, group=group)
    recv.view(-1).copy_(r)


def paf_by_byte_range(ops, paths: Sequence[str], targets: Sequence[str], map_qual: int, mq_cutoff: int, iden_percent: float,
                      world: int, rank: int, all_reduce_max, device: torch.device, group=None, via_host: bool = False,
                      slack: float = 1.3) -> Optional[list]:
    """The PAF files of a run, every rank parsing 1/world of their BYTES.

    GCI.py:211-254 scores a query over ALL its lines (of all files so far, in file order), and a query's lines may lie
    anywhere in a file -- so the work splits in two: (A) every rank tokenises and filters the lines that start in its byte range
    of every file (`ops.paf_hits_text`: the expensive half -- str.strip / split / int() / the f64 identity of every line);
    (B) the 80-byte hits travel to the rank that owns their query name ((hash >> 33) % world, the routing rule of the records:
    one all-to-all of hits and one of name slots per file, stable, so that a query's hits arrive in line order: the ranges ascend
    with the rank) and are scored there (`ops.paf_score_hits`).  -> one JoinInput per file holding the queries THIS rank owns,
    i.e. what `_own_names_only(paf_filter(whole files))` gives.

    None: some rank met a line (or a query) the reference raises on -- every rank returns None together and the caller runs the
    whole files the old way, where the reference's exception comes out with its exact type and line."""
    F = len(paths)
    bad = 0
    hits: List[torch.Tensor] = []
    d_text = None
    try:
        parts = []
        for p in paths:
            raw = np.memmap(p, dtype=np.uint8, mode="r") if _file_size(p) else np.zeros(0, np.uint8)
            lo, hi = byte_range_of_rank(raw, rank, world)
            parts.append(np.asarray(raw[lo:hi]))
        ends = np.cumsum([x.shape[0] for x in parts], dtype=np.uint64)
        text = np.concatenate(parts + [np.zeros(16, np.uint8)])
        d_text = ops.to_device(text)
        hits = ops.paf_hits_text(d_text, ends, targets, map_qual, mq_cutoff, iden_percent)
    except Exception as e:                                   # noqa: BLE001 -- agreed on below; the fallback raises it properly
        if not _is_data_error(e):
            raise
        bad = 1
    counts = [int(h.shape[0]) for h in hits] if not bad else [0] * F
    longest = 0
    if not bad:
        for h in hits:
            if int(h.shape[0]):
                longest = max(longest, int(h[:, 64:68].contiguous().view(torch.int32).max().item()))
    red = all_reduce_max([bad, longest] + counts)
    if red[0]:
        return None
    name_slot = max(16, (red[1] + 15) // 16 * 16)
    caps = [int(c * slack / world) + 256 for c in red[2:]]
    B = ops.PAF_HIT_BYTES
    while True:
        status = torch.full((F,), -1, dtype=torch.int64, device=device)
        recv_hits, recv_names = [], []
        for f in range(F):
            cap = caps[f]
            send_h = torch.zeros((world * (cap + 1), B), dtype=torch.uint8, device=device)
            send_n = torch.zeros(world * cap * name_slot, dtype=torch.uint8, device=device)
            ops.route_hits(hits[f], d_text, world, cap, send_h, send_n, name_slot, status[f:f + 1])
            r_h, r_n = torch.empty_like(send_h), torch.empty_like(send_n)
            all_to_all_bytes(r_h, send_h, group, via_host)
            all_to_all_bytes(r_n, send_n, group, via_host)
            recv_hits.append(r_h)
            recv_names.append(r_n)
        # a bucket that overflowed on ANY rank (queries hash unevenly): larger buckets everywhere, again
        over = int((status.cpu() != -1).any().item())
        if all_reduce_max([over])[0]:
            caps = [2 * c for c in caps]
            continue
        break
    # what arrived: per file, per source rank, `count` hits behind the header slot -> one dense array in (file, rank, line) order,
    # the names of all files in one buffer, qn_off pointing at the name slots
    d_names = torch.cat(recv_names + [torch.zeros(16, dtype=torch.uint8, device=device)])
    dense, upto, base = [], [0], 0
    for f in range(F):
        cap = caps[f]
        bk = recv_hits[f].view(world, cap + 1, B)
        cnt = bk[:, 0, 8:16].contiguous().view(torch.int64).reshape(-1).cpu().tolist()
        n_f = 0
        for d in range(world):
            c = int(cnt[d])
            if c:
                part = bk[d, 1:1 + c].clone()
                off = base + (d * cap + torch.arange(c, dtype=torch.int64, device=device)) * name_slot
                part[:, 0:8] = off.view(torch.uint8).view(c, 8)
                dense.append(part)
                n_f += c
        upto.append(upto[-1] + n_f)
        base += world * cap * name_slot
    d_hits = torch.cat(dense) if dense else torch.zeros((1, B), dtype=torch.uint8, device=device)
    bad, out = 0, None
    try:
        out = ops.paf_score_hits(d_names, d_hits, upto, targets)
    except Exception as e:                                   # noqa: BLE001
        if not _is_data_error(e):
            raise
        bad = 1
    if all_reduce_max([bad])[0]:
        return None
    return out


def _file_size(path: str) -> int:
    import os
    return os.path.getsize(path)


def _is_data_error(e: Exception) -> bool:
    """An error the INPUT causes (a line the reference raises on): the sharded PAF path hands those to the whole-file path.
    Anything else (HIP, memory, a bug) is raised where it happens."""
    from ._lib import GciError, GCI_E_MALFORMED, GCI_E_ZERO_DIV
    return isinstance(e, GciError) and e.status in (GCI_E_MALFORMED, GCI_E_ZERO_DIV)
