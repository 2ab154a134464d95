"""Multi-GPU sharding of the path: one process per GPU, contigs sharded, ONE real
exchange step.

* Depth build, gap mask, two-type max, issue scan, text and per-contig scoring are independent
  per contig -> each rank owns a set of contigs (longest-processing-time packing).
* The read-name join is NOT contig-local (a read aligned to different contigs in two files must
  be dropped, GCI.py:296-297; a repeated name keeps only its last record, GCI.py:269), so every
  rank decodes a slice of each file's records (K1) and the 32-byte compact records plus the
  name bytes are replicated with an RCCL all-gather; every rank then runs the full join and
  keeps the intervals of the contigs it owns (`contig_map`).
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
