#!/usr/bin/env python3
"""bench.py -- aligned Gbases/s through the filter+depth pipeline (BASELINE.json metric).

A "step" is one pass of the hot path over one batch of synthetic alignments whose inflated BAM
bytes are already resident in HBM:

    K1 record filter -> [N>1: exact cross-rank name check, one RCCL all-to-all of 8-byte name hashes; only if a name
       is shared between ranks: all-gather of compact records + names and the replicated join] -> K3 name join
    -> K4/K5 depth build with, fused into the same two passes over the per-tile event buckets, the
       per-contig sums, the issue-scan run boundaries and the decimal depth text (K8 / K10 / R15)
    -> [N>1: RCCL all-reduce of the int64 sum of depth]

Workload at N=1: BASELINE.json configs[1] -- CHM13 chr19 (61,707,364 bp), one 40x HiFi BAM.
At N>1 (weak scaling) every rank owns one chr19-sized contig of an N-contig assembly and the
records of that contig; results are identical to a single-GPU run over the same N contigs.

Launch:  python bench.py --gpus 1 --steps K --warmup W
         python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CHR19_LEN = 61_707_364
HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--contig-len", type=int, default=CHR19_LEN, help="per-rank contig length (default chr19)")
    ap.add_argument("--coverage", type=float, default=40.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--inflight", type=int, default=1,
                    help="steps in flight (1 GPU only): N > 1 runs consecutive steps on N streams, each with its own library "
                         "context and buffers -- the latency-bound filter / join of one step then overlaps the HBM-bound tile "
                         "build of another.  Default 1: every kernel runs alone and the roofline figure is that of the kernel")
    ap.add_argument("--heads", action="store_true",
                    help="feed the filter the heads stream (records without SEQ / QUAL, what the command line uploads: "
                         "gci_bam_heads + gci_bam_filter_heads) instead of the whole inflated stream")
    ap.add_argument("--force-exchange", action="store_true",
                    help="run the multi-GPU name check / exchange and the all-reduce even with one rank (self-test)")
    ap.add_argument("--force-replicated", action="store_true",
                    help="with the exchange: always take the replicated-join fallback (all-gather of records + names)")
    return ap.parse_args()


class Workload:
    """Per-rank resident inputs + preallocated outputs for one step."""

    def __init__(self, eng, rank, world, contig_len, coverage, exchange=False, replicated=False, heads=False):
        import torch
        from gci_amd import synth
        self.torch = torch
        self.eng, self.rank, self.world = eng, rank, world
        names = ["chr19"] if world == 1 else ["chr19_%d" % r for r in range(world)]
        self.contigs = tuple((n, contig_len) for n in names)
        # this rank's slice of the BAM: the records of its own contig
        own = ((names[rank], contig_len),)
        rs = synth.simulate_reads(own, coverage, "hifi", seed=synth.seed_for(2, rank))
        self.aligned_bases = rs.aligned_bases()
        self.n_rec = len(rs)
        self.name_bytes = int(np.char.str_len(rs.names).sum())
        # the BAM header lists all N contigs; refID of this rank's records = rank
        rs.ref_id[:] = rank
        rs.contigs = self.contigs
        stream, offs = synth.to_bam_stream(rs)
        self.stream_bytes = int(stream.shape[0])
        self.host_stream, self.host_offs = (stream, offs) if rank == 0 and world == 1 else (None, None)
        self.heads = heads
        if heads:                                          # through a BGZF file and the native host pipeline, as the CLI does
            import tempfile
            from gci_amd import hostio
            from gci_amd.formats import bam as bamfmt
            with tempfile.TemporaryDirectory(prefix="gci_bench_") as tmp:
                path = os.path.join(tmp, "r%d.bam" % rank)
                bamfmt.write_bam_stream(path, stream, level=1, threads=hostio.default_threads())
                with hostio.bam_heads(np.fromfile(path, dtype=np.uint8)) as hd:
                    assert hd.offsets.shape[0] == self.n_rec
                    self.d_bam, self.d_off = eng.to_device(hd.stream), eng.to_device(hd.offsets)
            self.stream_bytes = int(self.d_bam.shape[0])
        else:
            self.d_bam = eng.to_device(stream)
            self.d_off = eng.to_device(offs)
        del stream
        self.ref_sel = eng.to_device(np.arange(world, dtype=np.int32))
        eng.set_layout([contig_len])                       # local track: the contig this rank owns
        cmap = np.full(world, -1, dtype=np.int32)
        cmap[rank] = 0
        self.exchange = exchange or world > 1
        self.force_replicated = replicated
        self.contig_map = eng.to_device(cmap) if self.exchange else None
        dev = eng.device
        self.recs = torch.empty((self.n_rec, 32), dtype=torch.uint8, device=dev)
        self.track = eng.new_track()
        self.ivl = torch.empty((self.n_rec * (world if world > 1 else 1), 4), dtype=torch.int32, device=dev)
        self.count = torch.zeros(1, dtype=torch.int32, device=dev)
        self.keys = torch.empty(1 << 16, dtype=torch.int64, device=dev)
        self.nkeys = torch.zeros(1, dtype=torch.int32, device=dev)
        self.text_off = torch.zeros(2, dtype=torch.int64, device=dev)
        # [sum of depth (zeroed and written by every build: d_sums), bases of this rank, cross-rank name conflicts so far
        # (low word: the int32 counter of the name check)].  At N > 1 the step all-reduces the sum IN PLACE (the global
        # mean depth is that over the constant total of bases); the conflict count is all-reduced once, in check().
        self.totals_src = torch.zeros(3, dtype=torch.int64, device=dev)
        self.sums = self.totals_src[0:1]
        self.status = torch.zeros(2, dtype=torch.int64, device=dev)
        self.text = None
        self.rec_base = 0
        self.replicated_steps = 0
        from gci_amd._lib import BuildOpts
        o = BuildOpts()
        o.flank, o.want_text, o.counted = 15, 1, 1          # counted: the join has done the build's counting pass
        o.d_contig_text_off, o.d_sums = self.text_off.data_ptr(), self.sums.data_ptr()
        o.d_n_keys, o.d_keys, o.key_cap = self.nkeys.data_ptr(), self.keys.data_ptr(), int(self.keys.shape[0])
        o.issue_flank, o.lo, o.hi = 15, -1.0, 0.0
        self.opts = o
        if self.exchange:
            self._setup_exchange()

    # ---- multi-GPU: every rank needs every record's (hash, interval, name) for the join ----------
    def _setup_exchange(self):
        from gci_amd import shard
        self.ex = shard.RecordExchange(self.n_rec, self.name_bytes, self.eng.device)
        self.rec_base = self.ex.rec_idx_base
        self.recs = self.ex.send_recs                       # K1 writes straight into the send buffer
        self.ivl = self.torch.empty((self.world * self.ex.max_n, 4), dtype=self.torch.int32, device=self.eng.device)
        # Both collectives of the step are synchronous ops; the kernel trace shows their RCCL kernels on the hardware
        # queue of the step's own kernels, in line with them.  (Tried: the name check on a side stream and the
        # all-reduce as an async op next to the tile build -- both put the collective on another queue and made the
        # step 20 - 45 us slower on this chip.)
        self.check_names = shard.NameCheck(self.n_rec, self.eng.device, self.eng.hash_bucket, self.eng.hash_conflicts,
                                           alternate=True)
        self.check_names.n_conf = self.totals_src[2:3].view(self.torch.int32)[0:1]      # counted straight into the totals
        self.totals_src[1] = self.contigs[self.rank][1]
        self.replicated_steps = 0

    def _p(self, t):
        return ctypes.c_void_p(t.data_ptr()) if t is not None else None

    def step(self):
        eng, lib, ctx = self.eng, self.eng.lib, self.eng.ctx
        from gci_amd._lib import JoinFile
        chk = eng._chk
        k1 = lib.gci_bam_filter_heads if self.heads else lib.gci_bam_filter
        chk(k1(ctx, self._p(self.d_bam), self.stream_bytes, self._p(self.d_off), self.n_rec, self._p(self.ref_sel),
               self.world, 30, 50, 0.1, 0.9, self.rec_base, self._p(self.recs), self._p(self.status[0:1])), "gci_bam_filter")
        jf = (JoinFile * 1)()
        if not self.exchange:
            jf[0].d_recs, jf[0].n_recs, jf[0].name_delta = self.recs.data_ptr(), self.n_rec, 36
            jf[0].d_name_base, jf[0].d_name_off = self.d_bam.data_ptr(), self.d_off.data_ptr()
        elif not self.force_replicated:
            # Exact cross-rank name test (hash all-to-all, 8 bytes per record), enqueued without a host sync: the
            # step goes on SPECULATIVELY with the local join; check() reads the accumulated verdict after the
            # timed region and main() redoes everything with the replicated join if any step saw a conflict.
            self.check_names.enqueue(self.recs[:self.n_rec])          # adds to check_names.n_conf (device, local)
            jf[0].d_recs, jf[0].n_recs, jf[0].name_delta = self.recs.data_ptr(), self.n_rec, 36
            jf[0].d_name_base, jf[0].d_name_off = self.d_bam.data_ptr(), self.d_off.data_ptr()
        else:
            # a name occurs on two ranks: replicate records + names and join everything everywhere
            self.replicated_steps += 1
            ex = self.ex
            loc = (JoinFile * 1)()
            loc[0].d_recs, loc[0].n_recs, loc[0].name_delta = self.recs.data_ptr(), self.n_rec, 36
            loc[0].d_name_base = self.d_bam.data_ptr()
            loc[0].d_name_off = self.d_off.data_ptr()
            chk(lib.gci_pack_names(ctx, loc, self._p(ex.send_names), ex.name_cap, self._p(ex.send_off)),
                "gci_pack_names")
            g = ex.gather()
            self._g = g                                     # keep the index tensor alive until the join ran
            jf[0].d_recs, jf[0].n_recs, jf[0].name_delta = g.recs.data_ptr(), self.world * g.max_n, 0
            jf[0].d_name_base, jf[0].d_name_off = g.names.data_ptr(), g.name_index.data_ptr()
        # the join also does the counting pass of the depth build over the intervals it emits (gci_name_join_count)
        chk(lib.gci_name_join_count(ctx, jf, 1, 0.9, self._p(self.contig_map), self._p(self.ivl), int(self.ivl.shape[0]),
                                    self._p(self.count), self._p(self.status[1:2]), int(self.opts.flank)), "gci_name_join_count")
        # fused build: depth + per-contig sums + text byte offsets + issue-run boundaries from one pass over
        # the per-tile event buckets (no HBM re-read of the track), then depth + decimal text in the second
        o = self.opts
        chk(lib.gci_depth_build_begin(ctx, self._p(self.ivl), self._p(self.count), int(self.ivl.shape[0]),
                                      ctypes.byref(o)), "gci_depth_build_begin")
        if self.text is None:                              # first (warm-up) call sizes the text buffer
            total = int(self.text_off[1].item())
            self.text = self.torch.empty(total + (total >> 4) + 4096, dtype=self.torch.uint8, device=eng.device)
        chk(lib.gci_depth_build_finish(ctx, self._p(self.track), self._p(self.text), int(self.text.shape[0])),
            "gci_depth_build_finish")
        if self.exchange:
            import torch.distributed as dist
            # ONE integer all-reduce per step, in place: the genome-wide sum of depth (global mean depth = that / bases)
            dist.all_reduce(self.sums, op=dist.ReduceOp.SUM)

    def check(self):
        """Record-level status of the last step + issue-key capacity."""
        from gci_amd._lib import GciError
        for w, what in zip(self.status.cpu().numpy().view(np.uint64).tolist(), ("gci_bam_filter", "gci_name_join")):
            rec = ctypes.c_uint32(0)
            st = self.eng.lib.gci_decode_status(w, ctypes.byref(rec))
            if st != 0:
                raise GciError(st, "%s failed on record %d" % (what, rec.value))
        if int(self.nkeys.item()) > self.keys.shape[0] or int(self.count.item()) > self.ivl.shape[0]:
            raise GciError(-8, "bench output buffers too small")
        if self.exchange and not self.force_replicated:
            import torch.distributed as dist
            conf = self.totals_src[2:3].clone()                  # every rank's count of hashes seen from two ranks
            dist.all_reduce(conf, op=dist.ReduceOp.SUM)
            if int(conf.item()) > 0:
                return False      # a query name is shared between ranks: the speculative local joins were not exact
        return True


def cpu_baseline(w: Workload):
    """The oracle's single-thread C/Python restatement of the same step on the same chr19 input,
    timed on this host: filter -> dict -> slice-add depth -> run scan -> text -> sum."""
    from oracle import gci_oracle as O
    O.build()
    refs = [n for n, _ in w.contigs]
    tl = dict(w.contigs)
    t0 = time.perf_counter()
    d, hq = O.bam_file_dict(w.host_stream, w.host_offs, refs, refs, 30, 50, 0.1, 0.9)
    file1 = O.name_join([d], hq, 0.9)
    depths = O.depth_build(file1, tl, 15)
    bed = O.collapse_depth_range(depths, -1, 0, 15, 0)
    text = O.depth_text(depths)
    mean = O.mean_depth(depths)
    dt = time.perf_counter() - t0
    return dt, depths, bed, text, mean


def main():
    args = parse_args()
    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world == 1 and args.gpus > 1:
        sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    if world > 1 or args.force_exchange or args.force_replicated:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # RCCL logs to stdout: keep it off the channel on which rank 0 prints its ONE JSON line
        if os.environ.get("NCCL_DEBUG", "").upper() in ("VERSION", "INFO", "WARN"):
            os.environ.pop("NCCL_DEBUG")
        os.environ.setdefault("NCCL_DEBUG_FILE", "/dev/stderr")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from gci_amd.build import build_hip, needs_build
    if needs_build():
        if local_rank == 0:
            build_hip()
        if world > 1:
            dist.barrier()
    from gci_amd import _lib
    from gci_amd.device import Engine
    eng = Engine(local_rank)
    w = Workload(eng, rank, world, args.contig_len, args.coverage, exchange=args.force_exchange or args.force_replicated,
                 replicated=args.force_replicated, heads=args.heads)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # --inflight N: N - 1 more contexts on streams of their own; step k runs on context k % N
    lanes = [(eng, w, None)]
    if args.inflight > 1:
        if world > 1 or w.exchange:
            sys.exit("bench.py --inflight is for the single-GPU local path")
        for _ in range(args.inflight - 1):
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                e2 = Engine(local_rank, stream=st)
                w2 = Workload(e2, rank, world, args.contig_len, args.coverage, heads=args.heads)
                for _ in range(max(1, args.warmup)):
                    w2.step()
            lanes.append((e2, w2, st))

    def run_steps(n):
        for k in range(n):
            e_k, w_k, st = lanes[k % len(lanes)]
            if st is None:
                w_k.step()
            else:
                with torch.cuda.stream(st):
                    w_k.step()

    for _ in range(max(1, args.warmup)):
        w.step()
    fence()
    if not w.check():             # shared names between ranks: every further step takes the replicated join
        w.force_replicated = True
        w.check_names.reset()
        for _ in range(max(1, args.warmup)):
            w.step()
        fence()
        w.check()

    for e_k, _, _ in lanes:
        e_k.profile_enable(1 << _lib.PROF_DEPTH_SCAN)      # HIP events around the dominant kernel only
        e_k.profile_read(reset=True)
    fence()
    t0 = time.perf_counter()
    run_steps(args.steps)
    fence()
    dt = time.perf_counter() - t0
    prof = {}
    for e_k, _, _ in lanes:
        for name, (ms, n) in e_k.profile_read(reset=True).items():
            prof[name] = (prof.get(name, (0.0, 0))[0] + ms, prof.get(name, (0.0, 0))[1] + n)
        e_k.profile_enable(0)
    for _, w_k, _ in lanes:
        if not w_k.check():
            sys.exit("bench: a query name became shared between ranks during the timed region")
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=eng.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        ab = torch.tensor([w.aligned_bases], dtype=torch.int64, device=eng.device)
        dist.all_reduce(ab, op=dist.ReduceOp.SUM)
        aligned_total = int(ab.item())
    else:
        aligned_total = w.aligned_bases

    scan_ms, scan_n = prof.get("k_tile_build", (0.0, 0))
    scan_avg_ms = scan_ms / max(1, scan_n)
    # k_tile_build writes the int32 track (4 B/base) and the decimal text; it reads only the event buckets
    text_bytes = int(w.text_off[1].item())
    algo_bytes = 4.0 * args.contig_len + text_bytes       # DESIGN.md "algorithmic bytes"
    achieved = algo_bytes / (scan_avg_ms * 1e-3) / 1e9 if scan_avg_ms > 0 else 0.0

    # HBM bytes per launch of the dominant kernel from the committed PMC passes (cannot be collected from inside
    # this process: rocprofv3 wraps the command; see tools/prof_pmc.sh and profiles/)
    traffic = None
    try:
        tj = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_traffic.json"))
        if tj and args.contig_len == CHR19_LEN and args.coverage == 40.0:
            traffic = json.load(open(os.path.join(ROOT, "profiles", tj[-1])))["hbm_bytes_per_launch"]
    except Exception:
        traffic = None

    breakdown = None
    if True:
        eng.profile_enable((1 << _lib.PROF_COUNT) - 1)
        for _ in range(3):
            w.step()
        breakdown = {k: round(ms / n * 1e3, 2) for k, (ms, n) in eng.profile_read(reset=True).items()}   # us / launch
        eng.profile_enable(0)

    out = {
        "metric": "aligned Gbases/s through filter+depth pipeline (CHM13, 40x HiFi)",
        "value": aligned_total * args.steps / dt / 1e9,
        "unit": "Gbases/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "int32",
        "data": "synthetic",
        "config": {"workload": "CHM13 chr19 (%d bp) x %d contig(s), one %gx HiFi BAM, filter -> join -> depth -> "
                               "issue scan -> depth text" % (args.contig_len, world, args.coverage),
                   "records_per_gpu": w.n_rec, "aligned_bases_per_step": aligned_total,
                   "bam_input": "heads stream (records without SEQ / QUAL)" if w.heads else "whole inflated stream",
                   ("heads_bytes_per_gpu" if w.heads else "inflated_bam_bytes_per_gpu"): w.stream_bytes, "parallelism": "contig-sharded x%d" % world,
                   "steps_in_flight": len(lanes),
                   "join": ("local" if not w.exchange else
                            "local, validated by the exact cross-rank name check (hash all-to-all)" if not w.replicated_steps else
                            "replicated (all-gather of records + names)")},
        "roofline": {"bound": "hbm", "kernel": "k_tile_build (depth + text write)", "achieved": achieved, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "algorithmic_bytes_per_launch": algo_bytes, "avg_launch_ms": scan_avg_ms, "launches": scan_n},
        "kernel_us_per_launch": breakdown,
    }

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cdt, depths, bed, text, mean = cpu_baseline(w)
        out["cpu_baseline"] = {"value": w.aligned_bases / cdt / 1e9, "unit": "Gbases/s", "cores": 1, "kind": "port",
                               "sample": "one full step (all %d records, %d bp) through oracle/gci_oracle.{c,py}, "
                                         "%.1f s" % (w.n_rec, args.contig_len, cdt)}
        # the timed GPU result must equal the oracle's on the full-size input
        from gci_amd import pipeline
        tr = pipeline.DepthTracks(eng, dict(w.contigs), w.track)
        ok = np.array_equal(tr["chr19"], depths["chr19"])
        ok = ok and pipeline.collapse_depth_range(tr, -1, 0, 15, 0) == bed
        nk = int(w.nkeys.item())                              # the fused issue keys of the last timed step
        runs = eng._keys_to_runs(w.keys[:nk].cpu().numpy().view(np.uint64), 1)
        ok = ok and pipeline._issues_from_runs(runs[0], args.contig_len - 30, args.contig_len, 15, 0) == bed["chr19"]
        ok = ok and int(w.sums[0].item()) == int(depths["chr19"].sum())
        n_text = int(w.text_off[1].item())
        ok = ok and (b">chr19\n" + w.text[:n_text].cpu().numpy().tobytes()) == text
        ok = ok and tr.mean() == mean
        out["parity_vs_oracle_full_size"] = bool(ok)
        if not ok:
            print(json.dumps(out))
            sys.exit("PARITY FAILURE: GPU result differs from the oracle at full size")
    elif world == 1:
        out["cpu_baseline"] = None

    if rank == 0:
        sys.stdout.flush()
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
