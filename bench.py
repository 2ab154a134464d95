#!/usr/bin/env python3
"""bench.py -- aligned Gbases/s through the filter+depth pipeline (BASELINE.json metric).

A "step" is one pass of the hot path over one batch of synthetic alignments whose record bytes are already resident
in HBM:

    K1 record filter, once per input file -> [N>1: exact cross-rank name check, one RCCL all-to-all of 8-byte name
       hashes; only if a name is shared between ranks: all-gather of compact records + names, replicated join]
    -> K3 name join over the files (the `-op` overlap join, GCI.py:272-301)
    -> K4/K5 depth build with, fused into the same two passes over the per-tile event buckets, the per-contig sums,
       the issue-scan run boundaries and the decimal depth text (K8 / K10 / R15)
    -> [N>1: RCCL all-reduce of the int64 sums of depth]

Workload at N=1 (default): BASELINE.json configs[2] -- CHM13 whole genome (25 contigs, 3.117 Gb), HiFi 40x aligned
by two aligners (two BAM files of the same reads: the `-op` join has work to do), fed as heads streams (records
without SEQ / QUAL: what `gci_bam_heads` makes of a BGZF file and what the command line uploads).
`--workload chr19` keeps configs[1] (one chr19-sized contig, one 40x HiFi BAM, whole inflated stream in HBM).
At N>1 (weak scaling) every rank owns one CHM13-sized haplotype of an N-haplotype assembly and the records of its
contigs from both files.

Launch:  python bench.py --gpus N --steps K --warmup W      (N > 1: re-launches itself under torch.distributed.run, one process per GPU)
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

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")       # (gci_amd/__init__.py: streams that share a hardware queue run one after the other)
ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CHR19_LEN = 61_707_364
HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FILTER = (30, 50, 0.1, 0.9)    # -mq, -mc, -cp, -ip defaults (GCI.py:1040-1069)
OVLP, FLANK = 0.9, 15


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--paf-published-gb", default="",
                    help="--workload genome4 --cli: write the two PAF files at these sizes in GB, 'HIFI,ONT' (the reference's published CHM13 "
                         "run: '3.6,48'), by a cg:Z: tag behind every line; default: the PAF texts as generated (0.7 GB together)")
    ap.add_argument("--paf-gb", type=float, default=20.0,
                    help="--workload paf: size of the PAF text (the reference's CHM13 run reads a 48 GB ONT PAF and a 3.6 GB HiFi PAF)")
    ap.add_argument("--workload", choices=("genome", "chr19", "genome4", "diploid", "paf"), default="genome",
                    help="genome: BASELINE configs[2] (default); chr19: configs[1], one contig and one BAM per rank; genome4: configs[3] "
                         "on ONE GPU (CHM13, --hifi + --nano, per read type one BAM + one PAF: two filters, max, three issue scans, three "
                         "tracks); diploid: configs[4] on ONE GPU (mat + pat, 46 contigs, 6.2 Gb, HiFi 100x + ONT, 20 N gaps, -R regions)")
    ap.add_argument("--coverage-ont", type=float, default=None, help="genome4 / diploid: ONT coverage (default 40 / 20)")
    ap.add_argument("--scale", type=float, default=1.0, help="genome workload: shrink every contig (testing only)")
    ap.add_argument("--contig-len", type=int, default=CHR19_LEN, help="chr19 workload: per-rank contig length")
    ap.add_argument("--coverage", type=float, default=40.0)
    ap.add_argument("--reads", choices=["hifi", "ont"], default="hifi",
                    help="genome workload: read type of the simulated files (ont: 2 900 CIGAR ops per record on average, "
                         "the chunked long-CIGAR path; use with --scale: a whole-genome ONT heads stream is 40 GB per file)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true",
                    help="skip numbers (2) and (3) of SURVEY.md 8(d): the device pipeline incl. H2D / D2H and the "
                         "command-line wall time on a chr19 BAM with realistic SEQ / QUAL entropy")
    ap.add_argument("--only-step", action="store_true",
                    help="genome workload: the timed step, its roofline and the full-size parity check only -- no window / command-line-"
                         "shaped / in-flight legs (their k_tile_build launches write no text and would blur a profiler's per-kernel "
                         "averages), no end-to-end numbers, no CPU baseline: what tools/round_profile.sh wraps in rocprofv3")
    ap.add_argument("--cli", action="store_true",
                    help="genome4 / diploid: also run the two-read-type COMMAND LINE on real files (BGZF BAMs + PAFs on tmpfs) and hold its "
                         "files against the oracle (SURVEY 8(d) number 3 for configs[3]); kept out of the default run")
    ap.add_argument("--no-cli-genome", action="store_true",
                    help="genome workload: skip survey_8d.3_command_line_genome (the command line as a process of its own on the two "
                         "genome-size BGZF files: ~130 GB written to tmpfs first, minutes of host-side deflate)")
    ap.add_argument("--ingest-gb", type=float, default=64.0,
                    help="survey_8d.3b: GB of realistic-entropy BGZF streamed from host RAM through the command line's ingestion "
                         "(0: skip)")
    ap.add_argument("--inflight", type=int, default=1,
                    help="chr19 workload, 1 GPU: N > 1 runs consecutive steps on N streams with a context each")
    ap.add_argument("--heads", action="store_true", help="chr19 workload: feed the heads stream instead of the whole stream")
    ap.add_argument("--k1", choices=("pages", "stream"), default="pages",
                    help="record filter input: pages = RECORD PAGES (gci_bam_pages_*: what the command line's ingestion leaves on "
                         "the device; gci_bam_filter_pages), stream = the whole inflated stream + offset table (gci_bam_filter; chr19 workload only)")
    ap.add_argument("--backend", choices=("nccl", "gloo"), default="nccl",
                    help="torch.distributed backend: nccl = RCCL over xGMI (one GPU per rank); gloo stages the collectives through "
                         "host memory and lets several ranks share one GPU (GCI_DIST_DEVICE): correctness runs only")
    ap.add_argument("--scaling", choices=("strong", "weak"), default="strong",
                    help="N > 1, genome workload.  strong (default): the SAME CHM13 workload as N = 1, its contigs dealt to the ranks "
                         "(longest first onto the least loaded), every rank filters the records of its contigs, the join is sharded by "
                         "name hash (two all-to-alls per file + one of intervals) -- the split that replaces the reference's "
                         "Pool(threads).map(read_sam) (GCI.py:257-270).  weak: every rank one CHM13-sized haplotype of its own")
    ap.add_argument("--shared-names", type=float, default=0.0,
                    help="N > 1, genome workload: this fraction of every rank's second file carries read names of the NEXT rank "
                         "(a read aligned to contigs of two ranks): the name check then finds conflicts and every step takes "
                         "the replicated join.  Default 0: names are unique to their rank")
    ap.add_argument("--verify-oracle", action="store_true",
                    help="N > 1, small --scale only: rank 0 collects every rank's input and three of its contigs' tracks and holds "
                         "them against the oracle run over the whole (all ranks') files")
    ap.add_argument("--count-in-build", action="store_true",
                    help="A/B: the join does not do the depth build's counting pass (gci_name_join instead of gci_name_join_count)")
    ap.add_argument("--force-exchange", action="store_true",
                    help="run the multi-GPU name check / exchange and the all-reduce even with one rank (self-test)")
    ap.add_argument("--force-sharded", action="store_true",
                    help="one rank: run the STRONG-scaling step of N > 1 anyway -- the join sharded by name hash over a world of one "
                         "(every all-to-all and the all-reduce execute, through RCCL with --backend nccl): the self-test of the path an "
                         "8-GPU node takes, on one GPU")
    ap.add_argument("--force-replicated", action="store_true",
                    help="with the exchange: always take the replicated-join fallback (all-gather of records + names)")
    return ap.parse_args()


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


VIA_HOST = False      # --backend gloo: collectives on host copies


def all_reduce(t, op):
    import torch.distributed as dist
    if not VIA_HOST:
        dist.all_reduce(t, op=op)
        return
    h = t.cpu()
    dist.all_reduce(h, op=op)
    t.copy_(h)


class Workload:
    """Per-rank resident inputs + preallocated outputs for one step over F input files."""

    def __init__(self, eng, rank, world, contigs, files, heads, exchange=False, replicated=False, name="", algo=None, k1="pages",
                 sharded=False):
        """contigs: the (name, length) table of the WHOLE run (all ranks); `files`: this rank's slice of every input
        file as (stream uint8, offsets uint64, name_bytes) host arrays whose BAM header lists `contigs`;
        the rank owns the contigs `own` = indices into `contigs` (set by the caller through self.own before layout)."""
        import torch
        self.torch = torch
        self.eng, self.rank, self.world = eng, rank, world
        self.contigs, self.heads, self.name = contigs, heads, name
        self.algo = algo or {}
        self.sharded = bool(sharded)                     # strong scaling: the join sharded by name hash (shard.ShardedJoin)
        self.exchange = (exchange or world > 1) and not self.sharded
        self.force_replicated = replicated
        self.replicated_steps = 0
        dev = eng.device
        self.d_bam, self.d_off, self.n_rec, self.stream_bytes, self.name_bytes = [], [], [], [], []
        for stream, offs, nb in files:
            self.d_bam.append(eng.to_device(stream))
            self.d_off.append(eng.to_device(offs))
            self.n_rec.append(int(offs.shape[0]))
            self.stream_bytes.append(int(stream.shape[0]))
            self.name_bytes.append(int(nb))
        self.n_files = len(files)
        self.total_rec = sum(self.n_rec)
        # record pages: made on the device from the uploaded stream, once (ingestion, outside the timed step: it is the
        # last step of the walk over the inflated file); the stream itself is dropped
        if k1 != "pages" and heads:
            sys.exit("bench.py --k1 stream reads the whole inflated stream (--workload chr19 without --heads); a heads stream is "
                     "read through its record pages")
        self.k1 = k1
        self.pages = None
        # The inflated record bytes (the heads stream, or the whole stream) and the record offsets STAY resident: they are the
        # step's input ("inflated record bytes resident"), and laying them out as record pages -- the last step of the
        # record walk, gci_bam_pages_size / _write -- is done INSIDE every step (round 6; rounds 3 - 5 made the pages once, in front
        # of the clock).  pages_in_step = False keeps that older window (`n1_*` in the bench line: kernels over resident pages).
        self.pages_in_step = False
        self.keys_to_host = False
        self.keys_host = None
        self.in_stream, self.in_off, self.in_bytes = list(self.d_bam), list(self.d_off), list(self.stream_bytes)
        if k1 == "pages":
            self.pages = [eng.bam_pages(b, o, not heads) for b, o in zip(self.d_bam, self.d_off)]
            self.d_bam = [p.buf for p in self.pages]
            self.d_off = [torch.empty(max(n, 1), dtype=torch.int64, device=dev) for n in self.n_rec]   # K1 writes the name offsets
            self.stream_bytes = [int(p.buf.shape[0]) for p in self.pages]
            self.pages_in_step = True
        else:
            self.in_stream = self.in_off = None
        self.name_delta = 0 if k1 == "pages" else 36

    def layout(self, own):
        """own: indices (into self.contigs) of the contigs this rank builds, in header order."""
        torch, eng, dev = self.torch, self.eng, self.eng.device
        self.own = list(own)
        self.own_lengths = [self.contigs[c][1] for c in self.own]
        eng.set_layout(self.own_lengths)
        nc = len(self.own)
        n_all = len(self.contigs)
        self.ref_sel = eng.to_device(np.arange(n_all, dtype=np.int32))      # K1: refID -> global contig index
        cmap = np.full(n_all, -1, dtype=np.int32)
        cmap[self.own] = np.arange(nc, dtype=np.int32)
        # the join maps global contig -> local track index; entries < 0 drop the interval (contigs of other ranks)
        self.contig_map = eng.to_device(cmap) if (self.exchange or nc != n_all) and not self.sharded else None
        self.recs = [torch.empty((max(n, 1), 32), dtype=torch.uint8, device=dev) for n in self.n_rec]
        self.rec_base = [0] * self.n_files
        self.track = eng.new_track()
        self.ivl = torch.empty((max(self.total_rec, 1) * (self.world if self.exchange else 1), 4), dtype=torch.int32, device=dev)
        self.count = torch.zeros(1, dtype=torch.int32, device=dev)
        self.keys = torch.empty(1 << 18, dtype=torch.int64, device=dev)
        self.nkeys = torch.zeros(1, dtype=torch.int32, device=dev)
        self.text_off = torch.zeros(nc + 1, dtype=torch.int64, device=dev)
        # [per-contig sums of depth (zeroed and written by every build) | bases of this rank | cross-rank name conflicts]
        self.totals = torch.zeros(nc + 2, dtype=torch.int64, device=dev)
        self.sums = self.totals[0:nc]
        self.totals[nc] = int(sum(self.own_lengths))
        self.sum_total = torch.zeros(1, dtype=torch.int64, device=dev)    # sharded mode: this rank's sum of depth -> all-reduced
        self.status = torch.zeros(self.n_files + 1, dtype=torch.int64, device=dev)
        self.text = None
        from gci_amd._lib import BuildOpts
        o = BuildOpts()
        o.flank, o.want_text, o.counted = FLANK, 1, 1          # counted: the join has done the build's counting pass
        o.d_contig_text_off, o.d_sums = self.text_off.data_ptr(), self.sums.data_ptr()
        o.d_n_keys, o.d_keys, o.key_cap = self.nkeys.data_ptr(), self.keys.data_ptr(), int(self.keys.shape[0])
        o.issue_flank, o.lo, o.hi = FLANK, -1.0, 0.0
        self.opts = o
        if self.exchange:
            self._setup_exchange()
        if self.sharded:
            from gci_amd import shard
            self.opts.counted = 0                        # the intervals of the build arrive through an all-to-all, uncounted
            self.sj = shard.ShardedJoin(eng, self.n_rec, self.owner_of_contig, eng.device, via_host=VIA_HOST)
        return self

    # ---- multi-GPU: exact name check every step; the replicated join only when a name is shared -------------------
    def _setup_exchange(self):
        from gci_amd import shard
        nc = len(self.own)
        self.ex = [shard.RecordExchange(n, nb, self.eng.device, via_host=VIA_HOST) for n, nb in zip(self.n_rec, self.name_bytes)]
        self.rec_base = [e.rec_idx_base for e in self.ex]
        self.recs = [e.send_recs for e in self.ex]            # K1 writes straight into the send buffers
        self.ivl = self.torch.empty((sum(self.world * e.max_n for e in self.ex), 4), dtype=self.torch.int32,
                                    device=self.eng.device)
        # Both collectives of the step are synchronous ops on the hardware queue of the step's own kernels (tried: the
        # name check on a side stream, the all-reduce as an async op -- 20 - 45 us slower per step on this chip).
        self.check_names = shard.NameCheck(self.total_rec, self.eng.device, self.eng.hash_bucket, self.eng.hash_conflicts,
                                           alternate=True, via_host=VIA_HOST)
        self.check_names.n_conf = self.totals[nc + 1:nc + 2].view(self.torch.int32)[0:1]   # counted straight into the totals

    def step_records(self):
        """The record side of a step alone: K1 per file and the join (with the build's counting pass)."""
        self.step(records_only=True)

    def step(self, records_only=False):
        eng, lib, ctx = self.eng, self.eng.lib, self.eng.ctx
        from gci_amd._lib import JoinFile
        chk = eng._chk
        k1 = lib.gci_bam_filter                          # --k1 stream: the round-1 / 2 kernel over the whole inflated stream
        F = self.n_files
        for f in range(F):
            if self.pages is not None:
                pg = self.pages[f]
                if self.pages_in_step:
                    # record pages from the resident record bytes, into the buffer the first build sized (the same input: the same size)
                    h = (ctypes.c_uint64 * 3)()
                    chk(lib.gci_bam_pages_size(ctx, _p(self.in_stream[f]), self.in_bytes[f], _p(self.in_off[f]), self.n_rec[f], int(not self.heads),
                                               pg.page_bytes, h), "gci_bam_pages_size")
                    if int(h[1]) != self.stream_bytes[f] or int(h[0]) != pg.n_pages:
                        raise RuntimeError("bench: the record pages of file %d changed size between steps" % f)
                    chk(lib.gci_bam_pages_write(ctx, _p(self.in_stream[f]), self.in_bytes[f], _p(self.in_off[f]), self.n_rec[f], int(not self.heads),
                                                _p(pg.buf), int(h[1])), "gci_bam_pages_write")
                chk(lib.gci_bam_filter_pages(ctx, _p(pg.buf), self.stream_bytes[f], pg.page_bytes, pg.n_pages, self.n_rec[f],
                                             _p(self.ref_sel), len(self.contigs), FILTER[0], FILTER[1], FILTER[2], FILTER[3],
                                             self.rec_base[f], _p(self.recs[f]), _p(self.d_off[f]), _p(self.status[f:f + 1])),
                    "gci_bam_filter_pages")
                continue
            chk(k1(ctx, _p(self.d_bam[f]), self.stream_bytes[f], _p(self.d_off[f]), self.n_rec[f], _p(self.ref_sel),
                   len(self.contigs), FILTER[0], FILTER[1], FILTER[2], FILTER[3], self.rec_base[f], _p(self.recs[f]),
                   _p(self.status[f:f + 1])), "gci_bam_filter")
        if self.sharded:
            return self._step_sharded()
        jf = (JoinFile * F)()
        if self.exchange and self.force_replicated:
            # a name occurs on two ranks: replicate records + names and join everything everywhere
            self.replicated_steps += 1
            self._g = []
            for f in range(F):
                ex = self.ex[f]
                loc = (JoinFile * 1)()
                loc[0].d_recs, loc[0].n_recs, loc[0].name_delta = self.recs[f].data_ptr(), self.n_rec[f], self.name_delta
                loc[0].d_name_base, loc[0].d_name_off = self.d_bam[f].data_ptr(), self.d_off[f].data_ptr()
                chk(lib.gci_pack_names(ctx, loc, _p(ex.send_names), ex.name_cap, _p(ex.send_off)), "gci_pack_names")
                self.recs[f][:, 29] &= 3                       # the names travel packed: no longer GCI_REC_NAME16
                g = ex.gather()
                self._g.append(g)                                # keep the index tensors alive until the join ran
                jf[f].d_recs, jf[f].n_recs, jf[f].name_delta = g.recs.data_ptr(), self.world * g.max_n, 0
                jf[f].d_name_base, jf[f].d_name_off = g.names.data_ptr(), g.name_index.data_ptr()
        else:
            if self.exchange:
                # Exact cross-rank name test (hash all-to-all, 8 bytes per record), enqueued without a host sync: the
                # step goes on SPECULATIVELY with the local join; check() reads the accumulated verdict after the
                # timed region and main() redoes everything with the replicated join if any step saw a conflict.
                self.check_names.enqueue_files([self.recs[f][:self.n_rec[f]] for f in range(F)])
            for f in range(F):
                jf[f].d_recs, jf[f].n_recs, jf[f].name_delta = self.recs[f].data_ptr(), self.n_rec[f], self.name_delta
                jf[f].d_name_base, jf[f].d_name_off = self.d_bam[f].data_ptr(), self.d_off[f].data_ptr()
        # the join also does the counting pass of the depth build over the intervals it emits (gci_name_join_count)
        if self.opts.counted:
            chk(lib.gci_name_join_count(ctx, jf, F, OVLP, _p(self.contig_map), _p(self.ivl), int(self.ivl.shape[0]),
                                        _p(self.count), _p(self.status[F:F + 1]), int(self.opts.flank)), "gci_name_join_count")
        else:
            chk(lib.gci_name_join(ctx, jf, F, OVLP, _p(self.contig_map), _p(self.ivl), int(self.ivl.shape[0]),
                                  _p(self.count), _p(self.status[F:F + 1])), "gci_name_join")
        if records_only:
            return
        # fused build: depth + per-contig sums + text byte offsets + issue-run boundaries from one pass over
        # the per-tile event buckets (no HBM re-read of the track), then depth + decimal text in the second
        o = self.opts
        chk(lib.gci_depth_build_begin(ctx, _p(self.ivl), _p(self.count), int(self.ivl.shape[0]), ctypes.byref(o)),
            "gci_depth_build_begin")
        if self.text is None:                              # first (warm-up) call sizes the text buffer
            total = int(self.text_off[len(self.own)].item())
            self.text = self.torch.empty(total + (total >> 4) + 4096, dtype=self.torch.uint8, device=eng.device)
        chk(lib.gci_depth_build_finish(ctx, _p(self.track), _p(self.text), int(self.text.shape[0])),
            "gci_depth_build_finish")
        if self.exchange:
            import torch.distributed as dist
            # ONE integer all-reduce per step, in place: the sums of depth (global mean depth = their total / bases)
            all_reduce(self.sums, dist.ReduceOp.SUM)
        self._keys_home()

    def _keys_home(self):
        """The issue-run boundaries of the step on the HOST ("... and issue intervals on host"): count, then the keys."""
        if self.keys_to_host:
            n = int(self.nkeys.item())
            self.keys_host = self.keys[:min(n, int(self.keys.shape[0]))].cpu()

    def _step_sharded(self):
        """Strong scaling: records of this rank's contigs -> by name hash to the rank that owns the name (two all-to-alls per
        file) -> join of the names owned here -> intervals to the owners of their contigs (one all-to-all) -> build."""
        import torch.distributed as dist
        from gci_amd.device import JoinInput
        eng, lib, ctx, chk, sj = self.eng, self.eng.lib, self.eng.ctx, self.eng._chk, self.sj
        inputs = sj.exchange_files([JoinInput(self.recs[f][:self.n_rec[f]], self.d_bam[f], self.d_off[f], self.name_delta)
                                    for f in range(self.n_files)])
        ivl, n_slots = sj.join(inputs, OVLP)
        self.ivl_in_build, self.count_in_build = ivl, None
        o = self.opts
        chk(lib.gci_depth_build_begin(ctx, _p(ivl), None, n_slots, ctypes.byref(o)), "gci_depth_build_begin")
        if self.text is None:                              # first (warm-up) call sizes the text buffer
            total = int(self.text_off[len(self.own)].item())
            self.text = self.torch.empty(total + (total >> 4) + 4096, dtype=self.torch.uint8, device=eng.device)
        chk(lib.gci_depth_build_finish(ctx, _p(self.track), _p(self.text), int(self.text.shape[0])), "gci_depth_build_finish")
        self.torch.sum(self.sums, dim=0, keepdim=True, out=self.sum_total)
        all_reduce(self.sum_total, dist.ReduceOp.SUM)      # the genome-wide sum of depth: ONE integer all-reduce per step
        self._keys_home()

    def check(self):
        """Record-level status of the last step + output capacities; False when a name is shared between ranks."""
        from gci_amd._lib import GciError
        what = ["gci_bam_filter[%d]" % f for f in range(self.n_files)] + ["gci_name_join"]
        for w, name in zip(self.status.cpu().numpy().view(np.uint64).tolist(), what):
            rec = ctypes.c_uint32(0)
            st = self.eng.lib.gci_decode_status(w, ctypes.byref(rec))
            if st != 0:
                raise GciError(st, "%s failed on record %d" % (name, rec.value))
        if self.sharded:
            def decode(w, name):
                rec = ctypes.c_uint32(0)
                st = self.eng.lib.gci_decode_status(w, ctypes.byref(rec))
                if st != 0:
                    raise GciError(st, "%s: status %d (record %d) on rank %d" % (name, st, rec.value, self.rank))
            self.sj.check(decode)
            if int(self.nkeys.item()) > self.keys.shape[0]:
                raise GciError(-8, "bench output buffers too small")
            return True
        if int(self.nkeys.item()) > self.keys.shape[0] or int(self.count.item()) > self.ivl.shape[0]:
            raise GciError(-8, "bench output buffers too small")
        if self.exchange and not self.force_replicated:
            import torch.distributed as dist
            nc = len(self.own)
            conf = self.totals[nc + 1:nc + 2].clone()            # every rank's count of hashes seen from two ranks
            all_reduce(conf, dist.ReduceOp.SUM)
            if int(conf.item()) > 0:
                return False      # a query name is shared between ranks: the speculative local joins were not exact
        return True

    # ---- algorithmic bytes of one step ------------------------------------------
    def step_algorithmic_bytes(self):
        K = int(self.count.item())
        L = int(sum(self.own_lengths))
        T = int(self.text_off[len(self.own)].item())
        k1 = int(self.algo.get("k1_bytes", 0)) + 32 * self.total_rec
        join = 16 * self.n_rec[0] + 48 * sum(self.n_rec[1:]) + 16 * K
        build = 28 * K + 4 * L + T
        # record pages inside the step: the record bytes read once, the pages written once
        pages = (int(sum(self.in_bytes)) + int(sum(self.stream_bytes))) if (self.pages_in_step and self.pages is not None) else 0
        return {"record_pages": pages, "k1_record_filter": k1, "name_join": join, "depth_build_text": build, "total": pages + k1 + join + build,
                "intervals": K, "bases": L, "text_bytes": T}


class TwoTypeWorkload:
    """configs[3] / configs[4] on one GPU: what GCI() does with `--hifi` AND `--nano` (GCI.py:1007-1026).  A step =
    per read type [PAF filter on the device (K2) + paged record filter (K1) -> join (PAF first, GCI.py:272) -> fused depth
    build with sums and, when no gap mask follows, the issue runs] -> gap masks (GCI.py:315-329) -> per-base max of the two
    tracks (GCI.py:350) -> issue scans of the three tracks (GCI.py:356-419) -> `-R` regions scans (GCI.py:610-657)."""

    def __init__(self, eng, inp, name):
        import torch
        self.torch, self.eng, self.inp, self.name = torch, eng, inp, name
        self.rank, self.world, self.sharded, self.exchange, self.replicated_steps = 0, 1, False, False, 0
        self.contigs = inp.contigs
        names, lens = inp.names, inp.lengths
        eng.set_layout(lens)
        self.own, self.own_lengths = list(range(len(names))), lens
        self.ref_sel = eng.to_device(np.arange(len(names), dtype=np.int32))
        dev = eng.device
        self.types = []
        for t in (inp.hifi, inp.nano):
            d_s, d_o = eng.to_device(t.bam.stream), eng.to_device(t.bam.offsets)
            pages = eng.bam_pages(d_s, d_o, False)
            del d_s, d_o
            n = int(t.bam.offsets.shape[0])
            d = dict(pages=pages, n_rec=n, recs=torch.empty((max(n, 1), 32), dtype=torch.uint8, device=dev),
                     name_off=torch.empty(max(n, 1), dtype=torch.int64, device=dev),
                     paf=eng.to_device(t.paf) if t.paf is not None else None,
                     paf_end=np.asarray([t.paf.shape[0]], dtype=np.uint64) if t.paf is not None else None,
                     track=eng.new_track(), ivl=torch.empty((2 * max(n, 1) + 16, 4), dtype=torch.int32, device=dev),
                     sums=torch.zeros(len(names), dtype=torch.int64, device=dev),
                     count=torch.zeros(1, dtype=torch.int32, device=dev), k1_bytes=t.bam.k1_bytes)
            self.types.append(d)
        self.two = eng.new_track()
        self.gaps = None
        if inp.gaps:
            g = [(names.index(c), a, b, 0) for c, segs in inp.gaps.items() for a, b in segs]
            self.gaps = np.asarray(g, dtype=np.int32)           # (host rows: gci_two_type_tail takes the N runs from the host)
        self.tail_keys = torch.empty((3, 1 << 16), dtype=torch.int64, device=dev)
        self.tail_n = torch.zeros(3, dtype=torch.int32, device=dev)
        offs = eng.offsets
        self.windows = [(offs[names.index(c)] + a, offs[names.index(c)] + b) for c, a, b in inp.regions]
        self.aligned_bases = inp.aligned_bases
        self.stream_bytes = [int(d["pages"].buf.shape[0]) for d in self.types]
        self.n_rec = [d["n_rec"] for d in self.types]
        self.n_files = 2 + sum(1 for d in self.types if d["paf"] is not None)
        self.heads, self.pages = True, [d["pages"] for d in self.types]
        self.last = {}

    def step(self):
        eng = self.eng
        from gci_amd.device import JoinInput
        names = self.inp.names
        runs = []
        for d in self.types:
            inputs = eng.paf_filter_text(d["paf"], d["paf_end"], names, FILTER[0], FILTER[1], FILTER[3]) if d["paf"] is not None else []
            recs, noff = eng.bam_filter_pages(d["pages"], self.ref_sel, *FILTER, out=d["recs"], name_off=d["name_off"], check=False)
            d["status"] = eng._status.clone()
            inputs.append(JoinInput(recs, d["pages"].buf, noff, 0))
            ivl, cnt = eng.name_join(inputs, OVLP, out=d["ivl"], count=d["count"], check=False, count_flank=FLANK)
            d["jstatus"] = eng._status.clone()
            d["fused"] = eng.depth_build_fused(ivl, cnt, FLANK, d["track"], want_text=False, want_sums=d["sums"], issue=None, counted=True)
        # the tail in one pass (gci_two_type_tail): N-run masks of both tracks, their maximum, the issue runs of all three
        eng.two_type_tail(self.types[0]["track"], self.types[1]["track"], self.gaps, -1.0, 0.0, FLANK, out=self.two,
                          keys=self.tail_keys, n_keys=self.tail_n, read=False)
        n = self.tail_n.cpu().numpy()                           # (the one read of the tail: three counters, then the few keys)
        if int(n.max()) > int(self.tail_keys.shape[1]):
            raise RuntimeError("bench: issue-run key buffer too small")
        hk = self.tail_keys.cpu().numpy().view(np.uint64)
        runs = [eng._keys_to_runs(hk[x, :int(n[x])], len(names)) for x in range(3)]
        reg = [eng.issue_scan_windows(t, self.windows, -1.0, 0.0) for t in (self.types[0]["track"], self.types[1]["track"], self.two)] \
            if self.windows else None
        self.last = dict(runs=runs, regions=reg)

    def check(self):
        from gci_amd._lib import GciError
        for d in self.types:
            for w, what in ((d["status"], "gci_bam_filter_pages"), (d["jstatus"], "gci_name_join")):
                rec = ctypes.c_uint32(0)
                st = self.eng.lib.gci_decode_status(int(w.item()) & ((1 << 64) - 1), ctypes.byref(rec))
                if st != 0:
                    raise GciError(st, "%s failed on record %d" % (what, rec.value))
            if int(d["count"].item()) > d["ivl"].shape[0]:
                raise GciError(-8, "bench output buffers too small")
        return True

    def step_algorithmic_bytes(self):
        L = int(sum(self.own_lengths))
        K = [int(d["count"].item()) for d in self.types]
        k1 = sum(d["k1_bytes"] + 32 * d["n_rec"] for d in self.types)
        paf = sum(int(d["paf"].shape[0]) for d in self.types if d["paf"] is not None)
        join = sum(48 * d["n_rec"] + 16 * k for d, k in zip(self.types, K))
        build = sum(28 * k + 4 * L for k in K)
        rest = 12 * L                                            # the tail in one pass: two tracks read, their maximum written
        return {"k1_record_filter": k1, "paf_text": paf, "name_join": join, "depth_build": build, "max_and_scans": rest,
                "total": k1 + paf + join + build + rest, "intervals": K, "bases": L, "text_bytes": 0}


def parity_two_type(w, chosen):
    """The timed result of a two-type workload against the oracle on whole contigs at full size: per read type the exact
    restriction of filter() to those contigs (oracle.file1_on_contigs_mixed), the gap mask, the max, the three issue lists."""
    from oracle import gci_oracle as O
    from gci_amd import pipeline
    O.build()
    inp = w.inp
    names = inp.names
    chosen = [c for c in chosen if c in names]
    tl = {c: inp.lengths[names.index(c)] for c in chosen}
    ok = True
    tracks = []
    for t, d in zip((inp.hifi, inp.nano), w.types):
        file1 = O.file1_on_contigs_mixed([t.paf.tobytes()] if t.paf is not None else [], [(t.bam.stream, t.bam.offsets, names)], names, chosen,
                                         FILTER[0], FILTER[1], FILTER[2], FILTER[3], OVLP, heads=True)
        depths = O.depth_build(file1, tl, FLANK)
        O.merge_gaps_depths(depths, {c: v for c, v in inp.gaps.items() if c in tl} or None)
        tracks.append(depths)
    two = O.max2(tracks[0], tracks[1])
    for k, (want, got_track) in enumerate(zip(tracks + [two], [w.types[0]["track"], w.types[1]["track"], w.two])):
        bed = O.collapse_depth_range(want, -1, 0, FLANK, 0)
        for c in chosen:
            ci = names.index(c)
            o, L = w.eng.offsets[ci], tl[c]
            ok = ok and np.array_equal(got_track[o:o + L].cpu().numpy(), want[c])
            a, b = pipeline._slice_bound(FLANK, L), pipeline._slice_bound(L - FLANK, L)
            ok = ok and pipeline._issues_from_runs(w.last["runs"][k][ci], max(0, b - a), L, FLANK, 0) == bed[c]
        if w.last["regions"] is not None:
            for (c, a, b), got in zip(inp.regions, w.last["regions"][k]):
                if c in tl:
                    ok = ok and [tuple(x) for x in (np.asarray(got) + a).tolist()] == O.collapse_contig(want[c][a:b], -1, 0, 0, a)
    return bool(ok), chosen


def plot_front_end_number(w, window_size=50000, dmax=4.0):
    """N3 at the size of configs[4] ("`-p` windowed depth"): what plot_depth() computes before it draws (GCI.py:837-868, 660-705)
    -- per read type the global mean depth of the masked track (one read of it, gci_depth_sum) and the zero-delimited sliding-
    window means of EVERY contig (the zero runs: one gci_issue_scan_windows over all contigs; the window sums: one gci_range_sums)
    -- timed on the tracks the last step left, and held against the oracle's base-by-base loop on 2 Mb stretches (one of them
    behind 2^31 elements of the track)."""
    import torch
    from gci_amd import pipeline
    from oracle import gci_oracle as O
    inp, eng = w.inp, w.eng
    names, tl = inp.names, dict(inp.contigs)
    tracks = [pipeline.DepthTracks(eng, tl, d["track"]) for d in w.types]
    items = [(t, 0, None) for t in names]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    means, series = [], []
    for tr in tracks:
        means.append(tr.mean())
        series.append(pipeline.sliding_window_average_depth_many(tr, items, window_size, means[-1] * dmax))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    offs = eng.offsets
    far = max(range(len(names)), key=lambda c: offs[c])
    picks = [(names[far], 1_000_000, 3_000_000), (names[0], 0, 2_000_000), (names[len(names) // 2], inp.lengths[len(names) // 2] - 2_000_000, None)]
    for c, segs in list(inp.gaps.items())[:1]:                       # a stretch with an N run in it
        a = max(0, segs[0][0] - 1_000_000)
        picks.append((c, a, a + 2_000_000))
    ok = True
    for k, tr in enumerate(tracks):
        got = pipeline.sliding_window_average_depth_many(tr, picks, window_size, means[k] * dmax)
        for (c, a, b), (pos, val, _) in zip(picks, got):
            host = tr[c]
            b2 = len(host) if b is None else b
            want_pos, want_val = O.sliding_window_average_depth(host[a:b2], window_size, means[k] * dmax, a)
            ok = ok and pos == want_pos and np.array_equal(val, want_val)
            del host
        ok = ok and means[k] == float(sum(int(tr.sums()[i]) for i in range(len(names)))) / float(sum(inp.lengths))
    n_vals = sum(len(p) for s in series for p, _, _ in s)
    L = sum(inp.lengths)
    return {"seconds": dt, "tracks": len(tracks), "contigs": len(names), "window_size": window_size, "values": n_vals,
            "algorithmic_bytes": 3 * 4 * L * len(tracks), "hbm_frac": 3 * 4 * L * len(tracks) / dt / 1e9 / HBM_PEAK_GBS,
            "parity_vs_oracle": bool(ok), "parity_stretches": ["%s:%d-%s" % (c, a, b if b is not None else "end") for c, a, b in picks],
            "note": "per read type: mean depth (gci_depth_sum), zero runs of all contigs (one gci_issue_scan_windows), window sums (one "
                    "gci_range_sums); three reads of each track"}


def make_two_type_workload(eng_factory, rank, world, args, exchange, replicated):
    from gci_amd import workloads
    if world > 1:
        sys.exit("bench.py --workload %s runs on one GPU (configs[3] / configs[4] at full size fit its 288 GB)" % args.workload)
    config = 4 if args.workload == "genome4" else 5
    cov_h = args.coverage if args.workload == "genome4" else (args.coverage if args.coverage != 40.0 else 100.0)
    cov_o = args.coverage_ont if args.coverage_ont is not None else (40.0 if config == 4 else 20.0)
    inp = workloads.genome_two_type(config, args.scale, cov_h, cov_o, verbose=True)
    eng = eng_factory()
    name = ("CHM13 whole genome (25 contigs, %d bp), --hifi %gx + --nano %gx, per read type one BAM (record pages) + one PAF" if config == 4 else
            "diploid mat + pat (46 contigs, %d bp), --hifi %gx + --nano %gx, one BAM (record pages) per read type, 20 N gaps, -R regions") % (
                sum(inp.lengths), cov_h, cov_o)
    w = TwoTypeWorkload(eng, inp, name + ": 2 x (filters -> join -> depth build) -> gap mask -> max -> 3 issue scans")
    return eng, w


def smi_sample():
    """What the box says about itself (rocm-smi; none of it is in this program's hands): shader / memory clocks, power, temperatures
    of GPU 0.  Taken before and after the timed steps so that a slow box can be told from a slow kernel."""
    import subprocess
    try:
        r = subprocess.run(["rocm-smi", "-d", "0", "--showclocks", "--showpower", "--showtemp", "--json"], capture_output=True, text=True, timeout=20)
        d = json.loads(r.stdout)
        c = d.get("card0") or next(iter(d.values()))
        keep = {}
        for k, v in c.items():
            kl = k.lower()
            if any(t in kl for t in ("sclk", "mclk", "fclk", "power", "temperature")) and "level" not in kl.replace("clock level", ""):
                keep[k] = v
            elif "clock level" in kl and any(t in kl for t in ("sclk", "mclk")):
                keep[k] = v
        return keep or None
    except Exception:
        return None


def fill_ceiling_gbs(eng, nbytes, reps=6):
    """The rate this box, in this process, writes a buffer of the size of the dominant kernel's output with a plain fill -- the best of
    hipMemsetAsync (through the library) and torch's fill kernels: nothing read, 16-byte stores -- the ceiling that kernel can be held
    against when two boxes differ.  (A fill slows with its footprint on this chip: 6.6 TB/s over 4 GB, 5.4 - 6.0 over 22 GB,
    profiles/r05v_fill_methods.txt -- so the ceiling is taken at the kernel's own size.)"""
    import torch
    nbytes = int(min(nbytes, 24e9)) // 16 * 16
    best = 0.0
    with torch.cuda.stream(eng.stream):
        x = torch.empty(nbytes, dtype=torch.uint8, device=eng.device)
        methods = (lambda: eng.lib.gci_memset(eng.ctx, ctypes.c_void_p(x.data_ptr()), 0, nbytes), lambda: x.zero_(), lambda: x.view(torch.int32).fill_(7))
        for fn in methods:
            for _ in range(2):
                fn()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(eng.stream)
            for _ in range(reps):
                fn()
            b.record(eng.stream)
            torch.cuda.synchronize()
            best = max(best, nbytes / (a.elapsed_time(b) / reps * 1e-3) / 1e9)
    del x
    torch.cuda.empty_cache()
    return best


def genome_dual_once(args, rank, world, base):
    """The genome workload of this run.  At N > 1 every rank needs the same two files: rank 0 generates them ONCE on all of the host's
    cores and leaves them as .npy files in the directory the launcher made (GCI_BENCH_SHARED, tmpfs); the other ranks map them.
    Without such a directory (a launch by hand under torch.distributed.run) every rank generates with its share of the cores."""
    from gci_amd import workloads
    shared = os.environ.get("GCI_BENCH_SHARED")
    if world == 1 or not shared or not os.path.isdir(shared):
        return workloads.genome_dual(args.scale, args.coverage, contigs=base, verbose=(rank == 0), kind=args.reads,
                                     procs=max(1, workloads_default_procs() // max(1, world)))
    done = os.path.join(shared, "genome_dual.done")
    if rank == 0:
        inp = workloads.genome_dual(args.scale, args.coverage, contigs=base, verbose=True, kind=args.reads, procs=workloads_default_procs())
        meta = {"contigs": [[n, int(l)] for n, l in inp.contigs], "files": []}
        for i, f in enumerate(inp.files):
            np.save(os.path.join(shared, "f%d_stream.npy" % i), f.stream)
            np.save(os.path.join(shared, "f%d_offsets.npy" % i), f.offsets)
            np.save(os.path.join(shared, "f%d_per_contig.npy" % i), f.aligned_per_contig)
            meta["files"].append({"aligned_bases": int(f.aligned_bases), "k1_bytes": int(f.k1_bytes), "name_bytes": int(f.name_bytes)})
        with open(done + ".tmp", "w") as fh:
            json.dump(meta, fh)
        os.rename(done + ".tmp", done)
        return inp
    t0 = time.time()
    while not os.path.exists(done):
        if time.time() - t0 > 3600:
            sys.exit("bench: rank %d waited an hour for rank 0's workload in %s" % (rank, shared))
        time.sleep(0.2)
    meta = json.load(open(done))
    files = []
    for i, m in enumerate(meta["files"]):
        files.append(workloads.AlignmentFile(np.load(os.path.join(shared, "f%d_stream.npy" % i), mmap_mode="r"),
                                             np.load(os.path.join(shared, "f%d_offsets.npy" % i), mmap_mode="r"),
                                             m["aligned_bases"], m["k1_bytes"], m["name_bytes"],
                                             np.load(os.path.join(shared, "f%d_per_contig.npy" % i))))
    return workloads.GenomeInput(tuple((n, int(l)) for n, l in meta["contigs"]), files)


def make_genome_workload(eng_factory, rank, world, args, exchange, replicated):
    """configs[2]; at N>1 haplotype `rank` of an N-haplotype assembly (contig names h<r>_chrN, read names unique to the
    rank).  The host arrays are generated BEFORE the HIP context exists (worker processes are forked)."""
    from gci_amd import synth, workloads
    base = synth.CHM13
    inp = genome_dual_once(args, rank, world, base)
    nper = len(inp.contigs)
    contig_owner = None
    if (world > 1 or args.force_sharded) and args.scaling == "strong":
        # the same genome as N = 1: contigs dealt to the ranks, every rank keeps the records of ITS contigs (what its
        # index-driven ingestion would read of each file: pipeline.bam_records_of_contigs)
        from gci_amd import shard
        from gci_amd.formats import bam as bamfmt
        contig_owner = shard.lpt_assign(inp.lengths, world)
        own = [c for c, o in enumerate(contig_owner) if o == rank]
        mine = np.zeros(nper, dtype=bool)
        mine[own] = True
        for fobj in inp.files:
            first = bamfmt.parse_header(fobj.stream).first_record
            offs = fobj.offsets.astype(np.int64)
            ends = np.concatenate([offs[1:], [fobj.stream.shape[0]]])
            ref = fobj.stream[(offs[:, None] + np.arange(4, 8)[None, :])].copy().view("<i4").reshape(-1)
            keep = (ref >= 0) & mine[np.clip(ref, 0, nper - 1)]
            # records are sorted by contig: the kept ones are a few contiguous runs
            edges = np.flatnonzero(np.diff(np.concatenate([[0], keep.astype(np.int8), [0]])))
            parts, new_offs, size = [fobj.stream[:first]], [], first
            for a, b in zip(edges[0::2], edges[1::2]):
                lo, hi = int(offs[a]), int(ends[b - 1])
                parts.append(fobj.stream[lo:hi])
                new_offs.append(offs[a:b] - lo + size)
                size += hi - lo
            fobj.aligned_bases = int(fobj.aligned_per_contig[own].sum())
            fobj.k1_bytes = int(fobj.k1_bytes * (keep.sum() / max(1, keep.shape[0])))      # (share of the records: an estimate)
            fobj.stream = np.concatenate(parts)
            fobj.offsets = (np.concatenate(new_offs) if new_offs else np.zeros(0, np.int64)).astype(np.uint64)
        contigs = inp.contigs
    elif world > 1:
        # one header for the whole run: rank r's contigs are entries [r * nper, (r + 1) * nper); shift the refIDs
        from gci_amd.formats import bam as bamfmt
        all_contigs = tuple(("h%d_%s" % (r, n), l) for r in range(world) for n, l in inp.contigs)
        hdr = np.frombuffer(bamfmt.encode_header([n for n, _ in all_contigs], [l for _, l in all_contigs]), dtype=np.uint8)
        for fobj in inp.files:
            first = bamfmt.parse_header(fobj.stream).first_record
            body = fobj.stream[first:] if fobj.stream.flags.writeable else np.array(fobj.stream[first:])   # (a mapped file of rank 0's: copy)
            offs = fobj.offsets - np.uint64(first)
            ref = body[(offs[:, None] + np.arange(4, 8, dtype=np.uint64)[None, :]).astype(np.int64)].copy().view("<i4").reshape(-1)
            ref = np.where(ref >= 0, ref + rank * nper, ref).astype("<i4")
            body[(offs[:, None] + np.arange(4, 8, dtype=np.uint64)[None, :]).astype(np.int64)] = ref.view(np.uint8).reshape(-1, 4)
            # read names unique to the rank: the prefix "m64011_gNN/" becomes "m64RR_gNN/" with RR = the rank (same length)
            owner = np.full(offs.shape[0], rank, dtype=np.int64)
            if args.shared_names > 0 and fobj is inp.files[-1]:
                # ... except that this fraction of the LAST file's records are reads of the next rank (every rank generated
                # the same reads, so the name exists there): aligned to contigs of two ranks -> the join must drop them
                rng = np.random.Generator(np.random.PCG64(977 + rank))
                owner[rng.random(offs.shape[0]) < args.shared_names] = (rank + 1) % world
            body[(offs + np.uint64(36 + 5)).astype(np.int64)] = (ord("0") + owner % 10).astype(np.uint8)
            body[(offs + np.uint64(36 + 4)).astype(np.int64)] = (ord("0") + owner // 10 % 10).astype(np.uint8)
            fobj.stream = np.concatenate([hdr, body])
            fobj.offsets = offs + np.uint64(hdr.shape[0])
        contigs = all_contigs
        own = list(range(rank * nper, (rank + 1) * nper))
    else:
        contigs, own = inp.contigs, list(range(nper))
    full = None
    if contig_owner is not None and args.verify_oracle and rank == 0:
        full = workloads.genome_dual(args.scale, args.coverage, contigs=base, kind=args.reads,
                                     procs=max(1, workloads_default_procs() // max(1, world)))     # the undivided files, for the oracle
    eng = eng_factory()
    files = [(f.stream, f.offsets, f.name_bytes) for f in inp.files]
    w = Workload(eng, rank, world, contigs, files, heads=True, exchange=exchange, replicated=replicated,
                 name="CHM13 whole genome (%d contigs, %d bp)%s, HiFi %gx by two aligners (2 BAM files as heads streams, "
                      "-op join), filter x2 -> join -> depth -> issue scan -> depth text" % (
                          nper, sum(l for _, l in inp.contigs),
                          (" x %d haplotypes" % world if contig_owner is None else " over %d GPUs" % world) if world > 1 else "", args.coverage)
                      + ("" if args.reads == "hifi" else " [ONT reads]"),
                 algo={"k1_bytes": sum(f.k1_bytes for f in inp.files)}, k1=args.k1, sharded=contig_owner is not None)
    w.owner_of_contig = contig_owner
    w.full_input = full
    if rank == 0:
        print("bench: world %d, scaling %s, join %s" % (world, args.scaling if world > 1 else "-", "sharded by name hash" if w.sharded else
              "name check + local / replicated" if w.exchange else "local"), file=sys.stderr, flush=True)
    w.aligned_bases = inp.aligned_bases
    w.inp = inp if ((rank == 0 and world == 1) or args.verify_oracle) else None
    if w.inp is None:
        del inp
    return eng, w.layout(own)


def workloads_default_procs():
    from gci_amd import hostio
    return hostio.default_threads()


def make_chr19_workload(eng_factory, rank, world, args, exchange, replicated, eng=None):
    """configs[1]: rank r owns one chr19-sized contig of an N-contig assembly and the records of that contig."""
    from gci_amd import synth
    names = ["chr19"] if world == 1 else ["chr19_%d" % r for r in range(world)]
    contigs = tuple((n, args.contig_len) for n in names)
    rs = synth.simulate_reads(((names[rank], args.contig_len),), args.coverage, "hifi", seed=synth.seed_for(2, rank))
    aligned = rs.aligned_bases()
    name_bytes = int(np.char.str_len(rs.names).sum())
    from gci_amd.workloads import _k1_algorithmic_bytes
    k1b = _k1_algorithmic_bytes(rs)
    rs.ref_id[:] = rank
    rs.contigs = contigs
    stream, offs = synth.to_bam_stream(rs)
    host = (stream, offs) if rank == 0 and world == 1 else None
    if args.heads:                                          # through a BGZF file and the native host pipeline, as the CLI does
        import tempfile
        from gci_amd import hostio
        from gci_amd.formats import bam as bamfmt
        with tempfile.TemporaryDirectory(prefix="gci_bench_") as tmp:
            path = os.path.join(tmp, "r%d.bam" % rank)
            bamfmt.write_bam_stream(path, stream, level=1, threads=hostio.default_threads())
            with hostio.bam_heads(np.fromfile(path, dtype=np.uint8)) as hd:
                assert hd.offsets.shape[0] == len(rs)
                up = (hd.stream.copy(), hd.offsets.copy(), name_bytes)
    else:
        up = (stream, offs, name_bytes)
    eng = eng or eng_factory()
    w = Workload(eng, rank, world, contigs, [up], heads=args.heads, exchange=exchange, replicated=replicated,
                 name="CHM13 chr19 (%d bp) x %d contig(s), one %gx HiFi BAM (%s), filter -> join -> depth -> issue scan -> "
                      "depth text" % (args.contig_len, world, args.coverage,
                                      "heads stream" if args.heads else "whole inflated stream"),
                 algo={"k1_bytes": k1b}, k1=args.k1)
    w.aligned_bases = aligned
    w.host = host
    return eng, w.layout([rank])


# ---- CPU legs (oracle = test infrastructure; rank 0, N = 1 only) ----------------------------------------------------

def _track_contig(w, c):
    o = w.eng.offsets[c]
    return w.track[o:o + w.own_lengths[c]].cpu().numpy().astype(np.int64)


def parity_genome(w, chosen=("chr14", "chr22", "chrM")):
    """The timed result against the oracle on whole contigs at full size (exact: oracle.file1_on_contigs), plus
    the genome-wide sum of depth against the sum of the clipped join intervals."""
    from oracle import gci_oracle as O
    from gci_amd import pipeline
    O.build()
    inp = w.inp
    names = inp.names
    chosen = [c for c in chosen if c in names]
    bams = [(f.stream, f.offsets, names) for f in inp.files]
    file1 = O.file1_on_contigs(bams, names, chosen, *FILTER, OVLP, heads=True)
    tl = {c: inp.lengths[names.index(c)] for c in chosen}
    depths = O.depth_build(file1, tl, FLANK)
    bed = O.collapse_depth_range(depths, -1, 0, FLANK, 0)
    nk = int(w.nkeys.item())
    runs = w.eng._keys_to_runs(w.keys[:nk].cpu().numpy().view(np.uint64), len(names))
    toff = w.text_off.cpu().numpy()
    sums = w.sums.cpu().numpy()
    ok = True
    for c in chosen:
        ci = names.index(c)
        L = tl[c]
        got = _track_contig(w, ci)
        ok = ok and np.array_equal(got, depths[c])
        a, b = pipeline._slice_bound(FLANK, L), pipeline._slice_bound(L - FLANK, L)
        ok = ok and pipeline._issues_from_runs(runs[ci], max(0, b - a), L, FLANK, 0) == bed[c]
        ok = ok and int(sums[ci]) == int(depths[c].sum())
        ok = ok and w.text[int(toff[ci]):int(toff[ci + 1])].cpu().numpy().tobytes() == O.depth_text_contig(depths[c])
    K = int(w.count.item())
    iv = w.ivl[:K].cpu().numpy().astype(np.int64)
    Ls = np.asarray(inp.lengths, dtype=np.int64)[iv[:, 0]]
    a = np.clip(iv[:, 1] + FLANK, 0, Ls)
    b = np.clip(iv[:, 2] - FLANK + 1, 0, Ls)
    ok = ok and int(np.maximum(b - a, 0).sum()) == int(sums.sum())
    w.oracle_on_chosen = {"depths": depths, "bed": bed, "lengths": tl}      # (cli_genome_number holds the command line's files against it)
    return bool(ok), chosen


def _cpu_filter_task(args):
    f, c = args
    from oracle import gci_oracle as O
    inp, names = _CPU_CTX["inp"], _CPU_CTX["names"]
    lo, hi = _CPU_CTX["ranges"][f][c]
    fobj = inp.files[f]
    return f, c, O.bam_file_dict(fobj.stream, fobj.offsets[lo:hi], names, names, *FILTER, heads=True)


def _cpu_contig_task(args):
    c, s, e = args
    from oracle import gci_oracle as O
    L = _CPU_CTX["inp"].lengths[c]
    name = _CPU_CTX["names"][c]
    d = np.zeros(L, dtype=np.int64)
    O.lib().orc_depth_build(O._p(d), L, O._p(s), O._p(e), s.shape[0], FLANK)
    bed = O.collapse_depth_range({name: d}, -1, 0, FLANK, 0)
    return c, bed[name], len(O.depth_text_contig(d)), int(d.sum())


_CPU_CTX = {}


def cpu_baseline_genome(inp, target_cpu_s=20.0):
    """The oracle's C/Python restatement (`kind: port`) of the same step on a bounded sample of the same workload --
    the reference run with `--chrs <the last contigs of the header>` and `-t <cores>` -- with the reference's own
    parallel structure: a process pool over contigs for the record filter (GCI.py:257-270: Pool.map(read_sam) + dict
    merge in the parent), the join serial in the parent (GCI.py:272-301), a pool again for depth / scan / text."""
    import multiprocessing as mp
    from oracle import gci_oracle as O
    O.build()
    cores = workloads_default_procs()
    names = inp.names
    # sample: contigs from the end of the header until ~target_cpu_s of single-core work (2.7 aligned Gbases/s/core, r01)
    want_bases = target_cpu_s * 2.7e9 / (40.0 * len(inp.files))
    sample, acc = [], 0
    for c in range(len(names) - 1, -1, -1):
        sample.append(c)
        acc += inp.lengths[c]
        if acc >= want_bases:
            break
    sample.sort()
    ranges = []
    for fobj in inp.files:
        ref = fobj.stream[(fobj.offsets[:, None] + np.arange(4, 8, dtype=np.uint64)[None, :]).astype(np.int64)].copy().view("<i4").reshape(-1)
        ranges.append({c: (int(np.searchsorted(ref, c, "left")), int(np.searchsorted(ref, c, "right"))) for c in sample})
    _CPU_CTX.update(inp=inp, names=names, ranges=ranges)
    # metric numerator of the sample, counted like the GPU's: reference spans of its records with flag 0x4 clear
    aligned = int(sum(int(f.aligned_per_contig[c]) for f in inp.files for c in sample))
    t0 = time.perf_counter()
    ctx = mp.get_context("fork")
    with ctx.Pool(cores) as pool:
        parts = pool.map_async(_cpu_filter_task, [(f, c) for f in range(len(inp.files)) for c in sample], chunksize=1).get(600)
        dicts = [dict() for _ in inp.files]
        hq = set()
        for f, c, (d, h) in sorted(parts, key=lambda x: (x[0], x[1])):
            dicts[f].update(d)
            hq |= h
        file1 = O.name_join(dicts, hq, OVLP)
        by_c = {c: ([], []) for c in sample}
        cidx = {names[c]: c for c in sample}
        for seg in file1.values():
            k = by_c[cidx[seg[0]]]
            k[0].append(seg[1])
            k[1].append(seg[2])
        out = pool.map_async(_cpu_contig_task, [(c, np.asarray(by_c[c][0], dtype=np.int64), np.asarray(by_c[c][1], dtype=np.int64))
                                                for c in sample], chunksize=1).get(600)
    dt = time.perf_counter() - t0
    _CPU_CTX.clear()
    return {"seconds": dt, "cores": cores, "os_cpu_count": os.cpu_count(), "sample_contigs": [names[c] for c in sample],
            "sample_bases": int(sum(inp.lengths[c] for c in sample)), "sample_aligned_bases": aligned,
            "issue_runs": int(sum(len(x[1]) for x in out))}


def cpu_baseline_libgci_cpu(inp, gpu_track=None):
    """The same step on the WHOLE workload through libgci_cpu.so -- include/gci_hip.h compiled a second time, by g++, for host
    memory and host threads (gci_amd/csrc/cpu/gci_cpu.cpp) -- on every core of the host: record filter per
    file (heads streams), the join, the depth build, the issue scan, the depth text, the sums.  What it writes is held against the
    GPU's track, base for base."""
    from gci_amd import cpu
    cpu.build()
    e = cpu.CpuEngine(threads=os.cpu_count() or 1)
    e.set_layout(inp.lengths)
    e.heads(True)
    sel = np.arange(len(inp.names), dtype=np.int32)
    t0 = time.perf_counter()
    stages = {}

    def lap(name, t=[t0]):
        now = time.perf_counter()
        stages[name] = round(now - t[0], 3)
        t[0] = now

    files = []
    for f in inp.files:
        stream = np.ascontiguousarray(f.stream)
        files.append((e.bam_filter(stream, f.offsets, sel, *FILTER), stream, f.offsets, 36))
    lap("record filter x%d" % len(files))
    ivl = e.name_join(files, OVLP)
    lap("join")
    track = e.depth_build(ivl, FLANK)
    lap("depth build")
    keys = e.issue_keys(track, -1, 0, FLANK)
    lap("issue scan")
    text, toff = e.depth_text(track)
    lap("depth text")
    sums = e.depth_sum(track)
    lap("sums")
    dt = time.perf_counter() - t0
    out = {"seconds": dt, "cores": e.threads, "os_cpu_count": os.cpu_count(), "stages_s": stages, "intervals": int(ivl.shape[0]),
           "issue_run_keys": int(keys.shape[0]), "text_bytes": int(text.shape[0]), "sum_of_depth": int(sums.sum())}
    del text
    if gpu_track is not None:
        out["equal_to_the_gpu_track"] = bool(np.array_equal(track, gpu_track))
    return out


def cpu_baseline_chr19(w):
    """configs[1]: the oracle's single-thread restatement of the same step on the same chr19 input."""
    from oracle import gci_oracle as O
    O.build()
    refs = [n for n, _ in w.contigs]
    tl = dict(w.contigs)
    stream, offs = w.host
    t0 = time.perf_counter()
    d, hq = O.bam_file_dict(stream, offs, refs, refs, *FILTER)
    file1 = O.name_join([d], hq, OVLP)
    depths = O.depth_build(file1, tl, FLANK)
    bed = O.collapse_depth_range(depths, -1, 0, FLANK, 0)
    text = O.depth_text(depths)
    mean = O.mean_depth(depths)
    dt = time.perf_counter() - t0
    return dt, depths, bed, text, mean


def port_over_reference():
    """Measured in the build container (tools/port_vs_reference.py imports the unmodified reference there; it cannot
    travel): aligned Gbases/s of the oracle port / of the reference on configs[0]."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "port_vs_reference.json")))
    except Exception:
        return None


def verify_against_oracle_multi_rank(w, args):
    """--verify-oracle (N > 1): the oracle over the concatenation of all ranks' files (one global header, rank r's
    records on its own contigs) against three contigs of every rank's track.  -> bool on rank 0, None elsewhere."""
    import torch.distributed as dist
    from gci_amd.formats import bam as bamfmt
    nper = len(w.own)
    pick = [c for c in (nper - 1, nper - 4, nper - 5) if 0 <= c < nper]           # chrM, chr22, chr21 of this rank
    mine = {"files": [], "tracks": {w.own[c]: _track_contig(w, c) for c in pick}}
    for f in w.inp.files:
        first = bamfmt.parse_header(f.stream).first_record
        mine["files"].append((f.stream[first:].copy(), (f.offsets - np.uint64(first)).astype(np.uint64)))
    parts = [None] * w.world
    dist.all_gather_object(parts, mine)
    if w.rank != 0:
        return None
    from oracle import gci_oracle as O
    O.build()
    names = [n for n, _ in w.contigs]
    hdr = np.frombuffer(bamfmt.encode_header(names, [l for _, l in w.contigs]), dtype=np.uint8)
    dicts, hq = [], set()
    for k in range(len(mine["files"])):
        bodies, offs, size = [hdr], [], int(hdr.shape[0])
        for p in parts:
            b, o = p["files"][k]
            bodies.append(b)
            offs.append(o + np.uint64(size))
            size += int(b.shape[0])
        d, h = O.bam_file_dict(np.concatenate(bodies), np.concatenate(offs), names, names, *FILTER, heads=True)
        dicts.append(d)
        hq |= h
    file1 = O.name_join(dicts, hq, OVLP)
    ok = True
    for p in parts:
        for c, got in p["tracks"].items():
            t, L = w.contigs[c]
            want = O.depth_build({q: s for q, s in file1.items() if s[0] == t}, {t: L}, FLANK)[t]
            ok = ok and np.array_equal(got, want)
    return bool(ok)


def verify_strong_against_oracle(w):
    """--verify-oracle with --scaling strong (N > 1, small --scale): the oracle over the UNDIVIDED files (rank 0 generated them
    a second time) against every contig of every rank's track, and the all-reduced sum of depth against the oracle's total."""
    import torch.distributed as dist
    mine = {w.own[c]: _track_contig(w, c) for c in range(len(w.own))}
    parts = [None] * w.world
    dist.all_gather_object(parts, mine)
    if w.rank != 0:
        return None
    from oracle import gci_oracle as O
    O.build()
    inp = w.full_input
    names = inp.names
    dicts, hq = [], set()
    for f in inp.files:
        d, h = O.bam_file_dict(f.stream, f.offsets, names, names, *FILTER, heads=True)
        dicts.append(d)
        hq |= h
    want = O.depth_build(O.name_join(dicts, hq, OVLP), dict(inp.contigs), FLANK)
    ok = len(set(c for p in parts for c in p)) == len(names)
    for p in parts:
        for c, got in p.items():
            ok = ok and np.array_equal(got, want[names[c]])
    ok = ok and int(w.sum_total.item()) == int(sum(int(v.sum()) for v in want.values()))
    return bool(ok)


# ---- SURVEY.md 8(d) numbers (2) and (3) ---------------------------------------------------------------------------

def device_pipeline_number(eng, w):
    """(2): the same step INCLUDING the H2D of the heads streams + offsets and the D2H of what leaves the device (the
    issue-run keys and the .depth.gz members written by the GPU).  The host buffers are pinned (page-locked once, outside the
    window: what a host that streams files through fixed staging buffers has), the uploads run on a copy stream of their own,
    and file f + 1 travels while file f is laid out as record pages and filtered."""
    import torch
    from gci_amd.device import JoinInput
    inp = w.inp
    best = None
    ref_sel = w.ref_sel
    t_pin = time.perf_counter()
    pinned = [(torch.from_numpy(f.stream).pin_memory(), torch.from_numpy(f.offsets.view(np.int64)).pin_memory()) for f in inp.files]
    t_pin = time.perf_counter() - t_pin
    copy = torch.cuda.Stream()
    dev = eng.device
    for _ in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ups = []
        with torch.cuda.stream(copy):
            for h_s, h_o in pinned:
                d_s = torch.empty(h_s.shape, dtype=torch.uint8, device=dev)
                d_o = torch.empty(h_o.shape, dtype=torch.int64, device=dev)
                d_s.copy_(h_s, non_blocking=True)
                d_o.copy_(h_o, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(copy)
                ups.append((d_s, d_o, ev))
        ins = []
        for d_s, d_o, ev in ups:
            torch.cuda.current_stream().wait_event(ev)
            pg = eng.bam_pages(d_s, d_o, False)
            recs, noff = eng.bam_filter_pages(pg, ref_sel, *FILTER, check=False)
            ins.append(JoinInput(recs, pg.buf, noff, 0))
        ivl, cnt = eng.name_join(ins, OVLP, count_flank=FLANK, check=False, out=w.ivl)
        fused = eng.depth_build_fused(ivl, cnt, FLANK, w.track, want_text=False, want_sums=True, issue=(-1.0, 0.0, FLANK),
                                      counted=True)
        blobs = eng.depth_deflate(w.track)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
        del ins, ups
    h2d = int(sum(f.stream.shape[0] + 8 * f.offsets.shape[0] for f in inp.files))
    del pinned
    return {"seconds": best, "gbases_per_s": w.aligned_bases / best / 1e9, "h2d_bytes": h2d, "h2d_gb_per_s_if_all_of_it": h2d / best / 1e9,
            "d2h_depth_gz_member_bytes": int(sum(len(b) for b in blobs)), "pinning_seconds_outside_the_window": t_pin,
            "note": "pinned host memory, uploads on a copy stream: file 2 travels while file 1 is paged + filtered; record pages are "
                    "made on the device inside the window (gci_bam_pages_*); PCIe Gen5 x16 bounds it (63 GB/s spec)"}


def survey_window_step(eng, w, steps):
    """The step inside the window SURVEY.md section 8(d) defines -- "from inflated record bytes resident to all depth tracks
    final in device memory and issue intervals on host": filter x2 -> join -> depth build WITHOUT the decimal text (K10 is not
    in that window; the headline keeps it as the harder output), the issue-run keys copied to the host every step."""
    import torch
    from gci_amd import _lib
    o = w.opts
    keep = o.want_text
    o.want_text = 0
    try:
        def one():
            chk, lib, ctx = eng._chk, eng.lib, eng.ctx
            w.step_records()
            chk(lib.gci_depth_build_begin(ctx, _p(w.ivl), _p(w.count), int(w.ivl.shape[0]), ctypes.byref(o)), "gci_depth_build_begin")
            chk(lib.gci_depth_build_finish(ctx, _p(w.track), None, 0), "gci_depth_build_finish")
            n = int(w.nkeys.item())                               # (synchronises: the keys are on the host when the step ends)
            return w.keys[:min(n, int(w.keys.shape[0]))].cpu()
        one()
        torch.cuda.synchronize()
        eng.profile_enable(1 << _lib.PROF_DEPTH_SCAN)
        eng.profile_read(reset=True)
        t0 = time.perf_counter()
        for _ in range(steps):
            keys = one()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        ms, n = eng.profile_read(reset=True).get("k_tile_build", (0.0, 0))
        eng.profile_enable(0)
    finally:
        o.want_text = keep
    L = int(sum(w.own_lengths))
    algo = w.step_algorithmic_bytes()
    total = algo["k1_record_filter"] + algo["name_join"] + 28 * algo["intervals"] + 4 * L
    build_ms = ms / max(1, n)
    return {"ms_per_step": dt * 1e3, "gbases_per_s": w.aligned_bases / dt / 1e9, "issue_run_keys_to_host": int(keys.shape[0]),
            "algorithmic_bytes_per_step": total, "hbm_frac": total / dt / 1e9 / HBM_PEAK_GBS,
            "k_tile_build_avg_launch_ms": build_ms, "k_tile_build_hbm_frac": (4.0 * L / (build_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if build_ms else None,
            "note": "SURVEY.md 8(d)'s timed window: no decimal text (4 B per base out of the build instead of 7); the issue-run keys "
                    "reach the host inside every step"}


def cli_shaped_step(eng, w, steps):
    """The step as the command line runs it (GCI.py:99-143 through gci_amd/pipeline.py): no decimal text in HBM -- `.depth.gz`
    leaves the device as the gzip members the GPU writes from the build's run lists (gci_depth_deflate_*), D2H of those members included.
    The headline step keeps the text (the harder output)."""
    import torch
    o = w.opts
    keep = o.want_text
    o.want_text = 0
    o.want_runs = 1                      # as pipeline.filter asks when the members come from the device
    try:
        def one():
            chk, lib, ctx = eng._chk, eng.lib, eng.ctx
            w.step_records()
            chk(lib.gci_depth_build_begin(ctx, _p(w.ivl), _p(w.count), int(w.ivl.shape[0]), ctypes.byref(o)), "gci_depth_build_begin")
            chk(lib.gci_depth_build_finish(ctx, _p(w.track), None, 0), "gci_depth_build_finish")
            return eng.depth_deflate(w.track, from_build=True)
        blobs = one()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            blobs = one()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
    finally:
        o.want_text = keep
        o.want_runs = 0
    L = int(sum(w.own_lengths))
    members = int(sum(len(b) for b in blobs))
    algo = w.step_algorithmic_bytes()
    total = algo["k1_record_filter"] + algo["name_join"] + 28 * algo["intervals"] + 4 * L + 8 * 24 * (L // 4096) + members
    return {"ms_per_step": dt * 1e3, "gbases_per_s": w.aligned_bases / dt / 1e9, "depth_gz_member_bytes": members,
            "algorithmic_bytes_per_step": total, "hbm_frac": total / dt / 1e9 / HBM_PEAK_GBS,
            "note": "filter x2 -> join -> depth build without text -> .depth.gz members written by the GPU from the run lists the "
                    "build keeps (gci_build_opts.want_runs: the track is written once and not read) and copied to the host"}


def two_in_flight(eng, w, args, device_index):
    """The same K steps with TWO in flight: a second library context on a stream of its own, with its own copy of the
    inputs and its own outputs, takes every other step, so that the filter / join kernels (instruction- and latency-
    bound) of one step run beside the HBM-bound tile build of the other.  Reported next to the default line, which keeps
    one step in flight because `roofline` is a statement about the kernel running alone."""
    import torch
    from gci_amd import _lib
    from gci_amd.device import Engine
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        e2 = Engine(device_index, stream=st)
        w2 = Workload(e2, 0, 1, w.contigs, [(f.stream, f.offsets, f.name_bytes) for f in w.inp.files], heads=True, name=w.name, k1=w.k1)
        w2.layout(w.own)
        for _ in range(max(1, args.warmup)):
            w2.step()
    torch.cuda.synchronize()
    for e_k in (eng, e2):
        e_k.profile_enable(1 << _lib.PROF_DEPTH_SCAN)
        e_k.profile_read(reset=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        if k & 1:
            with torch.cuda.stream(st):
                w2.step()
        else:
            w.step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ms = n = 0
    for e_k in (eng, e2):
        m, c = e_k.profile_read(reset=True).get("k_tile_build", (0.0, 0))
        ms, n = ms + m, n + c
        e_k.profile_enable(0)
    ok = w2.check() and w.check()
    same = bool(torch.equal(w2.track, w.track)) and bool(torch.equal(w2.sums, w.sums))
    del w2, e2
    return {"ms_per_step": dt / args.steps * 1e3, "gbases_per_s": w.aligned_bases * args.steps / dt / 1e9,
            "k_tile_build_avg_launch_ms": ms / max(1, n), "status_ok": bool(ok), "same_track_as_one_in_flight": same}


def ingest_number(coverage, gb):
    """(3b): the ingestion at the size the metric is quoted on.  `gb` GB of BGZF with realistic SEQ / QUAL entropy in host RAM
    (the members of a chr19 40x HiFi BAM repeated: header once, its 3.66 GB of records as members of their own, k times) go
    through what the command line does with a file too large to inflate whole (pipeline._bam_join_input_gpu: runs of members
    uploaded, inflated on the device with the CRC checked, the records walked, laid out as record pages and filtered; the
    partial record a run ends in carried to the next) -- GB/s in and out, and the projection for the two 40x CHM13 files."""
    import shutil
    import tempfile
    import torch
    from gci_amd import hostio, pipeline, synth
    from gci_amd.formats import bam as bamfmt, bgzf
    from gci_amd.device import Engine
    rs = synth.simulate_reads(synth.CHR19, coverage, "hifi", seed=synth.seed_for(2, 0))
    stream, _ = synth.to_bam_stream(rs, seq_qual="random", seed=7)
    n_rec = len(rs)
    del rs
    first = bamfmt.parse_header(stream).first_record
    thr = hostio.default_threads()
    head = np.frombuffer(bgzf.compress(stream[:first].tobytes(), 1, 1)[:-len(bgzf.BGZF_EOF)], dtype=np.uint8)
    body = np.frombuffer(bgzf.compress(stream[first:].tobytes(), 1, thr)[:-len(bgzf.BGZF_EOF)], dtype=np.uint8)
    body_inflated = int(stream.shape[0]) - first
    del stream
    k = max(1, int(round(gb * 1e9 / body.shape[0])))
    raw = np.empty(head.shape[0] + k * body.shape[0] + len(bgzf.BGZF_EOF), dtype=np.uint8)
    raw[:head.shape[0]] = head
    for i in range(k):
        raw[head.shape[0] + i * body.shape[0]:head.shape[0] + (i + 1) * body.shape[0]] = body
    raw[-len(bgzf.BGZF_EOF):] = np.frombuffer(bgzf.BGZF_EOF, dtype=np.uint8)
    tmp = tempfile.mkdtemp(prefix="gci_ingest_")
    try:
        hp = os.path.join(tmp, "header.bam")                     # (the run-by-run path reads the BAM header from the file)
        with open(hp, "wb") as f:
            f.write(head.tobytes() + bgzf.BGZF_EOF)
        eng = Engine(0)
        targets = ["chr19"]

        def ref_sel_for(hdr):
            return eng.to_device(np.zeros(1, dtype=np.int32))
        t0 = time.perf_counter()
        pos, isz = hostio.bgzf_blocks(raw)
        t_table = time.perf_counter() - t0
        keep = pipeline.GPU_INFLATE_MAX
        pipeline.GPU_INFLATE_MAX = 0                             # run by run, as for a file of this size
        try:
            t0 = time.perf_counter()
            ji = pipeline._bam_join_input_gpu(eng, hp, raw, pipeline._Members(eng, pipeline.BAM_CHUNK_BYTES, pos, isz), ref_sel_for, FILTER)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        finally:
            pipeline.GPU_INFLATE_MAX = keep
        got = int(ji.recs.shape[0])
        rnd = eng.inflate_round()
        inflated = k * body_inflated + first
        out = {"seconds": dt, "member_table_seconds": t_table, "bgzf_bytes": int(raw.shape[0]), "inflated_bytes": inflated,
               "records": got, "records_expected": k * n_rec, "gb_per_s_in": raw.shape[0] / dt / 1e9, "gb_per_s_out": inflated / dt / 1e9,
               "kept_on_device_bytes": int(ji.name_base.shape[0] + ji.recs.shape[0] * 32 + ji.name_off.shape[0] * 8),
               "deflate_ratio": inflated / raw.shape[0],
               "projected_seconds_configs2_two_40x_files": 2 * 190e9 / (inflated / dt),
               "note": "pageable host memory; runs of at most %d MiB inflated and a whole number of the device's decode rounds (%d members "
                       "at a time), each: upload -> inflate + CRC on the device -> record walk -> record pages -> paged filter; the "
                       "bytes of run k + 1 travel (copy stream, two device buffers) while run k is inflated and filtered" % (
                           pipeline.BAM_CHUNK_BYTES >> 20, rnd)}
        del ji, eng
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def cli_number(coverage):
    """(3): wall time of the drop-in command line (python GCI.py -r ref.fa --hifi x.bam) on a chr19 40x HiFi BAM whose
    SEQ / QUAL have realistic entropy (synth.to_bam_stream(seq_qual='random')): BGZF inflate, FASTA scan, all kernels,
    .depth.gz / .bed / .gci emission and file I/O.  Run in-process twice; the second run (warm page cache) is reported."""
    import contextlib
    import io
    import shutil
    import tempfile
    import torch
    from gci_amd import cli, hostio, synth
    from gci_amd.formats import bam as bamfmt
    tmp = tempfile.mkdtemp(prefix="gci_cli_")
    try:
        rs = synth.simulate_reads(synth.CHR19, coverage, "hifi", seed=synth.seed_for(2, 0))
        aligned, n_rec = rs.aligned_bases(), len(rs)
        stream, _ = synth.to_bam_stream(rs, seq_qual="random", seed=7)
        bam, fa = os.path.join(tmp, "hifi.bam"), os.path.join(tmp, "ref.fa")
        bamfmt.write_bam_stream(bam, stream, level=1, threads=hostio.default_threads())
        inflated = int(stream.shape[0])
        del stream, rs
        synth.write_reference_fasta(fa, synth.CHR19)
        walls = []
        for k in range(2):
            od = os.path.join(tmp, "out%d" % k)
            t0 = time.perf_counter()
            with contextlib.redirect_stdout(io.StringIO()):
                cli.main(["GCI.py", "-r", fa, "--hifi", bam, "-d", od, "-t", str(hostio.default_threads())])
            torch.cuda.synchronize()
            walls.append(time.perf_counter() - t0)
        return {"seconds": walls[-1], "first_run_seconds": walls[0], "gbases_per_s": aligned / walls[-1] / 1e9,
                "workload": "chr19 61,707,364 bp, one %gx HiFi BAM, %d records" % (coverage, n_rec),
                "bam_file_bytes": os.path.getsize(bam), "inflated_bytes": inflated,
                "deflate_ratio": inflated / os.path.getsize(bam), "host_threads": hostio.default_threads()}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def cli_genome_number(inp, oracle_on_chosen, verbose=True):
    """(3) at the size the metric is quoted on: `python GCI.py -r chm13.fa --hifi a.bam b.bam` as a process of its own on the
    two 40x whole-genome BGZF BAM files of the resident workload (the same records as the timed step, written back as the files
    they came from: every record with SEQ / QUAL bytes of realistic entropy, deflated at level 1 in htslib's block geometry --
    workloads.write_bgzf_from_heads) and the 3.1 GB assembly, all on tmpfs (a warm page cache).  Wall time of the process, the
    per-phase split it logs (GCI_PHASES), and the files it wrote -- .depth.gz, .bed, .gci -- against the oracle on the contigs of
    `oracle_on_chosen` (parity_genome's)."""
    import gzip
    import shutil
    import subprocess
    import tempfile
    from gci_amd import hostio, synth, workloads
    from oracle import gci_oracle as O
    need = 0
    for f in inp.files:
        o = f.offsets.astype(np.int64)
        l_seq = f.stream[(o[:, None] + np.arange(20, 24)[None, :])].copy().view("<i4").reshape(-1).astype(np.int64)
        need += int((int(f.stream.shape[0]) + int(((l_seq + 1) // 2 + l_seq).sum())) / 2.2)
    need += int(sum(inp.lengths) * 1.02) + (2 << 30)
    base = None
    for d in ("/dev/shm", tempfile.gettempdir()):
        try:
            if shutil.disk_usage(d).free > need * 1.1:
                base = d
                break
        except OSError:
            pass
    mem = workloads._memory_budget_gb() * 1e9
    if base is None or (base == "/dev/shm" and mem < need * 1.25 + 40e9):
        return {"skipped": "needs %.0f GB of file space (tmpfs counts as memory: %.0f GB may still be taken)" % (need / 1e9, mem / 1e9)}
    tmp = tempfile.mkdtemp(prefix="gci_cli_genome_", dir=base)
    try:
        t0 = time.perf_counter()
        bams, made = [], []
        for k, f in enumerate(inp.files):
            p = os.path.join(tmp, "hifi_aligner%d.bam" % (k + 1))
            made.append(workloads.write_bgzf_from_heads(p, f.stream, f.offsets, seed=20250919 + k, verbose=verbose))
            bams.append(p)
        fa = os.path.join(tmp, "chm13.fa")
        synth.write_reference_fasta(fa, inp.contigs)
        t_gen = time.perf_counter() - t0
        od, ph = os.path.join(tmp, "out"), os.path.join(tmp, "phases.json")
        env = dict(os.environ, GCI_PHASES=ph, PYTHONPATH=ROOT)
        cmd = [sys.executable, os.path.join(ROOT, "GCI.py"), "-r", fa, "--hifi"] + bams + ["-d", od, "-t", str(hostio.default_threads())]
        # Twice: the files have just been written (by sixteen processes, into tmpfs), and the first pass of ANY process over freshly
        # written tmpfs pages is slower than every later one -- measured in round 6 (profiles/r06b_first_pass_experiments.txt): `cat`
        # of the fresh files to /dev/null runs at 13 GB/s, and a command line started behind that `cat` is as fast as a second pass
        # (1.59 s against 3.8 s at 0.3 of the genome); pread() instead of the mapping changes nothing (3.9 s).  The kernel's first
        # read access to a page it has only ever written, nothing of this program's.  Both are reported, the second is `seconds`;
        # a user whose BAM has just been written by samtools onto tmpfs sees the first.
        # (a pass gets three minutes: on a host whose other tenants keep its cores and its page cache busy a pass has been seen to
        # take a quarter of an hour, nearly all of it in front of and behind this program's own phases -- such a pass is given up,
        # said so, and the next one measured)
        walls, reps, given_up = [], [], 0
        for _ in range(4):
            shutil.rmtree(od, ignore_errors=True)
            t0 = time.perf_counter()
            try:
                r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=180)
            except subprocess.TimeoutExpired:
                given_up += 1
                if given_up >= 2:
                    return {"error": "two passes of GCI.py over the genome-size files did not finish in 180 s each (the host was busy with "
                                     "something else); input generation took %.0f s" % t_gen}
                continue
            walls.append(time.perf_counter() - t0)
            if r.returncode != 0:
                return {"error": "GCI.py exited with %d: %s" % (r.returncode, r.stderr[-1500:])}
            reps.append(json.load(open(ph)))
            if len(walls) == 2:
                break
        wall, rep = walls[1], reps[1]
        # ---- the files against the oracle, on whole contigs
        chosen = list(oracle_on_chosen["depths"])
        layout = [v for k, v in rep["notes"].items() if k.startswith("depth_gz_layout:")][0]
        ok = sorted(layout) == sorted(n for n, l in inp.contigs if l > 0)
        with open(os.path.join(od, "GCI.depth.gz"), "rb") as f:
            for c in chosen:
                a, b = layout[c]
                f.seek(a)
                got = gzip.decompress(f.read(b - a))
                ok = ok and got == (">%s\n" % c).encode() + O.depth_text_contig(oracle_on_chosen["depths"][c])
                del got
        bed_lines = {}
        for line in open(os.path.join(od, "GCI.0.depth.bed")):
            bed_lines.setdefault(line.split("\t", 1)[0], []).append(line)
        for c in chosen:
            ok = ok and "".join(bed_lines.get(c, [])) == O.bed_text({c: oracle_on_chosen["bed"][c]})
        tl = oracle_on_chosen["lengths"]
        want_rows = {}
        for c in chosen:                                       # a contig's row of the .gci depends on that contig alone
            text = O.compute_index_text({c: tl[c]}, [{c: oracle_on_chosen["bed"][c]}], ["HiFi"], FLANK, 0.005)[0]
            want_rows[c] = [l for l in text.split("\n") if l.startswith(c + "\t")]
        got_rows = {}
        for line in open(os.path.join(od, "GCI.gci")).read().split("\n"):
            got_rows.setdefault(line.split("\t", 1)[0], []).append(line)
        for c in chosen:
            ok = ok and got_rows.get(c) == want_rows[c] and len(want_rows[c]) == 1
        outputs = {fn: os.path.getsize(os.path.join(od, fn)) for fn in sorted(os.listdir(od))}
        aligned = inp.aligned_bases
        serial = ("process start (interpreter, numpy; the HIP runtime starts beside it -- no torch in this process); the table of the beginning of the first "
                  "file and the upload of its first run; after the last byte of the last file: join -> depth build -> .depth.gz "
                  "members -> D2H -> file writes.  Everything else -- the assembly's N scan, the rest of the member tables, every "
                  "later upload, the record walk, pages and filter of a run -- runs beside the inflate kernel")
        return {"seconds": wall, "gbases_per_s": aligned / wall / 1e9,
                "seconds_first_pass_over_freshly_written_files": walls[0], "passes_given_up_after_180_s": given_up,
                "seconds_in_front_of_the_phase_log": rep["notes"].get("process_age_s_when_the_phase_clock_started"),
                "seconds_behind_the_phase_report": (wall - rep["notes"]["process_age_s_at_the_report"]) if "process_age_s_at_the_report" in rep["notes"] else None,
                "first_pass_phases_wall_s": {k: round(v, 4) for k, v in reps[0]["wall_s"].items() if v >= 0.05},
                "process": "python GCI.py (a process of its own: interpreter, imports, HIP context and library load are inside the wall "
                           "time; it holds its HBM buffers itself -- gci_amd/hbm.py -- and ends through the interpreter's ordinary exit)",
                "startup_seconds_outside_the_phase_log": wall - rep["total_s"],
                "phases_wall_s": {k: round(v, 4) for k, v in rep["wall_s"].items()},
                "phases_device_s": {k: round(v, 4) for k, v in rep["gpu_s"].items()},
                # (the first and the longest call of every device stage, and how many there were: on some boxes the first inflate of the
                #  first file takes 1 - 2 s instead of 0.1 -- DESIGN.md section 8 -- and this is where a run says whether it met one)
                "phases_device_first_call_s": {k: round(v, 4) for k, v in rep.get("gpu_first_s", {}).items()},
                "phases_device_longest_call_s": {k: round(v, 4) for k, v in rep.get("gpu_max_s", {}).items()},
                "phases_device_calls": rep.get("gpu_calls", {}),
                "bgzf_bytes": rep["notes"].get("bgzf_bytes"), "inflated_bytes": rep["notes"].get("inflated_bytes"),
                "bgzf_members": rep["notes"].get("bgzf_members"),
                "deflate_ratio": rep["notes"].get("inflated_bytes", 0) / max(1, rep["notes"].get("bgzf_bytes", 1)),
                "files": "two 40x HiFi BGZF BAMs (%s bytes) + %d-byte FASTA on %s (warm page cache)" % (
                    " + ".join(str(m["bytes"]) for m in made), os.path.getsize(fa), base),
                "outputs_bytes": outputs, "parity_vs_oracle_on_contigs": chosen, "parity": bool(ok),
                "input_generation_seconds": t_gen, "serial_in_it": serial, "host_threads": hostio.default_threads()}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


PAF_PUBLISHED_GB = {}      # bench.py --paf-published-gb H,N -> {"hifi": H, "nano": N}


def cli_two_type_number(inp, chosen, verbose=True):
    """SURVEY.md 8(d) number (3) for the two-read-type command line (GCI.py:1007-1026 -- the run the reference's only published timing,
    images/RAM_t32.png, is about: per read type one BAM and one PAF): the inputs as real files on tmpfs -- BGZF BAMs with SEQ / QUAL of
    realistic entropy written from the heads streams, the PAF texts as they are, the assembly as FASTA --, `python GCI.py --hifi ...
    --nano ...` as a process of its own, twice; wall time, the process's own phase log, and the three depth files, the three BED files and
    the .gci rows of `chosen` whole contigs against the oracle (filter() restricted to those contigs, per read type; gap mask; max)."""
    import gzip
    import shutil
    import subprocess
    import tempfile
    from gci_amd import hostio, synth, workloads
    from oracle import gci_oracle as O
    O.build()
    names = inp.names
    chosen = [c for c in chosen if c in names]
    tl = {c: inp.lengths[names.index(c)] for c in chosen}
    base = "/dev/shm" if os.path.isdir("/dev/shm") else None
    tmp = tempfile.mkdtemp(prefix="gci_cli_two_type_", dir=base)
    try:
        t0 = time.perf_counter()
        files = {}
        for kind, t in (("hifi", inp.hifi), ("nano", inp.nano)):
            p = os.path.join(tmp, "%s.bam" % kind)
            made = workloads.write_bgzf_from_heads(p, t.bam.stream, t.bam.offsets, seed=20250919 + (0 if kind == "hifi" else 1), verbose=verbose)
            files[kind] = [p]
            if t.paf is not None:
                q = os.path.join(tmp, "%s.paf" % kind)
                # --paf-published-gb "3.6,48": the PAF files at the sizes of the reference's published CHM13 run (README.md:321,
                # images/RAM_t32.png) -- the same lines, every one with a cg:Z: tag of the length that makes up the size
                want = PAF_PUBLISHED_GB.get(kind)
                if want:
                    workloads.write_paf_at_size(q, t.paf, int(want * 1e9))
                else:
                    t.paf.tofile(q)
                files[kind].append(q)
            files[kind + "_made"] = made
        fa = os.path.join(tmp, "assembly.fa")
        synth.write_reference_fasta(fa, inp.contigs, gaps=inp.gaps) if inp.gaps else synth.write_reference_fasta(fa, inp.contigs)
        t_gen = time.perf_counter() - t0
        od, ph = os.path.join(tmp, "out"), os.path.join(tmp, "phases.json")
        env = dict(os.environ, GCI_PHASES=ph, PYTHONPATH=ROOT)
        cmd = [sys.executable, os.path.join(ROOT, "GCI.py"), "-r", fa, "--hifi"] + files["hifi"] + ["--nano"] + files["nano"] + \
              ["-d", od, "-t", str(hostio.default_threads())]
        walls, reps, given_up = [], [], 0
        for _ in range(4):
            shutil.rmtree(od, ignore_errors=True)
            t0 = time.perf_counter()
            try:
                r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
            except subprocess.TimeoutExpired:
                given_up += 1
                if given_up >= 2:
                    return {"error": "two passes did not finish in 300 s each; input generation took %.0f s" % t_gen}
                continue
            walls.append(time.perf_counter() - t0)
            if r.returncode != 0:
                return {"error": "GCI.py exited with %d: %s" % (r.returncode, r.stderr[-1500:])}
            reps.append(json.load(open(ph)))
            if len(walls) == 2:
                break
        wall, rep = walls[1], reps[1]
        # ---- the oracle on the chosen contigs, per read type; then the files
        tracks = []
        for t in (inp.hifi, inp.nano):
            file1 = O.file1_on_contigs_mixed([t.paf.tobytes()] if t.paf is not None else [], [(t.bam.stream, t.bam.offsets, names)], names, chosen,
                                             FILTER[0], FILTER[1], FILTER[2], FILTER[3], OVLP, heads=True)
            d = O.depth_build(file1, tl, FLANK)
            tracks.append(d)
        layouts = {os.path.basename(k.split(":", 1)[1]): v for k, v in rep["notes"].items() if k.startswith("depth_gz_layout:")}
        ok = True
        raw_tracks = [{c: v.copy() for c, v in d.items()} for d in tracks]              # (a read type's own .depth.gz is written before the gap mask)
        for d in tracks:
            O.merge_gaps_depths(d, {c: v for c, v in inp.gaps.items() if c in tl} or None)
        two = O.max2(tracks[0], tracks[1])
        for fn, want_text_of, want_bed_of in (("GCI_hifi", raw_tracks[0], tracks[0]), ("GCI_nano", raw_tracks[1], tracks[1]), ("GCI_two_type", two, two)):
            layout = layouts.get(fn + ".depth.gz")
            ok = ok and layout is not None
            if layout is None:
                continue
            with open(os.path.join(od, fn + ".depth.gz"), "rb") as f:
                for c in chosen:
                    a, b = layout[c]
                    f.seek(a)
                    got = gzip.decompress(f.read(b - a))
                    ok = ok and got == (">%s\n" % c).encode() + O.depth_text_contig(want_text_of[c])
                    del got
            bed = O.collapse_depth_range(want_bed_of, -1, 0, FLANK, 0)
            lines = {}
            for line in open(os.path.join(od, fn + ".0.depth.bed")):
                lines.setdefault(line.split("\t", 1)[0], []).append(line)
            for c in chosen:
                ok = ok and "".join(lines.get(c, [])) == O.bed_text({c: bed[c]})
        outputs = {fn: os.path.getsize(os.path.join(od, fn)) for fn in sorted(os.listdir(od))}
        aligned = int(inp.hifi.bam.aligned_bases + inp.nano.bam.aligned_bases + inp.hifi.paf_aligned_bases + inp.nano.paf_aligned_bases)
        return {"seconds": wall, "gbases_per_s": aligned / wall / 1e9, "seconds_first_pass_over_freshly_written_files": walls[0],
                "passes_given_up_after_300_s": given_up,
                "command": "python GCI.py -r assembly.fa --hifi hifi.bam hifi.paf --nano nano.bam nano.paf -d out (a process of its own)",
                "seconds_in_front_of_the_phase_log": rep["notes"].get("process_age_s_when_the_phase_clock_started"),
                "phases_wall_s": {k: round(v, 4) for k, v in rep["wall_s"].items()},
                "phases_device_s": {k: round(v, 4) for k, v in rep["gpu_s"].items()},
                "files": {k: {"bytes": [os.path.getsize(p) for p in v]} for k, v in files.items() if not k.endswith("_made")},
                "bam_members": {k[:-5]: v.get("members") for k, v in files.items() if k.endswith("_made")},
                "outputs_bytes": outputs, "parity_vs_oracle_on_contigs": chosen, "parity": bool(ok), "input_generation_seconds": t_gen,
                "reference_published": "6.03 h / 1.83 h for its CHM13 run on other hardware (images/RAM_t32.png): context, not a comparison"}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def paf_number(args):
    """K2 at the size of the reference's published run (README.md:321: a 48 GB ONT PAF + a 3.6 GB HiFi PAF): `--paf-gb` GB of PAF text
    through gci_paf_filter_device (the PAF path of filter(), GCI.py:211-254), text resident in HBM.  The text is a chunk of ~1 M
    lines -- the PAF view of simulated HiFi reads over CHM13, 2 % of them split into 2 - 3 blocks (synth.to_paf_text) -- repeated
    with the chunk number written into every query name, so every chunk has its own queries; a query's lines all lie in its chunk
    and the filter treats every query independently, so the oracle over chunk 0 alone is the exact expectation for chunk 0's
    queries (and, with the chunk number replaced, for every other chunk's)."""
    import tempfile
    import torch
    from gci_amd import synth
    from gci_amd.device import Engine, REC_DTYPE
    from oracle import gci_oracle as O
    contigs = synth.CHM13
    names = [n for n, _ in contigs]
    os.environ.setdefault("GCI_PAF_POOL_KEEP_GB", "256")          # the passes reuse K2's scratch (a command line makes one pass)
    t0 = time.perf_counter()
    rs = synth.simulate_reads(contigs, 6.0, "hifi", seed=synth.seed_for(4, 1), name_prefix="c0000/m64011_190830/")
    other = synth.perturb(rs, synth.seed_for(4, 3))
    other.contigs = tuple(contigs)
    chunk = synth.to_paf_text(other, synth.seed_for(4, 11))
    keep = ((other.flag & 0x4) == 0) & ((other.flag & 0x100) == 0)
    aligned_chunk = int(other.ref_span()[keep].sum())
    del rs, other
    nl = np.flatnonzero(chunk == 10)
    starts = np.concatenate([[0], nl[:-1] + 1]).astype(np.int64)
    assert bytes(chunk[:5]) == b"c0000"
    K = max(1, int(round(args.paf_gb * 1e9 / chunk.shape[0])))
    if K > 9999:
        sys.exit("bench.py --workload paf: at most 9999 chunks")
    text = np.empty(K * chunk.shape[0] + 16, dtype=np.uint8)
    text[-16:] = 0
    for k in range(K):
        seg = text[k * chunk.shape[0]:(k + 1) * chunk.shape[0]]
        seg[:] = chunk
        digits = np.frombuffer(b"%04d" % k, dtype=np.uint8)
        for j in range(4):
            seg[starts + 1 + j] = digits[j]
    n_bytes = K * int(chunk.shape[0])
    t_gen = time.perf_counter() - t0
    eng = Engine(0)
    t0 = time.perf_counter()
    d_text = eng.to_device(text)
    torch.cuda.synchronize()
    t_h2d = time.perf_counter() - t0
    ends = np.asarray([n_bytes], dtype=np.uint64)
    for _ in range(max(1, args.warmup)):
        out = eng.paf_filter_text(d_text, ends, names, FILTER[0], FILTER[1], FILTER[3])
        del out
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    passes = []
    for _ in range(args.steps):
        t1 = time.perf_counter()
        out = eng.paf_filter_text(d_text, ends, names, FILTER[0], FILTER[1], FILTER[3])
        torch.cuda.synchronize()
        passes.append(time.perf_counter() - t1)
        if _ + 1 < args.steps:
            del out
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    # ---- parity: chunk 0 (and the last chunk) against the oracle over the chunk's own text
    with tempfile.TemporaryDirectory() as tmp:
        p = os.path.join(tmp, "chunk0.paf")
        chunk.tofile(p)
        want, want_hq = O.paf_filter([p], names, FILTER[0], FILTER[1], FILTER[3])
    want = want[0]
    recs = out[0].recs.cpu().numpy().reshape(-1).view(REC_DTYPE)
    noff = out[0].name_off.cpu().numpy()
    ok = True
    for k in sorted({0, K - 1}):
        lo, hi = k * chunk.shape[0], (k + 1) * chunk.shape[0]
        sel = np.flatnonzero((noff >= lo) & (noff < hi))
        got = {}
        for i in sel.tolist():
            nm = bytes(text[int(noff[i]):int(noff[i]) + int(recs["name_len"][i])]).decode()
            got["c0000" + nm[5:]] = (names[int(recs["contig"][i])], int(recs["start"][i]), int(recs["end"][i]), int(recs["qlen"][i]),
                                     bool(recs["flags"][i] & 2))
            ok = ok and nm[:5] == "c%04d" % k
        exp = {q: (s[0], s[1], s[2], s[3], q in want_hq) for q, s in want.items()}
        ok = ok and got == exp
    n_lines = K * int(nl.shape[0])
    return {"seconds_per_pass": dt, "pass_seconds": [round(x, 4) for x in passes], "paf_bytes": n_bytes, "lines": n_lines, "queries_out": int(recs.shape[0]), "chunks": K,
            "ms_per_gb_of_text": dt * 1e3 / (n_bytes / 1e9), "text_gb_per_s": n_bytes / dt / 1e9, "lines_per_s": n_lines / dt,
            "aligned_bases": K * aligned_chunk, "parity_vs_oracle_chunks": sorted({0, K - 1}), "parity": bool(ok),
            "h2d_seconds_outside_the_pass": t_h2d, "generation_seconds": t_gen}


def main():
    args = parse_args()
    if args.paf_published_gb:
        h, n = (float(x) for x in args.paf_published_gb.split(","))
        PAF_PUBLISHED_GB.update({"hifi": h, "nano": n})
    if args.only_step:
        args.no_e2e = args.no_cpu_baseline = args.no_cli_genome = True
    if args.workload == "paf":
        r = paf_number(args)
        out = {"metric": "PAF filter alone (K2): aligned Gbases/s of its lines, text resident in HBM -- NOT the filter+depth pipeline's metric",
               "value": r["aligned_bases"] / r["seconds_per_pass"] / 1e9, "unit": "Gbases/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": r["seconds_per_pass"] * 1e3,
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int64 / f64", "data": "synthetic",
               "config": {"workload": "the PAF half of filter() alone (K2, GCI.py:211-254): %.1f GB of PAF text resident in HBM, one pass = "
                                      "line starts -> tokenise + filter -> query table -> per-query scoring" % (r["paf_bytes"] / 1e9),
                          "baseline_config": "the PAF inputs of configs[3] at the size of the reference's published run"},
               "roofline": {"bound": "hbm", "kernel": "K2 as a whole (seven k_paf_* kernels + scans)", "achieved": r["text_gb_per_s"],
                            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": r["text_gb_per_s"] / HBM_PEAK_GBS, "traffic": None,
                            "algorithmic_bytes_per_launch": r["paf_bytes"]},
               "cpu_baseline": None, "paf": r}
        print(json.dumps(out), flush=True)
        if not r["parity"]:
            sys.exit("PARITY FAILURE: the PAF filter differs from the oracle")
        return
    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    launched = all(k in os.environ for k in ("RANK", "LOCAL_RANK", "MASTER_ADDR"))
    if args.gpus > 1 and not launched:
        # `python bench.py --gpus N` on its own: become the launcher (one process per GPU over RCCL), as GCI.py --gpus N does
        # (gci_amd/cli.py).  The strong-scaling workload is generated ONCE, by rank 0, into a directory every rank maps.
        import tempfile
        port = int(os.environ.get("MASTER_PORT", str(29700 + os.getpid() % 2000)))
        shared = tempfile.mkdtemp(prefix="gci_bench_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
        env = dict(os.environ, GCI_BENCH_SHARED=shared)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        import subprocess
        rc = subprocess.call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
                              "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:], env=env)
        import shutil
        shutil.rmtree(shared, ignore_errors=True)
        sys.exit(rc)
    if not launched:
        rank, local_rank, world = 0, 0, 1
    if args.gpus != world:
        sys.exit("bench.py --gpus %d was launched with WORLD_SIZE %d" % (args.gpus, world))
    import torch.distributed as dist
    exchange = args.force_exchange or args.force_replicated

    from gci_amd.build import build_hip, needs_build
    if needs_build() and local_rank == 0:
        build_hip()

    global VIA_HOST
    VIA_HOST = args.backend == "gloo"
    device_index = int(os.environ.get("GCI_DIST_DEVICE", str(local_rank)))

    def eng_factory():
        """HIP context, process group and library context: created AFTER the host-side generation of the inputs."""
        torch.cuda.set_device(device_index)
        if world > 1 or exchange or args.force_sharded:
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            # RCCL logs to stdout: keep it off the channel on which rank 0 prints its ONE JSON line
            if os.environ.get("NCCL_DEBUG", "").upper() in ("VERSION", "INFO", "WARN"):
                os.environ.pop("NCCL_DEBUG")
            os.environ.setdefault("NCCL_DEBUG_FILE", "/dev/stderr")
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29531")
            if args.backend == "gloo":
                os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
                dist.init_process_group("gloo", rank=rank, world_size=world)
            else:
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device_index))
            dist.barrier()                                  # (also: rank 0 has finished building the library)
        from gci_amd.device import Engine
        return Engine(device_index)

    from gci_amd import _lib
    two_type = args.workload in ("genome4", "diploid")
    make = make_two_type_workload if two_type else make_genome_workload if args.workload == "genome" else make_chr19_workload
    eng, w = make(eng_factory, rank, world, args, exchange, args.force_replicated)
    if args.count_in_build and not two_type:
        w.opts.counted = 0

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # --inflight N: N - 1 more contexts on streams of their own; step k runs on context k % N
    lanes = [(eng, w, None)]
    if args.inflight > 1:
        if world > 1 or w.exchange or args.workload != "chr19":
            sys.exit("bench.py --inflight is for the single-GPU chr19 workload")
        from gci_amd.device import Engine
        for _ in range(args.inflight - 1):
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                e2 = Engine(device_index, stream=st)
                _, w2 = make_chr19_workload(None, rank, world, args, False, False, eng=e2)
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

    # The window of `value` (with the inputs in HBM as the bench contract wants them): inflated record bytes resident
    # on the device -> record pages -> K1 -> join -> depth build (+ decimal text) -> the issue-run boundaries on the host.
    for _, w_k, _ in [(eng, w, None)]:
        w_k.keys_to_host = not two_type
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
    smi_before = smi_sample() if rank == 0 else None
    fence()
    t0 = time.perf_counter()
    run_steps(args.steps)
    fence()
    dt = time.perf_counter() - t0
    smi_after = smi_sample() if rank == 0 else None
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
        all_reduce(t, dist.ReduceOp.MAX)
        dt = float(t.item())
        ab = torch.tensor([w.aligned_bases], dtype=torch.int64, device=eng.device)
        all_reduce(ab, dist.ReduceOp.SUM)
        aligned_total = int(ab.item())
    else:
        aligned_total = w.aligned_bases
    ms_per_step = dt / args.steps * 1e3

    # ---- roofline of the dominant kernel: k_tile_build writes the int32 track (4 B/base) and the decimal text; it
    # reads only the event buckets.  Average launch time from HIP events on the ctx stream over the timed region.
    scan_ms, scan_n = prof.get("k_tile_build", (0.0, 0))
    scan_avg_ms = scan_ms / max(1, scan_n)
    algo = w.step_algorithmic_bytes()
    algo_bytes = 4.0 * algo["bases"] + algo["text_bytes"]
    achieved = algo_bytes / (scan_avg_ms * 1e-3) / 1e9 if scan_avg_ms > 0 else 0.0
    step_achieved = algo["total"] / (ms_per_step * 1e-3) / 1e9

    # HBM bytes per launch of the dominant kernel from the committed PMC passes of this workload (they cannot be
    # collected from inside this process: rocprofv3 wraps the command; tools/prof_pmc.sh, profiles/)
    traffic, traffic_source = None, None
    try:
        tag = "genome" if args.workload == "genome" else "chr19"
        tj = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_%s_traffic.json" % tag))
        if tj and args.scale == 1.0 and args.coverage == 40.0 and (args.workload == "genome" or args.contig_len == CHR19_LEN):
            tf = json.load(open(os.path.join(ROOT, "profiles", tj[-1])))
            # the file must be about THIS kernel on THIS workload: its name, and HBM bytes within 10 % of the algorithmic ones
            # (a kernel that changed, or a workload of another size, makes the stored counters stale: then no traffic is reported)
            if tf.get("kernel") == "k_tile_build" and 0.9 * algo_bytes <= tf["hbm_bytes_per_launch"] <= 1.5 * algo_bytes:
                traffic = tf["hbm_bytes_per_launch"]
                traffic_source = "profiles/%s (stored: the PMC passes of an earlier run of this workload -- rocprofv3 wraps the command, this " \
                                 "process cannot collect them)" % tj[-1]
            else:
                traffic_source = "profiles/%s REFUSED: it is about kernel %r with %.3g HBM bytes per launch, this run's k_tile_build has %.3g " \
                                 "algorithmic bytes" % (tj[-1], tf.get("kernel"), tf.get("hbm_bytes_per_launch", 0.0), algo_bytes)
    except Exception:
        traffic = None

    eng.profile_enable((1 << _lib.PROF_COUNT) - 1)
    for _ in range(3):
        w.step()
    breakdown = {k: round(ms / n * 1e3, 2) for k, (ms, n) in eng.profile_read(reset=True).items()}   # us / launch
    eng.profile_enable(0)
    # number (1) of SURVEY 8(d) as rounds 3 - 5 reported it as `value`: the kernels alone over record pages made in front of the clock,
    # nothing copied to the host inside the step
    n1_ms = None
    if world == 1 and not two_type and getattr(w, "pages_in_step", False):
        w.pages_in_step, w.keys_to_host = False, False
        w.step()
        fence()
        t1 = time.perf_counter()
        for _ in range(max(3, args.steps // 2)):
            w.step()
        fence()
        n1_ms = (time.perf_counter() - t1) / max(3, args.steps // 2) * 1e3
        w.pages_in_step, w.keys_to_host = True, True
    fill_gbs = None
    if world == 1 and rank == 0 and algo_bytes > 0:
        try:
            fill_gbs = fill_ceiling_gbs(eng, algo_bytes)
        except Exception:
            fill_gbs = None

    out = {
        "metric": "aligned Gbases/s through filter+depth pipeline (CHM13, 40x HiFi)",
        "value": aligned_total * args.steps / dt / 1e9,
        "unit": "Gbases/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        # N = 1 is the first point of the curve --scaling asks for (strong by default: the same whole genome at every N)
        "scaling": ("strong" if w.sharded else "weak") if world > 1 else (args.scaling if args.workload == "genome" else "weak"),
        "vs_baseline": None,
        "dtype": "int32",
        "data": "synthetic",
        "config": {"workload": w.name + (" [scale %g]" % args.scale if args.workload != "chr19" and args.scale != 1.0 else ""),
                   "baseline_config": {"genome": "configs[2]", "chr19": "configs[1]", "genome4": "configs[3] on one GPU",
                                       "diploid": "configs[4] on one GPU"}[args.workload],
                   "input_files_per_gpu": w.n_files, "records_per_gpu": w.n_rec, "aligned_bases_per_step": aligned_total,
                   "bam_input": ("record pages (gci_bam_pages_*: %d-byte pages made on the device from the " % w.pages[0].page_bytes if w.pages else "")
                                + ("heads stream (records without SEQ / QUAL)" if w.heads else "whole inflated stream") + (")" if w.pages else ""),
                   "input_bytes_per_gpu": w.stream_bytes, "parallelism": "contig-sharded x%d" % world,
                   "steps_in_flight": len(lanes),
                   "value_is": ("SURVEY 8(d)'s window with the inputs resident in HBM: inflated record bytes (heads streams + record offsets) on the "
                                "device -> record pages (gci_bam_pages_*) -> K1 -> join -> depth build + decimal text -> issue-run boundaries "
                                "copied to the host, all inside every timed step; n1_* = the kernels alone over pages made in front of the "
                                "clock (what rounds 3 - 5 reported as value), n2_* = the same window with the record bytes coming from pinned "
                                "host memory (PCIe inside), n3_* = the command line on BGZF files" if getattr(w, "pages_in_step", False) else
                                "kernels only over resident inputs"),
                   "n1_kernels_only_ms_per_step": n1_ms,
                   "n1_kernels_only_gbases_per_s": (aligned_total / (n1_ms * 1e-3) / 1e9) if n1_ms else None,
                   "workload_generated": "once, by rank 0" if (world > 1 and os.environ.get("GCI_BENCH_SHARED")) else "by every rank" if world > 1 else "in this process",
                   "collectives_per_step": (3 if w.sharded else None),
                   "join": ("sharded by name hash: ONE all-to-all of the 32-byte records and 48-byte name slots of every file, one of the "
                            "16-byte intervals to the owners of their contigs, one all-reduce of the sum of depth (3 collectives per step, "
                            "whatever the number of files); %d bytes leave this rank per step" % w.sj.bytes_per_step()
                            if w.sharded else "local" if not w.exchange else
                            "local, validated by the exact cross-rank name check (hash all-to-all)" if not w.replicated_steps else
                            "replicated (all-gather of records + names)")},
        "roofline": {"bound": "hbm", "kernel": "k_tile_build (depth + text write)", "achieved": achieved, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                     "algorithmic_bytes_per_launch": algo_bytes, "avg_launch_ms": scan_avg_ms, "launches": scan_n,
                     # what a plain fill of as many bytes reaches on this box in this process, and the kernel against THAT
                     "fill_ceiling_gbs": fill_gbs, "frac_of_fill_ceiling": (achieved / fill_gbs) if fill_gbs else None,
                     "box": {"rocm_smi_before_steps": smi_before, "rocm_smi_after_steps": smi_after}},
        "step_roofline": {"bound": "hbm", "achieved": step_achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                          "frac": step_achieved / HBM_PEAK_GBS, "algorithmic_bytes_per_step": algo},
        "kernel_us_per_launch": breakdown,
    }

    forced = world == 1 and (exchange or args.force_sharded)          # one rank on the multi-rank path (the RCCL self-test)
    if (world > 1 or forced) and args.verify_oracle and args.workload == "genome":
        ok = verify_strong_against_oracle(w) if w.sharded else verify_against_oracle_multi_rank(w, args)
        if rank == 0:
            out["parity_vs_oracle_all_ranks"] = ok
            if not ok:
                print(json.dumps(out))
                sys.exit("PARITY FAILURE: the multi-rank result differs from the oracle over all ranks' files")
    if forced:
        out["config"]["collectives"] = {"backend": args.backend, "world": 1, "executed_per_step": (
            "all_to_all_single x 2 (records + name slots of all files in one, intervals) + all_reduce of the sum of depth"
            if w.sharded else "all_to_all_single (name-check hashes) + all_reduce of the per-contig sums"
            + (" + all_gather_into_tensor x 3 per file (replicated join)" if w.replicated_steps else ""))}
        out["cpu_baseline"] = None
    elif rank == 0 and world == 1 and two_type:
        names = w.inp.names
        chosen = ["chr14", "chr22", "chrM"] if args.workload == "genome4" else ["mat_chr14", "pat_chr21", "pat_chr22"]
        hit = [c for c, segs in w.inp.gaps.items() if c in names]                     # a contig with a gap, one with a -R region
        chosen += [c for c in (hit[:1] + [r[0] for r in w.inp.regions[:1]]) if c not in chosen and w.inp.lengths[names.index(c)] < 1.2e8]
        ok, chosen = parity_two_type(w, chosen)
        out["parity_vs_oracle_full_size"] = ok
        out["parity_contigs"] = chosen
        out["roofline"]["kernel"] = "k_tile_build (depth write; no text in this workload)"
        out["cpu_baseline"] = None
        if ok and args.cli:
            inp2 = w.inp
            out["survey_8d"] = {"1_kernels_only_gbases_per_s": out["value"], "3_command_line_two_read_types": cli_two_type_number(inp2, chosen)}
            if out["survey_8d"]["3_command_line_two_read_types"].get("parity") is False:
                print(json.dumps(out))
                sys.exit("PARITY FAILURE: the two-read-type command line's files differ from the oracle")
        if ok and args.workload == "diploid":
            out["plot_front_end_n3"] = plot_front_end_number(w)
            if not out["plot_front_end_n3"]["parity_vs_oracle"]:
                print(json.dumps(out))
                sys.exit("PARITY FAILURE: the -p front end differs from the oracle at full size")
        if not ok:
            print(json.dumps(out))
            sys.exit("PARITY FAILURE: GPU result differs from the oracle at full size")
    elif rank == 0 and world == 1:
        if args.workload == "genome":
            ok, chosen = parity_genome(w)
            out["parity_vs_oracle_full_size"] = ok
            out["parity_contigs"] = chosen
            if not ok:
                print(json.dumps(out))
                sys.exit("PARITY FAILURE: GPU result differs from the oracle at full size")
            survey = {"1_kernels_only_gbases_per_s": out["config"]["n1_kernels_only_gbases_per_s"], "value_window_gbases_per_s": out["value"]}
            if not args.no_e2e and args.inflight == 1:
                out["two_steps_in_flight"] = two_in_flight(eng, w, args, device_index)
            if not args.only_step:
                out["survey_window_step"] = survey_window_step(eng, w, max(3, args.steps // 2))
                out["cli_shaped_step"] = cli_shaped_step(eng, w, max(3, args.steps // 2))
            if not args.no_e2e:
                survey["2_device_pipeline_incl_h2d_d2h"] = device_pipeline_number(eng, w)
            if not args.no_cpu_baseline:
                # the whole workload through libgci_cpu.so on every core; the oracle's port on a sample stays beside it (the ratio
                # port / reference measured in the build container translates THAT one to the reference's Python)
                gpu_track = w.track.cpu().numpy()
                c = cpu_baseline_libgci_cpu(w.inp, gpu_track)
                del gpu_track
                if c.get("equal_to_the_gpu_track") is False:
                    out["cpu_baseline"] = c
                    print(json.dumps(out))
                    sys.exit("PARITY FAILURE: libgci_cpu.so and the GPU disagree on the depth track")
                o = cpu_baseline_genome(w.inp, target_cpu_s=8.0)
                frac = o["sample_bases"] / float(sum(w.inp.lengths))
                out["cpu_baseline"] = {
                    "value": aligned_total / c["seconds"] / 1e9, "unit": "Gbases/s", "cores": c["cores"], "kind": "port", "port": "libgci_cpu",
                    "sample": "100 %%: the whole workload (both files, all %d contigs) through libgci_cpu.so -- include/gci_hip.h compiled by g++ "
                              "for host memory and host threads -- on %d threads (os.cpu_count() = %s), %.2f s wall: %s; its depth track equals "
                              "the GPU's base for base.  It starts from the inflated record bytes in host memory (NO BGZF inflate): comparable with "
                              "`value`, n1 and n2 -- not with n3, the command line, which also inflates the files" % (len(w.inp.names), c["cores"], c["os_cpu_count"], c["seconds"],
                                                           ", ".join("%s %.2f" % kv for kv in c["stages_s"].items())),
                    "equal_to_the_gpu_track": c.get("equal_to_the_gpu_track"),
                    "oracle_port_on_a_sample": {
                        "value": o["sample_aligned_bases"] / o["seconds"] / 1e9, "cores": o["cores"],
                        "sample": "contigs %s (%.1f %% of the workload) through oracle/gci_oracle.{c,py} with the reference's structure (process "
                                  "pools over contigs, serial join), %.1f s on %d processes" % (",".join(o["sample_contigs"]), 100.0 * frac,
                                                                                               o["seconds"], o["cores"]),
                        "port_over_reference": port_over_reference()}}
            else:
                out["cpu_baseline"] = None
            if not args.no_e2e and not args.no_cli_genome and args.reads == "hifi":
                inp, orc = w.inp, w.oracle_on_chosen
                # The command line is a process of its own and wants most of the device for itself (150 GB at its peak): what this
                # process's allocator still caches from the legs above (a second workload, fill buffers, members) goes back to the
                # driver first -- with it held here the child's inflate ran 40 % longer (5.5 s of device time where the same command
                # line alone on the box takes 3.3: tools/hwtests/cli_trace.sh, profiles/r05i2_cli_genome_trace.txt).
                import gc
                gc.collect()
                torch.cuda.synchronize()
                torch.cuda.empty_cache()
                survey["3_command_line_genome"] = cli_genome_number(inp, orc)
                if survey["3_command_line_genome"].get("parity") is False:
                    out["survey_8d"] = survey
                    print(json.dumps(out))
                    sys.exit("PARITY FAILURE: the command line's files differ from the oracle at genome size")
                del inp, orc
            w.inp = w.oracle_on_chosen = None
            if not args.no_e2e:
                survey["3_command_line_chr19_realistic_bam"] = cli_number(args.coverage)
                if args.ingest_gb > 0:
                    torch.cuda.empty_cache()
                    survey["3b_ingest_genome"] = ingest_number(args.coverage, args.ingest_gb)
            out["survey_8d"] = survey
            # the numbers of SURVEY 8(d) as FLAT SCALARS of `config` (the driver's record keeps scalars of config, not nested objects)
            g3 = survey.get("3_command_line_genome") or {}
            p2 = survey.get("2_device_pipeline_incl_h2d_d2h") or {}
            cfg = out["config"]
            cfg["n2_device_pipeline_gbases_per_s"] = p2.get("gbases_per_s")
            cfg["n2_device_pipeline_s"] = p2.get("seconds")
            cfg["n3_cli_genome_s"] = g3.get("seconds")
            cfg["n3_cli_genome_gbases_per_s"] = g3.get("gbases_per_s")
            cfg["n3_cli_first_pass_s"] = g3.get("seconds_first_pass_over_freshly_written_files")
            cfg["n3_inflate_device_s"] = (g3.get("phases_device_s") or {}).get("bgzf_inflate + crc")
            cfg["n3_first_inflate_call_s"] = (g3.get("phases_device_first_call_s") or {}).get("bgzf_inflate + crc")
            cfg["n3_start_s"] = g3.get("startup_seconds_outside_the_phase_log")
            cfg["n3_exit_s"] = g3.get("seconds_behind_the_phase_report")
            cfg["n3_bgzf_gb"] = (g3.get("bgzf_bytes") or 0) / 1e9 if g3 else None
            cfg["n3_parity"] = g3.get("parity")
            cfg["cpu_baseline_excludes"] = "BGZF inflate: libgci_cpu.so starts from the inflated record bytes in host memory, as n1 / value / n2 do -- not comparable with n3"
        elif not args.no_cpu_baseline:
            cdt, depths, bed, text, mean = cpu_baseline_chr19(w)
            out["cpu_baseline"] = {"value": w.aligned_bases / cdt / 1e9, "unit": "Gbases/s", "cores": 1, "kind": "port",
                                   "sample": "one full step (all %d records, %d bp) through oracle/gci_oracle.{c,py}, "
                                             "%.1f s" % (w.n_rec[0], args.contig_len, cdt),
                                   "port_over_reference": port_over_reference()}
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
        else:
            out["cpu_baseline"] = None

    if rank == 0:
        sys.stdout.flush()
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
