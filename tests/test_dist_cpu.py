"""The N>1 path with world_size 2 on CPU (gloo): the record / name exchange that feeds the join,
the contig ownership maps and the integer all-reduce.  Per-rank K1 output is stood in for by the
oracle's decode (the HIP kernels need a GPU); what is checked is that after the exchange every
rank holds exactly the single-process record set, and that the sharded result equals the
unsharded one."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gci_amd import shard, synth
        from gci_amd.device import REC_DTYPE, name_hash_np
        from oracle import gci_oracle as O
        contigs = (("a", 300_000), ("b", 200_000), ("c", 90_000))
        targets = [n for n, _ in contigs]
        rs = synth.simulate_reads(contigs, 15, "hifi", seed=77)
        dup = rs.take(np.arange(0, len(rs), 9))                 # repeated names across contigs
        dup.ref_id[:] = (dup.ref_id + 1) % 3
        dup.pos[:] = np.minimum(dup.pos, 60_000)
        rs = synth.concat(rs, dup).sorted()
        stream, offs = synth.to_bam_stream(rs)
        a = O.bam_filter_arrays(stream, offs, np.arange(3, dtype=np.int32), 30, 50, 0.1, 0.9)
        lo, hi = shard.record_slices(len(rs), world)[rank]      # this rank decodes records [lo, hi)
        n = hi - lo
        names = [bytes(stream[int(o):int(o) + int(l)]) for o, l in zip(a["name_off"][lo:hi], a["name_len"][lo:hi])]
        ex = shard.RecordExchange(n, sum(len(x) for x in names), torch.device("cpu"))
        recs = np.zeros(n, dtype=REC_DTYPE)
        recs["name_hash"] = name_hash_np(names)
        for f in ("contig", "start", "end", "qlen"):
            recs[f] = a[f][lo:hi]
        recs["rec_idx"] = ex.rec_idx_base + np.arange(n)
        recs["flags"] = a["passed"][lo:hi] | (a["hq"][lo:hi] << 1)
        recs["name_len"] = a["name_len"][lo:hi]
        ex.send_recs[:n] = torch.from_numpy(recs.view(np.uint8).reshape(n, 32))
        blob = np.frombuffer(b"".join(names), dtype=np.uint8)
        ex.send_names[:blob.shape[0]] = torch.from_numpy(blob.copy())
        off = np.zeros(n + 1, dtype=np.int64)
        np.cumsum([len(x) for x in names], out=off[1:])
        ex.send_off[:n + 1] = torch.from_numpy(off)
        g = ex.gather()

        # every rank now holds every passing record, addressable by its global index, with its name
        allrec = g.recs.numpy().reshape(-1).view(REC_DTYPE)
        got = {}
        order = np.argsort((allrec["contig"].astype(np.int64) << 32) | allrec["rec_idx"], kind="stable")
        for i in order:
            r = allrec[i]
            if r["flags"] & 1:
                o = int(g.name_index[int(r["rec_idx"])])
                nm = bytes(g.names.numpy()[o:o + int(r["name_len"])]).decode()
                got[nm] = (targets[int(r["contig"])], int(r["start"]), int(r["end"]), int(r["qlen"]))
        want, hq = O.bam_file_dict(stream, offs, targets, targets, 30, 50, 0.1, 0.9)
        assert got == want, "gathered record set differs from the single-process dict"

        # contig-sharded depth == unsharded depth on the owned contigs; totals via integer all-reduce
        owner = shard.lpt_assign([l for _, l in contigs], world)
        cmap, mine = shard.contig_map_for(owner, rank)
        full = O.depth_build(O.name_join([want], hq, 0.9), dict(contigs), 15)
        local = O.depth_build({k: v for k, v in got.items() if cmap[targets.index(v[0])] >= 0},
                              {targets[c]: contigs[c][1] for c in mine}, 15)
        for c in mine:
            assert np.array_equal(local[targets[c]], full[targets[c]])
        s, nb = shard.allreduce_totals(sum(int(v.sum()) for v in local.values()), sum(contigs[c][1] for c in mine),
                                       torch.device("cpu"))
        assert s == sum(int(v.sum()) for v in full.values()) and nb == sum(l for _, l in contigs)
        issues = [(c, s0, e0) for c in mine for s0, e0 in O.collapse_contig(local[targets[c]], -1, 0, 15, 0)]
        allissues = shard.gather_interval_lists(issues)
        ref_issues = [(targets.index(t), s0, e0) for t, v in O.collapse_depth_range(full, -1, 0, 15, 0).items() for s0, e0 in v]
        assert allissues == sorted(ref_issues)
        # ---- NameCheck glue (all-to-all of padded hash buckets) with numpy stand-ins for the two kernels ----------
        def bucket_np(recs_t, n_parts, cap, out):
            rr = recs_t.numpy().reshape(-1).view(REC_DTYPE)
            o = out.view(n_parts, cap + 1)
            o[:, 0] = 0
            for h in rr["name_hash"][(rr["flags"] & 1) == 1].tolist():
                d = (h >> 33) % n_parts
                k = int(o[d, 0]); o[d, 0] += 1
                if k < cap:
                    o[d, 1 + k] = int(np.uint64(h).astype(np.int64))

        def conflicts_np(buckets, n_parts, cap, out_n):
            bb = buckets.view(n_parts, cap + 1)
            seen, n = {}, 0
            for src in range(n_parts):
                cnt = int(bb[src, 0])
                if cnt > cap:
                    n += 1
                for h in bb[src, 1:1 + min(cap, cnt)].tolist():
                    if h in seen and seen[h] != src:
                        n += 1
                    seen.setdefault(h, src)
            out_n[0] += n

        local = torch.from_numpy(recs.view(np.uint8).reshape(n, 32).copy())
        chk = shard.NameCheck(n, torch.device("cpu"), bucket_np, conflicts_np)
        got_conf = chk.conflicts(local)
        # truth: names held (as passing records) by both ranks' slices
        mine = {nm for nm, f in zip(names, recs["flags"]) if f & 1}
        allsets = [None, None]
        dist.all_gather_object(allsets, mine)
        assert (got_conf > 0) == (len(allsets[0] & allsets[1]) > 0), (got_conf, len(allsets[0] & allsets[1]))
        # and with disjoint name sets the verdict is "no conflict"
        uniq = recs.copy()
        uniq["name_hash"] = name_hash_np([b"rank%d/%d" % (rank, i) for i in range(n)])
        assert chk.conflicts(torch.from_numpy(uniq.view(np.uint8).reshape(n, 32).copy())) == 0
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_world2_gloo_exchange_and_sharding():
    from oracle import gci_oracle
    gci_oracle.build()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(60)
    for rank, msg in res:
        assert msg == "ok", "rank %d: %s" % (rank, msg)


# ---- the name-hash-sharded join (shard.ShardedJoin) with numpy stand-ins for the five device operations -------------------

class _NumpyOps:
    """What an Engine does for ShardedJoin, restated with numpy / the oracle: the bucket layouts of include/gci_hip.h."""

    def __init__(self, O, targets, rec_dtype):
        self.O, self.targets, self.dt = O, targets, rec_dtype

    def route_records(self, ji, n_parts, cap, out_recs, out_names, slot, status):
        recs = ji.recs.numpy().reshape(-1).view(self.dt)
        o = out_recs.numpy().reshape(-1).view(self.dt).reshape(n_parts, cap + 1)
        names = out_names.numpy().reshape(n_parts, cap, slot)
        names[:] = 0
        cnt = [0] * n_parts
        base, off = ji.name_base.numpy(), ji.name_off.numpy()
        for i, r in enumerate(recs):
            if not (r["flags"] & 1):
                continue
            d = (int(r["name_hash"]) >> 33) % n_parts
            k = cnt[d]
            cnt[d] += 1
            if k < cap:
                o[d, 1 + k] = r
                o[d, 1 + k]["flags"] |= 4
                a = int(off[i]) + ji.name_delta
                names[d, k, :int(r["name_len"])] = base[a:a + int(r["name_len"])]
        for d in range(n_parts):
            o[d, 0] = np.zeros((), dtype=self.dt)
            o[d, 0]["name_hash"], o[d, 0]["contig"] = cnt[d], -1
        status[0] = -1 if max(cnt) <= cap else 8

    def route_seal_records(self, recs, n_parts, cap, status):
        o = recs.numpy().reshape(-1).view(self.dt).reshape(n_parts, cap + 1)
        status[0] = -1
        for d in range(n_parts):
            c = int(o[d, 0]["name_hash"])
            o[d, 0]["flags"] = 0
            o[d, 1 + min(c, cap):]["flags"] = 0

    def name_join(self, inputs, ovlp, contig_map, out, count, check, count_flank, status=None):
        dicts, hq = [], set()
        for ji in inputs:
            recs = ji.recs.numpy().reshape(-1).view(self.dt)
            base, off = ji.name_base.numpy(), ji.name_off.numpy()
            order = sorted((int(r["contig"]), i) for i, r in enumerate(recs) if r["flags"] & 1)
            d = {}
            for _, i in order:
                r = recs[i]
                a = int(off[i]) + ji.name_delta
                q = bytes(base[a:a + int(r["name_len"])]).decode()
                d[q] = (self.targets[int(r["contig"])], int(r["start"]), int(r["end"]), int(r["qlen"]))
                if r["flags"] & 2:
                    hq.add(q)
            dicts.append(d)
        res = self.O.name_join(dicts, hq, ovlp)
        o = out.numpy()
        for k, (t, s, e) in enumerate(res.values()):
            o[k] = (self.targets.index(t), s, e, 0)
        count[0] = len(res)
        if status is not None:
            status[0] = -1

    def route_intervals(self, ivl, count, owner, n_parts, cap, out, status):
        o = out.numpy().reshape(n_parts, cap + 1, 4)
        cnt = [0] * n_parts
        for c, s, e, _ in ivl.numpy()[:int(count[0])].tolist():
            d = int(owner[c])
            if cnt[d] < cap:
                o[d, 1 + cnt[d]] = (c, s, e, 0)
            cnt[d] += 1
        for d in range(n_parts):
            o[d, 0] = (-1, cnt[d], 0, 0)
        status[0] = -1 if max(cnt) <= cap else 8

    def route_seal_intervals(self, ivl, n_parts, cap, cmap, status):
        o = ivl.numpy().reshape(n_parts, cap + 1, 4)
        status[0] = -1
        for d in range(n_parts):
            c = int(o[d, 0, 1])
            for k in range(1, cap + 1):
                o[d, k, 0] = int(cmap[o[d, k, 0]]) if k <= min(c, cap) else -1


def _worker_sharded_join(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gci_amd import shard, synth
        from gci_amd.device import REC_DTYPE, JoinInput, name_hash_np
        from oracle import gci_oracle as O
        contigs = (("a", 260_000), ("b", 200_000), ("c", 150_000), ("d", 60_000))
        targets = [n for n, _ in contigs]
        base = synth.simulate_reads(contigs, 12, "hifi", seed=91)
        sets = [base, synth.perturb(base, 92), synth.perturb(base, 93)]          # three files: reads move between contigs / ranks
        owner = shard.lpt_assign([l for _, l in contigs], world)
        ops = _NumpyOps(O, targets, REC_DTYPE)
        local, dicts, hq = [], [], set()
        for rs in sets:
            stream, offs = synth.to_bam_stream(rs)
            a = O.bam_filter_arrays(stream, offs, np.arange(len(targets), dtype=np.int32), 30, 50, 0.1, 0.9)
            d, h = O.bam_file_dict(stream, offs, targets, targets, 30, 50, 0.1, 0.9)
            dicts.append(d)
            hq |= h
            mine = np.flatnonzero(np.asarray(owner)[np.clip(a["contig"], 0, len(targets) - 1)] == rank)    # the records of this rank's contigs
            n = mine.shape[0]
            names = [bytes(stream[int(o):int(o) + int(l)]) for o, l in zip(a["name_off"][mine], a["name_len"][mine])]
            recs = np.zeros(n, dtype=REC_DTYPE)
            recs["name_hash"] = name_hash_np(names)
            for f in ("contig", "start", "end", "qlen", "name_len"):
                recs[f] = a[f][mine]
            recs["flags"] = a["passed"][mine] | (a["hq"][mine] << 1)
            recs["rec_idx"] = np.arange(n)
            local.append(JoinInput(torch.from_numpy(recs.view(np.uint8).reshape(n, 32).copy()), torch.from_numpy(stream.copy()),
                                   torch.from_numpy(a["name_off"][mine].astype(np.int64)), 0))
        sj = shard.ShardedJoin(ops, [int(ji.recs.shape[0]) for ji in local], owner, torch.device("cpu"))
        inputs = [sj.exchange_file(f, ji) for f, ji in enumerate(local)]
        ivl, n_slots = sj.join(inputs, 0.9)
        assert n_slots == ivl.shape[0] and sj.bytes_per_step() > 0
        sj.check(lambda w, what: None if w == (1 << 64) - 1 else (_ for _ in ()).throw(AssertionError((what, w))))
        cmap, mine_c = shard.contig_map_for(owner, rank)
        got = sorted((mine_c[c], s, e) for c, s, e, _ in ivl.numpy().tolist() if c >= 0)
        want = sorted((targets.index(t), s, e) for t, s, e in O.name_join(dicts, hq, 0.9).values() if owner[targets.index(t)] == rank)
        assert got == want and len(want) > 50, (len(got), len(want))
        q.put((rank, "ok"))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_world2_gloo_name_hash_sharded_join():
    """shard.ShardedJoin over gloo, two ranks: records routed by name hash, joined where their name is owned, intervals routed
    to the owner of their contig -- every rank ends with exactly the single-process join's intervals on ITS contigs."""
    from oracle import gci_oracle
    gci_oracle.build()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker_sharded_join, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(60)
    for rank, msg in res:
        assert msg == "ok", "rank %d: %s" % (rank, msg)
