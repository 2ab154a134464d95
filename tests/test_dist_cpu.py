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
        # ONE all-to-all for the records and names of every file (exchange_files) -- and the per-file form gives the same arrays
        counted = {"n": 0}
        real = shard.all_to_all_bytes

        def counting(*a, **kw):
            counted["n"] += 1
            return real(*a, **kw)
        shard.all_to_all_bytes = counting
        try:
            inputs = sj.exchange_files(local)
            assert counted["n"] == 1
            ivl, n_slots = sj.join(inputs, 0.9)
            assert counted["n"] == 2 and shard.ShardedJoin.COLLECTIVES_PER_STEP == 3      # (+ the all-reduce of the sums)
        finally:
            shard.all_to_all_bytes = real
        one_by_one = shard.ShardedJoin(ops, [int(ji.recs.shape[0]) for ji in local], owner, torch.device("cpu"))
        for f, ji in enumerate(local):
            x = one_by_one.exchange_file(f, ji)
            assert torch.equal(x.recs, inputs[f].recs) and torch.equal(x.name_base, inputs[f].name_base)
        assert n_slots == ivl.shape[0] and sj.bytes_per_step() > 0
        sj.check(lambda w, what: None if w == (1 << 64) - 1 else (_ for _ in ()).throw(AssertionError((what, w))))
        cmap, mine_c = shard.contig_map_for(owner, rank)
        got = sorted((mine_c[c], s, e) for c, s, e, _ in ivl.numpy().tolist() if c >= 0)
        want = sorted((targets.index(t), s, e) for t, s, e in O.name_join(dicts, hq, 0.9).values() if owner[targets.index(t)] == rank)
        assert got == want and len(want) > 50, (len(got), len(want))
        # capacity of what the join may emit: the routed buckets + the rows a caller passes beside them (PAF records), and never
        # fewer rows than its inputs have (intervals past the array were dropped without a word)
        more = shard.ShardedJoin(ops, [int(ji.recs.shape[0]) for ji in local], owner, torch.device("cpu"), extra_records=12345)
        assert more.ivl.shape[0] == sj.ivl.shape[0] + 12345
        sj.ivl = sj.ivl[:1].clone()
        ivl2, _ = sj.join(inputs, 0.9)
        sj.check(lambda w, what: None if w == (1 << 64) - 1 else (_ for _ in ()).throw(AssertionError((what, w))))
        assert sorted((mine_c[c], s, e) for c, s, e, _ in ivl2.numpy().tolist() if c >= 0) == want
        q.put((rank, "ok"))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_world2_gloo_name_hash_sharded_join(world):
    """shard.ShardedJoin over gloo, two and three ranks: records routed by name hash, joined where their name is owned, intervals routed
    to the owner of their contig -- every rank ends with exactly the single-process join's intervals on ITS contigs."""
    from oracle import gci_oracle
    gci_oracle.build()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() + 13 * world) % 2000
    procs = [ctx.Process(target=_worker_sharded_join, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(60)
    for rank, msg in res:
        assert msg == "ok", "rank %d: %s" % (rank, msg)


# ---- PAF files by byte range (shard.paf_by_byte_range) with plain-Python stand-ins for the three device operations ---------------

HIT_DTYPE = np.dtype([("qn_off", "<u8"), ("qhash", "<u8"), ("qlen", "<i8"), ("qs", "<i8"), ("qe", "<i8"), ("ts", "<i8"), ("te", "<i8"),
                      ("identity", "<f8"), ("qn_len", "<u4"), ("t", "<i4"), ("hq", "<u4"), ("slot", "<u4")])
assert HIT_DTYPE.itemsize == 80


class _PafOps:
    """gci_paf_hits_device / gci_route_hits / gci_paf_score_device restated over numpy + tests/paf_ref's scoring rules."""
    PAF_HIT_BYTES = 80

    def __init__(self):
        from gci_amd.device import REC_DTYPE, name_hash_np
        self.REC_DTYPE, self.hash = REC_DTYPE, name_hash_np

    def to_device(self, a):
        return torch.from_numpy(np.ascontiguousarray(a).copy())

    def paf_hits_text(self, d_text, ends, targets, map_qual, mq_cutoff, iden_percent):
        import re
        text = d_text.numpy().tobytes()
        out, lo = [], 0
        for hi in [int(e) for e in ends]:
            rows = []
            for m in re.finditer(rb"[^\r\n]*(?:\r\n|\r|\n|\Z)", text[lo:hi]):
                if m.end() == m.start():
                    continue
                raw = m.group(0)
                line = raw.decode().strip()
                col = line.split("\t")
                if col[5] not in targets:
                    continue
                ident = int(col[9]) / int(col[10])
                mapq = int(col[11])
                if mapq >= map_qual and ident >= iden_percent:
                    lead = len(raw) - len(raw.lstrip(b" \t\n\r\x0b\x0c"))
                    q = col[0].encode()
                    rows.append((lo + m.start() + lead, int(self.hash([q])[0]), int(col[1]), int(col[2]), int(col[3]), int(col[7]), int(col[8]),
                                 ident, len(q), targets.index(col[5]), 1 if mapq >= mq_cutoff else 0, 0))
            a = np.array(rows, dtype=HIT_DTYPE) if rows else np.zeros(0, dtype=HIT_DTYPE)
            out.append(torch.from_numpy(a.view(np.uint8).reshape(-1, 80).copy()))
            lo = hi
        return out

    def route_hits(self, hits, d_name_base, n_parts, cap, out_hits, out_names, name_slot, status):
        h = hits.numpy().reshape(-1).view(HIT_DTYPE) if int(hits.shape[0]) else np.zeros(0, dtype=HIT_DTYPE)
        base = d_name_base.numpy()
        oh = out_hits.numpy().reshape(-1).view(HIT_DTYPE).reshape(n_parts, cap + 1)
        on = out_names.numpy().reshape(n_parts, cap, name_slot)
        oh[:] = 0
        on[:] = 0
        dest = (h["qhash"] >> np.uint64(33)) % np.uint64(n_parts)
        for d in range(n_parts):
            idx = np.flatnonzero(dest == d)                      # stable: line order kept
            oh[d, 0]["qhash"] = idx.shape[0]
            oh[d, 0]["t"] = -1
            if idx.shape[0] > cap:
                status[0] = 8
                idx = idx[:cap]
            oh[d, 1:1 + idx.shape[0]] = h[idx]
            for k, i in enumerate(idx):
                o, l = int(h["qn_off"][i]), int(h["qn_len"][i])
                on[d, k, :l] = base[o:o + l]

    def paf_score_hits(self, d_names, d_hits, upto, targets):
        from gci_amd.device import JoinInput
        from gci_amd.pipeline import _merge_span
        names = d_names.numpy()
        h = d_hits.numpy().reshape(-1).view(HIT_DTYPE)
        blocks, hq, where = {}, set(), {}
        out = []
        for f in range(len(upto) - 1):
            for i in range(upto[f], upto[f + 1]):
                o, l = int(h["qn_off"][i]), int(h["qn_len"][i])
                q = names[o:o + l].tobytes()
                where.setdefault(q, o)
                blocks.setdefault(q, {}).setdefault(int(h["t"][i]), []).append(
                    (int(h["qlen"][i]), int(h["qs"][i]), int(h["qe"][i]), int(h["ts"][i]), int(h["te"][i]), float(h["identity"][i])))
                if h["hq"][i]:
                    hq.add(q)
            recs = np.zeros(len(blocks), dtype=self.REC_DTYPE)
            off = np.zeros(len(blocks), dtype=np.int64)
            for k, (q, by_t) in enumerate(blocks.items()):
                best_key, best = None, None
                for t, alns in by_t.items():
                    covered, _, _ = _merge_span([(a[1], a[2]) for a in alns])
                    total = 0
                    for a in alns:
                        total = total + a[5]
                    key = (total / len(alns) * (covered / alns[0][0]), targets[t])
                    if best_key is None or key > best_key:
                        _, s, e = _merge_span([(a[3], a[4]) for a in alns])
                        best_key, best = key, (t, s, e, alns[0][0])
                recs[k]["name_hash"] = int(self.hash([q])[0])
                recs[k]["contig"], recs[k]["start"], recs[k]["end"], recs[k]["qlen"] = best
                recs[k]["rec_idx"], recs[k]["name_len"] = k, len(q)
                off[k] = where[q]
            recs["flags"] = 1
            out.append((JoinInput(torch.from_numpy(recs.view(np.uint8).reshape(-1, 32).copy()), d_names, torch.from_numpy(off), 0), set(hq)))
        return out


def _worker_paf_ranges(rank, world, port, q, paths, newline):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gci_amd import shard
        from gci_amd.device import REC_DTYPE
        import paf_ref

        def all_reduce_max(values):
            t = torch.tensor([int(v) for v in values], dtype=torch.int64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return [int(x) for x in t.tolist()]
        targets = ["chr1", "chr2", "chr3"]
        ops = _PafOps()
        got = shard.paf_by_byte_range(ops, paths, targets, 30, 50, 0.9, world, rank, all_reduce_max, torch.device("cpu"), via_host=True)
        want, want_hq = paf_ref.paf_filter_py(paths, targets, 30, 50, 0.9)
        assert got is not None and len(got) == len(paths)
        n_mine = 0
        for f, (ji, hq) in enumerate(got):
            recs = ji.recs.numpy().reshape(-1).view(REC_DTYPE)
            names = ji.name_base.numpy()
            mine = {}
            for r in recs:
                o = int(ji.name_off[int(r["rec_idx"])])
                qn = names[o:o + int(r["name_len"])].tobytes().decode()
                assert (int(r["name_hash"]) >> 33) % world == rank            # every query on the rank that owns its name
                mine[qn] = (targets[int(r["contig"])], int(r["start"]), int(r["end"]), int(r["qlen"]))
            owned = {k: v for k, v in want[f].items() if (int(ops.hash([k.encode()])[0]) >> 33) % world == rank}
            assert mine == owned, (f, len(mine), len(owned))
            n_mine += len(mine)
        assert n_mine > 20
        q.put((rank, "ok"))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,newline", [(2, "\n"), (3, "\r\n"), (2, "\r")])
def test_gloo_paf_files_by_byte_range(tmp_path, world, newline):
    """shard.paf_by_byte_range over gloo: every rank parses the lines that start in its byte range of two PAF files (queries with
    lines in both halves of a file and in both files), the hits travel to the rank that owns the query -- the union over the ranks
    is exactly the whole-file filter's per-file dicts (tests/paf_ref.py), each query on the rank (name hash >> 33) % world."""
    rng = np.random.default_rng(5)
    targets = ["chr1", "chr2", "chr3", "other"]
    paths = []
    for f in range(2):
        lines = []
        for i in range(400):
            qn = "read%d/%d" % (f * 150 + int(rng.integers(0, 300)), int(rng.integers(0, 3)))      # repeats inside and across the files
            qlen = int(rng.integers(5000, 20000))
            qs = int(rng.integers(0, qlen // 2)); qe = int(rng.integers(qs + 1, qlen))
            ts = int(rng.integers(0, 100000)); te = ts + (qe - qs)
            alen = qe - qs + int(rng.integers(0, 50)); nm = int(alen * rng.uniform(0.85, 1.0))
            lines.append("\t".join(map(str, [qn, qlen, qs, qe, "+", targets[int(rng.integers(0, 4))], 200000, ts, te, nm, alen,
                                              int(rng.integers(0, 61)), "tp:A:P"])))
        p = tmp_path / ("f%d.paf" % f)
        p.write_bytes((newline.join(lines) + (newline if f == 0 else "")).encode())
        paths.append(str(p))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() + world * 7 + len(newline)) % 2000
    procs = [ctx.Process(target=_worker_paf_ranges, args=(r, world, port, q, paths, newline)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(60)
    for rank, msg in res:
        assert msg == "ok", "rank %d: %s" % (rank, msg)


def test_line_starts_for_byte_ranges():
    from gci_amd import shard
    raw = np.frombuffer(b"ab\ncd\r\nef\rgh\n", dtype=np.uint8)
    starts = [0, 3, 7, 10]
    for pos in range(len(raw) + 1):
        want = min([s for s in starts if s >= pos] + [len(raw)])
        assert shard.line_start_at_or_after(raw, pos) == want, pos
    for world in (1, 2, 3, 5, 20):
        cuts = [shard.byte_range_of_rank(raw, r, world) for r in range(world)]
        assert cuts[0][0] == 0 and cuts[-1][1] == len(raw)
        assert all(a[1] == b[0] for a, b in zip(cuts, cuts[1:])) and all(lo in starts + [len(raw)] for lo, _ in cuts)


def _bytes_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ["RANK"], os.environ["WORLD_SIZE"], os.environ["LOCAL_RANK"] = str(rank), str(world), "0"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gci_amd import shard
        ctx = shard.Context.__new__(shard.Context)           # (the collectives only: no device, no second init)
        ctx.rank, ctx.world, ctx.backend, ctx.device_index = rank, world, "gloo", 0            # (root is a property: rank == 0)
        rng = np.random.default_rng(rank)
        # rank r sends r + (r % 2) blobs of sizes 0 .. 70 000 (an empty list, empty blobs, one larger than the others' totals)
        mine = [rng.integers(0, 256, int(s), dtype=np.uint8) for s in ([0, 70_000, 5][:rank + (rank % 2)] if rank else [])]
        got = ctx.gather_bytes_to_root([m.tobytes() if k % 2 else m for k, m in enumerate(mine)])
        if rank == 0:
            ok = len(got) == world
            for r in range(world):
                want = [np.random.default_rng(r).integers(0, 256, int(s), dtype=np.uint8) for s in ([0, 70_000, 5][:r + (r % 2)] if r else [])]
                rg = np.random.default_rng(r)
                want = [rg.integers(0, 256, int(s), dtype=np.uint8) for s in ([0, 70_000, 5][:r + (r % 2)] if r else [])]
                ok = ok and len(got[r]) == len(want) and all(np.array_equal(a, b) for a, b in zip(got[r], want))
            q.put(bool(ok))
        else:
            q.put(got is None)
    finally:
        dist.destroy_process_group()


def test_members_travel_as_one_sized_gather_of_bytes():
    """shard.Context.gather_bytes_to_root (how the .depth.gz members of a contig-sharded run reach rank 0): ragged lists of blobs,
    an empty list, empty blobs -- rank 0 gets every rank's list byte for byte, the others None; nothing pickled."""
    world, port = 3, 29877
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    procs = [ctxm.Process(target=_bytes_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(60)
    assert all(res), res
