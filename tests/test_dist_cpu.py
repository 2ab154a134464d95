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
