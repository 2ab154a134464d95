#!/usr/bin/env python3
"""BASELINE configs[2] geometry end to end on one MI355X: CHM13 (25 contigs, 3.117 Gb), two 40x HiFi alignment files of
the same reads (the second one perturbed as another aligner would: exercises the `-op` join), as heads streams
(records without SEQ / QUAL -- the whole inflated files would be 2 x 190 GB): K1 x 2 -> join -> depth + text + sums +
issue runs.  Prints per-kernel HIP-event times, the wall time of the device step and two invariants.
Usage: exp_genome_full.py [scale]   (scale < 1 shrinks every contig; default 1.0)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gci_amd import synth, _lib
from gci_amd.device import Engine, JoinInput
from gci_amd.formats import bam as bamfmt

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
contigs = tuple((n, max(20_000, int(l * scale))) for n, l in synth.CHM13)
lens = np.array([l for _, l in contigs], dtype=np.int64)
names = [n for n, _ in contigs]
# generate group by group (a read set of the whole genome would need ~60 GB of host memory)
groups, cur, acc = [], [], 0
for i, (n, l) in enumerate(contigs):
    if cur and acc + l > 400_000_000 * max(scale, 0.05):
        groups.append(cur); cur, acc = [], 0
    cur.append(i); acc += l
groups.append(cur)
hdr = np.frombuffer(bamfmt.encode_header(names, [int(x) for x in lens]), dtype=np.uint8)
parts = [[hdr], [hdr]]
offs = [[], []]
size = [int(hdr.shape[0]), int(hdr.shape[0])]
aligned = 0
t0 = time.time()
for g, idx in enumerate(groups):
    sub = tuple(contigs[i] for i in idx)
    rs = synth.simulate_reads(sub, 40, "hifi", seed=synth.seed_for(3, 0) + 7 * g, name_prefix="m64011_g%02d/" % g)
    files = [rs, synth.perturb(rs, synth.seed_for(3, 1) + 7 * g)]
    for f, r in enumerate(files):
        aligned += int(r.ref_span()[(r.flag & 4) == 0].sum())
        r.ref_id = (r.ref_id + idx[0]).astype(np.int32)          # the group's contigs are consecutive in the header
        r.contigs = contigs
        s, o = synth.to_bam_stream(r, heads=True)
        first = bamfmt.parse_header(s).first_record
        parts[f].append(s[first:])
        offs[f].append(o - np.uint64(first) + np.uint64(size[f]))
        size[f] += int(s.shape[0]) - first
    print("group %d/%d: %d contigs, %d reads, %.0f s" % (g + 1, len(groups), len(idx), len(rs), time.time() - t0), file=sys.stderr, flush=True)
    del rs, files
streams = [np.concatenate(p) for p in parts]
offsets = [np.concatenate(o) for o in offs]
del parts, offs

e = Engine(0)
e.set_layout([int(x) for x in lens])
ref_sel = e.to_device(np.arange(len(contigs), dtype=np.int32))
d = [(e.to_device(s), e.to_device(o)) for s, o in zip(streams, offsets)]
track = e.new_track()

def step():
    ins = []
    for d_s, d_o in d:
        recs = e.bam_filter(d_s, d_o, ref_sel, 30, 50, 0.1, 0.9, heads=True, check=False)
        ins.append(JoinInput(recs, d_s, d_o, 36))
    ivl, cnt = e.name_join(ins, 0.9, count_flank=15)
    fused = e.depth_build_fused(ivl, cnt, 15, track, want_text=True, want_sums=True, issue=(-1.0, 0.0, 15), counted=True)
    return ivl, cnt, fused

ivl, cnt, fused = step()                                   # warm-up: scratch allocation
torch.cuda.synchronize()
e.profile_enable((1 << _lib.PROF_COUNT) - 1); e.profile_read()
walls = []
for _ in range(3):
    torch.cuda.synchronize(); t = time.perf_counter()
    ivl, cnt, fused = step()
    torch.cuda.synchronize(); walls.append(time.perf_counter() - t)
pr = {k: round(ms / n * 1e3, 1) for k, (ms, n) in e.profile_read().items()}
e.profile_enable(0)
K = int(cnt.item())
iv = ivl[:K].cpu().numpy().astype(np.int64)
L = lens[iv[:, 0]]
a = np.clip(iv[:, 1] + 15, 0, L); b = np.clip(iv[:, 2] - 15 + 1, 0, L)
want_sum = int(np.maximum(b - a, 0).sum())
got_sum = int(np.asarray(fused["sums"]).sum())
text_total = int(fused["text_off"][-1])
# .depth.gz members written by the device (gci_depth_deflate_*)
import gzip
torch.cuda.synchronize(); t = time.perf_counter()
blobs = e.depth_deflate(track)
torch.cuda.synchronize(); deflate_s = time.perf_counter() - t
small = int(np.argmin(lens)); mid = int(np.argsort(lens)[len(lens) // 3])
off = e.offsets
ok_gz = True
for c in (small, mid):
    d_c = track[off[c]:off[c] + int(lens[c])].cpu().numpy()
    ok_gz = ok_gz and gzip.decompress(blobs[c]) == ("\n".join(map(str, d_c.tolist())) + "\n").encode()
out = {"scale": scale, "contigs": len(contigs), "bases": int(lens.sum()), "records_per_file": [int(o.shape[0]) for o in offsets],
       "heads_bytes_per_file": [int(s.shape[0]) for s in streams], "aligned_bases_both_files": aligned, "intervals_after_join": K,
       "us_per_launch": pr, "kernel_sum_ms": round(sum(pr.values()) / 1e3, 3),
       "step_wall_ms": [round(w * 1e3, 2) for w in walls],
       "aligned_Gbases_per_s": round(aligned / min(walls) / 1e9, 1),
       "sum_depth_equals_sum_of_clipped_intervals": got_sum == want_sum, "sum_depth": got_sum, "text_bytes": text_total,
       "depth_gz_members_bytes": int(sum(len(x) for x in blobs)), "depth_gz_device_s": round(deflate_s, 4),
       "depth_gz_two_contigs_decompress_to_the_track": bool(ok_gz),
       "issue_runs": int(sum(len(r) for r in fused["runs"])) if fused["runs"] is not None else None}
print(json.dumps(out))
