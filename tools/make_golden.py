#!/usr/bin/env python3
"""Generate tests/golden/ by running the UNMODIFIED reference (/root/reference/GCI.py) in this
container.  Only data is written: synthetic inputs made by gci_amd.synth and the
files / return values the reference produced from them.  The reference source is never copied;
this script cannot run on the GPU box (no /root/reference there) and nothing at test time needs it.

    python tools/make_golden.py            # regenerates every case (about a minute)

Two kinds of fixtures:
  tests/golden/<case>/inputs/*, expected/*      end-to-end runs of GCI() on small synthetic inputs
  tests/golden/kats.json                        direct calls of single reference functions
"""
from __future__ import annotations

import contextlib
import gzip
import hashlib
import io
import json
import os
import shutil
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

import load_reference  # noqa: E402
from gci_amd import synth  # noqa: E402
from gci_amd.formats import paf as paffmt  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def write_inputs(case_dir, contigs, files, gaps=None, regions=None):
    """files: {name: ReadSet | list of PAF lines}"""
    inp = os.path.join(case_dir, "inputs")
    os.makedirs(inp, exist_ok=True)
    synth.write_reference_fasta(os.path.join(inp, "ref.fa"), contigs, gaps)
    for name, obj in files.items():
        path = os.path.join(inp, name)
        if name.endswith(".bam"):
            synth.write_bam_file(path, obj, level=9, threads=4)
        else:
            paffmt.write(path, obj)
    if regions:
        with open(os.path.join(inp, "regions.bed"), "w") as f:
            for t, s, e in regions:
                f.write(f"{t}\t{s}\t{e}\n")
    return inp


def run_reference(case, args, hifi=None, nano=None, regions=False):
    ref = load_reference.load()
    case_dir = os.path.join(GOLDEN, case)
    inp = os.path.join(case_dir, "inputs")
    exp = os.path.join(case_dir, "expected")
    shutil.rmtree(exp, ignore_errors=True)
    os.makedirs(exp)
    tmp = tempfile.mkdtemp(prefix="gci_golden_")
    kw = dict(hifi=[os.path.join(inp, f) for f in hifi] if hifi else None,
              nano=[os.path.join(inp, f) for f in nano] if nano else None,
              directory=tmp, prefix="GCI", reference=os.path.join(inp, "ref.fa"),
              regions=os.path.join(inp, "regions.bed") if regions else None, threads=1, force=True)
    kw.update(args)
    out = io.StringIO()
    with contextlib.redirect_stdout(out):
        ref.GCI(**kw)
    # R14: what GCI() printed, with the two directories that differ between runs replaced by placeholders
    transcript = out.getvalue().replace(tmp, "{OUT}").replace(inp, "{IN}")
    manifest = {"args": {k: v for k, v in args.items()}, "hifi": hifi, "nano": nano, "regions": bool(regions),
                "files": {}, "stdout": transcript}
    for fn in sorted(os.listdir(tmp)):
        src = os.path.join(tmp, fn)
        if not os.path.isfile(src):
            continue
        if fn.endswith(".depth.gz"):
            text = gzip.open(src, "rb").read()
            manifest["files"][fn] = {"sha256_decompressed": hashlib.sha256(text).hexdigest(), "bytes": len(text)}
            with gzip.GzipFile(os.path.join(exp, fn), "wb", 9, mtime=0) as g:      # reproducible container
                g.write(text)
        else:
            shutil.copy(src, os.path.join(exp, fn))
            manifest["files"][fn] = {"sha256": hashlib.sha256(open(src, "rb").read()).hexdigest()}
    if os.path.isdir(os.path.join(tmp, "images")):              # -p: the figures
        os.makedirs(os.path.join(exp, "images"))
        for fn in sorted(os.listdir(os.path.join(tmp, "images"))):
            shutil.copy(os.path.join(tmp, "images", fn), os.path.join(exp, "images", fn))
            manifest["files"]["images/" + fn] = {"sha256": hashlib.sha256(open(os.path.join(tmp, "images", fn), "rb").read()).hexdigest()}
    with open(os.path.join(case_dir, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    shutil.rmtree(tmp)
    print(case, "->", ", ".join(manifest["files"]))


def case_single_bam():
    contigs = (("ctg1", 300_000),)
    rs = synth.simulate_reads(contigs, 20, "hifi", seed=synth.seed_for(1, 0))
    write_inputs(os.path.join(GOLDEN, "c1_single_bam"), contigs, {"hifi.mm2.bam": rs})
    run_reference("c1_single_bam", {}, hifi=["hifi.mm2.bam"])


def case_two_bam():
    contigs = (("chrA", 220_000), ("chrB", 90_000), ("chrC", 9_000))
    a = synth.simulate_reads(contigs, 22, "hifi", seed=synth.seed_for(3, 0))
    b = synth.perturb(a, synth.seed_for(3, 1))
    write_inputs(os.path.join(GOLDEN, "c3_two_bam"), contigs, {"hifi.mm2.bam": a, "hifi.wm2.bam": b})
    run_reference("c3_two_bam", {}, hifi=["hifi.mm2.bam", "hifi.wm2.bam"])


def case_three_bam_chrs():
    contigs = (("chrA", 150_000), ("chrB", 120_000), ("chrC", 60_000))
    a = synth.simulate_reads(contigs, 18, "hifi", seed=synth.seed_for(3, 10))
    b = synth.perturb(a, synth.seed_for(3, 11))
    c = synth.perturb(a, synth.seed_for(3, 12))
    write_inputs(os.path.join(GOLDEN, "c3_three_bam_chrs"), contigs, {"a.bam": a, "b.bam": b, "c.bam": c})
    run_reference("c3_three_bam_chrs", {"chrs": "chrA,chrC", "map_qual": 20, "mq_cutoff": 45, "ovlp_percent": 0.95,
                                        "threshold": 1}, hifi=["a.bam", "b.bam", "c.bam"])


def case_paf_bam():
    contigs = (("chrA", 200_000), ("chrB", 80_000))
    a = synth.simulate_reads(contigs, 20, "hifi", seed=synth.seed_for(4, 0))
    b = synth.perturb(a, synth.seed_for(4, 1))
    write_inputs(os.path.join(GOLDEN, "c4_paf_bam"), contigs,
                 {"hifi.wm2.bam": a, "hifi.mm2.paf": synth.to_paf_lines(b, synth.seed_for(4, 2), split_frac=0.1)})
    run_reference("c4_paf_bam", {}, hifi=["hifi.mm2.paf", "hifi.wm2.bam"])


def case_two_paf():
    """Two PAF files + one BAM in one read type: pins the reference's quirk that `synteny` is created once, before
    the per-file loop (GCI.py:214-215), so the second PAF re-emits every query and block of the first."""
    contigs = (("chrA", 160_000), ("chrB", 70_000))
    a = synth.simulate_reads(contigs, 18, "hifi", seed=synth.seed_for(4, 20))
    b = synth.perturb(a, synth.seed_for(4, 21))
    c = synth.perturb(a, synth.seed_for(4, 22))
    write_inputs(os.path.join(GOLDEN, "c4_two_paf"), contigs,
                 {"hifi.wm2.bam": a, "hifi.mm2.paf": synth.to_paf_lines(b, synth.seed_for(4, 23), split_frac=0.15),
                  "hifi.other.paf": synth.to_paf_lines(c.take(np.arange(0, len(c), 2)), synth.seed_for(4, 24), split_frac=0.15)})
    run_reference("c4_two_paf", {"mq_cutoff": 40}, hifi=["hifi.mm2.paf", "hifi.other.paf", "hifi.wm2.bam"])


def case_two_type():
    contigs = (("mat_chr1", 180_000), ("pat_chr1", 160_000), ("mat_chr2", 70_000))
    gaps = {"mat_chr1": [(40_000, 40_500), (100_000, 100_001)], "pat_chr1": [(0, 120)], "mat_chr2": [(69_900, 70_000)]}
    h = synth.simulate_reads(contigs, 25, "hifi", seed=synth.seed_for(5, 0))
    h2 = synth.perturb(h, synth.seed_for(5, 1))
    n = synth.simulate_reads(contigs, 30, "ont", seed=synth.seed_for(5, 2), long_cigar_frac=0.0)
    n2 = synth.perturb(n, synth.seed_for(5, 3))
    regions = [("mat_chr1", 1000, 60_000), ("mat_chr1", 39_000, 41_000), ("pat_chr1", 0, 160_000),
               ("mat_chr2", 50_000, 70_000), ("pat_chr1", 100, 300)]
    write_inputs(os.path.join(GOLDEN, "c5_two_type"), contigs,
                 {"hifi.wm2.bam": h, "hifi.mm2.paf": synth.to_paf_lines(h2, synth.seed_for(5, 4), 0.05),
                  "ont.wm2.bam": n, "ont.mm2.bam": n2,
                  "ont.mm2.paf": synth.to_paf_lines(n2, synth.seed_for(5, 5), 0.05)},
                 gaps=gaps, regions=regions)
    run_reference("c5_two_type", {"threshold": 2, "dist_percent": 0.001, "flank_len": 10},
                  hifi=["hifi.wm2.bam", "hifi.mm2.paf"], nano=["ont.mm2.paf", "ont.wm2.bam", "ont.mm2.bam"],
                  regions=True)


def case_nano_only_long_cigar():
    contigs = (("tig", 400_000),)
    n = synth.simulate_reads(contigs, 12, "ont", seed=synth.seed_for(4, 7), long_cigar_frac=0.05)
    write_inputs(os.path.join(GOLDEN, "c4_nano_long_cigar"), contigs, {"ont.bam": n})
    run_reference("c4_nano_long_cigar", {"flank_len": 0}, nano=["ont.bam"])


# ----------------------------------------------------------------------------------------------
# known-answer tests from single reference functions
# ----------------------------------------------------------------------------------------------

def make_kats():
    ref = load_reference.load()
    rng = np.random.default_rng(20250919)
    kats = {}

    # R10 collapse_depth_range
    cases = []
    fixed = [([0] * 10, -1, 0, 2, 0), ([0, 0, 0, 0, 5, 5, 5, 5, 0, 0], -1, 0, 2, 0),
             ([0, 0, 0, 0, 0, 5, 5, 5, 0, 0], -1, 0, 2, 0), ([0, 0, 5, 5, 5, 5, 5, 0, 0, 0], -1, 0, 2, 0),
             ([0, 3, 0, 0, 3, 0], -1, 0, 0, 100), ([0], -1, 0, 0, 0), ([], -1, 0, 0, 0), ([0, 0, 0], -1, 0, 1, 7),
             ([1, 2, 3, 2, 1, 0, 1, 2, 3], 0, 2, 0, 0), ([0] * 5, -1, 0, 3, 0), ([2, 0, 0, 2] * 8, -1, 1.5, 1, 0)]
    for d, lo, hi, fl, sp in fixed:
        cases.append(dict(depth=d, lo=lo, hi=hi, fl=fl, sp=sp))
    for _ in range(60):
        L = int(rng.integers(1, 120))
        d = (rng.random(L) < rng.uniform(0.2, 0.9)).astype(int) * rng.integers(1, 5, L)
        cases.append(dict(depth=d.tolist(), lo=-1, hi=int(rng.integers(0, 3)), fl=int(rng.integers(0, 12)),
                          sp=int(rng.integers(0, 50))))
    for c in cases:
        c["out"] = ref.collapse_depth_range({"t": np.array(c["depth"], dtype=int)}, c["lo"], c["hi"], c["fl"], c["sp"])["t"]
    kats["collapse_depth_range"] = cases

    # R11 / R12 interval algebra
    alg = []
    for _ in range(80):
        L = int(rng.integers(40, 3000))
        fl = int(rng.integers(0, 16))
        n = int(rng.integers(0, 7))
        pts = np.sort(rng.choice(np.arange(fl, max(fl + 2 * n + 2, L - fl)), size=2 * n, replace=False)) if n else []
        segs = [(int(pts[2 * i]), int(pts[2 * i + 1])) for i in range(n)]
        dp = float(rng.choice([0.0, 0.001, 0.005, 0.02, 0.1]))
        explicit = bool(rng.random() < 0.3)
        s, e = (int(rng.integers(0, L // 2)), int(rng.integers(L // 2, L))) if explicit else (None, None)
        alg.append(dict(L=L, fl=fl, segs=segs, dp=dp, start=s, end=e,
                        complement=ref.complement_merged_depth({"t": segs}, {"t": L}, fl, s, e)["t"],
                        merged=[list(x) for x in ref.merge_merged_depth_bed({"t": segs}, {"t": L}, dp, fl, s, e)["t"]]))
    alg.append(dict(L=1000, fl=15, segs=[(15, 40), (100, 120), (130, 140), (900, 985)], dp=0.02, start=None, end=None,
                    complement=ref.complement_merged_depth({"t": [(15, 40), (100, 120), (130, 140), (900, 985)]},
                                                           {"t": 1000}, 15)["t"],
                    merged=[list(x) for x in ref.merge_merged_depth_bed(
                        {"t": [(15, 40), (100, 120), (130, 140), (900, 985)]}, {"t": 1000}, 0.02, 15)["t"]]))
    kats["interval_algebra"] = alg
    n50 = []
    for _ in range(40):
        v = rng.integers(0, 1000, int(rng.integers(0, 12))).tolist()
        n50.append(dict(lengths=v, out=int(ref.compute_n50(v))))
    n50 += [dict(lengths=x, out=int(ref.compute_n50(x))) for x in ([5, 4, 3, 2, 1], [], [3, 3, 2], [0, 0], [-10])]
    kats["compute_n50"] = n50

    # R6 numpy slice semantics (incl. the negative-stop wrap)
    sl = []
    for s, e, fl, L in [(0, 10, 15, 50), (0, 13, 15, 50), (0, 14, 15, 50), (0, 16, 15, 50), (20, 48, 15, 50),
                        (30, 80, 15, 50), (0, 50, 0, 50), (49, 50, 0, 50), (5, 200, 3, 50), (0, 29, 15, 50)]:
        d = np.zeros(L, dtype=int)
        d[s + fl:e - fl + 1] += 1
        sl.append(dict(s=s, e=e, fl=fl, L=L, out=d.tolist()))
    kats["slice_add"] = sl

    # R3 merge_alns_properties + scoring pieces
    mp = []
    for _ in range(60):
        n = int(rng.integers(1, 6))
        alns = []
        for _ in range(n):
            qs = int(rng.integers(0, 900)); qe = qs + int(rng.integers(1, 400))
            ts = int(rng.integers(0, 9000)); te = ts + int(rng.integers(1, 400))
            alns.append((1500, qs, qe, ts, te, float(rng.uniform(0.9, 1.0))))
        mp.append(dict(alns=[list(a) for a in alns], q=list(ref.merge_alns_properties(alns, 1, 2)),
                       t=list(ref.merge_alns_properties(alns, 3, 4)), avg=ref.get_average_identity(alns)))
    kats["merge_alns_properties"] = mp

    # R5 join fold on hand-made dicts (incl. resurrection with three files and qlen of the current file)
    joins = []
    def J(files, hq, op=0.9):
        import copy
        f = copy.deepcopy(files)
        # the join is inlined in filter(); restate its inputs as the per-file dicts and let the reference's
        # own code run by feeding filter()'s local logic through a tiny driver
        return None
    kats["_note_join"] = "R5 is inlined in filter(); it is pinned end-to-end by the c3_* / c4_* / c5_* cases"

    # R13 score formatting
    sc = []
    for on50, en50, on, en in [(266013, 45027022, 65, 1), (259735, 31921180, 850, 12), (100, 100, 1, 1), (0, 100, 3, 1),
                               (50, 100, 0, 1), (31921180, 31921180, 12, 12), (1, 3, 2, 1)]:
        from math import log2
        sc.append(dict(obs_n50=on50, exp_n50=en50, obs_n=on, exp_n=en,
                       out=repr(0 if on == 0 else round(100 * log2(on50 / en50 + 1) / log2(on / en + 1), 4))))
    kats["score_repr"] = sc

    # N3 the -p numeric front-end: sliding_window_average_depth / pre_plot_base (own generator: the cases above keep
    # their random inputs)
    rng3 = np.random.default_rng(20250928)
    sw = []
    fixed = [([3, 3, 0, 0, 2, 2, 2, 2, 2, 0, 1], 2, 10.0, 0), ([0] * 6, 2, 5.0, 3), ([4] * 10, 3, 3.5, 0), ([4] * 9, 3, 100.0, 1000),
             ([], 5, 1.0, 0), ([7], 1, 2.0, 0), ([1, 2, 3], 5, 10.0, 40), ([5, 0, 5, 0, 5], 1, 4.0, 0), ([2, 4, 6, 8, 0], 4, 4.9, 9)]
    for d, ws, md, st in fixed:
        sw.append(dict(depth=d, ws=ws, max_depth=md, start=st))
    for _ in range(40):
        L = int(rng3.integers(1, 400))
        d = rng3.poisson(rng3.uniform(0.3, 6.0), L)
        for _z in range(int(rng3.integers(0, 4))):
            a = int(rng3.integers(0, L))
            d[a:a + int(rng3.integers(1, 30))] = 0
        sw.append(dict(depth=d.tolist(), ws=int(rng3.integers(1, 60)), max_depth=float(rng3.uniform(0.5, 8.0)),
                       start=int(rng3.integers(0, 10_000_000))))
    for c in sw:
        with contextlib.redirect_stderr(io.StringIO()):
            pos, val = ref.sliding_window_average_depth(list(c["depth"]), c["ws"], c["max_depth"], c["start"], "t")
        c["pos"], c["val"] = [float(x) for x in pos], [float(x) for x in val]
    kats["sliding_window_average_depth"] = sw
    pp = []
    for n_types in (1, 2):
        for _ in range(6):
            targets = {"a": int(rng3.integers(30, 300)), "b": int(rng3.integers(30, 300))}
            dl = [{t: rng3.poisson(3.0 + 2 * i, L) for t, L in targets.items()} for i in range(n_types)]
            md = [float(rng3.uniform(3, 12)) for _ in range(n_types)]
            ws = int(rng3.integers(2, 25))
            with contextlib.redirect_stderr(io.StringIO()):
                av, y_frac, y_min, y_max = ref.pre_plot_base(dl, md, ws, 0)
            pp.append(dict(depths=[{t: v.tolist() for t, v in d.items()} for d in dl], max_depths=md, ws=ws,
                           y_frac=float(y_frac), y_min=float(y_min), y_max=float(y_max),
                           series=[{t: ([float(x) for x in p], [float(x) for x in v]) for t, (p, v) in a.items()} for a in av]))
    kats["pre_plot_base"] = pp

    with open(os.path.join(GOLDEN, "kats.json"), "w") as f:
        json.dump(kats, f, indent=0, sort_keys=True, default=lambda o: o.tolist() if hasattr(o, "tolist") else list(o))
    print("kats.json written")


def case_cli_plot():
    """The whole command line with -p: two read types, two contigs, two regions, 500-base windows."""
    contigs = (("ctgP", 60_000), ("ctgQ", 24_000))
    gaps = {"ctgP": [(30_000, 30_300)]}
    h = synth.simulate_reads(contigs, 28, "hifi", seed=synth.seed_for(6, 0))
    n = synth.simulate_reads(contigs, 22, "ont", seed=synth.seed_for(6, 1), long_cigar_frac=0.0)
    regions = [("ctgP", 8000, 12_000), ("ctgQ", 0, 24_000)]
    write_inputs(os.path.join(GOLDEN, "c6_plot"), contigs, {"hifi.mm2.bam": h, "ont.mm2.bam": n}, gaps=gaps, regions=regions)
    os.makedirs(os.path.join(tempfile.gettempdir(), "gci_plot_tmp"), exist_ok=True)
    run_reference("c6_plot", {"plot": True, "window_size": 500, "depth_min": 0.2, "depth_max": 3.0},
                  hifi=["hifi.mm2.bam"], nano=["ont.mm2.bam"], regions=True)


def case_t2t_geometry():
    """BASELINE.json configs[3] / [4] in small: the 25 contigs of CHM13 at 1/1250 of their lengths (chrM at its minimum),
    both read types with two files each (BAM + PAF of another aligner's view of the same reads), gaps, -R regions over
    several contigs and -p.  Also the case the two-rank runs shard over many contigs of very different lengths."""
    contigs = tuple((n, max(9_000, l // 1250)) for n, l in synth.CHM13)
    gaps = {"chr1": [(50_000, 50_400)], "chr9": [(0, 300), (60_000, 60_001)], "chrX": [(100_000, 101_000)]}
    h = synth.simulate_reads(contigs, 30, "hifi", seed=synth.seed_for(8, 0))
    h2 = synth.perturb(h, synth.seed_for(8, 1))
    n = synth.simulate_reads(contigs, 24, "ont", seed=synth.seed_for(8, 2), long_cigar_frac=0.0)
    n2 = synth.perturb(n, synth.seed_for(8, 3))
    regions = [("chr1", 0, 150_000), ("chr2", 20_000, 180_000), ("chr9", 0, 110_000), ("chr21", 1_000, 30_000),
               ("chrX", 0, 123_000), ("chrM", 0, 9_000), ("chr2", 30_000, 31_000)]
    write_inputs(os.path.join(GOLDEN, "c7_t2t_geometry"), contigs,
                 {"hifi.wm2.bam": h, "hifi.mm2.paf": synth.to_paf_lines(h2, synth.seed_for(8, 4), 0.05),
                  "ont.wm2.bam": n, "ont.mm2.paf": synth.to_paf_lines(n2, synth.seed_for(8, 5), 0.05)},
                 gaps=gaps, regions=regions)
    run_reference("c7_t2t_geometry", {"plot": True, "window_size": 2000, "threshold": 1},
                  hifi=["hifi.wm2.bam", "hifi.mm2.paf"], nano=["ont.wm2.bam", "ont.mm2.paf"], regions=True)


def case_plot():
    """N3 end to end: the figures the reference's plot_base draws for a small one-type and a small two-type input
    (tests/golden/plot/*.png) together with the depth arrays they were drawn from (inputs.npz)."""
    ref = load_reference.load()
    out = os.path.join(GOLDEN, "plot")
    shutil.rmtree(out, ignore_errors=True)
    os.makedirs(os.path.join(out, "images"))
    rng = np.random.default_rng(20250929)
    L = 60_000
    h = rng.poisson(30, L)
    n = rng.poisson(22, L)
    for d in (h, n):
        d[:15] = 0
        d[-15:] = 0
    h[9_000:9_700] = 0; h[20_000:20_400] = rng.integers(1, 3, 400); h[41_000:41_050] = 0
    n[9_200:9_500] = 0; n[50_000:52_000] = rng.integers(0, 2, 2000)
    np.savez_compressed(os.path.join(out, "inputs.npz"), hifi=h.astype(np.int32), nano=n.astype(np.int32))
    meta = {}
    for name, dl in (("one", [{"ctgP": h}]), ("two", [{"ctgP": h}, {"ctgP": n}])):
        means = [float(np.mean(d["ctgP"])) for d in dl]
        maxd = [m * 4.0 for m in means]
        with contextlib.redirect_stderr(io.StringIO()), contextlib.redirect_stdout(io.StringIO()):
            av, y_frac, y_min, y_max = ref.pre_plot_base(dl, maxd, 500, 0)
            ref.plot_base(dl, "ctgP", av, means, y_frac, 0, 0.1, 0.005, y_min, y_max, "png", out, name, L, False, 0)
            ref.pre_plot_base([{"ctgP": d["ctgP"][8000:12000]} for d in dl], maxd, 500, 8000)
            av2, y_frac2, y_min2, y_max2 = ref.pre_plot_base([{"ctgP": d["ctgP"][8000:12000]} for d in dl], maxd, 500, 8000)
            ref.plot_base([{"ctgP": d["ctgP"][8000:12000]} for d in dl], "ctgP", av2, means, y_frac2, 8000, 0.1, 0.005, y_min2,
                          y_max2, "png", out, name, 12000, True, 0)
        meta[name] = dict(means=means, y=[float(y_frac), float(y_min), float(y_max)], y_region=[float(y_frac2), float(y_min2), float(y_max2)])
    with open(os.path.join(out, "meta.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print("plot ->", sorted(os.listdir(os.path.join(out, "images"))))


def case_cli_errors():
    """R14: the reference's command line on inputs it refuses (and a few it accepts with a warning), run through its own
    `__main__` block: exit message / code, stdout and stderr of every scenario -> tests/golden/cli_errors.json.
    Inputs: tests/golden/cli_errors/inputs (tiny two-contig BAMs, a PAF, FASTA files with and without the BAMs' contigs,
    BED files).  Not covered: the unreadable / unwritable directory exits (GCI.py:917-920, 929-932) -- the build
    container runs as root, for which os.access() is always true."""
    load_reference.load()                                  # puts the pysam / Bio stand-ins on sys.path
    base = os.path.join(GOLDEN, "cli_errors")
    inp = os.path.join(base, "inputs")
    shutil.rmtree(base, ignore_errors=True)
    os.makedirs(inp)
    contigs = (("e1", 30_000), ("e2", 12_000))
    h = synth.simulate_reads(contigs, 8, "hifi", seed=synth.seed_for(7, 0))
    n = synth.simulate_reads(contigs, 8, "ont", seed=synth.seed_for(7, 1), long_cigar_frac=0.0)
    n_short = synth.simulate_reads((("e1", 30_000), ("e2", 11_000)), 8, "ont", seed=synth.seed_for(7, 2), long_cigar_frac=0.0)
    synth.write_bam_file(os.path.join(inp, "hifi.bam"), h, level=9, threads=2)
    synth.write_bam_file(os.path.join(inp, "ont.bam"), n, level=9, threads=2)
    synth.write_bam_file(os.path.join(inp, "ont_e2_short.bam"), n_short, level=9, threads=2)
    paffmt.write(os.path.join(inp, "hifi.paf"), synth.to_paf_lines(h, synth.seed_for(7, 3), 0.05))
    for fn in os.listdir(inp):
        if fn.endswith(".bai"):
            os.remove(os.path.join(inp, fn))
    synth.write_reference_fasta(os.path.join(inp, "ref.fa"), contigs)
    synth.write_reference_fasta(os.path.join(inp, "ref_gap.fa"), contigs, {"e1": [(5_000, 5_100)]})
    synth.write_reference_fasta(os.path.join(inp, "ref_extra.fa"), contigs + (("e3", 5_000),))
    for name, rows in (("regions_ok.bed", [("e1", 1000, 9000)]), ("regions_unknown.bed", [("zz", 0, 100)]),
                       ("regions_e2.bed", [("e2", 10, 2000)])):
        with open(os.path.join(inp, name), "w") as f:
            for t, a, b in rows:
                f.write(f"{t}\t{a}\t{b}\n")
    I = "{IN}/"
    scenarios = [
        ("no_arguments", []),
        ("version", ["-v"]),
        ("no_alignment_type", ["-r", I + "ref.fa"]),
        ("hifi_file_missing", ["-r", I + "ref.fa", "--hifi", I + "nope.bam"]),
        ("hifi_without_bam", ["-r", I + "ref.fa", "--hifi", I + "hifi.paf"]),
        ("nano_file_missing", ["-r", I + "ref.fa", "--hifi", I + "hifi.bam", "--nano", I + "nope.bam"]),
        ("nano_without_bam", ["-r", I + "ref.fa", "--nano", I + "hifi.paf"]),
        ("no_reference", ["--hifi", I + "hifi.bam"]),
        ("reference_missing", ["-r", I + "nope.fa", "--hifi", I + "hifi.bam"]),
        ("regions_missing", ["-r", I + "ref.fa", "--hifi", I + "hifi.bam", "-R", I + "nope.bed", "-d", "{OUT}"]),
        ("prefix_with_slash", ["-r", I + "ref.fa", "--hifi", I + "hifi.bam", "-o", "x/", "-d", "{OUT}"]),
        ("chrs_unknown", ["-r", I + "ref.fa", "--hifi", I + "hifi.bam", "--chrs", "e1,zz", "-d", "{OUT}"]),
        ("regions_unknown", ["-r", I + "ref.fa", "--hifi", I + "hifi.bam", "-R", I + "regions_unknown.bed", "-d", "{OUT}"]),
        ("chrs_regions_inconsistent", ["-r", I + "ref.fa", "--hifi", I + "hifi.bam", "--chrs", "e1", "-R", I + "regions_e2.bed", "-d", "{OUT}"]),
        ("hifi_targets_vs_reference", ["-r", I + "ref_extra.fa", "--hifi", I + "hifi.bam", "-d", "{OUT}"]),
        ("nano_targets_vs_reference", ["-r", I + "ref_extra.fa", "--nano", I + "ont.bam", "-d", "{OUT}"]),
        ("hifi_nano_lengths_differ", ["-r", I + "ref.fa", "--hifi", I + "hifi.bam", "--nano", I + "ont_e2_short.bam", "-d", "{OUT}"]),
        ("mapq_warning_then_runs", ["-r", I + "ref.fa", "--hifi", I + "hifi.bam", "-mq", "60", "--mq-cutoff", "50", "-d", "{OUT}/"]),
        ("refuses_to_overwrite_depth", ["-r", I + "ref.fa", "--hifi", I + "hifi.bam", "-d", "{OUT}"]),            # second run, same place
        ("refuses_to_overwrite_gaps", ["-r", I + "ref_gap.fa", "--hifi", I + "hifi.bam", "-d", "{OUT}2"]),
        ("refuses_to_overwrite_gaps_again", ["-r", I + "ref_gap.fa", "--hifi", I + "hifi.bam", "-d", "{OUT}2"]),
    ]
    tmp = tempfile.mkdtemp(prefix="gci_cli_err_")
    results = []
    os.environ["COLUMNS"] = "100"
    for name, argv_t in scenarios:
        argv = ["GCI.py"] + [a.replace("{IN}", inp).replace("{OUT}", os.path.join(tmp, "out")) for a in argv_t]
        so, se = io.StringIO(), io.StringIO()
        old = sys.argv
        sys.argv = argv
        code = "completed"
        try:
            with contextlib.redirect_stdout(so), contextlib.redirect_stderr(se):
                # as `python GCI.py ...` would: the file's code in a module registered as __main__ (its Pool pickles
                # functions by that name).  Not runpy.run_path: it replaces argv[0], which argparse prints in the usage line.
                import types
                mod = types.ModuleType("__main__")
                mod.__file__ = load_reference.REF
                saved_main = sys.modules["__main__"]
                sys.modules["__main__"] = mod
                try:
                    exec(compile(open(load_reference.REF).read(), load_reference.REF, "exec"), mod.__dict__)
                finally:
                    sys.modules["__main__"] = saved_main
        except SystemExit as e:
            code = e.code
        finally:
            sys.argv = old
        norm = lambda t: t.replace(os.path.join(tmp, "out"), "{OUT}").replace(inp, "{IN}")      # noqa: E731
        results.append({"name": name, "argv": argv_t, "exit": norm(code) if isinstance(code, str) else code,
                        "stdout": norm(so.getvalue()), "stderr": norm(se.getvalue())})
        print("cli_errors:", name, "->", repr(code)[:90])
    shutil.rmtree(tmp, ignore_errors=True)
    with open(os.path.join(GOLDEN, "cli_errors.json"), "w") as f:
        json.dump(results, f, indent=1)


def copy_reference_example():
    """The reference's own data triple (example/MH63.*) -- data files, not source."""
    dst = os.path.join(GOLDEN, "MH63")
    os.makedirs(dst, exist_ok=True)
    for fn in ("MH63.depth.gz", "MH63.0.depth.bed", "MH63.gci"):
        shutil.copy(os.path.join("/root/reference/example", fn), os.path.join(dst, fn))
        os.chmod(os.path.join(dst, fn), 0o644)
    print("MH63 triple copied")


if __name__ == "__main__":
    if not load_reference.available():
        sys.exit("needs /root/reference (build container only)")
    os.makedirs(GOLDEN, exist_ok=True)
    only = set(sys.argv[1:])
    todo = [("c1", case_single_bam), ("c3a", case_two_bam), ("c3b", case_three_bam_chrs), ("c4a", case_paf_bam),
            ("c4b", case_nano_only_long_cigar), ("c4c", case_two_paf), ("plot", case_plot), ("c6", case_cli_plot), ("c5", case_two_type), ("c7", case_t2t_geometry), ("kats", make_kats),
            ("mh63", copy_reference_example), ("cli", case_cli_errors)]
    for name, fn in todo:
        if not only or name in only:
            fn()
