#!/usr/bin/env python3
"""The command line as a process of its own on two whole-genome BGZF BAM files at a chosen scale (bench.py's
survey_8d.3_command_line_genome leg alone): files written to tmpfs from the heads streams of workloads.genome_dual, `python GCI.py`
timed with its phase log, outputs held against the oracle on three contigs.  Usage: exp_cli_genome.py [scale] [--keep DIR]
With GCI_EXP_PROFILE=1 the run is repeated under rocprofv3 --kernel-trace --stats (kernel table printed)."""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from gci_amd import workloads, synth
from oracle import gci_oracle as O

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
inp = workloads.genome_dual(scale, 40.0, contigs=synth.CHM13, verbose=True)
O.build()
names = inp.names
chosen = ["chr14", "chr22", "chrM"]
orc = None
if not os.environ.get("GCI_EXP_PROFILE"):                  # (the A/B mode times host-side switches: no oracle needed)
    file1 = O.file1_on_contigs([(f.stream, f.offsets, names) for f in inp.files], names, chosen, *bench.FILTER, bench.OVLP, heads=True)
    tl = {c: inp.lengths[names.index(c)] for c in chosen}
    depths = O.depth_build(file1, tl, bench.FLANK)
    orc = {"depths": depths, "bed": O.collapse_depth_range(depths, -1, 0, bench.FLANK, 0), "lengths": tl}
if os.environ.get("GCI_EXP_PROFILE"):
    # keep the files: run the command line once more under the profiler
    import tempfile, shutil
    from gci_amd import hostio
    tmp = tempfile.mkdtemp(prefix="gci_cli_prof_", dir="/dev/shm")
    try:
        bams = []
        for k, f in enumerate(inp.files):
            p = os.path.join(tmp, "a%d.bam" % k)
            workloads.write_bgzf_from_heads(p, f.stream, f.offsets, seed=20250919 + k)
            bams.append(p)
        fa = os.path.join(tmp, "ref.fa")
        synth.write_reference_fasta(fa, inp.contigs)
        # A/B of host-side switches on the same files and the same box, unprofiled: wall time and the phase log
        variants = (("product (staged upload)", {}), ("staged, pages kept", {"GCI_FORGET_PAGES": "0"}),
                    ("product again", {}))
        if os.environ.get("GCI_EXP_AB"):                   # e.g. '[["16 GiB runs", {"GCI_BAM_CHUNK_BYTES": "17179869184"}]]'
            variants = [("product", {})] + [(a, b) for a, b in json.loads(os.environ["GCI_EXP_AB"])] + [("product again", {})]
        if os.environ.get("GCI_EXP_FIRST"):                # the FIRST pass over the freshly written files under another environment
            lab, extra0 = json.loads(os.environ["GCI_EXP_FIRST"])
            variants = [("first pass: " + lab, extra0)] + list(variants)
        if os.environ.get("GCI_EXP_PRETOUCH") == "cat":    # the files read once by another process before the first pass
            t0 = time.perf_counter()
            for b in bams + [fa]:
                subprocess.run(["cat", b], stdout=subprocess.DEVNULL)
            print("pretouch: cat of %d files took %.2f s" % (len(bams) + 1, time.perf_counter() - t0), flush=True)
        for label, extra in variants:
            env = dict(os.environ, GCI_PHASES=os.path.join(tmp, "ph_ab.json"), PYTHONPATH=ROOT, GCI_STUCK_TRACE="15")
            env.update(extra)
            od = os.path.join(tmp, "out_ab")
            shutil.rmtree(od, ignore_errors=True)
            t0 = time.perf_counter()
            r = subprocess.run([sys.executable, os.path.join(ROOT, "GCI.py"), "-r", fa, "--hifi"] + bams + ["-d", od, "-t", str(hostio.default_threads())],
                               env=env, capture_output=True, text=True)
            wall = time.perf_counter() - t0
            ph = json.load(open(os.path.join(tmp, "ph_ab.json")))
            if wall - ph.get("total_s", 0.0) > 10.0:         # a process that sat somewhere outside its phases: what it said
                print("   stderr tail: " + r.stderr[-3000:], flush=True)
            if os.environ.get("GCI_EXP_SAVE"):               # the whole phase log (with its per-run trace under GCI_PHASES_TRACE=1)
                os.makedirs(os.environ["GCI_EXP_SAVE"], exist_ok=True)
                ph2 = dict(ph, notes={k: v for k, v in ph["notes"].items() if not k.startswith("depth_gz_layout")})
                json.dump(ph2, open(os.path.join(os.environ["GCI_EXP_SAVE"], "phases_%s.json" % label.replace(" ", "_")), "w"))
                open(os.path.join(os.environ["GCI_EXP_SAVE"], "stderr_%s.txt" % label.replace(" ", "_")), "w").write(r.stderr)
            keep = ("bam_ingest", "name_join", "filter[", "fasta", "bgzf_member", "wait")
            if os.environ.get("GCI_EXP_MEMINFO"):            # what the page cache looks like behind the run (huge pages of tmpfs?)
                mi = {l.split(":")[0]: l.split(":")[1].strip() for l in open("/proc/meminfo")}
                print("   meminfo: " + ", ".join("%s %s" % (k, mi.get(k)) for k in ("Shmem", "ShmemHugePages", "ShmemPmdMapped", "Active(file)", "Inactive(file)", "Active(anon)", "Inactive(anon)", "Mapped")), flush=True)
            print("%-26s rc %d wall %.2f s (in front of the phase log %.2f, behind its report %.2f) | " % (
                  label, r.returncode, wall, ph["notes"].get("process_age_s_when_the_phase_clock_started", -1),
                  wall - ph["notes"].get("process_age_s_at_the_report", wall)) + ", ".join("%s %.2f" % (k.strip()[:28], v) for k, v in ph["wall_s"].items() if k.strip().startswith(keep))
                  + " | gpu inflate %.2f" % ph["gpu_s"].get("bgzf_inflate + crc", 0)
                  + " | mem %s" % ", ".join("%s %.1f GB" % (k[:12], v / 1e9) for k, v in (ph["notes"].get("device_memory") or {}).items() if isinstance(v, int) and v > 1e6), flush=True)
        if os.environ.get("GCI_EXP_NO_ROCPROF"):
            raise SystemExit(0)
        env = dict(os.environ, GCI_PHASES=os.path.join(tmp, "ph.json"), PYTHONPATH=ROOT, TMPDIR="/tmp")
        out = os.path.join(ROOT, "gpurun_out", "cli_prof")
        shutil.rmtree(out, ignore_errors=True)
        cmd = ["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", out, "-o", "cli", "--",
               sys.executable, os.path.join(ROOT, "GCI.py"), "-r", fa, "--hifi"] + bams + ["-d", os.path.join(tmp, "out"), "-t", str(hostio.default_threads())]
        t0 = time.perf_counter()
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, cwd="/tmp")
        print("profiled run: rc %d, %.2f s" % (r.returncode, time.perf_counter() - t0), r.stderr[-500:] if r.returncode else "")
        ph = json.load(open(os.path.join(tmp, "ph.json")))
        ph["notes"] = {k: v for k, v in ph["notes"].items() if not k.startswith("depth_gz_layout")}
        print(json.dumps(ph, indent=1))
        if os.environ.get("GCI_EXP_CPROFILE"):
            # once more in this process under cProfile: where the HOST spends the time the kernels do not account for
            import cProfile, pstats, io, contextlib
            from gci_amd import cli
            pr = cProfile.Profile()
            with contextlib.redirect_stdout(io.StringIO()):
                pr.enable()
                cli.main(["GCI.py", "-r", fa, "--hifi"] + bams + ["-d", os.path.join(tmp, "out2"), "-t", str(hostio.default_threads())])
                pr.disable()
            st = io.StringIO()
            pstats.Stats(pr, stream=st).sort_stats("cumulative").print_stats(45)
            print(st.getvalue()[:9000])
        import csv, glob
        for fcsv in glob.glob(out + "/**/*kernel_stats.csv", recursive=True):
            rows = sorted(csv.DictReader(open(fcsv)), key=lambda r: -float(r["TotalDurationNs"]))
            for r in rows[:25]:
                print("%-70s %6s calls %10.3f ms total %8.3f ms avg" % (r["Name"][:70], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6))
        for fcsv in glob.glob(out + "/**/*kernel_trace.csv", recursive=True):
            os.remove(fcsv)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
else:
    print(json.dumps(bench.cli_genome_number(inp, orc), indent=1))
