#!/usr/bin/env python3
"""What the command line does between its last output file and the end of its process (GCI_EXIT_TRACE=1: stamps when main() has
returned, at the interpreter's atexit and at the C library's): usage exit_cost.py [scale] -- chr19-like input at `scale` of the
genome, three runs."""
import os, subprocess, sys, tempfile, time, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gci_amd import workloads, synth, hostio

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.02
inp = workloads.genome_dual(scale, 40.0, contigs=synth.CHM13, verbose=False)
tmp = tempfile.mkdtemp(prefix="gci_exit_", dir="/dev/shm")
try:
    bams = []
    for k, f in enumerate(inp.files):
        p = os.path.join(tmp, "a%d.bam" % k)
        workloads.write_bgzf_from_heads(p, f.stream, f.offsets, seed=20250919 + k)
        bams.append(p)
    fa = os.path.join(tmp, "ref.fa")
    synth.write_reference_fasta(fa, inp.contigs)
    here = os.path.dirname(os.path.abspath(__file__))
    sodir = tempfile.mkdtemp(prefix="gci_exit_so_", dir="/tmp")          # (/dev/shm is mounted noexec on the GPU boxes)
    late, first = os.path.join(sodir, "libstamp_late.so"), os.path.join(sodir, "libstamp_first.so")
    for out, name in ((late, "late"), (first, "preloaded")):
        subprocess.run(["gcc", "-shared", "-fPIC", "-O1", "-DSTAMP_NAME=\"%s\"" % name, os.path.join(here, "exit_stamp.c"), "-o", out], check=True)
    for label, extra in (("ordinary exit", {}), ("ordinary exit", {}), ("arena off", {"GCI_ARENA": "0"}), ("torch buffers", {"GCI_HBM": "torch"})):
        od = os.path.join(tmp, "out")
        shutil.rmtree(od, ignore_errors=True)
        env = dict(os.environ, GCI_EXIT_TRACE=late, LD_PRELOAD=first, PYTHONPATH=ROOT)
        env.update(extra)
        t0 = time.time()
        r = subprocess.run([sys.executable, os.path.join(ROOT, "GCI.py"), "-r", fa, "--hifi"] + bams + ["-d", od, "-t", str(hostio.default_threads())],
                           env=env, capture_output=True, text=True)
        t1 = time.time()
        st = {l.split()[1]: float(l.split()[2]) for l in r.stderr.splitlines() if l.startswith("exit-trace")}
        need = ("main_returned", "python_atexit", "late_atexit", "preloaded_destructor")
        if r.returncode or any(k not in st for k in need):
            print(label, "rc", r.returncode, sorted(st), r.stderr[-600:])
            continue
        print("%-14s wall %.3f s | main() returned at %.3f | -> interpreter's atexit +%.3f | -> interpreter finalised, C handlers start +%.3f | -> "
              "the preloaded library's destructor (HIP's teardown is over) +%.3f | -> process gone +%.3f" % (
                  label, t1 - t0, st["main_returned"] - t0, st["python_atexit"] - st["main_returned"], st["late_atexit"] - st["python_atexit"],
                  st["preloaded_destructor"] - st["late_atexit"], t1 - st["preloaded_destructor"]), flush=True)
finally:
    shutil.rmtree(tmp, ignore_errors=True)
