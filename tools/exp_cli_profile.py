#!/usr/bin/env python3
"""Where the drop-in command line spends its time at chr19 / 40x (cProfile of the second, warm run)."""
import cProfile, contextlib, io, os, pstats, shutil, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gci_amd import synth, cli
from gci_amd.formats import bam as bamfmt

threads = os.cpu_count() or 1
tmp = tempfile.mkdtemp(prefix="gci_prof_")
rs = synth.simulate_reads(synth.CHR19, 40, "hifi", seed=synth.seed_for(2, 0))
stream, offs = synth.to_bam_stream(rs)
bam, fa = os.path.join(tmp, "hifi.bam"), os.path.join(tmp, "ref.fa")
bamfmt.write_bam_stream(bam, stream, level=1, threads=threads)
synth.write_reference_fasta(fa, synth.CHR19)
del stream
for k in range(3):
    od = os.path.join(tmp, "o%d" % k)
    pr = cProfile.Profile() if k == 2 else None
    t = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        if pr:
            pr.enable()
        cli.main(["GCI.py", "-r", fa, "--hifi", bam, "-d", od, "-t", str(threads)])
        torch.cuda.synchronize()
        if pr:
            pr.disable()
    print("run %d: %.3f s" % (k, time.perf_counter() - t))
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
shutil.rmtree(tmp)
