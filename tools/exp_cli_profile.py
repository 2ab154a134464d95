#!/usr/bin/env python3
"""Where the wall time of the drop-in command line goes (bench.py number 3: chr19 40x HiFi BAM with realistic SEQ / QUAL
entropy): cProfile of the second in-process run, by cumulative time.  GPU work is asynchronous: kernels show up at the
call that waits for them.  Usage: exp_cli_profile.py [scale]"""
import contextlib, cProfile, io, os, pstats, shutil, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gci_amd import cli, hostio, synth
from gci_amd.formats import bam as bamfmt
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
contigs = (("chr19", int(61_707_364 * scale)),)
tmp = tempfile.mkdtemp(prefix="gci_cliprof_")
rs = synth.simulate_reads(contigs, 40, "hifi", seed=synth.seed_for(2, 0))
stream, _ = synth.to_bam_stream(rs, seq_qual="random", seed=7)
bam, fa = os.path.join(tmp, "hifi.bam"), os.path.join(tmp, "ref.fa")
bamfmt.write_bam_stream(bam, stream, level=1, threads=hostio.default_threads())
del stream, rs
synth.write_reference_fasta(fa, contigs)
def run(k):
    od = os.path.join(tmp, "out%d" % k)
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        cli.main(["GCI.py", "-r", fa, "--hifi", bam, "-d", od, "-t", str(hostio.default_threads())])
    torch.cuda.synchronize()
    return time.perf_counter() - t0
print("run 0: %.3f s" % run(0)); print("run 1: %.3f s" % run(1))
pr = cProfile.Profile(); pr.enable(); w = run(2); pr.disable()
print("run 2 (profiled): %.3f s" % w)
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45); print(s.getvalue()[:9000])
s2 = io.StringIO(); pstats.Stats(pr, stream=s2).sort_stats("tottime").print_stats(18); print(s2.getvalue()[:4000])
shutil.rmtree(tmp, ignore_errors=True)
