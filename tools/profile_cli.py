import cProfile, pstats, io, os, sys, tempfile, contextlib
sys.path.insert(0, '.')
import torch
from gci_amd import synth, cli
from gci_amd.formats import bam as bamfmt
tmp = tempfile.mkdtemp()
rs = synth.simulate_reads(synth.CHR19, 40, "hifi", seed=synth.seed_for(2, 0))
stream, offs = synth.to_bam_stream(rs)
bamfmt.write_bam_stream(tmp + "/h.bam", stream, level=1, threads=64)
synth.write_reference_fasta(tmp + "/ref.fa", synth.CHR19)
del stream
args = ["GCI.py", "-r", tmp + "/ref.fa", "--hifi", tmp + "/h.bam", "-d", tmp + "/o", "-t", "64"]
with contextlib.redirect_stdout(io.StringIO()):
    cli.main(args + ["-f"])
pr = cProfile.Profile(); pr.enable()
with contextlib.redirect_stdout(io.StringIO()):
    cli.main(args + ["-f"])
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue()[:6000])
