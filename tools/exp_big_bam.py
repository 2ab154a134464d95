#!/usr/bin/env python3
"""A BAM whose inflated stream is well beyond 4 GiB (chr1 of CHM13 at 40x HiFi with realistic SEQ / QUAL: 14.7 GB inflated,
~6 GB on disk) through the device ingestion, whole and run by run of members, against the host heads path: same records,
same names.  Checks the 64-bit offsets of the inflate kernel, the record walk and K1.  Usage: exp_big_bam.py [scale]"""
import json, os, shutil, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gci_amd import hostio, pipeline, synth
from gci_amd.formats import bam as bamfmt
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
contigs = (("chr1", int(248_387_328 * scale)),)
tmp = tempfile.mkdtemp(prefix="gci_big_")
t0 = time.time()
rs = synth.simulate_reads(contigs, 40, "hifi", seed=synth.seed_for(2, 7))
stream, offs = synth.to_bam_stream(rs, seq_qual="random", seed=11)
bam = os.path.join(tmp, "big.bam")
bamfmt.write_bam_stream(bam, stream, level=1, threads=hostio.default_threads())
out = {"records": len(rs), "inflated_bytes": int(stream.shape[0]), "bam_bytes": os.path.getsize(bam), "make_s": round(time.time() - t0, 1)}
del stream
eng = pipeline.default_engine()
eng.set_layout([contigs[0][1]])
filt = (30, 50, 0.1, 0.9)
def sig(ji):
    names, noff = eng.pack_names(ji)
    return ji.recs.cpu().numpy().copy(), names.cpu().numpy().copy(), noff.cpu().numpy().copy()
res = {}
for label, ingest, gmax, chunk in (("heads", "heads", None, None), ("gpu_whole", "gpu", 64 << 30, None), ("gpu_runs", "gpu", 0, 3 << 30)):
    if gmax is not None:
        pipeline.GPU_INFLATE_MAX = gmax
    if chunk is not None:
        pipeline.BAM_CHUNK_BYTES = chunk
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ji = pipeline.bam_join_input(eng, bam, ["chr1"], filt, threads=hostio.default_threads(), ingest=ingest)
    torch.cuda.synchronize(); out[label + "_s"] = round(time.perf_counter() - t0, 3)
    res[label] = sig(ji)
    del ji
    torch.cuda.empty_cache()
for label in ("gpu_whole", "gpu_runs"):
    out[label + "_equal_heads"] = all(np.array_equal(a, b) for a, b in zip(res[label], res["heads"]))
shutil.rmtree(tmp, ignore_errors=True)
print(json.dumps(out))
