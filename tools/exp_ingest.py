#!/usr/bin/env python3
"""BAM ingestion of a chr19 40x HiFi file with realistic SEQ / QUAL entropy (BGZF inflates ~2.4:1): the native host pipeline
(ingest "heads": gci_bam_heads on host threads, 54 MB uploaded) against inflate + record walk on the device (ingest "gpu":
the file's bytes uploaded, gci_bgzf_inflate_device, gci_bam_record_offsets_device), stage by stage, and the whole command
line with either.  Usage: exp_ingest.py [scale]"""
import contextlib, io, json, os, shutil, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gci_amd import synth, hostio, pipeline, cli
from gci_amd.formats import bam as bamfmt
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
contigs = (("chr19", int(61_707_364 * scale)),)
tmp = tempfile.mkdtemp(prefix="gci_ingest_")
rs = synth.simulate_reads(contigs, 40, "hifi", seed=synth.seed_for(2, 0))
stream, _ = synth.to_bam_stream(rs, seq_qual="random", seed=7)
bam, fa = os.path.join(tmp, "x.bam"), os.path.join(tmp, "ref.fa")
bamfmt.write_bam_stream(bam, stream, level=1, threads=hostio.default_threads())
synth.write_reference_fasta(fa, contigs)
out = {"records": len(rs), "inflated_bytes": int(stream.shape[0]), "bam_bytes": os.path.getsize(bam), "host_threads": hostio.default_threads()}
del stream
eng = pipeline.default_engine()
eng.set_layout([contigs[0][1]])
filt = (30, 50, 0.1, 0.9)
for ingest in ("heads", "gpu", "heads", "gpu"):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ji = pipeline.bam_join_input(eng, bam, ["chr19"], filt, threads=hostio.default_threads(), ingest=ingest)
    torch.cuda.synchronize(); out["join_input_%s_s" % ingest] = round(time.perf_counter() - t0, 4)
    del ji
# stages of the device path
raw = np.fromfile(bam, dtype=np.uint8)
t0 = time.perf_counter(); pos, isz = hostio.bgzf_blocks(raw); out["gpu_member_table_s"] = round(time.perf_counter() - t0, 4)
torch.cuda.synchronize(); t0 = time.perf_counter(); d_raw = eng.to_device(raw); torch.cuda.synchronize(); out["gpu_h2d_file_s"] = round(time.perf_counter() - t0, 4)
del d_raw
torch.cuda.synchronize(); t0 = time.perf_counter(); d = eng.bgzf_inflate(raw, pos, isz); torch.cuda.synchronize(); out["gpu_h2d_plus_inflate_s"] = round(time.perf_counter() - t0, 4)
hdr = bamfmt.read_header(bam)
torch.cuda.synchronize(); t0 = time.perf_counter(); offs, used, ok = eng.bam_record_offsets(d, hdr.first_record, 1); torch.cuda.synchronize()
out["gpu_record_walk_s"] = round(time.perf_counter() - t0, 4); out["walk_ok"] = bool(ok and used == d.shape[0] and offs.shape[0] == len(rs))
del d, offs
for ingest in ("heads", "gpu"):
    os.environ["GCI_BAM_INGEST"] = ingest
    for k in range(2):
        od = os.path.join(tmp, "out_%s_%d" % (ingest, k))
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            cli.main(["GCI.py", "-r", fa, "--hifi", bam, "-d", od, "-t", str(hostio.default_threads())])
        torch.cuda.synchronize(); out["cli_%s_s" % ingest] = round(time.perf_counter() - t0, 4)
same = all(open(os.path.join(tmp, "out_heads_1", f), "rb").read() == open(os.path.join(tmp, "out_gpu_1", f), "rb").read() for f in ("GCI.gci", "GCI.0.depth.bed"))
out["cli_outputs_equal"] = same
shutil.rmtree(tmp, ignore_errors=True)
print(json.dumps(out))
