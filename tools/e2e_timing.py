#!/usr/bin/env python3
"""End-to-end timing of the drop-in CLI path on one MI355X (SURVEY.md 8d asks for three numbers per config:
kernels only = bench.py; device pipeline incl. H2D / D2H; full CLI incl. BGZF inflate and .depth.gz emission).
Writes a JSON summary to stdout."""
import json, os, sys, time, tempfile, shutil, io, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gci_amd import synth, pipeline, cli, hostio
from gci_amd.formats import bam as bamfmt, bgzf

def run(config, contigs, cov):
    threads = os.cpu_count() or 1
    tmp = tempfile.mkdtemp(prefix="gci_e2e_")
    out = {"config": config, "host_threads": threads}
    t = time.perf_counter()
    rs = synth.simulate_reads(contigs, cov, "hifi", seed=synth.seed_for(2, 0))
    stream, offs = synth.to_bam_stream(rs)
    out["aligned_bases"] = rs.aligned_bases(); out["records"] = len(rs); out["inflated_bytes"] = int(stream.shape[0])
    bam_path = os.path.join(tmp, "hifi.bam"); fa = os.path.join(tmp, "ref.fa")
    bamfmt.write_bam_stream(bam_path, stream, level=1, threads=threads)
    synth.write_reference_fasta(fa, contigs)
    out["bam_file_bytes"] = os.path.getsize(bam_path); out["gen_s"] = time.perf_counter() - t
    eng = pipeline.default_engine()
    # stage timings
    nt = hostio.pick_threads(threads); out["host_threads_used"] = nt
    t = time.perf_counter(); s2 = hostio.read_bgzf_file(bam_path, threads=nt); out["inflate_s"] = time.perf_counter() - t
    t = time.perf_counter(); o2, _ = hostio.bam_record_offsets(s2); out["offsets_s"] = time.perf_counter() - t
    torch.cuda.synchronize(); t = time.perf_counter(); d_bam = eng.to_device(s2); d_off = eng.to_device(o2); torch.cuda.synchronize(); out["h2d_s"] = time.perf_counter() - t
    t = time.perf_counter(); del d_bam, d_off, s2; out["release_s"] = time.perf_counter() - t
    # the heads-stream ingestion the command line uses instead of the three stages above
    raw = np.fromfile(bam_path, dtype=np.uint8)
    t = time.perf_counter(); hd = hostio.bam_heads(raw, threads=nt); out["heads_s"] = time.perf_counter() - t
    out["heads_bytes"] = int(hd.stream.shape[0])
    torch.cuda.synchronize(); t = time.perf_counter(); d_h = eng.to_device(hd.stream); d_o = eng.to_device(hd.offsets); torch.cuda.synchronize(); out["heads_h2d_s"] = time.perf_counter() - t
    hd.close(); del d_h, d_o, raw
    # full CLI (twice: the second run has warm page cache and a built context)
    for k in ("cli_first_s", "cli_s"):
        od = os.path.join(tmp, k)
        t = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            cli.main(["GCI.py", "-r", fa, "--hifi", bam_path, "-d", od, "-t", str(threads)])
        torch.cuda.synchronize(); out[k] = time.perf_counter() - t
    out["depth_gz_bytes"] = os.path.getsize(os.path.join(od, "GCI.depth.gz"))
    out["cli_Gbases_per_s"] = out["aligned_bases"] / out["cli_s"] / 1e9
    shutil.rmtree(tmp)
    return out

if __name__ == "__main__":
    res = [run("C1: ctg1 5 Mb, 30x HiFi", synth.CTG1, 30)]
    if len(sys.argv) > 1 and sys.argv[1] == "c2":
        res.append(run("C2: chr19 61.7 Mb, 40x HiFi", synth.CHR19, 40))
    print(json.dumps(res, indent=1))
