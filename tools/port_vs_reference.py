#!/usr/bin/env python3
"""Speed ratio of the oracle port (oracle/gci_oracle.{c,py}) to the UNMODIFIED reference on BASELINE configs[0]
(one 5 Mb contig, 30x simulated HiFi BAM, `-t 1`), measured in THIS container (the reference cannot travel to the GPU
box; bench.py prints this ratio next to its cpu_baseline so readers can translate "port" into "reference-equivalent").
Writes profiles/port_vs_reference.json.  The reference runs through tools/ref_shim (pysam / Bio stand-ins built on the
repo's own BAM / FASTA readers), so its BAM decode is the shim's, not htslib's -- stated in the output."""
import contextlib, io, json, os, shutil, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import load_reference
from gci_amd import synth
from gci_amd.formats import bam as bamfmt
from oracle import gci_oracle as O

tmp = tempfile.mkdtemp(prefix="gci_pvr_")
try:
    rs = synth.simulate_reads(synth.CTG1, 30, "hifi", seed=synth.seed_for(1, 0))
    aligned = rs.aligned_bases()
    bam, fa = os.path.join(tmp, "hifi.bam"), os.path.join(tmp, "ref.fa")
    synth.write_bam_file(bam, rs, level=1, threads=4)
    synth.write_reference_fasta(fa, synth.CTG1)
    ref = load_reference.load()
    walls = {}
    for t in (1, 8):
        od = os.path.join(tmp, "ref_t%d" % t)
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            ref.GCI(hifi=[bam], nano=None, directory=od, prefix="GCI", reference=fa, threads=t, force=True)
        walls[t] = time.perf_counter() - t0
    # the port on the same file: inflate + offsets through the repo's pure-Python container code, then the oracle path
    O.build()
    t0 = time.perf_counter()
    stream, hdr, offs = bamfmt.read_bam(bam)
    t_read = time.perf_counter() - t0
    out = O.run_path(hifi=dict(bam=[(stream, offs, list(hdr.references))]), references=list(hdr.references),
                     lengths=list(hdr.lengths))
    port = time.perf_counter() - t0
    res = {"config": "configs[0]: ctg1 5,000,000 bp, 30x HiFi, one BAM", "records": len(rs), "aligned_bases": aligned,
           "reference_seconds_t1": walls[1], "reference_seconds_t8": walls[8], "port_seconds_1core": port, "port_seconds_1core_in_memory": port - t_read,
           "port_in_memory_over_reference_t1": walls[1] / (port - t_read),
           "port_in_memory_gbases_per_s_1core": aligned / (port - t_read) / 1e9,
           "reference_gbases_per_s_t1": aligned / walls[1] / 1e9, "reference_gbases_per_s_t8": aligned / walls[8] / 1e9,
           "port_gbases_per_s_1core": aligned / port / 1e9, "port_over_reference_t1": walls[1] / port,
           "port_over_reference_t8": walls[8] / port, "host": "build container, %d cores" % (os.cpu_count() or 0),
           "note": ("reference = /root/reference/GCI.py unmodified, GCI() whole run incl. write_depth, through tools/ref_shim "
                    "(pysam stand-in: BAM decode is the shim's, not htslib's); port = oracle/gci_oracle.{c,py} run_path on the "
                    "same file; port_seconds_1core includes reading the BGZF container with the repo's pure-Python reader, "
                    "*_in_memory = from the inflated stream on, which is what bench.py's cpu_baseline times")}
    json.dump(res, open(os.path.join(ROOT, "profiles", "port_vs_reference.json"), "w"), indent=1)
    print(json.dumps(res, indent=1))
finally:
    shutil.rmtree(tmp, ignore_errors=True)
