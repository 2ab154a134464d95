#!/usr/bin/env python3
"""Which of bam_join_input's locals costs 18 ms when it goes out of scope (cProfile books it on the caller, pipeline.filter): the
steps of the device-ingest branch one by one on the chr19 40x file with realistic SEQ / QUAL, each `del` timed."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gci_amd import hostio, pipeline, synth
from gci_amd.device import Engine
from gci_amd.formats import bam as bamfmt
rs = synth.simulate_reads((("chr19", 61_707_364),), 40, "hifi", seed=synth.seed_for(2, 0))
stream, _ = synth.to_bam_stream(rs, seq_qual="random", seed=7)
p = os.path.join(tempfile.mkdtemp(), "x.bam")
bamfmt.write_bam_stream(p, stream, level=1, threads=hostio.default_threads())
del stream, rs
eng = Engine(0)
filt = (30, 50, 0.1, 0.9)
for it in range(3):
    T = [("start", time.perf_counter())]
    mark = lambda s: T.append((s, time.perf_counter()))      # noqa: E731
    raw = np.memmap(p, dtype=np.uint8, mode="r"); mark("memmap")
    upload = eng.start_upload(raw, parts=2); mark("start_upload")
    pos, isz = hostio.bgzf_blocks(np.asarray(raw)); mark("bgzf_blocks")
    ji = pipeline._bam_join_input_gpu(eng, p, raw, pipeline._Members(eng, pipeline.BAM_CHUNK_BYTES, pos, isz), lambda hdr: eng.to_device(np.zeros(1, np.int32)), filt, upload)
    mark("_bam_join_input_gpu")
    torch.cuda.synchronize(); mark("sync")
    up_keys = list(upload.keys())
    d_raw = upload.pop("d_raw"); del d_raw; mark("del d_raw")
    pool = upload.pop("pool"); pool.shutdown(); del pool; mark("pool.shutdown")
    del upload; mark("del upload")
    del raw; mark("del raw (munmap)")
    del ji; mark("del ji")
    torch.cuda.synchronize(); mark("sync")
    print("iteration %d: " % it + ", ".join("%s %.1f" % (s, (t - T[k][1]) * 1e3) for k, (s, t) in enumerate(T[1:])) + " ms", flush=True)
