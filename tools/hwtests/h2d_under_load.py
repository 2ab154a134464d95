#!/usr/bin/env python3
"""Host-to-device rate of pinned 64 MB pieces on one / two / four copy streams, with the device idle and with the wave inflate running
beside them (what the staging ring meets at genome size)."""
import os, sys, tempfile, threading, time
HERE = os.path.dirname(os.path.abspath(__file__)); sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import numpy as np, torch
from gci_amd import synth, hostio
from gci_amd.device import Engine
from gci_amd.formats import bam as bamfmt
rs = synth.simulate_reads((("chr19", int(61_707_364 * 0.25)),), 40, "hifi", seed=synth.seed_for(2, 0))
stream, _ = synth.to_bam_stream(rs, seq_qual="random", seed=7)
p = os.path.join(tempfile.mkdtemp(), "x.bam"); bamfmt.write_bam_stream(p, stream, level=1, threads=hostio.default_threads())
raw = np.fromfile(p, dtype=np.uint8); pos, isz = hostio.bgzf_blocks(raw)
e = Engine(0); d_raw = e.upload_padded(raw); torch.cuda.synchronize()
SL = 64 << 20; N = 48
src = [torch.empty(SL, dtype=torch.uint8).pin_memory() for _ in range(8)]
dst = torch.empty(N * SL, dtype=torch.uint8, device="cuda")
def copies(n_streams):
    ss = [torch.cuda.Stream() for _ in range(n_streams)]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(N):
        with torch.cuda.stream(ss[k % n_streams]):
            dst[k * SL:(k + 1) * SL].copy_(src[k % 8], non_blocking=True)
    for s in ss: s.synchronize()
    return N * SL / (time.perf_counter() - t0) / 1e9
stop = False
def load():
    with torch.cuda.stream(e.stream):
        while not stop:
            e.bgzf_inflate(None, pos, isz, check_crc=True, d_raw=d_raw)
            e.stream.synchronize()
for n in (1, 2, 4): print("idle device, %d copy stream(s): %.1f GB/s" % (n, max(copies(n) for _ in range(2))), flush=True)
th = threading.Thread(target=load); th.start(); time.sleep(0.3)
for n in (1, 2, 4): print("beside the inflate, %d copy stream(s): %.1f GB/s" % (n, max(copies(n) for _ in range(2))), flush=True)
stop = True; th.join()
