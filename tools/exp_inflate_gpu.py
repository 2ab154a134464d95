#!/usr/bin/env python3
"""gci_bgzf_inflate_device alone on a HiFi BAM with realistic SEQ / QUAL entropy: time per launch (HIP events around the call)
and output rate.  Usage: exp_inflate_gpu.py [scale] [level]"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gci_amd import synth, hostio
from gci_amd.device import Engine
from gci_amd.formats import bam as bamfmt
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.25
level = int(sys.argv[2]) if len(sys.argv) > 2 else 1
contigs = (("chr19", int(61_707_364 * scale)),)
rs = synth.simulate_reads(contigs, 40, "hifi", seed=synth.seed_for(2, 0))
stream, _ = synth.to_bam_stream(rs, seq_qual="random", seed=7)
p = os.path.join(tempfile.mkdtemp(), "x.bam")
bamfmt.write_bam_stream(p, stream, level=level, threads=hostio.default_threads())
raw = np.fromfile(p, dtype=np.uint8)
pos, isz = hostio.bgzf_blocks(raw)
e = Engine(0)
d = e.bgzf_inflate(raw, pos, isz)
assert os.environ.get('GCI_EXP_NOCHECK') or np.array_equal(d.cpu().numpy(), stream)
for crc in (True, False):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    d = e.bgzf_inflate(raw, pos, isz, check_crc=crc)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("level %d crc %s: %d members, %.1f MB -> %.1f MB in %.4f s = %.1f GB/s out (incl. H2D of the file)" % (
        level, crc, isz.shape[0], raw.shape[0] / 1e6, stream.shape[0] / 1e6, dt, stream.shape[0] / dt / 1e9), flush=True)
t0 = time.perf_counter(); h = hostio.bgzf_inflate(raw, threads=hostio.default_threads(), check_crc=True); dt = time.perf_counter() - t0
print("host zlib, %d threads: %.4f s = %.1f GB/s" % (hostio.default_threads(), dt, stream.shape[0] / dt / 1e9))
