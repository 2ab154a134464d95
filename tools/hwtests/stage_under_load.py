#!/usr/bin/env python3
"""The library's staging ring (gci_stage_send: tmpfs file -> pinned slots -> device) with the device idle and with the wave inflate
running beside it on another stream of the same process."""
import ctypes, os, sys, tempfile, threading, time
HERE = os.path.dirname(os.path.abspath(__file__)); sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import numpy as np, torch
from gci_amd import synth, hostio
from gci_amd.device import Engine
from gci_amd.formats import bam as bamfmt
rs = synth.simulate_reads((("chr19", int(61_707_364 * float(sys.argv[1]) if len(sys.argv) > 1 else 15_426_841)),), 40, "hifi", seed=synth.seed_for(2, 0))
stream, _ = synth.to_bam_stream(rs, seq_qual="random", seed=7)
pb = os.path.join(tempfile.mkdtemp(), "x.bam"); bamfmt.write_bam_stream(pb, stream, level=1, threads=hostio.default_threads())
raw = np.fromfile(pb, dtype=np.uint8); pos, isz = hostio.bgzf_blocks(raw)
e = Engine(0); lib = e.lib; d_raw = e.upload_padded(raw); torch.cuda.synchronize()
p = "/dev/shm/_stage_bw.bin"
blk = np.random.default_rng(1).integers(0, 256, 64 << 20, dtype=np.uint8).tobytes()
with open(p, "wb") as f:
    for _ in range(128): f.write(blk)                       # 8 GB
n = os.path.getsize(p)
dst = torch.empty(n, dtype=torch.uint8, device="cuda"); st = torch.cuda.Stream()
h = ctypes.c_void_p(); assert lib.gci_stage_create(e.ctx, 64 << 20, 4, 12, ctypes.byref(h)) == 0
def send():
    rawm = np.memmap(p, dtype=np.uint8, mode="r")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    assert lib.gci_stage_send(e.ctx, h, ctypes.c_void_p(rawm.ctypes.data), n, ctypes.c_void_p(dst.data_ptr()), ctypes.c_void_p(st.cuda_stream), 1, 1) == 0
    st.synchronize(); dt = time.perf_counter() - t0; del rawm
    return n / dt / 1e9
print("device idle: %.1f, %.1f GB/s" % (send(), send()), flush=True)
stop = False
def load():
    with torch.cuda.stream(e.stream):
        while not stop:
            for _ in range(8): e.bgzf_inflate(None, pos, isz, check_crc=True, d_raw=d_raw)
            e.stream.synchronize()
th = threading.Thread(target=load); th.start(); time.sleep(0.5)
print("beside the inflate: %.1f, %.1f, %.1f GB/s" % (send(), send(), send()), flush=True)
stop = True; th.join()
print("device idle again: %.1f GB/s" % send(), flush=True)
os.remove(p)
