#!/usr/bin/env python3
"""Build tools/hwtests/inflate_wave.hip (a WAVE per BGZF member: lanes at guessed bit offsets, stitched) and run it on the GPU against
a HiFi BAM with SEQ / QUAL of realistic entropy: every member it reports as decoded is compared with the known inflated stream, the
others are counted by reason; time per launch next to the product's lane-per-member kernel.
NOT YET RUN when committed (round 4 ended without GPU minutes); its logic is the one tools/hwtests/inflate_wave_host_check.cpp plays
on the host with the same helpers (5 459 of 5 468 members decoded and equal to zlib, 9 handed back).
Usage: inflate_wave.py [scale of chr19] [grid]"""
import ctypes, os, subprocess, sys, tempfile, time
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import numpy as np, torch
from gci_amd import synth, hostio
from gci_amd.device import Engine
from gci_amd.formats import bam as bamfmt

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.25
grid = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
stage = os.environ.get("IW_STAGE", "4")
so = os.path.join(HERE, "libinflate_wave_s%s.so" % stage)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-DIW_STAGE=" + stage, "-o", so,
                       os.path.join(HERE, "inflate_wave.hip")])
lib = ctypes.CDLL(so)
lib.inflate_wave_launch.restype = ctypes.c_int
lib.inflate_wave_launch.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p]
cap = int(lib.inflate_wave_match_cap())

rs = synth.simulate_reads((("chr19", int(61_707_364 * scale)),), 40, "hifi", seed=synth.seed_for(2, 0))
stream, _ = synth.to_bam_stream(rs, seq_qual="random", seed=7)
p = os.path.join(tempfile.mkdtemp(), "x.bam")
bamfmt.write_bam_stream(p, stream, level=1, threads=hostio.default_threads())
raw = np.fromfile(p, dtype=np.uint8)
pos, isz = hostio.bgzf_blocks(raw)
n = int(isz.shape[0])
off = np.concatenate([[0], np.cumsum(isz)]).astype(np.uint64)
dev = torch.device("cuda", 0)
d_raw = torch.zeros(raw.shape[0] + 16, dtype=torch.uint8, device=dev); d_raw[:raw.shape[0]] = torch.from_numpy(raw).to(dev)
d_pos = torch.from_numpy(pos.view(np.int64)).to(dev); d_off = torch.from_numpy(off.view(np.int64)).to(dev)
d_out = torch.zeros(int(off[-1]) + 16, dtype=torch.uint8, device=dev)
d_matches = torch.empty(grid * cap * 2, dtype=torch.int32, device=dev)
d_status = torch.full((n,), -1, dtype=torch.int32, device=dev)
st = torch.cuda.current_stream().cuda_stream
ms = []
for _ in range(3):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    rc = lib.inflate_wave_launch(d_raw.data_ptr(), d_pos.data_ptr(), d_off.data_ptr(), n, d_out.data_ptr(), d_matches.data_ptr(), grid,
                                 d_status.data_ptr(), ctypes.c_void_p(st))
    b.record(); torch.cuda.synchronize()
    assert rc == 0, rc
    ms.append(a.elapsed_time(b))
status = d_status.cpu().numpy()
out = d_out.cpu().numpy()
names = ["ok", "header", "no meeting point", "false end of block", "undecodable", "capacity", "length", "lanes", "cut"]
print("IW_STAGE", stage)
print("%d members, %.1f MB -> %.1f MB; launches %s ms (v0 copies: one match at a time)" % (n, raw.shape[0] / 1e6, stream.shape[0] / 1e6, ["%.1f" % x for x in ms]))
print("by status:", {names[k]: int((status == k).sum()) for k in range(9) if (status == k).any()})
bad = 0
for m in np.flatnonzero(status == 0).tolist():
    a, b = int(off[m]), int(off[m + 1])
    if not np.array_equal(out[a:b], stream[a:b]):
        bad += 1
print("members reported ok that differ from the stream:", bad)
e = Engine(0)
torch.cuda.synchronize(); t0 = time.perf_counter(); d = e.bgzf_inflate(raw, pos, isz, check_crc=False); torch.cuda.synchronize()
print("product (lane per member) incl. upload: %.1f ms" % ((time.perf_counter() - t0) * 1e3))
