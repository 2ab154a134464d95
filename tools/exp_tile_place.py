"""Does k_tile_build's time (3.3 - 4.1 ms for the same binary, run to run) depend on WHERE its two outputs lie?  The genome-scale
L-side (CHM13, 40x: 12.5 GB track + 9.3 GB text per launch) with track and text carved out of one arena at controlled offsets,
and from separate allocations made in different orders.  One box, one call."""
import ctypes
import json
import sys

import numpy as np
import torch

sys.path.insert(0, '.')
from gci_amd import synth, _lib                 # noqa: E402
from gci_amd.device import Engine               # noqa: E402
from gci_amd._lib import BuildOpts              # noqa: E402

e = Engine(0)
lens = np.array([l for _, l in synth.CHM13], dtype=np.int64)
rng = np.random.default_rng(7)
n = int(40 * lens.sum() / 18000)
c = np.searchsorted(np.cumsum(lens), rng.integers(0, lens.sum(), n), side="right").astype(np.int32)
L = lens[c]
span = np.minimum(np.clip(rng.normal(18_000, 2_500, n), 5_000, 30_000).astype(np.int64), np.maximum(L - 1, 1))
s = (rng.random(n) * (L - span)).astype(np.int64)
ivl = np.stack([c, s.astype(np.int32), (s + span).astype(np.int32), np.zeros(n, np.int32)], axis=1).astype(np.int32)
e.set_layout(lens.tolist())
d_ivl = e.to_device(ivl)
nc = len(lens)
toff = torch.zeros(nc + 1, dtype=torch.int64, device=e.device)
sums = torch.zeros(nc, dtype=torch.int64, device=e.device)
keys = torch.empty(1 << 20, dtype=torch.int64, device=e.device)
nk = torch.zeros(1, dtype=torch.int32, device=e.device)
o = BuildOpts()
o.flank, o.want_text, o.d_contig_text_off, o.d_sums = 15, 1, toff.data_ptr(), sums.data_ptr()
o.d_n_keys, o.d_keys, o.key_cap, o.issue_flank, o.lo, o.hi = nk.data_ptr(), keys.data_ptr(), 1 << 20, 15, -1.0, 0.0
p = lambda t: ctypes.c_void_p(t.data_ptr())     # noqa: E731
first = e.new_track()
e._chk(e.lib.gci_depth_build_begin(e.ctx, p(d_ivl), None, n, ctypes.byref(o)), "b")
tb = int(toff[nc].item())
del first
track_bytes = int(e.total) * 4
e.profile_enable(1 << _lib.PROF_DEPTH_SCAN)


def run(track, text, label):
    times = []
    for _ in range(4):
        e.profile_read()
        e._chk(e.lib.gci_depth_build_begin(e.ctx, p(d_ivl), None, n, ctypes.byref(o)), "b")
        e._chk(e.lib.gci_depth_build_finish(e.ctx, p(track), p(text), int(text.shape[0])), "f")
        pr = e.profile_read()
        times.append(round(pr["k_tile_build"][0] / pr["k_tile_build"][1] * 1e3))
    print("%-46s track %#x  text %#x  k_tile_build us %s" % (label, track.data_ptr(), text.data_ptr(), times), flush=True)


# separate allocations, both orders
for order in ("track first", "text first"):
    torch.cuda.empty_cache()
    if order == "track first":
        tr = torch.empty(track_bytes // 4, dtype=torch.int32, device=e.device)
        tx = torch.empty(tb + 64, dtype=torch.uint8, device=e.device)
    else:
        tx = torch.empty(tb + 64, dtype=torch.uint8, device=e.device)
        tr = torch.empty(track_bytes // 4, dtype=torch.int32, device=e.device)
    run(tr, tx, "separate allocations, " + order)
    del tr, tx
torch.cuda.empty_cache()
arena = torch.empty(track_bytes + tb + (6 << 30), dtype=torch.uint8, device=e.device)
base = arena.data_ptr()
al = lambda x, a: (x + a - 1) // a * a           # noqa: E731
for gap_name, gap in (("0", 0), ("4 KiB", 4096), ("64 KiB", 65536), ("1 MiB", 1 << 20), ("2 MiB + 4 KiB", (2 << 20) + 4096), ("1 GiB", 1 << 30),
                      ("1 GiB + 1 MiB", (1 << 30) + (1 << 20)), ("3 GiB + 128 KiB", (3 << 30) + (128 << 10))):
    t0 = al(base, 2 << 20) - base                # track at a 2 MiB boundary
    x0 = al(t0 + track_bytes, 2 << 20) + gap     # text behind it, at 2 MiB + gap
    tr = arena[t0:t0 + track_bytes].view(torch.int32)
    tx = arena[x0:x0 + tb + 64]
    run(tr, tx, "one arena, text at 2 MiB boundary + " + gap_name)
print(json.dumps({"track_bytes": track_bytes, "text_bytes": tb}))
