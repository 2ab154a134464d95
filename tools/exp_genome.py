"""L-side of the path at BASELINE configs[2] geometry (CHM13, 3.117 Gb, 40x => 6.9 M intervals): timing of the fused
depth build far outside the 256 MB Infinity Cache (track 12.5 GB, text ~9 GB)."""
import sys, json, numpy as np, torch, ctypes
sys.path.insert(0, '.')
from gci_amd import synth, _lib
from gci_amd.device import Engine
from gci_amd._lib import BuildOpts
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
track = e.new_track()
nc = len(lens)
toff = torch.zeros(nc + 1, dtype=torch.int64, device=e.device); sums = torch.zeros(nc, dtype=torch.int64, device=e.device)
keys = torch.empty(1 << 20, dtype=torch.int64, device=e.device); nk = torch.zeros(1, dtype=torch.int32, device=e.device)
o = BuildOpts(); o.flank = 15; o.want_text = 1; o.d_contig_text_off = toff.data_ptr(); o.d_sums = sums.data_ptr()
o.d_n_keys = nk.data_ptr(); o.d_keys = keys.data_ptr(); o.key_cap = 1 << 20; o.issue_flank = 15; o.lo = -1.0; o.hi = 0.0
p = lambda t: ctypes.c_void_p(t.data_ptr())
e._chk(e.lib.gci_depth_build_begin(e.ctx, p(d_ivl), None, n, ctypes.byref(o)), "b")
tb = int(toff[nc].item())
text = torch.empty(tb + 64, dtype=torch.uint8, device=e.device)
e.profile_enable((1 << _lib.PROF_COUNT) - 1); e.profile_read()
for _ in range(5):
    e._chk(e.lib.gci_depth_build_begin(e.ctx, p(d_ivl), None, n, ctypes.byref(o)), "b")
    e._chk(e.lib.gci_depth_build_finish(e.ctx, p(track), p(text), tb + 64), "f")
pr = {k: round(ms / cnt * 1e3, 1) for k, (ms, cnt) in e.profile_read().items()}
t2 = pr["k_tile_build"]
print(json.dumps({"intervals": n, "bases": int(lens.sum()), "text_bytes": tb, "us": pr,
                  "tile_build2_GBps": round((4 * lens.sum() + tb) / (t2 * 1e-6) / 1e9, 1),
                  "aligned_Gbases_per_s_Lside": round(float(span.sum()) / (sum(v for k, v in pr.items()) * 1e-6) / 1e9, 1)}))
