"""Experiment: split of k_tile_build between the depth write and the text render."""
import sys, ctypes, numpy as np, torch
sys.path.insert(0, '.')
from gci_amd import synth, _lib
from gci_amd.device import Engine, JoinInput
from gci_amd._lib import BuildOpts
e = Engine(0)
L = 61_707_364
rs = synth.simulate_reads((("chr19", L),), 40, "hifi", seed=synth.seed_for(2, 0))
stream, offs = synth.to_bam_stream(rs)
e.set_layout([L])
d_bam, d_off = e.to_device(stream), e.to_device(offs)
recs = e.bam_filter(d_bam, d_off, e.to_device(np.zeros(1, np.int32)), 30, 50, 0.1, 0.9)
ivl, cnt = e.name_join([JoinInput(recs, d_bam, d_off, 36)], 0.9)
track = e.new_track()
p = lambda t: ctypes.c_void_p(t.data_ptr())
toff = torch.zeros(2, dtype=torch.int64, device=e.device)
sums = torch.zeros(1, dtype=torch.int64, device=e.device)
def run(label, want_text, want_sums):
    o = BuildOpts(); o.flank = 15; o.want_text = int(want_text)
    o.d_contig_text_off = toff.data_ptr() if want_text else None
    o.d_sums = sums.data_ptr() if want_sums else None
    e._chk(e.lib.gci_depth_build_begin(e.ctx, p(ivl), p(cnt), int(ivl.shape[0]), ctypes.byref(o)), "b")
    text = torch.empty(int(toff[1].item()) + 64, dtype=torch.uint8, device=e.device) if want_text else None
    e.profile_enable((1 << _lib.PROF_COUNT) - 1); e.profile_read()
    for _ in range(10):
        e._chk(e.lib.gci_depth_build_begin(e.ctx, p(ivl), p(cnt), int(ivl.shape[0]), ctypes.byref(o)), "b")
        e._chk(e.lib.gci_depth_build_finish(e.ctx, p(track), p(text) if want_text else None, int(text.shape[0]) if want_text else 0), "f")
    pr = e.profile_read()
    print(label, {k: round(ms / n * 1e3, 1) for k, (ms, n) in pr.items()})
run("depth only          ", False, False)
run("depth + sums        ", False, True)
run("depth + text + sums ", True, True)
