"""The single-GPU command line holds its HBM buffers without a tensor library (gci_amd/hbm.py over the gci_dev_* exports of
libgci_hip.so): the provider's own behaviour, an Engine made with it against the oracle and against the torch-backed Engine, and
`python GCI.py ...` as a process of its own -- every golden case of the unmodified reference, byte for byte, in a process that never
imports torch and leaves through the interpreter's ordinary exit."""
import os
import subprocess
import sys
import time

import numpy as np
import pytest

from golden_util import CASES, GOLDEN, cli_args, expected, images, manifest, read_outputs

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def native_engine():
    from gci_amd.device import Engine
    e = Engine(0, backend="native")
    yield e
    e.close()


def test_provider_buffers_streams_and_the_allocator():
    from gci_amd import hbm
    T = hbm.native()
    assert T.is_available()
    a = np.arange(10_000, dtype=np.int64)
    d = T.from_numpy(a)
    assert d.shape == (10_000,) and d.dtype is T.int64 and np.array_equal(d.cpu().numpy(), a)
    assert int(d[77].item()) == 77 and np.array_equal(d[100:200].cpu().numpy(), a[100:200]) and d[5000:].shape == (5000,)
    assert np.array_equal(T.add_i64(d, -3).cpu().numpy(), a - 3)
    u = T.from_numpy(np.array([5, 0, 0xFFFFFFFF, 7], dtype=np.uint32))
    assert T.scan_u32_u64(u).cpu().numpy().tolist() == [0, 5, 5, 5 + 0xFFFFFFFF, 12 + 0xFFFFFFFF]
    big = np.random.default_rng(1).integers(0, 2**32, 50_000, dtype=np.uint64).astype(np.uint32)
    assert np.array_equal(T.scan_u32_u64(T.from_numpy(big)).cpu().numpy()[1:], np.cumsum(big.astype(np.uint64)).astype(np.int64))
    recs = T.from_numpy(np.full((1000, 32), 0xFF, dtype=np.uint8))
    T.rec_flags_and(recs, 3)
    h = recs.cpu().numpy()
    assert (h[:, 29] == 3).all() and (np.delete(h, 29, axis=1) == 0xFF).all()
    # two-dimensional buffers: slices and rows along the first axis, views of the last
    m = T.from_numpy(np.arange(12, dtype=np.int32).reshape(3, 4))
    assert m[1:].shape == (2, 4) and m[2].cpu().numpy().tolist() == [8, 9, 10, 11] and m.view(T.uint8).shape == (3, 16)
    c = T.cat([m[:1], m[1:]])
    assert np.array_equal(c.cpu().numpy(), np.arange(12, dtype=np.int32).reshape(3, 4))
    z = T.zeros(33, T.uint8)
    z[3:9] = T.from_numpy(np.arange(6, dtype=np.uint8))
    z[20:24] = 0xAB
    want = np.zeros(33, np.uint8); want[3:9] = np.arange(6); want[20:24] = 0xAB
    assert np.array_equal(z.cpu().numpy(), want) and np.array_equal(z.clone().cpu().numpy(), want)
    st = T.zeros(4, T.int64)
    st[2] = -1
    assert st.cpu().numpy().tolist() == [0, 0, -1, 0]
    # the allocator: a block given back is handed to the next request of its size class on the same stream, not to another stream's
    x = T.empty(5 << 20, T.uint8)
    p = x.data_ptr()
    del x
    y = T.empty((5 << 20) - 1000, T.uint8)
    assert y.data_ptr() == p
    side = T.Stream(0)
    del y
    with T.stream(side):
        w = T.empty(5 << 20, T.uint8)
    assert w.data_ptr() != p
    # record_stream: the block comes back with an event of the other stream; the next owner waits for it on the device
    main = T.current_stream(0)
    src = T.from_numpy(np.full(8 << 20, 7, dtype=np.uint8))
    buf = T.empty(8 << 20, T.uint8)
    q = buf.data_ptr()
    buf.record_stream(side)
    side.wait_stream(main)
    with T.stream(side):
        buf.copy_(src)
        got = buf.cpu().numpy()
    del buf
    again = T.empty(8 << 20, T.uint8)
    assert again.data_ptr() == q and (got == 7).all()
    again.zero_()
    assert int(again.cpu().numpy().max()) == 0
    # pinned host memory and events
    pin = T.pinned(1 << 20)
    pin.numpy()[:] = 9
    dev = T.empty(1 << 20, T.uint8)
    dev.copy_(pin, non_blocking=True)
    back = T.pinned(1 << 20)
    back[:1 << 20].copy_(dev, non_blocking=True)
    a0, a1 = T.Event(enable_timing=True), T.Event(enable_timing=True)
    a0.record(); dev.zero_(); a1.record(); a1.synchronize()
    assert a0.elapsed_time(a1) >= 0 and (back.numpy() == 9).all()
    held = hbm.memory_held(0)
    del d, u, recs, m, c, z, st, w, src, again, dev
    hbm.empty_cache(0)
    assert hbm.memory_held(0) < held


def test_arena_cuts_blocks_from_slabs_and_coalesces():
    """gci_dev_malloc / gci_dev_free (the arena of k_hbm.hip): blocks come out of slabs without a driver call each, a freed block
    merges with its free neighbours (the slab's space is whole again after any order of frees), a request beyond every slab makes a
    slab of its own, and the library's contexts draw from the same arena."""
    import ctypes
    from gci_amd import _lib
    lib = _lib.load()

    def info():
        r, u, n = ctypes.c_uint64(0), ctypes.c_uint64(0), ctypes.c_uint32(0)
        assert lib.gci_dev_arena_info(0, ctypes.byref(r), ctypes.byref(u), ctypes.byref(n)) == 0
        return r.value, u.value, n.value

    def alloc(nbytes):
        p = ctypes.c_void_p()
        assert lib.gci_dev_malloc(0, nbytes, ctypes.byref(p)) == 0
        return p.value

    if os.environ.get("GCI_ARENA", "1") == "0":
        pytest.skip("arena off")
    first = alloc(1)                                           # (at least one slab exists from here on)
    r0, u0, n0 = info()
    sizes = [4096, 100_000, 3 << 20, 1, 40 << 20, 12_345_678, 65_536, 2 << 20]
    ptrs = [alloc(s) for s in sizes]
    r1, u1, n1 = info()
    assert len(set(ptrs)) == len(ptrs) and all(p % 4096 == 0 for p in ptrs)
    assert u1 - u0 == sum((s + 4095) // 4096 * 4096 for s in sizes)
    spans = sorted((p, (s + 4095) // 4096 * 4096) for p, s in zip(ptrs, sizes))
    assert all(a + n <= b for (a, n), (b, _) in zip(spans, spans[1:]))          # no two blocks overlap
    for k in (3, 0, 7, 5, 1, 6, 2, 4):                                           # freed in a scrambled order
        assert lib.gci_dev_free(0, ctypes.c_void_p(ptrs[k])) == 0
    r2, u2, n2 = info()
    assert u2 == u0 and r2 == r1
    again = alloc(sum(sizes))                                                    # fits only where the freed blocks have merged
    r3, _, n3 = info()
    assert (r3, n3) == (r2, n2) or n3 == n2 + 1                                  # (a small first slab may have been outgrown: then one more)
    assert lib.gci_dev_free(0, ctypes.c_void_p(again)) == 0
    big = alloc(r3 + (1 << 20))                                                  # larger than everything reserved: a slab of its own
    r4, _, n4 = info()
    assert n4 == n3 + 1 and r4 >= 2 * r3
    assert lib.gci_dev_free(0, ctypes.c_void_p(big)) == 0 and lib.gci_dev_free(0, ctypes.c_void_p(first)) == 0
    got = ctypes.c_uint64(0)
    assert lib.gci_dev_reserve(0, r4 // 2, ctypes.byref(got)) == 0 and got.value == r4      # already there: nothing new
    # a context's scratch comes from the arena too
    from gci_amd.device import Engine
    _, ua, _ = info()
    e = Engine(0, backend="native")
    e.set_layout([5_000_000])
    _, ub, _ = info()
    e.close()
    _, uc, _ = info()
    assert ub > ua and uc <= ua + (64 << 20)


def test_native_engine_against_the_oracle_and_the_torch_engine(native_engine, engine, oracle, tmp_path):
    """The smoke path -- pages, paged filter, join, depth build, issue scan, text, .depth.gz members, BGZF inflate -- through an
    Engine whose buffers are the library's own: equal to the oracle, and byte-equal to the torch-backed Engine's outputs."""
    import gzip
    from gci_amd import hbm, pipeline, synth, hostio
    from gci_amd.device import JoinInput
    from gci_amd.formats import bam as bamfmt
    assert native_engine.T is hbm.native() and engine.T.name == "torch"
    contigs = (("ctgA", 1_500_000), ("ctgB", 500_000), ("ctgC", 4097))
    rs = synth.simulate_reads(contigs, 20, "hifi", seed=synth.seed_for(0, 3))
    stream, offs = synth.to_bam_stream(rs)
    targets, tl = [n for n, _ in contigs], dict(contigs)
    outs = []
    for eng in (native_engine, engine):
        eng.set_layout([l for _, l in contigs])
        pages = eng.bam_pages(eng.to_device(stream), eng.to_device(offs), True)
        recs, noff = eng.bam_filter_pages(pages, eng.to_device(np.arange(len(contigs), dtype=np.int32)), 30, 50, 0.1, 0.9)
        ivl, cnt = eng.name_join([JoinInput(recs, pages.buf, noff, 0)], 0.9, count_flank=15)
        track = eng.new_track()
        fused = eng.depth_build_fused(ivl, cnt, 15, track, want_text=True, want_sums=True, issue=(-1, 0, 15), counted=True)
        depths = pipeline.DepthTracks(eng, tl, track)
        bed = pipeline.collapse_depth_range(depths, -1, 0, 15, 0)
        members = [bytes(b) for b in eng.depth_deflate(track)]
        outs.append(dict(recs=recs.cpu().numpy().copy(), n=int(cnt.item()), track=track.cpu().numpy().copy(), text=fused["text"].cpu().numpy().tobytes(),
                         sums=np.asarray(fused["sums"]).copy(), runs=[r.tolist() for r in fused["runs"]], bed=bed, members=members,
                         mean=depths.mean(), host={t: depths[t] for t in targets}))
    nat, tor = outs
    for k in ("n", "text", "runs", "bed", "members", "mean"):
        assert nat[k] == tor[k], k
    assert np.array_equal(nat["recs"], tor["recs"]) and np.array_equal(nat["track"], tor["track"]) and np.array_equal(nat["sums"], tor["sums"])
    d, hq = oracle.bam_file_dict(stream, offs, targets, targets, 30, 50, 0.1, 0.9)
    want = oracle.depth_build(oracle.name_join([d], hq, 0.9), tl, 15)
    for t in targets:
        assert np.array_equal(nat["host"][t], want[t]), t
    assert nat["bed"] == oracle.collapse_depth_range(want, -1, 0, 15, 0)
    assert nat["text"] == b"".join(oracle.depth_text_contig(want[t]) for t in targets)
    for c, t in enumerate(targets):
        assert gzip.decompress(nat["members"][c]) == oracle.depth_text_contig(want[t])
    # N1 through the native provider: the file's bytes uploaded, inflated on the device, CRC checked
    p = str(tmp_path / "n.bam")
    bamfmt.write_bam_stream(p, stream, level=6, threads=2)
    raw = np.fromfile(p, dtype=np.uint8)
    pos, isz = hostio.bgzf_blocks(raw)
    assert native_engine.bgzf_inflate(raw, pos, isz).cpu().numpy().tobytes() == stream.tobytes()
    # ... and the whole of filter() (ingestion in runs of members, carried records, concatenated parts) on either provider
    got = []
    for eng in (native_engine, engine):
        ji = pipeline.bam_join_input(eng, p, targets, (30, 50, 0.1, 0.9), threads=2, chunk_bytes=None)
        got.append(ji.recs.cpu().numpy().copy())
    assert np.array_equal(got[0], got[1]) and np.array_equal(got[0], nat["recs"])


def _run_cli(args, env_extra=None, timeout=600):
    env = dict(os.environ, GCI_ASSERT_NO_TORCH="1", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    env.pop("GCI_HBM", None)
    env.update(env_extra or {})
    t = time.perf_counter()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "GCI.py")] + args[1:], capture_output=True, text=True, timeout=timeout, env=env)
    return r, time.perf_counter() - t


@pytest.mark.parametrize("case", CASES)
def test_command_line_as_a_process_without_torch(case, tmp_path):
    """`python GCI.py ...` on every golden case of the unmodified reference: its files byte for byte, its figures pixel for pixel,
    its transcript -- from a process that has not imported torch (GCI_ASSERT_NO_TORCH makes the command line check that itself
    before it leaves) and ends through the interpreter's own exit."""
    out = str(tmp_path / "out")
    r, _ = _run_cli(cli_args(case, out))
    assert r.returncode == 0, r.stderr[-3000:]
    got, want = read_outputs(out), expected(case)
    assert sorted(got) == sorted(want)
    for fn in want:
        assert got[fn] == want[fn], fn
    got_img, want_img = images(out), images(os.path.join(GOLDEN, case, "expected"))
    assert sorted(got_img) == sorted(want_img)
    for fn in want_img:
        assert np.array_equal(got_img[fn], want_img[fn]), fn
    first, _, rest = r.stdout.partition("\n")
    assert first.startswith("Used arguments:{")
    inp = os.path.join(GOLDEN, case, "inputs")
    assert rest.replace(out, "{OUT}").replace(inp, "{IN}") == manifest(case)["stdout"]
    # the overwrite guard exits with status 1 and the reference's message, from the same kind of process
    r2, _ = _run_cli(cli_args(case, out))
    assert r2.returncode == 1 and "exists" in r2.stderr and "--force" in r2.stderr


def test_command_line_provider_switch_and_streamed_ingestion(tmp_path):
    """GCI_HBM=torch runs the same command line on torch buffers (what a rank of --gpus N uses): the same files; and a BAM that
    goes through the device run by run of members (GCI_GPU_INFLATE_MAX / GCI_BAM_CHUNK_BYTES made small: carried records, the
    upload of run k + 1 beside the inflate of run k, parts put together) on the native provider."""
    case = "c3_two_bam"
    base = str(tmp_path / "a")
    r, _ = _run_cli(cli_args(case, base))
    assert r.returncode == 0, r.stderr[-2000:]
    want = read_outputs(base)
    for k, env in enumerate((dict(GCI_HBM="torch", GCI_ASSERT_NO_TORCH="0"),
                             dict(GCI_GPU_INFLATE_MAX=str(1 << 16), GCI_BAM_CHUNK_BYTES=str(1 << 20), GCI_FIRST_RUN_AHEAD="1"))):
        out = str(tmp_path / ("b%d" % k))
        r, _ = _run_cli(cli_args(case, out), env)
        assert r.returncode == 0, r.stderr[-2000:]
        assert read_outputs(out) == want, env


def test_paf_files_through_the_staging_ring(tmp_path):
    """PAF files of GBs travel from their mappings through the ring of pinned slots to their places in one device buffer
    (Engine.paf_filter; GCI_PAF_STAGE_MIN makes the golden cases' small files go that way): the same files as the small-file way."""
    for case in ("c4_two_paf", "c5_two_type"):
        out = str(tmp_path / case)
        r, _ = _run_cli(cli_args(case, out), dict(GCI_PAF_STAGE_MIN="1"))
        assert r.returncode == 0, r.stderr[-2000:]
        got, want = read_outputs(out), expected(case)
        assert sorted(got) == sorted(want)
        for fn in want:
            assert got[fn] == want[fn], (case, fn)
