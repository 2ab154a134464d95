"""Genome-scale synthetic inputs for BASELINE.json configs[2]: CHM13 geometry (25 contigs,
3.117 Gb), two 40x HiFi alignment files of the same reads -- the second one perturbed the way another aligner's
output differs (`synth.perturb`), which is what the `-op` join (GCI.py:272-301) exists for.

The files are produced as HEADS STREAMS (the records without SEQ / QUAL, what `gci_bam_heads` makes of a BGZF file and
what the command line uploads): the whole inflated files would be 2 x 190 GB.  Generation runs group by group of
contigs (a read set of the whole genome would need ~60 GB of host memory), the groups in worker processes.

Used by bench.py (the driver-timed workload), tests/test_gpu_genome.py and tools/.
"""
from __future__ import annotations

import os
import sys
import time
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import synth
from .formats import bam as bamfmt


@dataclass
class AlignmentFile:
    """One input file as its heads stream."""
    stream: np.ndarray            # uint8: BAM header + records without SEQ / QUAL
    offsets: np.ndarray           # uint64 [R]: offset of every record's block_size word
    aligned_bases: int            # metric numerator: sum of reference spans of records with flag 0x4 clear
    k1_bytes: int                 # algorithmic bytes K1 reads of it: 36 + l_read_name + 4 n_cigar + NM tag
    name_bytes: int
    aligned_per_contig: Optional[np.ndarray] = None   # int64 [n_contigs]: aligned_bases split by the record's contig


@dataclass
class GenomeInput:
    contigs: Tuple[Tuple[str, int], ...]
    files: List[AlignmentFile]

    @property
    def names(self) -> List[str]:
        return [n for n, _ in self.contigs]

    @property
    def lengths(self) -> List[int]:
        return [int(l) for _, l in self.contigs]

    @property
    def aligned_bases(self) -> int:
        return sum(f.aligned_bases for f in self.files)


def contig_groups(contigs: Sequence[Tuple[str, int]], max_bases: float) -> List[List[int]]:
    """Consecutive contigs (header order) packed into groups of at most `max_bases`."""
    groups, cur, acc = [], [], 0
    for i, (_, l) in enumerate(contigs):
        if cur and acc + l > max_bases:
            groups.append(cur)
            cur, acc = [], 0
        cur.append(i)
        acc += l
    groups.append(cur)
    return groups


def _k1_algorithmic_bytes(rs: synth.ReadSet) -> int:
    n_ops = np.diff(rs.cigar_off)
    n_field = np.where(n_ops > 65535, 2, n_ops)
    nm_sz = np.where(rs.nm < 256, 1, np.where(rs.nm < 65536, 2, 4))
    return int((36 + np.char.str_len(rs.names) + 1 + 4 * n_field + 3 + nm_sz).sum())


def _gen_group(args):
    """Worker: the two files' records of one group of contigs -> per file (record bytes, offsets relative to the first
    record, aligned bases, K1 bytes, name bytes).  Seeds: SURVEY.md 8d, config 3 (index of configs[2] counted from 1)."""
    contigs, idx, g, coverage, kind, n_files = args
    sub = tuple(contigs[i] for i in idx)
    rs = synth.simulate_reads(sub, coverage, kind, seed=synth.seed_for(3, 0) + 7 * g, name_prefix="m64011_g%02d/" % g)
    files = [rs] + [synth.perturb(rs, synth.seed_for(3, f) + 7 * g) for f in range(1, n_files)]
    out = []
    for r in files:
        r.ref_id = (r.ref_id + idx[0]).astype(np.int32)          # the group's contigs are consecutive in the header
        mapped = (r.flag & 4) == 0
        per_contig = np.bincount(r.ref_id[mapped], weights=np.maximum(r.ref_span(), 1)[mapped].astype(np.float64),
                                 minlength=len(contigs)).astype(np.int64)          # spans < 2^53: exact
        aligned = int(per_contig.sum())
        r.contigs = tuple(contigs)
        s, o = synth.to_bam_stream(r, heads=True)
        first = bamfmt.parse_header(s).first_record
        out.append((s[first:].copy(), (o - np.uint64(first)).astype(np.uint64), aligned, _k1_algorithmic_bytes(r),
                    int(np.char.str_len(r.names).sum()), per_contig))
    return g, out


def genome_dual(scale: float = 1.0, coverage: float = 40.0, contigs: Optional[Sequence[Tuple[str, int]]] = None,
                n_files: int = 2, procs: Optional[int] = None, verbose: bool = False, kind: str = "hifi") -> GenomeInput:
    """configs[2]: `n_files` alignment files of the same simulated reads over CHM13 (every contig scaled by `scale`,
    at least 20 kb).  The result is deterministic (independent of `procs`)."""
    base = tuple(contigs) if contigs is not None else synth.CHM13
    ctg = tuple((n, max(20_000, int(l * scale))) for n, l in base) if scale != 1.0 else tuple(base)
    total = sum(l for _, l in ctg)
    groups = contig_groups(ctg, max(2.0e7, min(2.6e8, total / 12.0)))
    if procs is None:
        from . import hostio
        procs = hostio.default_threads()
    procs = max(1, min(int(procs), len(groups)))
    tasks = [(ctg, idx, g, coverage, kind, n_files) for g, idx in enumerate(groups)]
    t0 = time.time()
    results = {}
    if procs == 1:
        for t in tasks:
            g, out = _gen_group(t)
            results[g] = out
    else:
        import multiprocessing as mp
        from concurrent.futures import ProcessPoolExecutor
        # fork: the workers run numpy only (they never touch HIP, and leave through os._exit like DataLoader workers)
        with ProcessPoolExecutor(procs, mp_context=mp.get_context("fork")) as ex:
            for g, out in ex.map(_gen_group, sorted(tasks, key=lambda t: -sum(ctg[i][1] for i in t[1]))):
                results[g] = out
                if verbose:
                    print("workload: group %d/%d done, %.0f s" % (len(results), len(groups), time.time() - t0),
                          file=sys.stderr, flush=True)
    hdr = np.frombuffer(bamfmt.encode_header([n for n, _ in ctg], [l for _, l in ctg]), dtype=np.uint8)
    files = []
    for f in range(n_files):
        parts, offs, size = [hdr], [], int(hdr.shape[0])
        aligned = k1 = nb = 0
        per_contig = np.zeros(len(ctg), dtype=np.int64)
        for g in range(len(groups)):
            s, o, a, kb, n, pc = results[g][f]
            per_contig += pc
            parts.append(s)
            offs.append(o + np.uint64(size))
            size += int(s.shape[0])
            aligned += a
            k1 += kb
            nb += n
            results[g][f] = None
        files.append(AlignmentFile(np.concatenate(parts), np.concatenate(offs) if offs else np.zeros(0, np.uint64),
                                   aligned, k1, nb, per_contig))
    if verbose:
        print("workload: %d contigs, %d bp, records per file %s, %.0f s on %d processes" % (
            len(ctg), total, [int(f.offsets.shape[0]) for f in files], time.time() - t0, procs), file=sys.stderr, flush=True)
    return GenomeInput(ctg, files)


# ---- BASELINE configs[3] / configs[4]: two read types, BAM + PAF inputs ---------------------------------------------------------

@dataclass
class ReadTypeInput:
    """One read type's inputs (`--hifi` or `--nano`): one BAM (heads stream) and, optionally, one PAF (the file's bytes) of the
    same reads as another aligner reports them."""
    bam: AlignmentFile
    paf: Optional[np.ndarray]            # uint8: the PAF text
    paf_aligned_bases: int               # sum of (tend - tstart) over its lines (the metric's numerator counts PAF lines so)
    n_reads: int


@dataclass
class TwoTypeInput:
    contigs: Tuple[Tuple[str, int], ...]
    hifi: ReadTypeInput
    nano: ReadTypeInput
    gaps: dict                            # contig -> [(start, end)] N runs of the assembly (configs[4]: 20 of them)
    regions: List[Tuple[str, int, int]]   # -R regions (configs[4])

    @property
    def names(self) -> List[str]:
        return [n for n, _ in self.contigs]

    @property
    def lengths(self) -> List[int]:
        return [int(l) for _, l in self.contigs]

    @property
    def aligned_bases(self) -> int:
        return sum(t.bam.aligned_bases + t.paf_aligned_bases for t in (self.hifi, self.nano))


def diploid_contigs(scale: float = 1.0) -> Tuple[Tuple[str, int], ...]:
    """configs[4] geometry: two haplotype copies of the 23 nuclear CHM13 contigs with +-0.5 % length
    jitter, `mat_chrN` / `pat_chrN` -- 46 contigs, ~6.2 Gb."""
    rng = np.random.Generator(np.random.PCG64(synth.seed_for(5, 99)))
    out = []
    for hap in ("mat", "pat"):
        for n, l in synth.CHM13[:23]:
            out.append(("%s_%s" % (hap, n), max(20_000, int(l * scale * (1.0 + rng.uniform(-0.005, 0.005))))))
    return tuple(out)


def _memory_budget_gb() -> float:
    """What this process and its children may still allocate: the memory cgroup's limit less its current use when there is one
    (the GPU boxes: 300 GiB in a 3 TB machine -- MemAvailable alone says 2.9 TB), else MemAvailable."""
    avail = 64.0
    try:
        avail = [int(l.split()[1]) for l in open("/proc/meminfo") if l.startswith("MemAvailable:")][0] / 1e6
    except Exception:
        pass
    try:
        lim = open("/sys/fs/cgroup/memory.max").read().strip()
        if lim != "max":
            cur = int(open("/sys/fs/cgroup/memory.current").read().strip())
            avail = min(avail, (int(lim) - cur) / 1e9)
    except Exception:
        pass
    return max(1.0, avail)


class _MemoryGuard:
    """Watches the resident memory of this process and its children while a pool generates a workload; past `limit_gb` it
    terminates the pool's workers (its own children, by PID) so that the generation fails instead of the machine."""

    def __init__(self, executor, limit_gb: float):
        import threading
        self.ex, self.limit, self.tripped, self.peak = executor, float(limit_gb), False, 0.0
        self._stop = threading.Event()
        self._t = threading.Thread(target=self._run, daemon=True)
        self._t.start()

    def _run(self):
        try:
            import psutil
        except Exception:
            return
        me = psutil.Process()
        while not self._stop.wait(0.5):
            try:
                rss = me.memory_info().rss + sum(c.memory_info().rss for c in me.children(recursive=True))
            except Exception:
                continue
            self.peak = max(self.peak, rss / 1e9)
            if rss / 1e9 > self.limit:
                self.tripped = True
                for p in list(getattr(self.ex, "_processes", {}).values()):
                    try:
                        p.terminate()
                    except Exception:
                        pass
                return

    def stop(self):
        self._stop.set()


def _gen_group_two(args):
    """Worker: one group of contigs -> per read type (BAM heads bytes, offsets, aligned, k1 bytes, name bytes, per-contig
    aligned, PAF text, PAF aligned, reads)."""
    contigs, idx, g, cov, config, want_paf = args[:6]
    j, k = args[6:8] if len(args) > 6 else (0, 1)              # part j of k of the group's reads (coverage / k each: bounded memory)
    sub = tuple(contigs[i] for i in idx)
    tag = "g%02d" % g if k == 1 else "g%02dp%02d" % (g, j)
    out = []
    for t, (kind, c) in enumerate((("hifi", cov[0]), ("ont", cov[1]))):
        rs = synth.simulate_reads(sub, c / k, kind, seed=synth.seed_for(config, 2 * t) + 7 * g + 1000003 * j,
                                  long_cigar_frac=0.0005 if kind == "ont" else 0.0,
                                  name_prefix=("m64011_%s/" % tag) if kind == "hifi" else None)
        if kind == "ont":                                       # names unique across groups (the generator numbers from 0)
            rs.names = np.char.add(("%s-" % tag).encode(), rs.names).astype("S")
        rs.ref_id = (rs.ref_id + idx[0]).astype(np.int32)
        rs.contigs = tuple(contigs)
        mapped = (rs.flag & 4) == 0
        per_contig = np.bincount(rs.ref_id[mapped], weights=np.maximum(rs.ref_span(), 1)[mapped].astype(np.float64),
                                 minlength=len(contigs)).astype(np.int64)
        s, o = synth.to_bam_stream(rs, heads=True)
        first = bamfmt.parse_header(s).first_record
        paf, paf_al = None, 0
        if want_paf:
            other = synth.perturb(rs, synth.seed_for(config, 2 * t + 1) + 7 * g + 1000003 * j)
            other.contigs = tuple(contigs)
            paf = synth.to_paf_text(other, synth.seed_for(config, 10 + t) + g + 1000003 * j)
            keep = ((other.flag & 0x4) == 0) & ((other.flag & 0x100) == 0)
            paf_al = int(other.ref_span()[keep].sum())
        out.append((s[first:].copy(), (o - np.uint64(first)).astype(np.uint64), int(per_contig.sum()), _k1_algorithmic_bytes(rs),
                    int(np.char.str_len(rs.names).sum()), per_contig, paf, paf_al, len(rs)))
    return (g, j), out


def genome_two_type(config: int = 4, scale: float = 1.0, cov_hifi: float = 40.0, cov_ont: float = 40.0, procs: Optional[int] = None,
                    verbose: bool = False) -> TwoTypeInput:
    """config 4 = BASELINE configs[3]: CHM13, `--hifi` + `--nano`, per read type one BAM and one PAF (4 inputs).
    config 5 = BASELINE configs[4]: diploid mat + pat (46 contigs), per read type one BAM, 20 N gaps, `-R` regions."""
    ctg = (tuple((n, max(20_000, int(l * scale))) for n, l in synth.CHM13) if scale != 1.0 else tuple(synth.CHM13)) if config == 4 \
        else diploid_contigs(scale)
    total = sum(l for _, l in ctg)
    groups = contig_groups(ctg, max(2.0e7, min(2.6e8, total / 12.0)))
    # A worker holds ~0.35 GB per Mb of reference while it makes ONT CIGARs at 40x (chr1 whole: ~90 GB; the GPU boxes run in a
    # 300 GiB memory cgroup): a group's reads are made in k parts of coverage / k each, at most TASK_GB per task, and as many
    # tasks at a time as half the memory this process may still take carries.
    TASK_GB = 6.0
    per_mb = 0.35e-3 * max(cov_ont, 1.0) / 40.0 + 0.04e-3 * max(cov_hifi, 1.0) / 40.0
    split = [max(1, int(np.ceil(sum(ctg[i][1] for i in g) * 1e-6 * per_mb * 1e3 / TASK_GB))) for g in groups]
    budget = _memory_budget_gb()
    if procs is None:
        from . import hostio
        procs = max(1, min(hostio.default_threads(), 16, int(budget / 2.0 / (TASK_GB + 1.0))))
    tasks = [(ctg, idx, g, (cov_hifi, cov_ont), config, config == 4, j, split[g]) for g, idx in enumerate(groups) for j in range(split[g])]
    procs = max(1, min(int(procs), len(tasks)))
    t0 = time.time()
    results = {}
    if procs == 1:
        for t in tasks:
            key, out = _gen_group_two(t)
            results[key] = out
    else:
        import multiprocessing as mp
        from concurrent.futures import ProcessPoolExecutor
        with ProcessPoolExecutor(procs, mp_context=mp.get_context("fork")) as ex:
            guard = _MemoryGuard(ex, 0.8 * budget)
            try:
                for key, out in ex.map(_gen_group_two, sorted(tasks, key=lambda t: -sum(ctg[i][1] for i in t[1]) / t[7])):
                    results[key] = out
                    if verbose:
                        print("workload: part %d/%d done, %.0f s" % (len(results), len(tasks), time.time() - t0), file=sys.stderr, flush=True)
            except Exception:
                if guard.tripped:
                    raise MemoryError("workload generation went past %.0f GB (peak seen %.0f GB): stopped before the box did" % (
                        0.8 * budget, guard.peak)) from None
                raise
            finally:
                guard.stop()
    # parts of a group side by side, groups in header order
    order = [(g, j) for g in range(len(groups)) for j in range(split[g])]
    hdr = np.frombuffer(bamfmt.encode_header([n for n, _ in ctg], [l for _, l in ctg]), dtype=np.uint8)
    types = []
    for t in range(2):
        parts, offs, pafs, size = [hdr], [], [], int(hdr.shape[0])
        aligned = k1 = nb = paf_al = reads = 0
        per_contig = np.zeros(len(ctg), dtype=np.int64)
        for g in order:
            s, o, a, kb, n, pc, paf, pal, nr = results[g][t]
            parts.append(s)
            offs.append(o + np.uint64(size))
            size += int(s.shape[0])
            aligned, k1, nb, paf_al, reads = aligned + a, k1 + kb, nb + n, paf_al + pal, reads + nr
            per_contig += pc
            if paf is not None:
                pafs.append(paf)
            results[g][t] = None
        bam = AlignmentFile(np.concatenate(parts), np.concatenate(offs) if offs else np.zeros(0, np.uint64), aligned, k1, nb, per_contig)
        types.append(ReadTypeInput(bam, np.concatenate(pafs) if pafs else None, paf_al, reads))
    gaps, regions = {}, []
    if config == 5:
        rng = np.random.Generator(np.random.PCG64(synth.seed_for(5, 77)))
        for c in rng.choice(len(ctg), size=min(20, len(ctg)), replace=False):
            n, l = ctg[int(c)]
            a = int(rng.integers(l // 10, l // 2))
            gaps.setdefault(n, []).append((a, a + int(rng.integers(1, 50_000))))
        for c in sorted(rng.choice(len(ctg), size=min(12, len(ctg)), replace=False).tolist()):
            n, l = ctg[c]
            a = int(rng.integers(0, l // 2))
            regions.append((n, a, min(l, a + int(rng.integers(10_000, max(20_000, l // 3))))))
    if verbose:
        print("workload: %d contigs, %d bp, reads hifi %d ont %d, %.0f s on %d processes" % (
            len(ctg), total, types[0].n_reads, types[1].n_reads, time.time() - t0, procs), file=sys.stderr, flush=True)
    return TwoTypeInput(ctg, types[0], types[1], gaps, regions)


# ---- a heads stream back as the BGZF file it came from (bench.py: the command line at genome size, SURVEY.md 8d number 3) ----

_BG = {}          # what the forked workers of write_bgzf_from_heads read: heads stream, offsets, shared member sizes


def _bgzf_member(payload, level: int = 1) -> bytes:
    import struct
    import zlib
    comp = zlib.compressobj(level, zlib.DEFLATED, -15)
    body = comp.compress(payload) + comp.flush()
    if 12 + 6 + len(body) + 8 - 1 > 0xFFFF:                  # would not fit BSIZE: stored
        comp = zlib.compressobj(0, zlib.DEFLATED, -15)
        body = comp.compress(payload) + comp.flush()
    head = struct.pack("<BBBBIBBHBBHH", 0x1F, 0x8B, 8, 4, 0, 0, 0xFF, 6, 66, 67, 2, 12 + 6 + len(body) + 8 - 1)
    return head + body + struct.pack("<II", zlib.crc32(payload) & 0xFFFFFFFF, len(payload))


def _bgzf_part(k: int):
    """Worker: records [lo, hi) of the heads stream -> the records with their SEQ / QUAL bytes (random bases, HiFi-like
    qualities: synth.to_bam_stream(seq_qual='random')) -> BGZF members cut the way htslib's writer cuts them (a record that
    still fits the 0xFF00-byte block goes into it whole, bgzf_flush_try; only a record larger than a block is split) ->
    written at this part's place in the file, which is known once every earlier part's size is (shared array)."""
    import os
    g = _BG
    heads, offs, n_stream = g["heads"], g["offs"], g["n_stream"]
    lo, hi = g["parts"][k]
    qlut = synth._hifi_qual_lut()
    rng = np.random.Generator(np.random.PCG64(g["seed"] + 7919 * k))
    a = offs[lo:hi].astype(np.int64)
    e = np.concatenate([a[1:], [int(offs[hi]) if hi < offs.shape[0] else n_stream]]).astype(np.int64)
    size = e - a                                                # bytes of the record in the heads stream (block_size word included)
    lrn = heads[a + 12].astype(np.int64)
    ncig = heads[a + 16].astype(np.int64) | (heads[a + 17].astype(np.int64) << 8)
    l_seq = (heads[a + 20].astype(np.int64) | (heads[a + 21].astype(np.int64) << 8) | (heads[a + 22].astype(np.int64) << 16)
             | (heads[a + 23].astype(np.int64) << 24))
    head_len = 36 + lrn + 4 * ncig                              # everything in front of SEQ
    aux_len = size - head_len
    nseq = (l_seq + 1) // 2
    out_size = size + nseq + l_seq
    out_off = np.zeros(hi - lo + 1, dtype=np.int64)
    np.cumsum(out_size, out=out_off[1:])
    total = int(out_off[-1])
    out = qlut[rng.integers(0, 256, total, dtype=np.uint8)]     # quality-like bytes everywhere; heads, SEQ and aux are laid over them
    seq_all = synth._SEQ_LUT[rng.integers(0, 16, int(nseq.sum()), dtype=np.uint8)]
    sp = 0
    for r in range(hi - lo):
        o, s, h, q = int(out_off[r]), int(a[r]), int(head_len[r]), int(nseq[r])
        out[o:o + h] = heads[s:s + h]
        out[o + h:o + h + q] = seq_all[sp:sp + q]
        sp += q
        x = int(aux_len[r])
        t = o + h + q + int(l_seq[r])
        out[t:t + x] = heads[s + h:s + h + x]
    out[(out_off[:-1, None] + np.arange(4)[None, :]).ravel()] = (out_size - 4).astype("<i4").view(np.uint8).reshape(-1, 4).ravel()
    # member cuts
    BLOCK = 0xFF00
    cuts, fill = [0], 0
    for r in range(hi - lo):
        sz = int(out_size[r])
        if fill and fill + sz > BLOCK:
            cuts.append(int(out_off[r]))
            fill = 0
        while sz > BLOCK - fill:                                # a record larger than what is left of an empty block: split
            take = BLOCK - fill
            cuts.append(cuts[-1] + take if fill == 0 and cuts[-1] >= int(out_off[r]) else int(out_off[r]) + take)
            sz -= take
            fill = 0
        fill += sz
    if cuts[-1] != total:
        cuts.append(total)
    mv = memoryview(out)
    pieces = []
    if k == 0:                                                  # the BAM header: members of its own in front
        hdr = bytes(heads[:g["first"]])
        pieces += [_bgzf_member(hdr[i:i + BLOCK]) for i in range(0, len(hdr), BLOCK)]
    pieces += [_bgzf_member(mv[c0:c1]) for c0, c1 in zip(cuts[:-1], cuts[1:]) if c1 > c0]
    data = b"".join(pieces)
    sizes = g["sizes"]
    sizes[k] = len(data)
    import time as _t
    while True:                                                 # every earlier part is running or done (the pool hands parts out in order)
        prev = sizes[:k]
        if all(v >= 0 for v in prev):
            break
        _t.sleep(0.005)
    fd = os.open(g["path"], os.O_WRONLY)
    try:
        os.pwrite(fd, data, int(sum(prev)))
    finally:
        os.close(fd)
    return k, len(data), len(pieces), total


def write_bgzf_from_heads(path: str, heads: np.ndarray, offsets: np.ndarray, seed: int = 7, procs: Optional[int] = None,
                          part_records: int = 24576, verbose: bool = False) -> dict:
    """The BGZF BAM file whose heads stream (gci_bam_heads) is `heads` / `offsets`: every record gets SEQ / QUAL bytes of realistic
    entropy back (they are what costs inflate time; nothing on the path reads them), the stream is deflated at level 1 by
    `procs` forked workers, part by part of `part_records` records, each part written at its place in the file as soon as the
    sizes of all earlier parts are known.  -> {"bytes", "inflated_bytes", "members", "seconds"}."""
    import multiprocessing as mp
    from concurrent.futures import ProcessPoolExecutor
    from .formats import bgzf
    R = int(offsets.shape[0])
    first = bamfmt.parse_header(heads).first_record
    parts = [(lo, min(R, lo + part_records)) for lo in range(0, R, part_records)] or [(0, 0)]
    if procs is None:
        from . import hostio
        procs = hostio.default_threads()
    procs = max(1, min(int(procs), len(parts)))
    sizes = mp.get_context("fork").Array("q", [-1] * len(parts), lock=False)
    _BG.clear()
    _BG.update(heads=heads, offs=offsets, n_stream=int(heads.shape[0]), parts=parts, seed=int(seed), first=int(first), sizes=sizes,
               path=path)
    with open(path, "wb"):
        pass
    t0 = time.time()
    done, members, inflated, nbytes = 0, 0, int(first), 0
    try:
        if procs == 1:
            res = map(_bgzf_part, range(len(parts)))
        else:
            ex = ProcessPoolExecutor(procs, mp_context=mp.get_context("fork"))
            res = ex.map(_bgzf_part, range(len(parts)))
        for k, n, m, t in res:
            done += 1
            members += m
            inflated += t
            nbytes += n
            if verbose and done % 32 == 0:
                print("bgzf: part %d/%d of %s, %.0f s" % (done, len(parts), path, time.time() - t0), file=sys.stderr, flush=True)
        if procs != 1:
            ex.shutdown()
    finally:
        _BG.clear()
    with open(path, "r+b") as f:
        f.seek(nbytes)
        f.write(bgzf.BGZF_EOF)
    return {"bytes": nbytes + len(bgzf.BGZF_EOF), "inflated_bytes": inflated, "members": members, "seconds": time.time() - t0}


def write_paf_at_size(path: str, paf_text: np.ndarray, target_bytes: int, chunk_lines: int = 4096) -> int:
    """The PAF text `paf_text` (its lines end in '\\n') written with a `cg:Z:` tag behind every line so that the file is about
    `target_bytes` long -- the shape of the PAFs the reference's published CHM13 run read (3.6 GB for HiFi, 48 GB for ONT: minimap2 -c
    writes every alignment's CIGAR as text, tens of KB per ONT line).  The twelve mandatory columns, the only ones GCI.py:218-229 reads,
    are untouched; the tag's text is a repeating CIGAR-like pattern (its content is never looked at).  Written chunk by chunk: the
    file never exists in this process's memory.  -> bytes written."""
    text = np.ascontiguousarray(paf_text, dtype=np.uint8)
    ends = np.flatnonzero(text == 10) + 1
    n = int(ends.shape[0])
    if n == 0:
        text.tofile(path)
        return int(text.shape[0])
    starts = np.concatenate([[0], ends[:-1]])
    extra = max(0, (int(target_bytes) - int(text.shape[0])) // n - 6)
    pattern = np.frombuffer((b"1234=1X56=2I789=1D" * (extra // 18 + 2))[:max(extra, 0)], dtype=np.uint8)
    tag = np.concatenate([np.frombuffer(b"\tcg:Z:", dtype=np.uint8), pattern, np.frombuffer(b"\n", dtype=np.uint8)]) if extra > 0 else None
    written = 0
    with open(path, "wb") as f:
        if tag is None:
            f.write(text.tobytes())
            return int(text.shape[0])
        ends_l, starts_l = ends.tolist(), starts.tolist()
        for a in range(0, n, chunk_lines):
            b = min(n, a + chunk_lines)
            parts = []
            for i in range(a, b):                                    # the line without its newline, then the tag (which brings one)
                parts.append(text[starts_l[i]:ends_l[i] - 1])
                parts.append(tag)
            buf = np.concatenate(parts)
            f.write(buf.data)
            written += int(buf.shape[0])
    return written
