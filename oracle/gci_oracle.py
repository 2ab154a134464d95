"""oracle/gci_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement of the reference's alignment-filter -> per-base-depth -> issue-scan path
(/root/reference/GCI.py), row by row of SURVEY.md section 8(a).  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; nothing under
gci_amd/ does.

* Per-record / per-base loops live in oracle/gci_oracle.c (plain C, single thread) and are
  reached through ctypes; small pure-Python twins (`*_py`) restate the same lines once more
  so the C can be checked against them on tiny inputs.
* Dict / set logic (PAF grouping, the cross-file join, interval algebra, the score) is plain
  Python, statement for statement what the reference does, with its dict-order and
  float-order behaviour.

Pinning (see tests/test_oracle_*.py): R6-R13 reproduce the reference's own
example/MH63.{depth.gz,0.depth.bed,gci} byte for byte and match golden vectors produced by the
unmodified reference in the build container (tools/make_golden.py).  R1 is PARITY UNPINNED at
the pysam/htslib boundary: htslib is not part of /root/reference (README.md:33 asks for a
"stable version"), so its decode rules are restated from the SAM/BAM spec and htslib's
documented behaviour and anchored on hand-assembled records.
"""
from __future__ import annotations

import ctypes
import os
import re
from math import log2
from typing import Dict, Iterable, List, Optional, Sequence, Set, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libgci_oracle.so")
_lib = None

E_NO_NM, E_ZERO_DIV, E_BAD_NM_TYPE, E_NO_END, E_MALFORMED = -3, -4, -5, -6, -7


class OracleRecordError(Exception):
    """The reference would have raised while processing this record."""

    def __init__(self, status: int, rec: int):
        self.status, self.rec = status, rec
        kind = {E_NO_NM: "KeyError (no NM tag, GCI.py:163)", E_ZERO_DIV: "ZeroDivisionError (GCI.py:165)",
                E_BAD_NM_TYPE: "non-integer NM", E_NO_END: "reference_end is None", E_MALFORMED: "malformed record"}
        super().__init__("%s at record %d" % (kind.get(status, status), rec))


def build(force: bool = False) -> str:
    """gcc -O2 the C restatement next to this file (no -ffast-math: f64 divisions must be IEEE)."""
    import subprocess
    src = os.path.join(_HERE, "gci_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.run(["gcc", "-O2", "-std=c11", "-fPIC", "-shared", "-Wall", "-o", _LIB_PATH, src], check=True)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        vp, u64, u32, i64, i32, dbl = (ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int64,
                                       ctypes.c_int32, ctypes.c_double)
        L.orc_bam_filter.argtypes = [vp, u64, vp, u32, vp, i32, ctypes.c_int, ctypes.c_int, dbl, dbl,
                                     vp, vp, vp, vp, vp, vp, vp, vp, vp, ctypes.c_int]
        L.orc_bam_filter.restype = ctypes.c_int
        L.orc_depth_build.argtypes = [vp, i64, vp, vp, u64, i64]
        L.orc_depth_build.restype = None
        L.orc_zero_range.argtypes = [vp, i64, i64, i64]
        L.orc_zero_range.restype = None
        L.orc_max2.argtypes = [vp, vp, i64, vp]
        L.orc_max2.restype = None
        L.orc_collapse.argtypes = [vp, i64, dbl, dbl, i64, i64, vp, u64]
        L.orc_collapse.restype = u64
        L.orc_depth_text.argtypes = [vp, i64, vp]
        L.orc_depth_text.restype = u64
        L.orc_sum.argtypes = [vp, i64]
        L.orc_sum.restype = i64
        L.orc_parse_depth_lines.argtypes = [vp, u64, vp, i64]
        L.orc_parse_depth_lines.restype = i64
        _lib = L
    return _lib


def _p(a: np.ndarray):
    return a.ctypes.data_as(ctypes.c_void_p)


Segment = Tuple[str, int, int, int]          # (contig, start, end, query_length)


# ==============================================================================================
# R1 + R4: read_sam over every chunk of every selected contig (GCI.py:146-169, 257-270)
# ==============================================================================================

def bam_filter_arrays(stream: np.ndarray, rec_off: np.ndarray, ref_sel: np.ndarray, map_qual: int,
                      mq_cutoff: int, clip_percent: float, iden_percent: float, heads: bool = False) -> Dict[str, np.ndarray]:
    """Per-record decision arrays from the C restatement (see orc_bam_filter).  heads: `stream` is a heads stream
    (the records without their SEQ / QUAL bytes -- what a genome-scale test can hold in memory)."""
    R = int(rec_off.shape[0])
    stream = np.ascontiguousarray(stream, dtype=np.uint8)
    rec_off = np.ascontiguousarray(rec_off, dtype=np.uint64)
    ref_sel = np.ascontiguousarray(ref_sel, dtype=np.int32)
    out = dict(passed=np.zeros(R, np.uint8), hq=np.zeros(R, np.uint8), contig=np.zeros(R, np.int32),
               start=np.zeros(R, np.int32), end=np.zeros(R, np.int32), qlen=np.zeros(R, np.int32),
               name_off=np.zeros(R, np.uint64), name_len=np.zeros(R, np.uint32))
    bad = ctypes.c_uint32(0)
    st = lib().orc_bam_filter(_p(stream), stream.shape[0], _p(rec_off), R, _p(ref_sel), ref_sel.shape[0],
                              int(map_qual), int(mq_cutoff), float(clip_percent), float(iden_percent),
                              _p(out["passed"]), _p(out["hq"]), _p(out["contig"]), _p(out["start"]), _p(out["end"]),
                              _p(out["qlen"]), _p(out["name_off"]), _p(out["name_len"]), ctypes.byref(bad),
                              0 if heads else 1)
    if st != 0:
        raise OracleRecordError(st, int(bad.value))
    return out


def read_names(stream: np.ndarray, name_off: np.ndarray, name_len: np.ndarray) -> List[str]:
    mv = memoryview(np.ascontiguousarray(stream))
    return [bytes(mv[int(o):int(o) + int(n)]).decode(errors="replace") for o, n in zip(name_off, name_len)]


def bam_file_dict(stream: np.ndarray, rec_off: np.ndarray, references: Sequence[str], targets: Sequence[str],
                  map_qual: int, mq_cutoff: int, clip_percent: float, iden_percent: float, heads: bool = False
                  ) -> Tuple[Dict[str, Segment], Set[str]]:
    """samfile_dicts[i] and the high-quality names one BAM contributes (GCI.py:257-270).

    Tasks run contig by contig in `targets` order and records arrive in file order inside a
    contig, so for a repeated query name the later (contig order, file order) record wins
    (dict.update at GCI.py:269, with -t 1 chunking)."""
    tindex = {t: i for i, t in enumerate(targets)}
    ref_sel = np.array([tindex.get(r, -1) for r in references], dtype=np.int32)
    a = bam_filter_arrays(stream, rec_off, ref_sel, map_qual, mq_cutoff, clip_percent, iden_percent, heads)
    return _dict_from_arrays(stream, a, targets, None)


def _dict_from_arrays(stream: np.ndarray, a: Dict[str, np.ndarray], targets: Sequence[str],
                      keep: Optional[Set[str]]) -> Tuple[Dict[str, Segment], Set[str]]:
    """The dict / set of GCI.py:166-168, 268-270 from the per-record decisions; keep != None: only those names."""
    idx = np.flatnonzero(a["passed"])
    order = idx[np.argsort(a["contig"][idx], kind="stable")]
    names = read_names(stream, a["name_off"][order], a["name_len"][order])
    contig, start, end, qlen, is_hq = (a[k][order].tolist() for k in ("contig", "start", "end", "qlen", "hq"))
    d: Dict[str, Segment] = {}
    hq: Set[str] = set()
    for k, q in enumerate(names):
        if keep is not None and q not in keep:
            continue
        d[q] = (targets[contig[k]], start[k], end[k], qlen[k])
        if is_hq[k]:
            hq.add(q)
    return d, hq


def file1_on_contigs(bams, targets: Sequence[str], chosen: Sequence[str], map_qual: int, mq_cutoff: int,
                     clip_percent: float, iden_percent: float, ovlp_percent: float, heads: bool = False) -> Dict[str, tuple]:
    """filter()'s `file1` (GCI.py:272-301) restricted to the entries that lie on the contigs `chosen`, for inputs too
    large to push every name through Python dicts: `bams` = [(stream, rec_off, references)].

    Exact, because the join treats every query name independently: an entry of file1 on contig X can only come from
    a name that has a passing record on X in some file, so it is enough to run the dict logic over ALL records (of all
    files, on any contig) of the names seen on the chosen contigs."""
    tindex = {t: i for i, t in enumerate(targets)}
    want = np.array([tindex[c] for c in chosen], dtype=np.int32)
    arrays = []
    keep: Set[str] = set()
    for stream, off, refs in bams:
        ref_sel = np.array([tindex.get(r, -1) for r in refs], dtype=np.int32)
        a = bam_filter_arrays(stream, off, ref_sel, map_qual, mq_cutoff, clip_percent, iden_percent, heads)
        arrays.append(a)
        on = np.flatnonzero((a["passed"] != 0) & np.isin(a["contig"], want))
        keep.update(read_names(stream, a["name_off"][on], a["name_len"][on]))
    dicts, hq = [], set()
    for (stream, off, refs), a in zip(bams, arrays):
        d, h = _dict_from_arrays(stream, a, targets, keep)
        dicts.append(d)
        hq |= h
    cs = set(chosen)
    return {q: seg for q, seg in name_join(dicts, hq, ovlp_percent).items() if seg[0] in cs}


def bam_filter_record_py(rec, references: Sequence[str], targets: Sequence[str], map_qual: int, mq_cutoff: int,
                         clip_percent: float, iden_percent: float):
    """Pure-Python twin of one iteration of read_sam's loop (GCI.py:152-168) on a decoded record
    (gci_amd.formats.bam.RecordView with the CG-restored CIGAR).  Returns None (filtered) or
    (name, segment, is_high_qual).  Raises what the reference raises."""
    if rec.ref_id < 0 or references[rec.ref_id] not in targets:
        return None
    if rec.flag & 0x4 or rec.flag & 0x100 or rec.flag & 0x800 or rec.mapq < map_qual:
        return None
    tot = [0] * 16
    for op, ln in rec.cigar:
        tot[op] += ln
    M, I, D, S, eq, X = tot[0], tot[1], tot[2], tot[4], tot[7], tot[8]
    if "NM" not in rec.aux:
        raise KeyError("NM")
    NM = rec.aux["NM"][1]
    mm = NM - (I + D)
    if (S / (M + eq + X + I + S) <= clip_percent) and ((M + eq + X - mm) / (M + eq + X + I + D) >= iden_percent):
        rlen = tot[0] + tot[2] + tot[3] + tot[7] + tot[8]
        end = rec.pos + (rlen if rlen > 0 else 1)
        return rec.name, (references[rec.ref_id], rec.pos, end, rec.l_seq), rec.mapq >= mq_cutoff
    return None


# ==============================================================================================
# R3: PAF path (GCI.py:211-254) with merge_alns_properties (64-96) and get_average_identity (49-61)
# ==============================================================================================

def _merge_blocks(alns: Sequence[tuple], x: int, y: int) -> Tuple[int, int, int]:
    """GCI.py:64-96: sort [a[x], a[y]] pairs, merge while `high_est >= low` (touching merges),
    return (total merged length, start and end of the LONGEST merged block; ties -> lowest start
    because sorted(..., reverse=True) is stable on the already ascending list)."""
    blocks = sorted([a[x], a[y]] for a in alns)
    merged: List[Tuple[int, int, int]] = []
    total = 0
    lo, hi = blocks[0]
    for b_lo, b_hi in blocks:
        if hi >= b_lo:
            if hi < b_hi:
                hi = b_hi
        else:
            merged.append((hi - lo, lo, hi))
            total += hi - lo
            lo, hi = b_lo, b_hi
    merged.append((hi - lo, lo, hi))
    total += hi - lo
    best = sorted(merged, key=lambda t: t[0], reverse=True)[0]
    return total, best[1], best[2]


def paf_filter(paf_paths: Sequence[str], targets: Sequence[str], map_qual: int, mq_cutoff: int,
               iden_percent: float) -> Tuple[List[Dict[str, Segment]], Set[str]]:
    """paf_lines (one dict per file) and the high-quality names (GCI.py:211-254).  `synteny` is
    created once, outside the per-file loop (GCI.py:214-215), so file i re-emits every query and
    block of files < i -- kept."""
    tset = set(targets)
    hq: Set[str] = set()
    out: List[Dict[str, Segment]] = [{} for _ in paf_paths]
    synteny: Dict[str, Dict[str, list]] = {}
    for i, path in enumerate(paf_paths):
        with open(path, "r") as f:
            for line in f:
                c = line.strip().split("\t")
                target = c[5]
                if target not in tset:
                    continue
                query, qlen, qs, qe = c[0], int(c[1]), int(c[2]), int(c[3])
                ts, te, nmatch, alnlen, mapq = int(c[7]), int(c[8]), int(c[9]), int(c[10]), int(c[11])
                identity = nmatch / alnlen
                if mapq >= map_qual and identity >= iden_percent:
                    synteny.setdefault(query, {}).setdefault(target, []).append((qlen, qs, qe, ts, te, identity))
                    if mapq >= mq_cutoff:
                        hq.add(query)
        for query, per_target in synteny.items():
            results = {}
            for target, alns in per_target.items():
                aligned, _, _ = _merge_blocks(alns, 1, 2)
                qlen = alns[0][0]
                alignrate = aligned / qlen
                ids = [a[-1] for a in alns]
                avg = sum(ids) / len(alns)                       # sequential f64 sum in file order
                score = avg * alignrate
                _, start, end = _merge_blocks(alns, 3, 4)
                results[target] = (score, start, end, qlen)
            best = sorted(results, key=lambda k: (results[k][0], k), reverse=True)[0]
            r = results[best]
            out[i][query] = (best, r[1], r[2], r[3])
    return out, hq


# ==============================================================================================
# R5: cross-file join (GCI.py:272-301)
# ==============================================================================================

def name_join(files: Sequence[Dict[str, Segment]], high_qual: Set[str], ovlp_percent: float) -> Dict[str, tuple]:
    """`files` = PAF dicts then BAM dicts, each in command-line order (GCI.py:272).  Returns
    file1: name -> (contig, start, end[, qlen])."""
    if len(files) <= 1:
        return dict(files[0])
    comm = set.intersection(*[set(f.keys()) for f in files])
    final = high_qual | comm
    file1: Dict[str, tuple] = {q: seg for q, seg in files[0].items() if q in final}
    for f in files[1:]:
        for q, seg in f.items():
            if q in file1:
                seg1 = file1[q]
                if seg[0] == seg1[0]:
                    s1, e1, s2, e2 = seg[1], seg[2], seg1[1], seg1[2]
                    ovlp = min(e1, e2) - max(s1, s2)
                    if ovlp / seg[-1] < ovlp_percent:      # qlen of the CURRENT file's record
                        del file1[q]
                    else:
                        file1[q] = (seg1[0], max(s1, s2), min(e1, e2))
                else:
                    del file1[q]
            elif q in high_qual:
                file1[q] = (seg[0], seg[1], seg[2])       # can resurrect a deleted query
    return file1


# ==============================================================================================
# R2 + R6: allocation and depth accumulation (GCI.py:201-208, 302-306)
# ==============================================================================================

def depth_build(file1: Dict[str, tuple], targets_length: Dict[str, int], flank_len: int) -> Dict[str, np.ndarray]:
    depths = {t: np.zeros(L, dtype=np.int64) for t, L in targets_length.items()}
    by_t: Dict[str, Tuple[list, list]] = {t: ([], []) for t in targets_length}
    for seg in file1.values():
        by_t[seg[0]][0].append(seg[1])
        by_t[seg[0]][1].append(seg[2])
    for t, (ss, ee) in by_t.items():
        s = np.asarray(ss, dtype=np.int64)
        e = np.asarray(ee, dtype=np.int64)
        lib().orc_depth_build(_p(depths[t]), depths[t].shape[0], _p(s), _p(e), s.shape[0], int(flank_len))
    return depths


def depth_build_py(intervals: Iterable[tuple], targets_length: Dict[str, int], flank_len: int) -> Dict[str, np.ndarray]:
    """Literal numpy twin (GCI.py:302-306), including Python's negative-stop wrap."""
    depths = {t: np.zeros(L, dtype=np.int64) for t, L in targets_length.items()}
    for seg in intervals:
        start = seg[1] + flank_len
        end = seg[2] - flank_len
        depths[seg[0]][start:end + 1] += 1
    return depths


# ==============================================================================================
# R7: depth text (GCI.py:99-143) -- the decompressed stream only
# ==============================================================================================

def depth_text_contig(depth: np.ndarray) -> bytes:
    d = np.ascontiguousarray(depth, dtype=np.int64)
    buf = np.empty(d.shape[0] * 21 + 1, dtype=np.uint8)
    n = lib().orc_depth_text(_p(d), d.shape[0], _p(buf))
    return buf[:n].tobytes()


def depth_text(depths: Dict[str, np.ndarray]) -> bytes:
    parts = []
    for t, d in depths.items():
        parts.append((">%s\n" % t).encode())
        parts.append(depth_text_contig(d))
    return b"".join(parts)


def depth_text_py(depths: Dict[str, np.ndarray]) -> bytes:
    out = []
    for t, d in depths.items():
        out.append(f">{t}\n")
        out.extend(f"{int(v)}\n" for v in d)
    return "".join(out).encode()


def parse_depth_text(text: bytes) -> Dict[str, np.ndarray]:
    """Test helper: inverse of depth_text (same grammar utility/GCI_score.py:25-37 reads)."""
    data = np.frombuffer(text, dtype=np.uint8)
    out: Dict[str, np.ndarray] = {}
    hdr = np.flatnonzero(data == ord(">"))
    for k, h in enumerate(hdr.tolist()):
        e = h + int(np.argmax(data[h:] == 10))
        name = bytes(data[h + 1:e]).decode()
        stop = int(hdr[k + 1]) if k + 1 < hdr.shape[0] else data.shape[0]
        body = np.ascontiguousarray(data[e + 1:stop])
        cap = int(np.count_nonzero(body == 10)) + 1
        vals = np.empty(cap, dtype=np.int64)
        n = int(lib().orc_parse_depth_lines(_p(body), body.shape[0], _p(vals), cap))
        if n < 0:
            raise ValueError("malformed depth text in contig %s" % name)
        out[name] = vals[:n].copy()
    return out


# ==============================================================================================
# R8: gap mask (GCI.py:18-46, 315-329)
# ==============================================================================================

def n_runs_of(seq: str) -> List[Tuple[int, int]]:
    return [(m.start(), m.end()) for m in re.finditer(r"(?i)N+", seq)]


def merge_gaps_depths(depths: Dict[str, np.ndarray], ns_bed: Optional[Dict[str, List[Tuple[int, int]]]]):
    if ns_bed is not None:
        for t, segs in ns_bed.items():
            if t in depths:
                for a, b in segs:
                    lib().orc_zero_range(_p(depths[t]), depths[t].shape[0], int(a), int(b))
    return depths


# ==============================================================================================
# R9: two-type merge (GCI.py:350)
# ==============================================================================================

def max2(hifi: Dict[str, np.ndarray], nano: Dict[str, np.ndarray]) -> Dict[str, np.ndarray]:
    out = {}
    for t, h in hifi.items():                       # iterates the HiFi dict's order
        n = nano[t]
        m = min(h.shape[0], n.shape[0])             # zip() stops at the shorter
        o = np.empty(m, dtype=np.int64)
        lib().orc_max2(_p(np.ascontiguousarray(h[:m])), _p(np.ascontiguousarray(n[:m])), m, _p(o))
        out[t] = o
    return out


# ==============================================================================================
# R10: issue scan (GCI.py:356-419)
# ==============================================================================================

def collapse_contig(depth: np.ndarray, leftmost: float, rightmost: float, flank_len: int, start_pos: int
                    ) -> List[Tuple[int, int]]:
    d = np.ascontiguousarray(depth, dtype=np.int64)
    cap = 1024
    while True:
        pairs = np.empty(2 * cap, dtype=np.int64)
        n = int(lib().orc_collapse(_p(d), d.shape[0], float(leftmost), float(rightmost), int(flank_len),
                                   int(start_pos), _p(pairs), cap))
        if n <= cap:
            return [(int(pairs[2 * k]), int(pairs[2 * k + 1])) for k in range(n)]
        cap = n


def collapse_depth_range(depths: Dict[str, np.ndarray], leftmost=-1, rightmost=0, flank_len=15, start_pos=0
                         ) -> Dict[str, List[Tuple[int, int]]]:
    return {t: collapse_contig(d, leftmost, rightmost, flank_len, start_pos) for t, d in depths.items()}


def collapse_contig_py(depth_list, leftmost, rightmost, flank_len, start_pos) -> List[Tuple[int, int]]:
    """Pure-Python twin of GCI.py:371-389 for tiny inputs."""
    out = []
    opened, closed = False, True
    n = len(depth_list)
    start = 0
    for i, d in enumerate(depth_list[flank_len:n - flank_len]):
        if leftmost < d <= rightmost:
            if not opened:
                start, opened, closed = i + flank_len, True, False
            if i == n - 2 * flank_len - 1:
                out.append((start + start_pos, i + flank_len + 1 + start_pos))
        elif not closed:
            if i > flank_len:
                out.append((start + start_pos, i + flank_len + start_pos))
            closed, opened = True, False
    return out


def bed_text(merged: Dict[str, List[Tuple[int, int]]]) -> str:
    return "".join(f"{t}\t{s}\t{e}\n" for t, segs in merged.items() for s, e in segs)


# ==============================================================================================
# R11-R13: interval algebra and the score (GCI.py:422-657)
# ==============================================================================================

def complement_merged_depth(merged, targets_length, flank_len=15, start=None, end=None):
    explicit = start is not None and end is not None
    out = {}
    for t, L in targets_length.items():
        if not explicit:
            start, end = flank_len, L - flank_len
        lens: List[int] = []
        last = start
        segs = merged[t]
        n = len(segs)
        if n == 0:
            lens.append(end - start)
        for i, (s, e) in enumerate(segs):
            if s > last:
                lens.append(s - last)
            if i != n - 1:
                last = e
            elif end > e:
                lens.append(end - e)
        out[t] = lens
    return out


def compute_n50(lengths) -> int:
    lengths = sorted(lengths, reverse=True)
    cum = np.cumsum(lengths)
    for i, c in enumerate(cum):
        if c >= cum[-1] / 2:
            return lengths[i]
    return 0


def merge_merged_depth_bed(merged, targets_length, dist_percent=0.005, flank_len=15, start=None, end=None):
    explicit = start is not None and end is not None
    out = {}
    for t, L in targets_length.items():
        dist = L * dist_percent
        if not explicit:
            start, end = flank_len, L - flank_len
        cur = (start, start)
        res = []
        for seg in merged[t]:
            if seg[0] - cur[1] <= dist:
                cur = (cur[0], seg[1])
            else:
                res.append(cur)
                cur = seg
        if end - cur[1] <= dist:
            cur = (cur[0], end)
        res.append(cur)
        out[t] = res
    return out


_RULE = "-" * 136 + "\n\n\n"
_GCI_HEADER = ("Chromosome\tTheoretical maximum N50\tCurated N50\tTheoretical minimum contigs number\t"
               "Curated contigs number\tGCI score\n")


def _score(obs_n50, exp_n50, obs_n, exp_n):
    if obs_n == 0:
        return 0
    return round(100 * log2(obs_n50 / exp_n50 + 1) / log2(obs_n / exp_n + 1), 4)


def compute_index_text(targets_length: Dict[str, int], merged_list, type_list, flank_len=15, dist_percent=0.005,
                       regions_bed=None, depths_list=None, threshold=0, chrs_list=()) -> Tuple[str, Optional[str]]:
    """-> (.gci text, .regions.gci text or None)   (GCI.py:522-657)"""
    regions_bed = regions_bed or {}
    genome = "Genome" if len(chrs_list) == 0 else "All_chromosomes"
    exp_n50 = dict(targets_length)
    exp_n = {t: 1 for t in targets_length}
    exp_n50[genome] = compute_n50(list(targets_length.values()))
    exp_n[genome] = len(targets_length)
    gci = []
    for i, merged in enumerate(merged_list):
        obs_len = complement_merged_depth(merged, targets_length, flank_len)
        obs_n50 = {t: compute_n50(v) for t, v in obs_len.items()}
        obs_n50[genome] = compute_n50([x for v in obs_len.values() for x in v])
        merged2 = merge_merged_depth_bed(merged, targets_length, dist_percent, flank_len)
        obs_len2 = complement_merged_depth(merged2, targets_length, flank_len)
        obs_n = {t: len(v) for t, v in obs_len2.items()}
        obs_n[genome] = sum(len(v) for v in obs_len2.values())
        gci.append(f"{type_list[i]}:\n")
        gci.append(_GCI_HEADER)
        for t in exp_n50:
            gci.append(f"{t}\t{exp_n50[t]}\t{obs_n50[t]}\t{exp_n[t]}\t{obs_n[t]}\t"
                       f"{_score(obs_n50[t], exp_n50[t], obs_n[t], exp_n[t])}\n")
        gci.append(_RULE)
    regions_text = None
    if len(regions_bed) > 0:
        r = ["Chromosome\tStart\tEnd\t" + "\t".join(type_list) + "\n"]
        all_len: List[int] = []
        all_obs: List[List[int]] = [[] for _ in depths_list]
        all_n = [0 for _ in depths_list]
        for t, segs in regions_bed.items():
            for (start, end) in segs:
                e50 = end - start
                if e50 > 0:
                    all_len.append(e50)
                row = []
                for i, dd in enumerate(depths_list):
                    sub = dd[t][start:end]
                    m = {t: collapse_contig(sub, -1, threshold, 0, start)}
                    ol = complement_merged_depth(m, {t: e50}, start, start, end)
                    o50 = compute_n50(ol[t])
                    if e50 > 0:
                        all_obs[i] += ol[t]
                    m2 = merge_merged_depth_bed(m, {t: e50}, dist_percent, start, start, end)
                    on = len(complement_merged_depth(m2, {t: e50}, start, start, end)[t])
                    if e50 > 0:
                        all_n[i] += on
                    row.append(_score(o50, e50, on, 1))
                r.append(f"{t}\t{start}\t{end}\t" + "\t".join(map(str, row)) + "\n")
        a50 = compute_n50(all_len)
        an = len(all_len)
        tot = []
        for i in range(len(depths_list)):
            tot.append(_score(compute_n50(all_obs[i]), a50, all_n[i], an))
        r.append(_RULE)
        r.append("All_regions\t*\t*\t" + "\t".join(map(str, tot)) + "\n")
        regions_text = "".join(r)
    return "".join(gci), regions_text


# ==============================================================================================
# R15: global mean depth (GCI.py:862-868)
# ==============================================================================================

def sliding_window_average_depth(depths, window_size=50000, max_depth=None, start=0):
    """The `-p` numeric front-end, restated from /root/reference/GCI.py:660-705: one pass over a contig (or a region
    slice of it); the window restarts at every zero-depth base (:680-689), a region shorter than the window uses a
    window of 1 (:675-677), means are clamped to max_depth (:683-684, :695-696, :702-703).
    -> (positions in Mb: list of float, values: float64 array)."""
    pos, val, win = [], [], []
    n = len(depths)
    if n < window_size:
        window_size = 1
    i = -1
    for i in range(n):
        d = int(depths[i])
        if d == 0:
            if win:                                        # flush the partial window in front of the zero
                a = sum(win) / len(win)
                val.append(max_depth if a > max_depth else a)
                pos.append((i + start - 1) / 1e6)
                win = []
            val.append(0)
            pos.append((i + start) / 1e6)
        else:
            win.append(d)
            if len(win) == window_size:
                a = sum(win) / window_size
                val.append(max_depth if a > max_depth else a)
                pos.append((i + start) / 1e6)
                win = []
    if win:
        a = sum(win) / len(win)
        val.append(max_depth if a > max_depth else a)
        pos.append((i + start) / 1e6)
    return pos, np.array(val, dtype=np.float64)


def pre_plot_base(depths_list, max_depths, window_size=50000, start=0):
    """/root/reference/GCI.py:708-739: the averaged series of every contig of every read type and the y-axis split of
    the figure (y_max from the first type, y_min from the second when there are two)."""
    averaged = [{} for _ in depths_list]
    maxima = [[] for _ in depths_list]
    for target in depths_list[0].keys():
        for i, depthss in enumerate(depths_list):
            p, v = sliding_window_average_depth(depthss[target], window_size, max_depths[i], start)
            averaged[i][target] = (p, v)
            maxima[i].append(max(v))
    y_max = max(maxima[0]) + 10
    y_min = 0 if len(depths_list) == 1 else max(maxima[1]) + 10
    return averaged, y_min / (y_max + y_min), y_min, y_max


def mean_depth(depths: Dict[str, np.ndarray]) -> float:
    tot = sum(int(lib().orc_sum(_p(np.ascontiguousarray(d, dtype=np.int64)), d.shape[0])) for d in depths.values())
    n = sum(d.shape[0] for d in depths.values())
    return float(tot) / n


def file1_on_contigs_mixed(paf_texts: Sequence[bytes], bams, targets: Sequence[str], chosen: Sequence[str], map_qual: int,
                           mq_cutoff: int, clip_percent: float, iden_percent: float, ovlp_percent: float, heads: bool = False,
                           tmp_dir: Optional[str] = None) -> Dict[str, tuple]:
    """file1_on_contigs() for one filter() call that has PAF files as well (GCI.py:211-254, 272-301): `paf_texts` = the files'
    bytes, `bams` = [(stream, rec_off, references)].  Exact for the same reason: the PAF scoring treats every query
    independently (all lines of the query, in file order) and so does the join, so it is enough to push through the dict
    logic every PAF line and every BAM record of the names that show up on the chosen contigs."""
    import tempfile
    tindex = {t: i for i, t in enumerate(targets)}
    want = np.array([tindex[c] for c in chosen], dtype=np.int32)
    cs = set(chosen)
    keep: Set[str] = set()
    arrays = []
    for stream, off, refs in bams:
        ref_sel = np.array([tindex.get(r, -1) for r in refs], dtype=np.int32)
        a = bam_filter_arrays(stream, off, ref_sel, map_qual, mq_cutoff, clip_percent, iden_percent, heads)
        arrays.append(a)
        on = np.flatnonzero((a["passed"] != 0) & np.isin(a["contig"], want))
        keep.update(read_names(stream, a["name_off"][on], a["name_len"][on]))
    split = [t.decode().splitlines(keepends=True) for t in paf_texts]
    for lines in split:
        for ln in lines:
            c = ln.split("\t", 6)
            if len(c) > 5 and c[5] in cs:
                keep.add(c[0])
    with tempfile.TemporaryDirectory(prefix="gci_oracle_", dir=tmp_dir) as tmp:
        paths = []
        for k, lines in enumerate(split):
            path = os.path.join(tmp, "f%d.paf" % k)
            with open(path, "w") as f:
                f.writelines(ln for ln in lines if ln.split("\t", 1)[0] in keep)
            paths.append(path)
        paf_dicts, hq = paf_filter(paths, targets, map_qual, mq_cutoff, iden_percent) if paths else ([], set())
    dicts = list(paf_dicts)
    for (stream, off, refs), a in zip(bams, arrays):
        d, h = _dict_from_arrays(stream, a, targets, keep)
        dicts.append(d)
        hq |= h
    return {q: seg for q, seg in name_join(dicts, hq, ovlp_percent).items() if seg[0] in cs}


# ==============================================================================================
# R14: the whole path as GCI() strings it together (GCI.py:991-1026), in memory
# ==============================================================================================

def filter_track(paf_paths, bams, targets_length, map_qual, mq_cutoff, iden_percent, clip_percent, ovlp_percent,
                 flank_len):
    """One filter() call (GCI.py:172-312).  `bams` = [(stream, rec_off, references)].
    Returns (depths, depth text of the un-masked track)."""
    targets = list(targets_length)
    paf_dicts, hq = paf_filter(paf_paths, targets, map_qual, mq_cutoff, iden_percent) if paf_paths else ([], set())
    bam_dicts = []
    for stream, off, refs in bams:
        d, h = bam_file_dict(stream, off, refs, targets, map_qual, mq_cutoff, clip_percent, iden_percent)
        bam_dicts.append(d)
        hq |= h
    file1 = name_join(paf_dicts + bam_dicts, hq, ovlp_percent)
    depths = depth_build(file1, targets_length, flank_len)
    return depths, file1


def run_path(hifi=None, nano=None, references=None, lengths=None, ns_bed=None, chrs_list=(), regions_bed=None,
             map_qual=30, mq_cutoff=50, iden_percent=0.9, ovlp_percent=0.9, clip_percent=0.1, flank_len=15,
             threshold=0, dist_percent=0.005, prefix="GCI") -> Dict[str, bytes]:
    """hifi / nano = dict(paf=[paths], bam=[(stream, rec_off, references)]) or None.  Returns
    {output file name: bytes} with `.depth.gz` entries holding the DECOMPRESSED text."""
    out: Dict[str, bytes] = {}
    tl = {r: l for r, l in zip(references, lengths) if (not chrs_list or r in chrs_list)}
    kw = dict(map_qual=map_qual, mq_cutoff=mq_cutoff, iden_percent=iden_percent, clip_percent=clip_percent,
              ovlp_percent=ovlp_percent, flank_len=flank_len)
    if ns_bed:
        out[f"{prefix}.gaps.bed"] = "".join(f"{t}\t{a}\t{b}\n" for t, segs in ns_bed.items() for a, b in segs).encode()

    def one(kind, pfx):
        depths, _ = filter_track(kind.get("paf", []), kind["bam"], tl, **kw)
        out[f"{pfx}.depth.gz"] = depth_text(depths)              # written before the gap mask
        return merge_gaps_depths(depths, ns_bed or None)

    if nano is None or hifi is None:
        kind, label = (hifi, "HiFi") if nano is None else (nano, "Nano")
        depths = one(kind, prefix)
        merged = collapse_depth_range(depths, -1, threshold, flank_len, 0)
        out[f"{prefix}.{threshold}.depth.bed"] = bed_text(merged).encode()
        g, r = compute_index_text(tl, [merged], [label], flank_len, dist_percent, regions_bed, [depths], threshold,
                                  chrs_list)
    else:
        h = one(hifi, prefix + "_hifi")
        n = one(nano, prefix + "_nano")
        two = max2(h, n)
        out[f"{prefix}_two_type.depth.gz"] = depth_text(two)     # written after the gap mask
        two = merge_gaps_depths(two, ns_bed or None)
        ms = []
        for d, pfx in ((h, "_hifi"), (n, "_nano"), (two, "_two_type")):
            m = collapse_depth_range(d, -1, threshold, flank_len, 0)
            out[f"{prefix}{pfx}.{threshold}.depth.bed"] = bed_text(m).encode()
            ms.append(m)
        g, r = compute_index_text(tl, ms, ["HiFi", "Nano", "HiFi + Nano"], flank_len, dist_percent, regions_bed,
                                  [h, n, two], threshold, chrs_list)
    out[f"{prefix}.gci"] = g.encode()
    if r is not None:
        out[f"{prefix}.regions.gci"] = r.encode()
    return out
