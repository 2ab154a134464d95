"""Deterministic synthetic long-read alignments.

No real CHM13 data is reachable offline, so every configuration in BASELINE.json is a
simulated read set over a hard-coded contig table.  Everything here is vectorised numpy so
that the chr19 / 40x case (~137 k HiFi records, ~3.7 GB of inflated BAM bytes) is assembled
in seconds on the GPU box's host; the records are byte-for-byte what a BGZF reader would
hand over after inflate.

Read model: HiFi lengths ~ N(18 kb, 2.5 kb) clipped to [5 kb, 30 kb]; ONT ~ lognormal(median
30 kb, sigma 0.6) clipped to [5 kb, 300 kb]; uniform starts; MAPQ {60: 88 %, 30-49: 4 %,
1-29: 5 %, 0: 3 %}; flags {secondary 3 % (SEQ '*'), supplementary 4 %, unmapped-placed 1 %};
CIGARs alternate match runs with X / I / D events ('=' / 'X' style) or I / D events with
planted mismatches under 'M' ('M' style); 6 % of reads soft-clipped 0.5-40 % at one end, 1 %
at both, 1 % hard-clipped; NM = I + D + X (+ planted) stored as C / S / I by magnitude, first
in the aux block for 90 % of records and last for the rest; reads overlapping planted
coverage holes are removed.
"""
from __future__ import annotations

from dataclasses import dataclass, field, replace
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from .formats import bam as bamfmt

OP_M, OP_I, OP_D, OP_N, OP_S, OP_H, OP_P, OP_EQ, OP_X = range(9)

# T2T-CHM13v2.0 sequence lengths (public release figures; used as *synthetic geometry* only).
CHM13 = (
    ("chr1", 248387328), ("chr2", 242696752), ("chr3", 201105948), ("chr4", 193574945),
    ("chr5", 182045439), ("chr6", 172126628), ("chr7", 160567428), ("chr8", 146259331),
    ("chr9", 150617247), ("chr10", 134758134), ("chr11", 135127769), ("chr12", 133324548),
    ("chr13", 113566686), ("chr14", 101161492), ("chr15", 99753195), ("chr16", 96330374),
    ("chr17", 84276897), ("chr18", 80542538), ("chr19", 61707364), ("chr20", 66210255),
    ("chr21", 45090682), ("chr22", 51324926), ("chrX", 154259566), ("chrY", 62460029),
    ("chrM", 16569),
)
CHR19 = (("chr19", 61707364),)
CTG1 = (("ctg1", 5_000_000),)


def seed_for(config: int, file_index: int) -> int:
    return 20250919 + 1000 * config + file_index


@dataclass
class ReadSet:
    """Per-record arrays, coordinate-sorted by (ref_id, pos)."""
    contigs: Tuple[Tuple[str, int], ...]
    ref_id: np.ndarray      # int32
    pos: np.ndarray         # int32
    mapq: np.ndarray        # uint8
    flag: np.ndarray        # uint16
    l_seq: np.ndarray       # int32
    nm: np.ndarray          # int64
    nm_last: np.ndarray     # bool: NM after the dummy tags instead of before
    names: np.ndarray       # bytes (S dtype)
    cigar: np.ndarray       # uint32 flat (len << 4 | op)
    cigar_off: np.ndarray   # int64 [R + 1]
    holes: Dict[int, List[Tuple[int, int]]] = field(default_factory=dict)

    def __len__(self) -> int:
        return int(self.ref_id.shape[0])

    def op_totals(self) -> np.ndarray:
        """int64 [R, 9] base totals per CIGAR op code (M I D N S H P = X)."""
        R = len(self)
        seg = np.repeat(np.arange(R), np.diff(self.cigar_off))
        out = np.zeros((R, 16), dtype=np.int64)
        np.add.at(out, (seg, (self.cigar & 0xF).astype(np.int64)), (self.cigar >> 4).astype(np.int64))
        return out[:, :9]

    def ref_span(self) -> np.ndarray:
        t = self.op_totals()
        return t[:, OP_M] + t[:, OP_D] + t[:, OP_N] + t[:, OP_EQ] + t[:, OP_X]

    def aligned_bases(self) -> int:
        """The metric's numerator: sum of reference spans of records with flag 0x4 clear."""
        span = np.maximum(self.ref_span(), 1)
        return int(span[(self.flag & 0x4) == 0].sum())

    def take(self, idx: np.ndarray) -> "ReadSet":
        idx = np.asarray(idx)
        n_ops = np.diff(self.cigar_off)[idx]
        new_off = np.zeros(idx.shape[0] + 1, dtype=np.int64)
        np.cumsum(n_ops, out=new_off[1:])
        src = np.repeat(self.cigar_off[:-1][idx], n_ops) + _within(n_ops)
        return replace(self, ref_id=self.ref_id[idx], pos=self.pos[idx], mapq=self.mapq[idx],
                       flag=self.flag[idx], l_seq=self.l_seq[idx], nm=self.nm[idx],
                       nm_last=self.nm_last[idx], names=self.names[idx],
                       cigar=self.cigar[src], cigar_off=new_off)

    def sorted(self) -> "ReadSet":
        return self.take(np.lexsort((self.pos, self.ref_id)))


def concat(a: "ReadSet", b: "ReadSet") -> "ReadSet":
    """Records of `a` followed by records of `b` (same contig table); not re-sorted."""
    off = np.concatenate([a.cigar_off, a.cigar_off[-1] + b.cigar_off[1:]])
    w = max(a.names.dtype.itemsize, b.names.dtype.itemsize)
    return replace(a, ref_id=np.concatenate([a.ref_id, b.ref_id]), pos=np.concatenate([a.pos, b.pos]),
                   mapq=np.concatenate([a.mapq, b.mapq]), flag=np.concatenate([a.flag, b.flag]),
                   l_seq=np.concatenate([a.l_seq, b.l_seq]), nm=np.concatenate([a.nm, b.nm]),
                   nm_last=np.concatenate([a.nm_last, b.nm_last]),
                   names=np.concatenate([a.names.astype("S%d" % w), b.names.astype("S%d" % w)]),
                   cigar=np.concatenate([a.cigar, b.cigar]), cigar_off=off)


def _within(counts: np.ndarray) -> np.ndarray:
    """[0..c0-1, 0..c1-1, ...] for a vector of counts."""
    counts = np.asarray(counts, dtype=np.int64)
    total = int(counts.sum())
    if total == 0:
        return np.zeros(0, dtype=np.int64)
    starts = np.cumsum(counts) - counts
    return np.arange(total, dtype=np.int64) - np.repeat(starts, counts)


def plant_holes(contigs: Sequence[Tuple[str, int]], rng: np.random.Generator, flank: int = 15,
                per_mb: float = 0.1) -> Dict[int, List[Tuple[int, int]]]:
    holes: Dict[int, List[Tuple[int, int]]] = {}
    for ci, (_, L) in enumerate(contigs):
        hs: List[Tuple[int, int]] = []
        if L >= 200_000:
            hs.append((0, int(rng.integers(flank + 1, 2 * flank + 1))))       # ends inside [fl, 2fl]
            hs.append((L - int(rng.integers(1, 5000)), L))                   # touches the far end
            n = max(1, int(L / 1e6 * per_mb))
            for _ in range(n):
                ln = int(np.exp(rng.uniform(0.0, np.log(200_000))))
                a = int(rng.integers(50_000, max(50_001, L - 50_000 - ln)))
                hs.append((a, a + max(1, ln)))
        holes[ci] = sorted(hs)
    return holes


def simulate_reads(contigs: Sequence[Tuple[str, int]], coverage: float, kind: str = "hifi",
                   seed: int = 0, style: Optional[str] = None, holes: bool = True,
                   long_cigar_frac: Optional[float] = None, name_prefix: Optional[str] = None) -> ReadSet:
    """kind: 'hifi' | 'ont'.  style: '=' ('=' / 'X' ops) or 'M'; default by seed parity."""
    rng = np.random.Generator(np.random.PCG64(seed))
    contigs = tuple((str(n), int(l)) for n, l in contigs)
    lens = np.array([l for _, l in contigs], dtype=np.int64)
    total = int(lens.sum())
    if style is None:
        style = "M" if seed % 2 else "="
    if kind == "hifi":
        mean_len, rate = 18_000.0, 0.002
    else:
        mean_len, rate = 30_000.0 * np.exp(0.18), 0.04
    R = max(1, int(round(coverage * total / mean_len)))

    if kind == "hifi":
        span = np.clip(rng.normal(18_000.0, 2_500.0, R), 5_000, 30_000).astype(np.int64)
    else:
        span = np.clip(np.exp(rng.normal(np.log(30_000.0), 0.6, R)), 5_000, 300_000).astype(np.int64)
    ref_id = np.searchsorted(np.cumsum(lens), rng.integers(0, total, R), side="right").astype(np.int32)
    L = lens[ref_id]
    span = np.minimum(span, np.maximum(L // 2, 1))
    pos = (rng.random(R) * (L - span + 1)).astype(np.int64)
    pos = np.minimum(pos, L - span)

    # ---- CIGAR events ----------------------------------------------------------------------
    k = rng.poisson(rate * span).astype(np.int64)
    k = np.minimum(k, np.maximum(span // 8, 0))
    if long_cigar_frac is None:
        long_cigar_frac = 0.001 if kind == "ont" else 0.0
    if long_cigar_frac > 0:
        pick = rng.random(R) < long_cigar_frac
        want = np.full(R, 33_500, dtype=np.int64)          # 2k+1 > 65535 ops
        k = np.where(pick & (L >= 8 * want), want, k)
    E = int(k.sum())
    ev_off = np.zeros(R + 1, dtype=np.int64)
    np.cumsum(k, out=ev_off[1:])
    u = rng.random(E)
    if style == "=":
        ev_op = np.where(u < 0.5, OP_X, np.where(u < 0.75, OP_I, OP_D)).astype(np.uint32)
    else:
        ev_op = np.where(u < 0.5, OP_I, OP_D).astype(np.uint32)
    ev_len = np.where(ev_op == OP_X, 1, 1 + rng.geometric(0.6, E) - 1 + 0).astype(np.int64)
    ev_len = np.clip(ev_len, 1, 4)
    ev_seg = np.repeat(np.arange(R), k)
    ref_ev = np.zeros(R, dtype=np.int64)
    np.add.at(ref_ev, ev_seg, np.where((ev_op == OP_D) | (ev_op == OP_X), ev_len, 0))
    # ensure every match run is >= 1: grow the span where events eat it up
    need = ref_ev + (k + 1)
    span = np.maximum(span, need)
    over = pos + span > L
    pos = np.where(over, np.maximum(L - span, 0), pos)
    bad = pos + span > L                      # contig shorter than the read: drop the events
    if bad.any():
        keep_ev = ~bad[ev_seg]
        ev_op, ev_len, ev_seg = ev_op[keep_ev], ev_len[keep_ev], ev_seg[keep_ev]
        k = np.where(bad, 0, k)
        np.cumsum(k, out=ev_off[1:])
        ref_ev = np.where(bad, 0, ref_ev)
        span = np.where(bad, np.minimum(span, L), span)
        pos = np.where(bad, 0, pos)
    match_total = span - ref_ev
    base = match_total // (k + 1)
    rem = match_total - base * (k + 1)

    # ---- clips -----------------------------------------------------------------------------
    q_aln = match_total.copy()
    np.add.at(q_aln, ev_seg, np.where((ev_op == OP_I) | (ev_op == OP_X), ev_len, 0))
    uc = rng.random(R)
    frac = np.exp(rng.uniform(np.log(0.005), np.log(0.4), R))
    clip_len = np.maximum(1, (frac * q_aln / (1.0 - frac)).astype(np.int64))
    s_lead = np.where(uc < 0.03, clip_len, 0)
    s_trail = np.where((uc >= 0.03) & (uc < 0.06), clip_len, 0)
    both = (uc >= 0.06) & (uc < 0.07)
    s_lead = np.where(both, np.maximum(1, clip_len // 8), s_lead)
    s_trail = np.where(both, np.maximum(1, clip_len // 8), s_trail)
    uh = rng.random(R)
    h_lead = np.where(uh < 0.005, rng.integers(10, 500, R), 0)
    h_trail = np.where((uh >= 0.005) & (uh < 0.01), rng.integers(10, 500, R), 0)

    a = (s_lead > 0).astype(np.int64) + (h_lead > 0)
    b = (s_trail > 0).astype(np.int64) + (h_trail > 0)
    n_ops = a + 2 * k + 1 + b
    cig_off = np.zeros(R + 1, dtype=np.int64)
    np.cumsum(n_ops, out=cig_off[1:])
    seg = np.repeat(np.arange(R), n_ops)
    j = _within(n_ops)
    c = j - a[seg]
    core = (c >= 0) & (c < 2 * k[seg] + 1)
    is_match = core & ((c & 1) == 0)
    is_event = core & ((c & 1) == 1)
    op = np.zeros(seg.shape[0], dtype=np.uint32)
    ln = np.zeros(seg.shape[0], dtype=np.int64)
    match_op = OP_EQ if style == "=" else OP_M
    op[is_match] = match_op
    ln[is_match] = base[seg[is_match]] + np.where(c[is_match] == 0, rem[seg[is_match]], 0)
    ev_idx = ev_off[:-1][seg[is_event]] + (c[is_event] - 1) // 2
    op[is_event] = ev_op[ev_idx]
    ln[is_event] = ev_len[ev_idx]
    lead = c < 0
    # lead ops: H first (if any) then S
    lead_is_h = lead & (j == 0) & (h_lead[seg] > 0)
    lead_is_s = lead & ~lead_is_h
    op[lead_is_h], ln[lead_is_h] = OP_H, h_lead[seg[lead_is_h]]
    op[lead_is_s], ln[lead_is_s] = OP_S, s_lead[seg[lead_is_s]]
    trail = c >= 2 * k[seg] + 1
    t_idx = c - (2 * k[seg] + 1)
    trail_is_s = trail & (t_idx == 0) & (s_trail[seg] > 0)
    trail_is_h = trail & ~trail_is_s
    op[trail_is_s], ln[trail_is_s] = OP_S, s_trail[seg[trail_is_s]]
    op[trail_is_h], ln[trail_is_h] = OP_H, h_trail[seg[trail_is_h]]
    cigar = ((ln.astype(np.uint64) << 4) | op).astype(np.uint32)

    # ---- NM, flags, mapq, names --------------------------------------------------------------
    nm = np.zeros(R, dtype=np.int64)
    np.add.at(nm, ev_seg, ev_len)
    if style == "M":
        nm += rng.binomial(np.maximum(match_total, 1), 0.001)
    l_seq = (q_aln + s_lead + s_trail).astype(np.int64)
    uf = rng.random(R)
    flag = np.where(rng.random(R) < 0.5, 0x10, 0).astype(np.uint16)
    flag = np.where(uf < 0.03, flag | 0x100, flag)
    flag = np.where((uf >= 0.03) & (uf < 0.07), flag | 0x800, flag)
    flag = np.where((uf >= 0.07) & (uf < 0.08), flag | 0x4, flag).astype(np.uint16)
    l_seq = np.where((flag & 0x100) != 0, 0, l_seq).astype(np.int32)     # secondary: SEQ '*'
    um = rng.random(R)
    mapq = np.where(um < 0.88, 60,
                    np.where(um < 0.92, rng.integers(30, 50, R),
                             np.where(um < 0.97, rng.integers(1, 30, R), 0))).astype(np.uint8)
    nm_last = rng.random(R) < 0.10
    ids = np.arange(R)
    if kind == "hifi":
        prefix = name_prefix or "m64011_190830_220126/"
        names = np.char.add(np.char.add(prefix, ids.astype(str)), "/ccs").astype("S")
    else:
        hx = rng.integers(0, 1 << 32, (R, 4), dtype=np.uint64)
        prefix = name_prefix or ""
        names = np.array([prefix + "%08x-%04x-4%03x-%04x-%08x%04x" % (
            h[0], h[1] & 0xFFFF, h[1] >> 20 & 0xFFF, h[2] & 0xFFFF, h[3], h[2] >> 16 & 0xFFFF)
            for h in hx.tolist()], dtype="S")

    rs = ReadSet(contigs, ref_id, pos.astype(np.int32), mapq, flag, l_seq, nm, nm_last, names,
                 cigar, cig_off)
    if holes:
        hl = plant_holes(contigs, rng)
        end = rs.pos.astype(np.int64) + span
        keep = np.ones(R, dtype=bool)
        for ci, hs in hl.items():
            sel = rs.ref_id == ci
            for (ha, hb) in hs:
                keep &= ~(sel & (rs.pos < hb) & (end > ha))
        rs = rs.take(np.flatnonzero(keep))
        rs.holes = hl
    return rs.sorted()


def perturb(rs: ReadSet, seed: int) -> ReadSet:
    """Second-aligner view of the same reads: {identical 90 %, shifted 1-50 bp 5 %, shifted by
    more than 15 % of the read 2 %, other contig 1 %, dropped 2 %}, MAPQ re-drawn."""
    rng = np.random.Generator(np.random.PCG64(seed))
    R = len(rs)
    u = rng.random(R)
    lens = np.array([l for _, l in rs.contigs], dtype=np.int64)
    span = np.maximum(rs.ref_span(), 1)
    pos = rs.pos.astype(np.int64).copy()
    ref = rs.ref_id.astype(np.int64).copy()
    small = (u >= 0.90) & (u < 0.95)
    pos = np.where(small, pos + rng.integers(1, 51, R) * rng.choice([-1, 1], R), pos)
    big = (u >= 0.95) & (u < 0.97)
    pos = np.where(big, pos + (0.15 * span).astype(np.int64) + rng.integers(50, 5000, R), pos)
    other = (u >= 0.97) & (u < 0.98) & (len(rs.contigs) > 1)
    ref = np.where(other, (ref + 1 + rng.integers(0, max(1, len(rs.contigs) - 1), R)) % len(rs.contigs), ref)
    L = lens[ref]
    pos = np.clip(pos, 0, np.maximum(L - span, 0))
    um = rng.random(R)
    mapq = np.where(um < 0.88, 60,
                    np.where(um < 0.92, rng.integers(30, 50, R),
                             np.where(um < 0.97, rng.integers(1, 30, R), 0))).astype(np.uint8)
    out = replace(rs, ref_id=ref.astype(np.int32), pos=pos.astype(np.int32), mapq=mapq)
    keep = ~(u >= 0.98) & (pos + span <= L)
    return out.take(np.flatnonzero(keep)).sorted()


# ----------------------------------------------------------------------------------------------
# Byte assembly
# ----------------------------------------------------------------------------------------------

_DUMMY_TAGS = (b"msi" + np.int32(0).tobytes() + b"ASi" + np.int32(0).tobytes() +
               b"tpAP" + b"def" + np.float32(0.0).tobytes() + b"rlC\x00")


def _nm_bytes(nm: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """-> (flat bytes of 'NM<type><value>' per record, per-record length)."""
    R = nm.shape[0]
    size = np.where(nm < 256, 1, np.where(nm < 65536, 2, 4)).astype(np.int64)
    typ = np.where(nm < 256, ord("C"), np.where(nm < 65536, ord("S"), ord("I"))).astype(np.uint8)
    ln = 3 + size
    off = np.cumsum(ln) - ln
    flat = np.zeros(int(ln.sum()), dtype=np.uint8)
    flat[off] = ord("N")
    flat[off + 1] = ord("M")
    flat[off + 2] = typ
    v = nm.astype(np.uint64)
    for b in range(4):
        sel = size > b
        flat[off[sel] + 3 + b] = ((v[sel] >> (8 * b)) & 0xFF).astype(np.uint8)
    return flat, ln


def _hifi_qual_lut() -> np.ndarray:
    """256-entry table: a uniform random byte -> a HiFi-like base quality (85 % of the bases at the Q93 cap, the rest
    spread over Q2..Q92 with the mass towards the high end): ~1.5 bits of entropy per quality and 2 per base, so that
    a BGZF-compressed file inflates about 3.5:1 like real HiFi data instead of 100:1."""
    q = np.empty(256, dtype=np.uint8)
    q[:218] = 93
    rest = np.round(92.0 - 90.0 * (np.arange(38) / 37.0) ** 1.7).astype(np.uint8)
    q[218:] = np.clip(rest, 2, 92)
    return q


_SEQ_NIBBLES = np.array([1, 2, 4, 8], dtype=np.uint8)                      # A C G T in BAM's 4-bit code
_SEQ_LUT = (_SEQ_NIBBLES[:, None] << 4 | _SEQ_NIBBLES[None, :]).reshape(16).astype(np.uint8)


def to_bam_stream(rs: ReadSet, chunk: int = 1 << 18, header_text: Optional[str] = None, heads: bool = False,
                  seq_qual: str = "const", seed: int = 0):
    """-> (inflated BAM stream uint8[n], record offsets uint64[R]).

    seq_qual = "const": SEQ / QUAL are 0xFF fill (the path never reads them; the file deflates 100:1).
    seq_qual = "random": uniformly random bases and HiFi-like qualities (`_hifi_qual_lut`), so that BGZF inflate costs
    what it costs on real data (bench.py's command-line timing).

    heads=True: the heads stream of the same file (gci_bam_heads: every record without its SEQ / QUAL bytes,
    l_seq unchanged) -- what a genome-scale experiment can hold in memory.

    Records whose CIGAR exceeds 65535 ops are written the spec's way (placeholder CIGAR plus
    a CG:B,I tag at the end of the aux block).  SEQ and QUAL are constant 0xFF fill: the
    path never reads them, they only have to occupy their bytes."""
    names = [n for n, _ in rs.contigs]
    lens = [l for _, l in rs.contigs]
    hdr = np.frombuffer(bamfmt.encode_header(names, lens, header_text), dtype=np.uint8)
    R = len(rs)
    n_ops = np.diff(rs.cigar_off)
    is_long = n_ops > 65535
    n_cig_field = np.where(is_long, 2, n_ops).astype(np.int64)
    name_len = np.char.str_len(rs.names).astype(np.int64) + 1
    l_seq = rs.l_seq.astype(np.int64)
    nm_flat, nm_len = _nm_bytes(rs.nm)
    nm_off = np.cumsum(nm_len) - nm_len
    dummy = np.frombuffer(_DUMMY_TAGS, dtype=np.uint8)
    cg_len = np.where(is_long, 8 + 4 * n_ops, 0)
    aux_len = nm_len + dummy.shape[0] + cg_len
    seq_bytes = np.zeros_like(l_seq) if heads else (l_seq + 1) // 2 + l_seq
    body = 32 + name_len + 4 * n_cig_field + seq_bytes + aux_len
    size = 4 + body
    offs = np.zeros(R + 1, dtype=np.int64)
    np.cumsum(size, out=offs[1:])
    offs += hdr.shape[0]
    if seq_qual == "random" and not heads:
        rng = np.random.Generator(np.random.PCG64(seed))
        out = np.empty(int(offs[-1]), dtype=np.uint8)
        qlut = _hifi_qual_lut()
        step = 1 << 26
        for a in range(0, out.shape[0], step):                  # everything gets quality-like bytes; SEQ is redone below
            b = min(out.shape[0], a + step)
            out[a:b] = qlut[rng.integers(0, 256, b - a, dtype=np.uint8)]
    else:
        out = np.full(int(offs[-1]), 0xFF, dtype=np.uint8)
    out[:hdr.shape[0]] = hdr
    span = np.maximum(rs.ref_span(), 1)

    core = np.zeros(R, dtype=np.dtype([
        ("block_size", "<i4"), ("ref_id", "<i4"), ("pos", "<i4"), ("l_read_name", "u1"), ("mapq", "u1"),
        ("bin", "<u2"), ("n_cigar", "<u2"), ("flag", "<u2"), ("l_seq", "<i4"), ("next_ref", "<i4"),
        ("next_pos", "<i4"), ("tlen", "<i4")]))
    core["block_size"] = body
    core["ref_id"] = rs.ref_id
    core["pos"] = rs.pos
    core["l_read_name"] = name_len
    core["mapq"] = rs.mapq
    core["bin"] = 4680           # placeholder bin: nothing on the path reads it
    core["n_cigar"] = n_cig_field
    core["flag"] = rs.flag
    core["l_seq"] = rs.l_seq
    core["next_ref"] = -1
    core["next_pos"] = -1
    core_b = core.view(np.uint8).reshape(R, 36)
    name_w = rs.names.dtype.itemsize
    names_b = np.frombuffer(rs.names.tobytes(), dtype=np.uint8).reshape(R, name_w) if R else np.zeros((0, 1), np.uint8)

    for lo in range(0, R, chunk):
        hi = min(R, lo + chunk)
        sl = slice(lo, hi)
        base = offs[lo:hi]
        out[(base[:, None] + np.arange(36)[None, :]).ravel()] = core_b[sl].ravel()
        # names (NUL terminated; the S dtype pads with NULs already)
        nl = name_len[sl]
        w = _within(nl)
        rows = np.repeat(np.arange(hi - lo), nl)
        src = np.where(w < name_w, names_b[lo:hi][rows, np.minimum(w, name_w - 1)], 0)
        out[np.repeat(base + 36, nl) + w] = src
        # cigar
        p_cig = base + 36 + nl
        short = ~is_long[sl]
        nco = n_ops[sl] * short
        w = _within(nco * 4)
        src_idx = np.repeat(rs.cigar_off[lo:hi] * 4, nco * 4) + w
        out[np.repeat(p_cig, nco * 4) + w] = rs.cigar.view(np.uint8)[src_idx]
        for r in np.flatnonzero(~short):
            g = lo + r
            ph = np.array([(int(l_seq[g]) << 4) | OP_S, (int(span[g]) << 4) | OP_N], dtype="<u4").view(np.uint8)
            out[p_cig[r]:p_cig[r] + 8] = ph
        if seq_qual == "random" and not heads:
            p_seq = p_cig + 4 * n_cig_field[sl]
            nb = (l_seq[sl] + 1) // 2
            for r in range(hi - lo):
                if nb[r]:
                    out[p_seq[r]:p_seq[r] + nb[r]] = _SEQ_LUT[rng.integers(0, 16, int(nb[r]), dtype=np.uint8)]
        # aux
        p_aux = p_cig + 4 * n_cig_field[sl] + seq_bytes[sl]
        last = rs.nm_last[sl]
        p_nm = np.where(last, p_aux + dummy.shape[0], p_aux)
        p_dm = np.where(last, p_aux, p_aux + nm_len[sl])
        w = _within(nm_len[sl])
        out[np.repeat(p_nm, nm_len[sl]) + w] = nm_flat[np.repeat(nm_off[sl], nm_len[sl]) + w]
        out[(p_dm[:, None] + np.arange(dummy.shape[0])[None, :]).ravel()] = np.tile(dummy, hi - lo)
        for r in np.flatnonzero(~short):
            g = lo + r
            p = int(p_aux[r] + nm_len[g] + dummy.shape[0])
            ops = rs.cigar[rs.cigar_off[g]:rs.cigar_off[g + 1]]
            out[p:p + 4] = np.frombuffer(b"CGBI", dtype=np.uint8)
            out[p + 4:p + 8] = np.array([ops.shape[0]], dtype="<i4").view(np.uint8)
            out[p + 8:p + 8 + 4 * ops.shape[0]] = ops.astype("<u4").view(np.uint8)
    return out, offs[:-1].astype(np.uint64)


def to_paf_lines(rs: ReadSet, seed: int, split_frac: float = 0.02) -> List[str]:
    """PAF view of the mapped primary-like records; `split_frac` of the reads are reported as
    2-3 collinear blocks.  Supplementary / secondary / unmapped records are skipped (a PAF from
    minimap2 would hold them as extra lines; a few are kept as lower-identity decoys)."""
    from .formats.paf import format_line
    rng = np.random.Generator(np.random.PCG64(seed))
    tot = rs.op_totals()
    span = rs.ref_span()
    lines: List[str] = []
    names = [n.decode() for n in rs.names.tolist()]
    for i in range(len(rs)):
        fl = int(rs.flag[i])
        if fl & 0x4 or fl & 0x100:
            continue
        cname, clen = rs.contigs[int(rs.ref_id[i])]
        t = tot[i]
        qlen = int(t[OP_M] + t[OP_I] + t[OP_S] + t[OP_EQ] + t[OP_X] + t[OP_H])
        aln_q = int(t[OP_M] + t[OP_I] + t[OP_EQ] + t[OP_X])
        q0 = int(cig_lead_clip(rs, i))
        ts, te = int(rs.pos[i]), int(rs.pos[i] + span[i])
        alnlen = int(t[OP_M] + t[OP_I] + t[OP_D] + t[OP_EQ] + t[OP_X])
        nmatch = max(0, alnlen - int(rs.nm[i]))
        strand = "-" if fl & 0x10 else "+"
        mq = int(rs.mapq[i])
        parts = 1
        if rng.random() < split_frac and span[i] > 3000:
            parts = int(rng.integers(2, 4))
        if parts == 1:
            lines.append(format_line(names[i], qlen, q0, q0 + aln_q, strand, cname, clen, ts, te, nmatch, alnlen, mq,
                                     ("tp:A:P",)))
        else:
            cuts_t = np.linspace(ts, te, parts + 1).astype(int)
            cuts_q = np.linspace(q0, q0 + aln_q, parts + 1).astype(int)
            for b in range(parts):
                al = int(cuts_t[b + 1] - cuts_t[b])
                nmb = max(0, al - int(rs.nm[i]) // parts)
                lines.append(format_line(names[i], qlen, int(cuts_q[b]), int(cuts_q[b + 1]), strand, cname, clen,
                                         int(cuts_t[b]), int(cuts_t[b + 1]), nmb, al, mq, ("tp:A:P",)))
    return lines


def to_paf_text(rs: ReadSet, seed: int, split_frac: float = 0.02) -> np.ndarray:
    """The PAF view of a read set as the file's bytes (uint8), built column by column with numpy -- to_paf_lines() formats
    one Python string per line, minutes at genome scale.  The same kind of content: mapped, non-secondary records; `split_frac`
    of the reads as 2 - 3 collinear blocks; `tp:A:P` behind the twelve mandatory columns."""
    rng = np.random.Generator(np.random.PCG64(seed))
    keep = np.flatnonzero(((rs.flag & 0x4) == 0) & ((rs.flag & 0x100) == 0))
    n_ops = np.diff(rs.cigar_off)
    tot = rs.op_totals()[keep]
    span = rs.ref_span()[keep]
    first = rs.cigar[np.minimum(rs.cigar_off[:-1], max(rs.cigar.shape[0] - 1, 0))][keep].astype(np.int64) if rs.cigar.shape[0] else np.zeros(keep.shape[0], np.int64)
    q0 = np.where((n_ops[keep] > 0) & np.isin(first & 0xF, (OP_S, OP_H)), first >> 4, 0)          # leading clip (one op in these reads)
    qlen = tot[:, OP_M] + tot[:, OP_I] + tot[:, OP_S] + tot[:, OP_EQ] + tot[:, OP_X] + tot[:, OP_H]
    aln_q = tot[:, OP_M] + tot[:, OP_I] + tot[:, OP_EQ] + tot[:, OP_X]
    alnlen = tot[:, OP_M] + tot[:, OP_I] + tot[:, OP_D] + tot[:, OP_EQ] + tot[:, OP_X]
    ts = rs.pos[keep].astype(np.int64)
    te = ts + span
    nm = rs.nm[keep].astype(np.int64)
    parts = np.where((rng.random(keep.shape[0]) < split_frac) & (span > 3000), rng.integers(2, 4, keep.shape[0]), 1)
    row = np.repeat(np.arange(keep.shape[0]), parts)                # one PAF line per block
    b = np.arange(row.shape[0]) - np.repeat(np.cumsum(parts) - parts, parts)
    p = parts[row]
    cut = lambda lo, hi, k: lo + (hi - lo) * k // p                  # noqa: E731
    c_ts, c_te = cut(ts[row], te[row], b), cut(ts[row], te[row], b + 1)
    c_qs, c_qe = cut(q0[row], (q0 + aln_q)[row], b), cut(q0[row], (q0 + aln_q)[row], b + 1)
    c_al = np.where(p == 1, alnlen[row], c_te - c_ts)
    c_nm = np.maximum(0, c_al - nm[row] // p)
    ref = rs.ref_id[keep][row].astype(np.int64)
    tlen = np.asarray([l for _, l in rs.contigs], dtype=np.int64)[ref]
    tnames = [n.encode() for n, _ in rs.contigs]
    tn_len = np.asarray([len(x) for x in tnames], dtype=np.int64)[ref]
    tn_w = max(len(x) for x in tnames)
    tn_mat = np.zeros((len(tnames), tn_w), dtype=np.uint8)
    for i, x in enumerate(tnames):
        tn_mat[i, :len(x)] = np.frombuffer(x, dtype=np.uint8)
    names = rs.names[keep][row]
    nw = names.dtype.itemsize
    name_mat = np.frombuffer(names.tobytes(), dtype=np.uint8).reshape(-1, nw) if row.shape[0] else np.zeros((0, 1), np.uint8)
    qn_len = np.char.str_len(names).astype(np.int64)
    strand = np.where((rs.flag[keep][row] & 0x10) != 0, ord("-"), ord("+")).astype(np.uint8)
    mapq = rs.mapq[keep][row].astype(np.int64)
    nums = [qlen[row], c_qs, c_qe, None, None, tlen, c_ts, c_te, c_nm, c_al, mapq]     # columns 1..11 (None: strand, tname)
    ndig = lambda v: np.where(v == 0, 1, np.floor(np.log10(np.maximum(v, 1))).astype(np.int64) + 1)     # noqa: E731
    widths = [qn_len] + [ndig(v) if v is not None else None for v in nums]
    widths[4], widths[5] = np.ones_like(qn_len), tn_len
    tail = np.frombuffer(b"\ttp:A:P\n", dtype=np.uint8)
    line_len = sum(widths) + 11 + tail.shape[0]
    off = np.cumsum(line_len) - line_len
    out = np.full(int(line_len.sum()), ord("\t"), dtype=np.uint8)
    at = off.copy()
    for k in range(nw):                                             # qname
        m = qn_len > k
        out[at[m] + k] = name_mat[m, k]
    at = at + qn_len + 1
    for col in range(1, 12):
        v = nums[col - 1]
        if col == 4:
            out[at] = strand
        elif col == 5:
            for k in range(tn_w):
                m = tn_len > k
                out[at[m] + k] = tn_mat[ref[m], k]
        else:
            w, x = widths[col], v.copy()
            for d in range(int(w.max()) if w.shape[0] else 0):      # digit d from the right
                m = w > d
                out[at[m] + w[m] - 1 - d] = (ord("0") + x[m] % 10).astype(np.uint8)
                x //= 10
        at = at + widths[col] + 1
    at -= 1                                                          # the tab behind column 11 is the tail's own
    for k in range(tail.shape[0]):
        out[at + k] = tail[k]
    return out


def cig_lead_clip(rs: ReadSet, i: int) -> int:
    ops = rs.cigar[rs.cigar_off[i]:rs.cigar_off[i + 1]]
    q = 0
    for v in ops[:2].tolist():
        if (v & 0xF) in (OP_S, OP_H):
            q += v >> 4
        else:
            break
    return q


def write_bam_file(path: str, rs: ReadSet, level: int = 1, threads: int = 4) -> None:
    stream, _ = to_bam_stream(rs)
    bamfmt.write_bam_stream(path, stream, level=level, threads=threads)
    bamfmt.write_bai(path + ".bai", len(rs.contigs), bam_path=path)


def write_reference_fasta(path: str, contigs: Sequence[Tuple[str, int]], gaps: Optional[Dict[str, List[Tuple[int, int]]]] = None,
                          width: int = 80) -> None:
    """A/C/G/T filler with optional N runs; only ids and N runs matter to the path."""
    gaps = gaps or {}
    with open(path, "wb") as f:
        for name, L in contigs:
            seq = np.frombuffer((b"ACGT" * (L // 4 + 1))[:L], dtype=np.uint8).copy()
            for k, (a, b) in enumerate(gaps.get(name, [])):
                seq[a:b] = ord("N") if k % 2 == 0 else ord("n")
            f.write(b">" + name.encode() + b" synthetic\n")
            nfull = L // width
            if nfull:
                body = np.empty((nfull, width + 1), dtype=np.uint8)
                body[:, :width] = seq[:nfull * width].reshape(nfull, width)
                body[:, width] = 10
                f.write(body.tobytes())
            if L % width:
                f.write(seq[nfull * width:].tobytes() + b"\n")
