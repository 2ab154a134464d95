#!/usr/bin/env python3
"""How fast does a DEFLATE decoder that starts at a WRONG bit offset fall into step with the real symbol sequence?  (CPU experiment for
the open item "a wave per BGZF member": lanes start at guessed offsets inside a block, with the block's own code tables, and a stitch
pass keeps what lines up.)  Members of a HiFi BAM with SEQ / QUAL of realistic entropy, deflated at zlib level 1 as htslib does; a pure
Python inflate records where every literal / length symbol of a block starts; decoders are then started at arbitrary bit offsets of
the block body and followed until they stand on a true symbol start.  Prints the distribution of that distance and the block
structure (blocks per member, symbols and bits per block, share of matches).  Usage: exp_inflate_resync.py [members] [starts per block]"""
import os, sys, tempfile, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gci_amd import synth, hostio
from gci_amd.formats import bam as bamfmt

N_MEMBERS = int(sys.argv[1]) if len(sys.argv) > 1 else 40
STARTS = int(sys.argv[2]) if len(sys.argv) > 2 else 48

LEN_BASE = [3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258]
LEN_EXTRA = [0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0]
DIST_EXTRA = [0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13]
DIST_BASE = [1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577]
MATCHES = []          # (length, distance) of every match of the true sequences (filled by step(..., keep=True))
CLEN_ORDER = [16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15]


class Bits:
    def __init__(self, data, pos=0):
        self.v = int.from_bytes(data, "little")
        self.n = 8 * len(data)
        self.pos = pos

    def take(self, k):
        x = (self.v >> self.pos) & ((1 << k) - 1)
        self.pos += k
        return x


def make_code(lens):
    """{(length, code): symbol} of a canonical Huffman code (RFC 1951 3.2.2)."""
    cnt = [0] * 16
    for l in lens:
        cnt[l] += 1
    cnt[0] = 0
    nxt, code = [0] * 16, 0
    for l in range(1, 16):
        code = (code + cnt[l - 1]) << 1
        nxt[l] = code
    table = {}
    for s, l in enumerate(lens):
        if l:
            table[(l, nxt[l])] = s
            nxt[l] += 1
    return table


def decode_sym(b, table):
    """One symbol, bit by bit (codes are packed most significant bit first); None: no such code within 15 bits / past the end."""
    code = 0
    for l in range(1, 16):
        if b.pos >= b.n:
            return None
        code = (code << 1) | b.take(1)
        s = table.get((l, code))
        if s is not None:
            return s
    return None


def read_block_header(b):
    """-> (last, type, lit table, dist table) with b behind the header; stored blocks: tables None."""
    last, typ = b.take(1), b.take(2)
    if typ == 0:
        return last, 0, None, None
    if typ == 1:
        return last, 1, make_code([8] * 144 + [9] * 112 + [7] * 24 + [8] * 8), make_code([5] * 30)
    hlit, hdist, hclen = b.take(5) + 257, b.take(5) + 1, b.take(4) + 4
    cl = [0] * 19
    for i in range(hclen):
        cl[CLEN_ORDER[i]] = b.take(3)
    ct = make_code(cl)
    lens = []
    while len(lens) < hlit + hdist:
        s = decode_sym(b, ct)
        if s < 16:
            lens.append(s)
        elif s == 16:
            lens += [lens[-1]] * (3 + b.take(2))
        elif s == 17:
            lens += [0] * (3 + b.take(3))
        else:
            lens += [0] * (11 + b.take(7))
    return last, 2, make_code(lens[:hlit]), make_code(lens[hlit:hlit + hdist])


def step(b, lit, dist, keep=False):
    """One literal / length symbol with everything that belongs to it -> ('lit' | 'match' | 'eob' | None = not decodable here)."""
    s = decode_sym(b, lit)
    if s is None or s > 285:
        return None
    if s < 256:
        return "lit"
    if s == 256:
        return "eob"
    ln = LEN_BASE[s - 257] + b.take(LEN_EXTRA[s - 257])
    d = decode_sym(b, dist)
    if d is None or d > 29:
        return None
    dv = DIST_BASE[d] + b.take(DIST_EXTRA[d])
    if keep:
        MATCHES.append((ln, dv))
    return "match"


def main():
    contigs = (("chr19", 1_200_000),)
    rs = synth.simulate_reads(contigs, 40, "hifi", seed=synth.seed_for(2, 0))
    stream, _ = synth.to_bam_stream(rs, seq_qual="random", seed=7)
    p = os.path.join(tempfile.mkdtemp(), "x.bam")
    bamfmt.write_bam_stream(p, stream, level=1, threads=4)
    raw = np.fromfile(p, dtype=np.uint8)
    pos, isz = hostio.bgzf_blocks(raw)
    rng = np.random.default_rng(5)
    blocks_per_member, syms_per_block, bits_per_block, match_share, out_per_sym = [], [], [], [], []
    dist_bits, dist_syms, failed = [], [], 0
    members = [m for m in range(1, len(isz) - 1) if isz[m] > 60000][:N_MEMBERS]
    for m in members:
        data = bytes(raw[int(pos[m]) + 18:int(pos[m + 1]) - 8])
        assert len(zlib.decompress(data, -15)) == int(isz[m])
        b = Bits(data)
        nb = 0
        while True:
            last, typ, lit, dist = read_block_header(b)
            nb += 1
            if typ == 0:
                b.pos = (b.pos + 7) & ~7
                ln = b.take(16); b.take(16); b.pos += 8 * ln
            else:
                body0 = b.pos
                starts, kinds = [], []
                while True:
                    starts.append(b.pos)
                    k = step(b, lit, dist, keep=True)
                    kinds.append(k)
                    if k == "eob":
                        break
                body1 = b.pos
                true = set(starts)
                syms_per_block.append(len(starts)); bits_per_block.append(body1 - body0)
                match_share.append(kinds.count("match") / len(kinds))
                # decoders started anywhere in the body
                for p0 in rng.integers(body0 + 1, max(body0 + 2, body1 - 64), STARTS).tolist():
                    if p0 in true:
                        continue
                    sb = Bits(data, p0)
                    n_sym, ok = 0, False
                    while sb.pos < body1 and n_sym < 20000:
                        if sb.pos in true:
                            ok = True
                            break
                        at = sb.pos
                        if step(sb, lit, dist) is None:
                            sb.pos = at + 1                      # not a code here: one bit on
                        n_sym += 1
                    if ok:
                        dist_bits.append(sb.pos - p0); dist_syms.append(n_sym)
                    else:
                        failed += 1
            if last:
                break
        blocks_per_member.append(nb)
        out_per_sym.append(int(isz[m]) / max(1, sum(syms_per_block[-nb:])))
    q = lambda a, f: float(np.quantile(np.asarray(a), f))
    print("%d members of ~64 KiB (zlib level 1, HiFi BAM with SEQ / QUAL of realistic entropy)" % len(members))
    print("blocks per member: mean %.2f (min %d, max %d); symbols per block: median %.0f; body bits per block: median %.0f; matches: %.1f %% of the symbols; output bytes per symbol %.2f"
          % (np.mean(blocks_per_member), min(blocks_per_member), max(blocks_per_member), q(syms_per_block, .5), q(bits_per_block, .5), 100 * np.mean(match_share), np.mean(out_per_sym)))
    print("decoders started at %d wrong bit offsets: %d never met the true sequence inside their block" % (len(dist_bits) + failed, failed))
    print("bits until in step:    median %.0f   90 %% %.0f   99 %% %.0f   max %.0f" % (q(dist_bits, .5), q(dist_bits, .9), q(dist_bits, .99), max(dist_bits)))
    print("symbols until in step: median %.0f   90 %% %.0f   99 %% %.0f   max %.0f" % (q(dist_syms, .5), q(dist_syms, .9), q(dist_syms, .99), max(dist_syms)))
    ml, md = np.asarray([x[0] for x in MATCHES]), np.asarray([x[1] for x in MATCHES])
    print("matches: length mean %.1f (median %.0f, 90 %% %.0f); distance median %.0f, 10 %% %.0f, 1 %% %.0f; distance < 64: %.1f %%, < 256: %.1f %%, < 1024: %.1f %%; distance < length (overlapping): %.2f %%"
          % (ml.mean(), q(ml, .5), q(ml, .9), q(md, .5), q(md, .1), q(md, .01), 100 * (md < 64).mean(), 100 * (md < 256).mean(), 100 * (md < 1024).mean(), 100 * (md < ml).mean()))
    # how deep are the chains of matches that copy from what a match wrote?  (rounds a batch-parallel copy needs)
    sub = q(bits_per_block, .5) / 64
    print("a block cut into 64 pieces: %.0f bits each; a lane that starts wrong wastes the median %.0f bits = %.0f %% of its piece"
          % (sub, q(dist_bits, .5), 100 * q(dist_bits, .5) / sub))


if __name__ == "__main__":
    main()
