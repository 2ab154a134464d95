#!/usr/bin/env python3
"""An executable statement of the "wave per BGZF member" inflate (open items) on the CPU, held against zlib:
  1. per DEFLATE block: the header and the code tables (serial), the rest of the payload cut into 64 pieces of equal bit length
     (of at least MIN_PIECE bits: the short last block of a member gets fewer lanes);
  2. lane k decodes from the START of piece k as if a literal / length symbol began there (lane 0 does begin on one) and notes the
     symbol starts it visits in the first WINDOW bits of its piece;
  3. stitch: lane k runs on past the end of its piece until it stands on a position lane k + 1 noted -- from there lane k + 1's
     sequence is the true one.  A lane that finds no such position within WINDOW bits takes the neighbour's piece over as well (the
     model counts these; a kernel would hand the member to the one-lane-per-member decoder).  The lane that meets the end-of-block
     symbol ends the block: the lanes behind it decoded the next block's bits with the wrong tables and are dropped;
  4. every lane's share of the output is known now: offsets by a prefix sum; literals are written, matches listed;
  5. copies: the matches in output order, 64 at a time; a match copies when every byte of its source is final, else waits for the
     next pass over the batch (overlapping matches copy byte by byte inside one pass).
Checks the result byte for byte against zlib for every member and prints what a kernel design needs: lanes that had to take a piece
over, passes per batch of copies.  Usage: model_inflate_wave.py [members]"""
import os, sys, tempfile, zlib
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import exp_inflate_resync as R
from gci_amd import synth, hostio
from gci_amd.formats import bam as bamfmt

LANES, WINDOW, MIN_PIECE = 64, 1024, 2048       # (a piece shorter than the window a lane needs to fall into step is no piece)


def tokens_from(data, pos, lit, dist, stop_at, limit_bits):
    """Decode from bit `pos` until the position reaches stop_at (a symbol that starts in front of it is finished), end of block, or an
    undecodable spot -> [(start bit, kind, a, b)], kind 0 literal (a = byte), 1 match (a = length, b = distance), 2 end of block."""
    b = R.Bits(data, pos)
    out = []
    while b.pos < stop_at and b.pos < limit_bits:
        at = b.pos
        s = R.decode_sym(b, lit)
        if s is None or s > 285:
            b.pos = at + 1
            out.append((at, 3, 0, 0))                      # nothing decodable here: one bit on (only ever on a wrong path)
            continue
        if s < 256:
            out.append((at, 0, s, 0))
        elif s == 256:
            out.append((at, 2, 0, 0))
            break
        else:
            ln = R.LEN_BASE[s - 257] + b.take(R.LEN_EXTRA[s - 257])
            d = R.decode_sym(b, dist)
            if d is None or d > 29:
                b.pos = at + 1
                out.append((at, 3, 0, 0))
                continue
            out.append((at, 1, ln, R.DIST_BASE[d] + b.take(R.DIST_EXTRA[d])))
    return out, b.pos


def inflate_member(data, stats):
    nbits = 8 * len(data)
    b = R.Bits(data)
    toks = []                                              # the member's true token sequence, block after block
    while True:
        last, typ, lit, dist = R.read_block_header(b)
        if typ == 0:
            b.pos = (b.pos + 7) & ~7
            ln = b.take(16); b.take(16)
            for i in range(ln):
                toks.append((0, b.take(8), 0))
        else:
            body0 = b.pos
            piece = max(MIN_PIECE, -(-(nbits - body0) // LANES))
            bound = [min(nbits, body0 + k * piece) for k in range(LANES + 1)]
            # 2. every lane over its own piece (+ the tail of its last symbol)
            seqs, ends = [], []
            for k in range(LANES):
                if bound[k] >= nbits:
                    seqs.append([]); ends.append(bound[k]); continue
                t, e = tokens_from(data, bound[k], lit, dist, bound[k + 1], nbits)
                seqs.append(t); ends.append(e)
            # 3. stitch
            k, true_from = 0, body0                        # lane k's sequence is the true one from bit `true_from`
            block_end = None
            while block_end is None:
                mine = [t for t in seqs[k] if t[0] >= true_from]
                assert all(t[1] != 3 for t in mine), "an undecodable spot on the true path"
                eob = next((i for i, t in enumerate(mine) if t[1] == 2), None)
                if eob is not None:
                    toks += [(t[1], t[2], t[3]) for t in mine[:eob]]
                    stats["lanes_used"].append(k + 1)
                    # where the block ends: behind the end-of-block code
                    bb = R.Bits(data, mine[eob][0]); R.decode_sym(bb, lit); block_end = bb.pos
                    break
                toks += [(t[1], t[2], t[3]) for t in mine]
                pos = ends[k]                              # lane k stands here, inside piece k + 1 (or at its start)
                nxt = k + 1
                assert nxt < LANES, "ran out of lanes before the end-of-block code"
                noted = {t[0] for t in seqs[nxt] if t[0] < bound[nxt] + WINDOW}
                run, p = [], pos
                while p not in noted:
                    if p >= bound[nxt] + WINDOW:           # no meeting point: take the neighbour's piece over
                        stats["taken_over"] += 1
                        t, p2 = tokens_from(data, p, lit, dist, bound[nxt + 1], nbits)
                        run += t
                        seqs[nxt], ends[nxt] = [], p2       # its own decode is void
                        p = None
                        break
                    t, p = tokens_from(data, p, lit, dist, p + 1, nbits)   # one symbol
                    run += t
                    if t and t[-1][1] == 2:
                        break
                stats["overrun_bits"].append((p if p is not None else ends[nxt]) - bound[nxt])
                if run and run[-1][1] == 2:                # the block ended in the overrun
                    toks += [(t[1], t[2], t[3]) for t in run[:-1]]
                    stats["lanes_used"].append(k + 1)
                    bb = R.Bits(data, run[-1][0]); R.decode_sym(bb, lit); block_end = bb.pos
                    break
                toks += [(t[1], t[2], t[3]) for t in run]
                if p is None:                              # taken over: go on behind the neighbour's piece
                    k, true_from = nxt, 1 << 62            # (nothing of its own sequence counts)
                    seqs[nxt] = []
                    ends[nxt] = ends[nxt]
                    true_from = ends[nxt]
                    # the lane after it is met from where this one stands now
                    k = nxt
                    continue
                k, true_from = nxt, p
            b.pos = block_end
        if last:
            break
    # 4. + 5.: offsets, literals, copies in batches
    sizes = [1 if t[0] == 0 else t[1] for t in toks]
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    out = bytearray(int(off[-1]))
    final = np.zeros(int(off[-1]) + 1, dtype=bool)
    matches = []
    for i, t in enumerate(toks):
        if t[0] == 0:
            out[off[i]] = t[1]; final[off[i]] = True
        else:
            matches.append((int(off[i]), t[1], t[2]))
    for j in range(0, len(matches), LANES):
        batch, passes = matches[j:j + LANES], 0
        todo = list(range(len(batch)))
        while todo:
            passes += 1
            ready = []
            for i in todo:                                   # all decisions of a pass are taken on the state in front of it
                o, ln, d = batch[i]
                src_final = final[o - d:min(o, o - d + ln)].all()   # (bytes of the source that lie inside the match itself come from it)
                if src_final:
                    ready.append(i)
            assert ready, "a batch that cannot move"
            for i in ready:
                o, ln, d = batch[i]
                for x in range(ln):
                    out[o + x] = out[o + x - d]
                final[o:o + ln] = True
            todo = [i for i in todo if i not in ready]
        stats["passes"].append(passes)
    return bytes(out)


def main():
    n_members = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    rs = synth.simulate_reads((("chr19", 1_200_000),), 40, "hifi", seed=synth.seed_for(2, 0))
    stream, _ = synth.to_bam_stream(rs, seq_qual="random", seed=7)
    p = os.path.join(tempfile.mkdtemp(), "x.bam")
    bamfmt.write_bam_stream(p, stream, level=1, threads=4)
    raw = np.fromfile(p, dtype=np.uint8)
    pos, isz = hostio.bgzf_blocks(raw)
    stats = {"taken_over": 0, "overrun_bits": [], "passes": [], "lanes_used": []}
    members = [0] + [m for m in range(1, len(isz) - 1)][:n_members - 2] + [len(isz) - 1]      # the header's member and the EOF block too
    for m in members:
        data = bytes(raw[int(pos[m]) + 18:int(pos[m + 1]) - 8])
        want = zlib.decompress(data, -15)
        got = inflate_member(data, stats)
        assert got == want, "member %d differs from zlib" % m
    ps, ov = np.asarray(stats["passes"]), np.asarray(stats["overrun_bits"])
    print("%d members inflated by the model, every one equal to zlib's output" % len(members))
    print("stitches: %d, pieces taken over for want of a meeting point within %d bits: %d; overrun into the neighbour's piece: median %d bits, 99 %% %d, max %d"
          % (len(ov), WINDOW, stats["taken_over"], np.median(ov), np.quantile(ov, .99), ov.max()))
    print("lanes in use when a block ended: median %d of %d (the lanes behind the end of the block decoded the next block with the wrong tables)"
          % (np.median(stats["lanes_used"]), LANES))
    print("copies: %d batches of up to %d matches; passes per batch: mean %.2f, 90 %% %d, max %d"
          % (len(ps), LANES, ps.mean(), np.quantile(ps, .9), ps.max()))


if __name__ == "__main__":
    main()
