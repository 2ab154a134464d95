// inflate_wave.hip -- EXPERIMENT (not part of libgci_hip.so): a WAVE per BGZF member instead of a lane per member.
//
// State when committed (end of round 4): compiles for gfx950; NOT YET RUN ON HARDWARE -- the GPU budget of the round was spent
// when the measurements that justify it came in (profiles/r04_inflate_resync.txt).  The algorithm is the one
// tools/model_inflate_wave.py executes on the CPU and holds against zlib; tools/hwtests/inflate_wave.py builds this file,
// runs it on a BAM with SEQ / QUAL of realistic entropy and compares every member with zlib.
//
// Per DEFLATE block of a member:
//   1. lane 0 reads the block header's code lengths; the wave sorts the symbols by code (ballots) and fills the primary tables (every
//      lane its entries) -- one set of tables per wave in LDS: 1.4 KB, not 1.3 KB per member-lane;
//   2. the rest of the payload is cut into 64 pieces of equal bit length (>= MIN_PIECE); lane k decodes from the START of piece k as
//      if a literal / length symbol began there, counts output bytes and matches, and notes the symbol starts it visits in the
//      first WINDOW bits of its piece as a bitmap in LDS;
//   3. stitch: lane k runs on behind the end of its piece until it stands on a bit position lane k + 1 noted: from there on lane
//      k + 1's sequence is the true one (a decoder that starts wrong is in step after 64 bits in the median, 413 at the 99th
//      percentile).  Lane k + 1 re-decodes its first few symbols up to that position to know what to leave out;
//   4. the lane whose true range holds the end-of-block code ends the block; the lanes behind it are void;
//   5. prefix sums over the lanes give every lane its place in the output and in the member's match list; a second pass over the
//      same bits writes the literals and lists the matches {destination, length, distance};
//   6. the copies: 64 matches at a time, one per lane; a lane copies when the earlier lanes of the batch whose destination meets its
//      source are done (5.96 passes per batch on the host).  v1 (this file): the bytes in global memory, a round trip per pass; the
//      member's window in LDS is the next step.
// Known costs of this first version, for whoever runs it first: (a) the ~320 code lengths of a block header are lane 0's work alone
// (they are a bit stream); the sorting by code and the primary tables are the wave's; (b) every symbol costs an unaligned 8-byte load from global memory (a piece is
// ~220 bytes: the lanes of a wave read 14 KB side by side, L1 should hold it; the payload in LDS would take 27 KB per wave);
// (c) the copies make a round trip to memory per pass, ~6 passes per 64 matches, ~190 batches per member.
// A member that does not stitch (no meeting point within WINDOW bits, an end-of-block code on a wrong path, an undecodable spot on
// the true path, a capacity) is reported in status[] and left to the lane-per-member kernel.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "inflate_wave_core.hpp"
#ifndef IW_STAGE
#define IW_STAGE 4
#endif

namespace {

// build_code() by the whole wave (one workgroup = one wave): the counts per code length and the place of every symbol among the symbols
// of its length by ballots -- lane s + 64 c holds symbol s of chunk c, a ballot per (length, chunk) --, the fifteen-step prefix over the
// lengths by lane 0.  Same Canon and same `sorted` as build_code() (the host check plays it lane by lane and compares).
__device__ bool build_code_wave(const uint8_t* lens, int n, Canon& cn, uint16_t* sorted, int lane, uint32_t* s_ok)
{
    uint32_t my[5];
#pragma unroll
    for (int c = 0; c < 5; c++) { const int s = lane + 64 * c; my[c] = s < n ? lens[s] : 0u; }
    uint32_t cnt = 0;
    for (uint32_t L = 1; L < 16; L++) {
        uint32_t t = 0;
#pragma unroll
        for (int c = 0; c < 5; c++) t += (uint32_t)__builtin_popcountll(__ballot(my[c] == L));
        if ((uint32_t)lane == L) cnt = t;
    }
    if (lane < 16) cn.next[lane] = (uint16_t)(lane ? cnt : 0u);
    __syncthreads();
    if (lane == 0) {
        uint32_t code = 0, idx = 0, left = 1u << 15, prev = 0;
        bool ok = true;
        cn.limit[0] = 0; cn.off[0] = 0;
        for (int l = 1; l < 16; l++) {
            const uint32_t k = cn.next[l];
            code = (code + prev) << 1;
            cn.limit[l] = (uint16_t)((code + k) << (15 - l));
            cn.off[l] = (int16_t)((int)idx - (int)code);
            cn.next[l] = (uint16_t)idx;
            idx += k; prev = k;
            const uint32_t need = k << (15 - l);
            if (need > left) ok = false; else left -= need;
        }
        *s_ok = ok ? 1u : 0u;
    }
    __syncthreads();
    for (uint32_t L = 1; L < 16; L++) {
        uint32_t at = cn.next[L];
#pragma unroll
        for (int c = 0; c < 5; c++) {
            const unsigned long long mask = __ballot(my[c] == L);
            if (my[c] == L) sorted[at + (uint32_t)__builtin_popcountll(mask & ((1ull << lane) - 1ull))] = (uint16_t)(lane + 64 * c);
            at += (uint32_t)__builtin_popcountll(mask);
        }
        if (lane == 0) cn.next[L] = (uint16_t)at;                        // one past the last symbol of length L, as build_code() leaves it
    }
    __syncthreads();
    return *s_ok != 0u;
}

__device__ __forceinline__ uint32_t wave_excl_sum(uint32_t v, int lane, uint32_t& total)
{
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)inc, d, 64); if (lane >= d) inc += o; }
    total = (uint32_t)__shfl((int)inc, 63, 64);
    return inc - v;
}

}  // namespace

// One wave per member, the workgroups go over the members in strides of the grid.  matches: gridDim.x * MATCH_CAP entries.
extern "C" __global__ __launch_bounds__(64) void k_inflate_wave(const uint8_t* __restrict__ raw, const uint64_t* __restrict__ member_pos,
                                                                const uint64_t* __restrict__ out_off, uint32_t n_members,
                                                                uint8_t* __restrict__ out, uint2* __restrict__ matches,
                                                                uint32_t* __restrict__ status)
{
    __shared__ Tabs T;
    __shared__ uint32_t note[64][WINDOW / 32];
    __shared__ uint32_t s_hdr[8];                                         // type, body start, last, hlit | hdist << 16, flag of build_code_wave
    const int lane = threadIdx.x;
    uint2* const mlist = matches + (size_t)blockIdx.x * MATCH_CAP;
    for (uint32_t m = blockIdx.x; m < n_members; m += gridDim.x) {
        const uint64_t pos0 = member_pos[m], pos1 = member_pos[m + 1];
        const uint32_t isize = (uint32_t)(out_off[m + 1] - out_off[m]);
        uint8_t* const dst = out + out_off[m];
        uint32_t st = ST_OK;
        if (pos1 < pos0 + 26 || isize > 65536u) { if (lane == 0) status[m] = ST_HEADER; continue; }
        const uint32_t xlen = (uint32_t)raw[pos0 + 10] | ((uint32_t)raw[pos0 + 11] << 8);
        const uint8_t* base = raw + pos0 + 12 + xlen;
        const uint32_t nbits = 8u * (uint32_t)(pos1 - 8 - (pos0 + 12 + xlen));
        uint32_t bpos = 0, out_pos = 0, n_match = 0;                      // (uniform over the wave)
        for (bool last = false; !last && st == ST_OK;) {
            __syncthreads();
            if (lane == 0) {
                uint32_t p = bpos;
                bool l = false;
                int hlit = 0, hdist = 0;
                const int type = block_header(base, p, nbits, T, l, hlit, hdist, false);
                s_hdr[0] = (uint32_t)type; s_hdr[1] = p; s_hdr[2] = l ? 1u : 0u; s_hdr[3] = (uint32_t)hlit | ((uint32_t)hdist << 16);
            }
            for (uint32_t i = lane; i < 64u * (WINDOW / 32); i += 64) (&note[0][0])[i] = 0u;
            __syncthreads();
            const uint32_t type = s_hdr[0], body0 = s_hdr[1];
            last = s_hdr[2] != 0u;
            if (type == 3u) { st = ST_HEADER; break; }
            if (type != 0u) {                                             // the two codes, then the primary tables: every lane its share
                const int hlit = (int)(s_hdr[3] & 0xFFFFu), hdist = (int)(s_hdr[3] >> 16);
                const bool ok_l = build_code_wave(T.lens + 32, hlit, T.lit_cn, T.lit_sorted, lane, &s_hdr[4]);
                const bool ok_d = build_code_wave(T.lens + 32 + hlit, hdist, T.dist_cn, T.dist_sorted, lane, &s_hdr[4]);
                if (!ok_l || !ok_d) { st = ST_HEADER; break; }
                for (uint32_t k = lane; k < (1u << LIT_BITS); k += 64) T.lit_tab[k] = table_entry(k, LIT_BITS, T.lit_cn, T.lit_sorted);
                if (lane < (1 << DIST_BITS)) T.dist_tab[lane] = table_entry((uint32_t)lane, DIST_BITS, T.dist_cn, T.dist_sorted);
                __syncthreads();
            }
            if (type == 0u) {                                             // stored
                const uint32_t byte = (body0 + 7u) >> 3;
                if (8u * (byte + 4u) > nbits) { st = ST_HEADER; break; }
                const uint32_t len = (uint32_t)base[byte] | ((uint32_t)base[byte + 1] << 8), nlen = (uint32_t)base[byte + 2] | ((uint32_t)base[byte + 3] << 8);
                if ((len ^ 0xFFFFu) != nlen || out_pos + len > isize || 8u * (byte + 4u + len) > nbits) { st = ST_HEADER; break; }
                for (uint32_t i = lane; i < len; i += 64) dst[out_pos + i] = base[byte + 4u + i];
                out_pos += len;
                bpos = 8u * (byte + 4u + len);
                continue;
            }
#if IW_STAGE < 1
            if (__ballot(T.lit_tab[lane] == 0xFFFFu) == ~0ull) { st = ST_LANES; } else { st = ST_LANES + 1; } break;
#endif
            // ---- 2. every lane over its own piece ---------------------------------------------------------------------------------
            uint32_t piece = (nbits - body0 + 63u) / 64u;
            if (piece < MIN_PIECE) piece = MIN_PIECE;
            const uint32_t start = body0 + (uint32_t)lane * piece;
            const bool active = start < nbits;
            const uint32_t bound = start + piece;                        // the next piece begins here (if there is one)
            // (an end-of-block code does not stop the lane: on a wrong path it is a false one -- one wrong symbol in ~16 000 --, and behind
            // the true one the lanes are void anyway.  The first two are remembered with the counts in front of them.)
            uint32_t p = start, ob = 0, om = 0;
            uint32_t eob_at = NONE, eob_end = 0, eob_ob = 0, eob_om = 0, eob2_at = NONE, eob2_end = 0, eob2_ob = 0, eob2_om = 0;
            if (active) {
                while (p < bound && p < nbits) {
                    const uint32_t rel = p - start;
                    if (rel < WINDOW) note[lane][rel >> 5] |= 1u << (rel & 31u);
                    const Sym s = step(base, p, nbits, T);
                    if (s.kind == 2u) {
                        if (eob_at == NONE) { eob_at = p; eob_end = p + s.used; eob_ob = ob; eob_om = om; }
                        else if (eob2_at == NONE) { eob2_at = p; eob2_end = p + s.used; eob2_ob = ob; eob2_om = om; }
                    }
                    p += s.used;
                    if (s.kind == 0u) ob += 1u; else if (s.kind == 1u) { ob += s.a; om += 1u; }
                }
            }
            const uint32_t e_k = p;                                       // the lane stands here (behind its piece, or on its end-of-block code)
            __syncthreads();
#if IW_STAGE < 2
            if (__ballot(ob == 0xFFFFFFFFu)) { st = ST_LANES; } else { st = ST_LANES + 1; } break;
#endif
            // ---- 3. stitch: on into the neighbour's piece until a position it noted ------------------------------------------------
            uint32_t meet = NONE, xb = 0, xm = 0, x_eob_end = 0;         // meeting point, bytes / matches of the overrun
            bool x_eob = false, x_fail = false;
            const bool has_next = lane < 63 && active && start + piece < nbits;
            if (has_next) {
                uint32_t q = e_k;
                for (;;) {
                    const uint32_t rel = q - bound;
                    if (rel >= WINDOW) { x_fail = true; break; }
                    if ((note[lane + 1][rel >> 5] >> (rel & 31u)) & 1u) { meet = q; break; }
                    const Sym s = step(base, q, nbits, T);
                    if (s.kind == 3u) { x_fail = true; break; }
                    if (s.kind == 2u) { x_eob = true; x_eob_end = q + s.used; break; }
                    q += s.used;
                    if (s.kind == 0u) xb += 1u; else { xb += s.a; xm += 1u; }
                }
            }
            // what the lane before this one found is where THIS lane's true sequence begins
            uint32_t from = (uint32_t)__shfl_up((int)meet, 1, 64);
            if (lane == 0) from = body0;
            // the lane's own end-of-block code: the first one at or behind `from` (one in front of it was met on the wrong path)
            if (eob_at != NONE && from != NONE && eob_at < from) { eob_at = eob2_at; eob_end = eob2_end; eob_ob = eob2_ob; eob_om = eob2_om; }
            // the first lane that ends the block (its own end-of-block code behind `from`, or one in its overrun)
            const bool own_eob = active && eob_at != NONE && from != NONE && eob_at >= from;
            const bool ends = own_eob || x_eob;
            const unsigned long long ends_mask = __ballot(ends);
            if (ends_mask == 0ull) { st = ST_LANES; break; }              // (no end-of-block code in reach: damaged, or a bug)
            const int E = __ffsll((long long)ends_mask) - 1;
            // every lane up to E must be on the true path: it has a beginning, did not stop on a false end-of-block code in front of
            // it, and (but for E) found its neighbour
            bool bad_lane = false;
            if (lane <= E) {
                if (from == NONE) bad_lane = true;
                else if (eob_at != NONE && eob_at < from) bad_lane = true;                      // a false end of block cut the lane's decode short
                else if (lane < E && (x_fail || meet == NONE)) bad_lane = true;
            }
            const unsigned long long bad_mask = __ballot(bad_lane);
            if (bad_mask) {
                const int b = __ffsll((long long)bad_mask) - 1;
                const uint32_t why = (uint32_t)__shfl((int)(from == NONE || x_fail || meet == NONE ? ST_NO_MEETING : ST_FALSE_EOB), b, 64);
                st = why;
                break;
            }
            // what the lane decoded in front of `from` does not count: the same symbols again, counted
            uint32_t sb = 0, sm = 0;
            bool undec = false;
            if (lane >= 1 && lane <= E) {
                uint32_t q = start;
                while (q < from) {
                    const Sym s = step(base, q, nbits, T);
                    q += s.used;
                    if (s.kind == 0u) sb += 1u; else if (s.kind == 1u) { sb += s.a; sm += 1u; }
                                                                          // (an end-of-block code in front of `from` is a false one: on)
                }
                if (q != from) undec = true;                             // the lane did not pass through `from` after all
            }
            if (__ballot(undec)) { st = ST_UNDECODABLE; break; }
#if IW_STAGE < 3
            if (__ballot(sb == 0xFFFFFFFFu)) { st = ST_LANES; } else { st = ST_LANES + 1; } break;
#endif
            // ---- 4. / 5. every lane's share, its place, the second pass --------------------------------------------------------------
            uint32_t cb = 0, cm = 0, stop = 0;                           // bytes, matches, and where the lane's share ends
            if (lane <= E) {
                if (lane == E && own_eob) { cb = eob_ob - sb; cm = eob_om - sm; }          // up to its end-of-block code
                else { cb = ob - sb + xb; cm = om - sm + xm; }                              // its piece and its overrun
                stop = lane < E ? meet : (own_eob ? eob_at : x_eob_end);  // (E with x_eob: up to the end-of-block code of its overrun)
            }
            uint32_t tot_b = 0, tot_m = 0;
            const uint32_t off_b = wave_excl_sum(cb, lane, tot_b), off_m = wave_excl_sum(cm, lane, tot_m);
            if (out_pos + tot_b > isize) { st = ST_LENGTH; break; }
            if (n_match + tot_m > MATCH_CAP) { st = ST_CAPACITY; break; }
            bool w_bad = false;
            if (lane <= E) {
                uint32_t q = from, o = out_pos + off_b, mi = n_match + off_m;
                for (;;) {
                    if (lane < E && q == stop) break;
                    const Sym s = step(base, q, nbits, T);
                    if (s.kind == 2u) break;                             // lane E ends here
                    if (s.kind == 3u) { w_bad = true; break; }
                    q += s.used;
                    if (s.kind == 0u) dst[o++] = (uint8_t)s.a;
                    else { mlist[mi++] = make_uint2(o | (s.a << 17), s.b); o += s.a; }     // destination (17 bits) | length, distance
                    if (lane < E && q > stop) { w_bad = true; break; }
                }
                if (o != out_pos + off_b + cb || mi != n_match + off_m + cm) w_bad = true;
            }
            if (__ballot(w_bad)) { st = ST_UNDECODABLE; break; }
            out_pos += tot_b; n_match += tot_m;
            // the block ends behind the end-of-block code lane E met
            bpos = (uint32_t)__shfl((int)(own_eob ? eob_end : x_eob_end), E, 64);
        }
        if (st == ST_OK && out_pos != isize) st = ST_LENGTH;
        // ---- 6. the copies: 64 matches at a time, one per lane, in output order ------------------------------------------------------
        // What lies in front of a batch is final (earlier batches, literals).  Lane i depends on the earlier lanes of the batch whose
        // destination meets its source: a contiguous range of lanes, as the destinations ascend (two binary searches over the batch's
        // destinations in LDS).  A pass copies every lane whose range is done -- decided on the state in front of the pass, the lanes of
        // a pass copy side by side --; 5.96 passes per batch on the host (inflate_wave_host_check.cpp).  v1: the bytes in global memory
        // (a round trip per pass); the window of the member in LDS is the next step.
        if (st == ST_OK && IW_STAGE >= 4) {
            __shared__ uint32_t s_o[64], s_end[64];
            __threadfence_block();
            bool c_bad = false;
            for (uint32_t b0 = 0; b0 < n_match; b0 += 64) {
                const uint32_t nb = n_match - b0 < 64u ? n_match - b0 : 64u;
                const bool have = (uint32_t)lane < nb;
                const uint2 mm = have ? mlist[b0 + lane] : make_uint2(0u, 1u);
                const uint32_t o = mm.x & 0x1FFFFu, len = mm.x >> 17, d = mm.y;
                if (have && (d == 0u || d > o)) c_bad = true;
                s_o[lane] = have ? o : 0xFFFFFFFFu; s_end[lane] = have ? o + len : 0xFFFFFFFFu;
                __syncthreads();
                const uint32_t s0 = o - d, s1 = s0 + len;
                uint32_t lo = 0, hi = nb;                                  // first lane whose end > s0; first lane whose start >= s1
                for (uint32_t a = 0, b = nb; a < b;) { const uint32_t mid = (a + b) >> 1; if (s_end[mid] > s0) b = mid; else a = mid + 1; lo = a < b ? b : a; }
                for (uint32_t a = 0, b = nb; a < b;) { const uint32_t mid = (a + b) >> 1; if (s_o[mid] >= s1) b = mid; else a = mid + 1; hi = a < b ? b : a; }
                if (hi > (uint32_t)lane) hi = (uint32_t)lane;
                const unsigned long long dep = hi > lo ? (((1ull << hi) - 1ull) & ~((1ull << lo) - 1ull)) : 0ull;
                const unsigned long long all = nb == 64u ? ~0ull : ((1ull << nb) - 1ull);
                unsigned long long done = 0ull;
                if (__ballot(c_bad)) break;
                while (done != all) {
                    const bool ready = have && !((done >> lane) & 1ull) && (dep & ~done) == 0ull;
                    const unsigned long long rmask = __ballot(ready);
                    if (rmask == 0ull) { c_bad = true; break; }
                    if (ready) for (uint32_t x = 0; x < len; x++) dst[o + x] = dst[s0 + x];       // (its own bytes when d < len: in order)
                    __threadfence_block();
                    done |= rmask;
                }
                if (__ballot(c_bad)) break;
                __syncthreads();
            }
            if (__ballot(c_bad)) st = ST_UNDECODABLE;
        }
        if (lane == 0) status[m] = st;
        __syncthreads();
    }
}

// raw: readable up to 16 bytes behind its last member; matches: grid * 22016 * 8 bytes; status: n_members words
extern "C" int inflate_wave_launch(const uint8_t* d_raw, const uint64_t* d_member_pos, const uint64_t* d_out_off, uint32_t n_members,
                                   uint8_t* d_out, void* d_matches, uint32_t grid, uint32_t* d_status, hipStream_t stream)
{
    hipLaunchKernelGGL(k_inflate_wave, dim3(grid), dim3(64), 0, stream, d_raw, d_member_pos, d_out_off, n_members, d_out, (uint2*)d_matches,
                       d_status);
    return (int)hipGetLastError();
}

extern "C" uint32_t inflate_wave_match_cap(void) { return MATCH_CAP; }
