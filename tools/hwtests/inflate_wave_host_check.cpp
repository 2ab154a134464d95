// inflate_wave_host_check.cpp -- the kernel of tools/hwtests/inflate_wave.hip played lane by lane on the host, with the very helpers
// the kernel uses (inflate_wave_core.hpp compiled for the host), every member of a BGZF file held against zlib.
//   g++ -O2 -std=c++17 -o /tmp/iw_check tools/hwtests/inflate_wave_host_check.cpp -lz && /tmp/iw_check file.bam [max members]
// The phases are the kernel's (same variables, a loop over the 64 lanes where the kernel has the wave); what it cannot show is what
// only hardware shows: the ordering of the wave's stores and loads in the copies, and speed.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#include <vector>

static inline uint32_t brev32(uint32_t v)
{
    v = ((v >> 1) & 0x55555555u) | ((v & 0x55555555u) << 1);
    v = ((v >> 2) & 0x33333333u) | ((v & 0x33333333u) << 2);
    v = ((v >> 4) & 0x0F0F0F0Fu) | ((v & 0x0F0F0F0Fu) << 4);
    v = ((v >> 8) & 0x00FF00FFu) | ((v & 0x00FF00FFu) << 8);
    return (v >> 16) | (v << 16);
}
#define IW_DEV
#define IW_INLINE static inline
#define IW_CONST static const
#define IW_BREV(x) brev32(x)
#include "inflate_wave_core.hpp"

struct Match { uint32_t o, len, d; };

// -> status (ST_*); out: isize bytes
static uint32_t member(const uint8_t* base, uint32_t nbits, uint32_t isize, std::vector<uint8_t>& dst, uint64_t stats[8])
{
    static Tabs T;
    static uint32_t note[64][WINDOW / 32];
    dst.assign(isize + 8, 0);
    std::vector<Match> mlist;
    uint32_t bpos = 0, out_pos = 0;
    for (bool last = false; !last;) {
        uint32_t p0 = bpos;
        int hlit = 0, hdist = 0;
        const int type = block_header(base, p0, nbits, T, last, hlit, hdist, true);
        if (type == 1 || type == 2) {                                              // build_code_wave(), lane by lane == build_code()
            for (int which = 0; which < 2; which++) {
                const uint8_t* lens = T.lens + 32 + (which ? hlit : 0);
                const int n = which ? hdist : hlit;
                const Canon& want = which ? T.dist_cn : T.lit_cn;
                const uint16_t* want_sorted = which ? T.dist_sorted : T.lit_sorted;
                Canon cn;
                uint16_t sorted[320] = {0};
                uint32_t cnt[16] = {0};
                for (int c = 0; c < 5; c++) for (int lane = 0; lane < 64; lane++) { const int sy = lane + 64 * c; if (sy < n) cnt[lens[sy]]++; }
                cnt[0] = 0;
                uint32_t code = 0, idx = 0, prev = 0;
                cn.limit[0] = 0; cn.off[0] = 0; cn.next[0] = 0;
                for (int l = 1; l < 16; l++) {
                    code = (code + prev) << 1;
                    cn.limit[l] = (uint16_t)((code + cnt[l]) << (15 - l)); cn.off[l] = (int16_t)((int)idx - (int)code); cn.next[l] = (uint16_t)idx;
                    idx += cnt[l]; prev = cnt[l];
                }
                int used = 0;
                for (uint32_t L = 1; L < 16; L++) {
                    uint32_t at = cn.next[L];
                    for (int c = 0; c < 5; c++) {
                        uint64_t mask = 0;
                        for (int lane = 0; lane < 64; lane++) { const int sy = lane + 64 * c; if (sy < n && lens[sy] == L) mask |= 1ull << lane; }
                        for (int lane = 0; lane < 64; lane++)
                            if ((mask >> lane) & 1ull) { sorted[at + (uint32_t)__builtin_popcountll(mask & ((1ull << lane) - 1ull))] = (uint16_t)(lane + 64 * c); used++; }
                        at += (uint32_t)__builtin_popcountll(mask);
                    }
                    cn.next[L] = (uint16_t)at;
                }
                if (memcmp(cn.limit, want.limit, sizeof cn.limit) || memcmp(cn.off, want.off, sizeof cn.off) || memcmp(cn.next, want.next, sizeof cn.next) ||
                    memcmp(sorted, want_sorted, sizeof(uint16_t) * (size_t)used)) { fprintf(stderr, "the wave's code differs from build_code()\n"); return ST_HEADER; }
            }
        }
        const uint32_t body0 = p0;
        memset(note, 0, sizeof note);
        if (type == 3) return ST_HEADER;
        if (type != 0) {                                                           // the tables entry by entry (the kernel: every lane its entries) ...
            for (uint32_t k = 0; k < (1u << LIT_BITS); k++) T.lit_tab[k] = table_entry(k, LIT_BITS, T.lit_cn, T.lit_sorted);
            for (uint32_t k = 0; k < (1u << DIST_BITS); k++) T.dist_tab[k] = table_entry(k, DIST_BITS, T.dist_cn, T.dist_sorted);
            uint16_t lt[1 << LIT_BITS], dt[1 << DIST_BITS];                        // ... == one lane's loop over the codes
            build_table(LIT_BITS, T.lit_cn, T.lit_sorted, lt);
            build_table(DIST_BITS, T.dist_cn, T.dist_sorted, dt);
            if (memcmp(lt, T.lit_tab, sizeof lt) || memcmp(dt, T.dist_tab, sizeof dt)) { fprintf(stderr, "a primary table differs from build_table()\n"); return ST_HEADER; }
        }
        if (type == 0) {
            const uint32_t byte = (body0 + 7u) >> 3;
            if (8u * (byte + 4u) > nbits) return ST_HEADER;
            const uint32_t len = (uint32_t)base[byte] | ((uint32_t)base[byte + 1] << 8), nlen = (uint32_t)base[byte + 2] | ((uint32_t)base[byte + 3] << 8);
            if ((len ^ 0xFFFFu) != nlen || out_pos + len > isize || 8u * (byte + 4u + len) > nbits) return ST_HEADER;
            memcpy(dst.data() + out_pos, base + byte + 4, len);
            out_pos += len;
            bpos = 8u * (byte + 4u + len);
            continue;
        }
        uint32_t piece = (nbits - body0 + 63u) / 64u;
        if (piece < MIN_PIECE) piece = MIN_PIECE;
        uint32_t start[64], e_k[64], ob[64], om[64], eob_at[64], eob_end[64], eob_ob[64], eob_om[64], eob2_at[64], eob2_end[64], eob2_ob[64], eob2_om[64], meet[64], xb[64], xm[64], x_eob_end[64], from[64];
        bool active[64], x_eob[64], x_fail[64];
        for (int lane = 0; lane < 64; lane++) {                                       // 2.
            start[lane] = body0 + (uint32_t)lane * piece;
            active[lane] = start[lane] < nbits;
            const uint32_t bound = start[lane] + piece;
            uint32_t p = start[lane];
            ob[lane] = om[lane] = 0; eob_at[lane] = eob2_at[lane] = NONE; eob_end[lane] = eob2_end[lane] = eob_ob[lane] = eob_om[lane] = eob2_ob[lane] = eob2_om[lane] = 0;
            if (active[lane]) {
                while (p < bound && p < nbits) {
                    const uint32_t rel = p - start[lane];
                    if (rel < WINDOW) note[lane][rel >> 5] |= 1u << (rel & 31u);
                    const Sym s = step(base, p, nbits, T);
                    if (s.kind == 2u) {
                        if (eob_at[lane] == NONE) { eob_at[lane] = p; eob_end[lane] = p + s.used; eob_ob[lane] = ob[lane]; eob_om[lane] = om[lane]; }
                        else if (eob2_at[lane] == NONE) { eob2_at[lane] = p; eob2_end[lane] = p + s.used; eob2_ob[lane] = ob[lane]; eob2_om[lane] = om[lane]; }
                    }
                    p += s.used;
                    if (s.kind == 0u) ob[lane] += 1u; else if (s.kind == 1u) { ob[lane] += s.a; om[lane] += 1u; }
                }
            }
            e_k[lane] = p;
        }
        for (int lane = 0; lane < 64; lane++) {                                       // 3.
            meet[lane] = NONE; xb[lane] = xm[lane] = 0; x_eob_end[lane] = 0; x_eob[lane] = x_fail[lane] = false;
            const uint32_t bound = start[lane] + piece;
            const bool has_next = lane < 63 && active[lane] && start[lane] + piece < nbits;
            if (has_next) {
                uint32_t q = e_k[lane];
                for (;;) {
                    const uint32_t rel = q - bound;
                    if (rel >= WINDOW) { x_fail[lane] = true; break; }
                    if ((note[lane + 1][rel >> 5] >> (rel & 31u)) & 1u) { meet[lane] = q; break; }
                    const Sym s = step(base, q, nbits, T);
                    if (s.kind == 3u) { x_fail[lane] = true; break; }
                    if (s.kind == 2u) { x_eob[lane] = true; x_eob_end[lane] = q + s.used; break; }
                    q += s.used;
                    if (s.kind == 0u) xb[lane] += 1u; else { xb[lane] += s.a; xm[lane] += 1u; }
                }
                stats[2] += e_k[lane] >= bound ? (meet[lane] != NONE ? meet[lane] - bound : 0) : 0;
                stats[3]++;
            }
        }
        int E = -1;
        bool own_eob[64];
        for (int lane = 0; lane < 64; lane++) {
            from[lane] = lane ? meet[lane - 1] : body0;
            if (eob_at[lane] != NONE && from[lane] != NONE && eob_at[lane] < from[lane]) {
                eob_at[lane] = eob2_at[lane]; eob_end[lane] = eob2_end[lane]; eob_ob[lane] = eob2_ob[lane]; eob_om[lane] = eob2_om[lane];
            }
            own_eob[lane] = active[lane] && eob_at[lane] != NONE && from[lane] != NONE && eob_at[lane] >= from[lane];
            if (E < 0 && (own_eob[lane] || x_eob[lane])) E = lane;
        }
        if (E < 0) return ST_LANES;
        for (int lane = 0; lane <= E; lane++) {
            if (from[lane] == NONE) return ST_NO_MEETING;
            if (eob_at[lane] != NONE && eob_at[lane] < from[lane]) return ST_FALSE_EOB;
            if (lane < E && (x_fail[lane] || meet[lane] == NONE)) return ST_NO_MEETING;
        }
        uint32_t sb[64] = {0}, sm[64] = {0};
        for (int lane = 1; lane <= E; lane++) {
            uint32_t q = start[lane];
            while (q < from[lane]) {
                const Sym s = step(base, q, nbits, T);
                q += s.used;
                if (s.kind == 0u) sb[lane] += 1u; else if (s.kind == 1u) { sb[lane] += s.a; sm[lane] += 1u; }
            }
            if (q != from[lane]) return ST_UNDECODABLE;
        }
        uint32_t cb[64] = {0}, cm[64] = {0}, stop[64] = {0}, off_b[64], off_m[64], tot_b = 0, tot_m = 0;
        for (int lane = 0; lane <= E; lane++) {
            if (lane == E && own_eob[lane]) { cb[lane] = eob_ob[lane] - sb[lane]; cm[lane] = eob_om[lane] - sm[lane]; }
            else { cb[lane] = ob[lane] - sb[lane] + xb[lane]; cm[lane] = om[lane] - sm[lane] + xm[lane]; }
            stop[lane] = lane < E ? meet[lane] : (own_eob[lane] ? eob_at[lane] : x_eob_end[lane]);
        }
        for (int lane = 0; lane < 64; lane++) { off_b[lane] = tot_b; off_m[lane] = tot_m; tot_b += cb[lane]; tot_m += cm[lane]; }
        if (out_pos + tot_b > isize) return ST_LENGTH;
        if (mlist.size() + tot_m > MATCH_CAP) return ST_CAPACITY;
        const uint32_t n_match = (uint32_t)mlist.size();
        mlist.resize(n_match + tot_m);
        for (int lane = 0; lane <= E; lane++) {                                       // 5.
            uint32_t q = from[lane], o = out_pos + off_b[lane], mi = n_match + off_m[lane];
            for (;;) {
                if (lane < E && q == stop[lane]) break;
                const Sym s = step(base, q, nbits, T);
                if (s.kind == 2u) break;
                if (s.kind == 3u) return ST_UNDECODABLE;
                q += s.used;
                if (s.kind == 0u) dst[o++] = (uint8_t)s.a;
                else { mlist[mi++] = {o, s.a, s.b}; o += s.a; }
                if (lane < E && q > stop[lane]) return ST_UNDECODABLE;
            }
            if (o != out_pos + off_b[lane] + cb[lane] || mi != n_match + off_m[lane] + cm[lane]) return ST_UNDECODABLE;
        }
        out_pos += tot_b;
        bpos = own_eob[E] ? eob_end[E] : x_eob_end[E];
        stats[4] += (uint64_t)(E + 1);
        stats[5]++;
    }
    if (out_pos != isize) return ST_LENGTH;
    // 6. the copies, as the NEXT version of the kernel is to make them: 64 matches at a time, one per lane, in output order.  What lies
    // in front of the batch is final (earlier batches, literals).  Lane i depends on the earlier lanes of the batch whose destination
    // meets its source -- a contiguous range of lanes, as destinations ascend --; a pass copies every lane whose range is done (decided
    // on the state in front of the pass: the lanes of a pass copy side by side), until the batch is done.
    for (size_t b0 = 0; b0 < mlist.size(); b0 += 64) {
        const int nb = (int)(mlist.size() - b0 < 64 ? mlist.size() - b0 : 64);
        uint64_t dep[64], done = 0, all = nb == 64 ? ~0ull : ((1ull << nb) - 1ull);
        for (int i = 0; i < nb; i++) {
            const Match& mi = mlist[b0 + i];
            if (mi.d == 0 || mi.d > mi.o) return ST_UNDECODABLE;
            const uint32_t s0 = mi.o - mi.d, s1 = s0 + mi.len;              // (the part of the source inside the match itself is the lane's own)
            dep[i] = 0;
            for (int j = 0; j < i; j++) {
                const Match& mj = mlist[b0 + j];
                if (mj.o < s1 && mj.o + mj.len > s0) dep[i] |= 1ull << j;
            }
            // the same as a range of lanes, by two binary searches over the batch's ascending destinations (the kernel's way)
            int lo = 0, hi = nb;
            for (int a = 0, b = nb; a < b;) { const int mid = (a + b) >> 1; if (mlist[b0 + mid].o + mlist[b0 + mid].len > s0) b = mid; else a = mid + 1; lo = b; if (a >= b) lo = a; }
            for (int a = 0, b = nb; a < b;) { const int mid = (a + b) >> 1; if (mlist[b0 + mid].o >= s1) b = mid; else a = mid + 1; hi = b; if (a >= b) hi = a; }
            if (hi > i) hi = i;
            const uint64_t range = hi > lo ? (((hi == 64 ? 0ull : (1ull << hi)) - 1ull) & ~((1ull << lo) - 1ull)) : 0ull;
            if (range != dep[i]) { fprintf(stderr, "dependency range differs from the pairwise test\n"); return ST_UNDECODABLE; }
        }
        uint32_t passes = 0;
        while (done != all) {
            uint64_t ready = 0;
            for (int i = 0; i < nb; i++) if (!((done >> i) & 1ull) && (dep[i] & ~done) == 0) ready |= 1ull << i;
            if (!ready) return ST_UNDECODABLE;
            for (int i = 0; i < nb; i++)
                if ((ready >> i) & 1ull) {
                    const Match& mm = mlist[b0 + i];
                    for (uint32_t x = 0; x < mm.len; x++) dst[mm.o + x] = dst[mm.o - mm.d + x];
                }
            done |= ready;
            passes++;
        }
        stats[7] += passes;
        stats[1]++;
    }
    stats[6] += mlist.size();
    return ST_OK;
}

int main(int argc, char** argv)
{
    if (argc < 2) { fprintf(stderr, "usage: %s file.bgzf [max members]\n", argv[0]); return 2; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 2; }
    fseek(f, 0, SEEK_END);
    const long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> raw((size_t)n + 64, 0);
    if (fread(raw.data(), 1, (size_t)n, f) != (size_t)n) return 2;
    fclose(f);
    const long max_members = argc > 2 ? atol(argv[2]) : 1L << 40;
    uint64_t stats[8] = {0};
    long members = 0, same = 0, by_status[8] = {0};
    std::vector<uint8_t> got, want;
    for (uint64_t pos = 0; pos + 26 <= (uint64_t)n && members < max_members; members++) {
        const uint8_t* h = raw.data() + pos;
        if (h[0] != 0x1F || h[1] != 0x8B) { fprintf(stderr, "not a gzip member at %llu\n", (unsigned long long)pos); return 1; }
        const uint32_t xlen = h[10] | (h[11] << 8), bsize = (h[16] | (h[17] << 8)) + 1u;
        const uint8_t* base = h + 12 + xlen;
        const uint32_t pay = bsize - 12 - xlen - 8;
        uint32_t isize;
        memcpy(&isize, h + bsize - 4, 4);
        want.assign(isize + 8, 0);
        z_stream z;
        memset(&z, 0, sizeof z);
        inflateInit2(&z, -15);
        z.next_in = (Bytef*)base; z.avail_in = pay; z.next_out = want.data(); z.avail_out = isize + 8;
        const int zr = inflate(&z, Z_FINISH);
        inflateEnd(&z);
        if (zr != Z_STREAM_END || z.total_out != isize) { fprintf(stderr, "zlib refuses member %ld\n", members); return 1; }
        const uint32_t st = member(base, 8u * pay, isize, got, stats);
        by_status[st < 8 ? st : 7]++;
        if (st == ST_OK) {
            if (memcmp(got.data(), want.data(), isize) == 0) same++;
            else { fprintf(stderr, "member %ld DIFFERS from zlib\n", members); return 1; }
        }
        pos += bsize;
    }
    printf("%ld members: %ld decoded by the wave scheme and equal to zlib byte for byte; by status: ok %ld, header %ld, no meeting point %ld, false end of block %ld, "
           "undecodable %ld, capacity %ld, length %ld, lanes %ld (anything but ok goes to the lane-per-member kernel)\n",
           members, same, by_status[0], by_status[1], by_status[2], by_status[3], by_status[4], by_status[5], by_status[6], by_status[7]);
    printf("stitches %llu, mean overrun into the neighbour's piece %.0f bits; lanes in use per block: mean %.1f; matches %llu in %llu batches of 64, "
           "%.2f passes per batch\n",
           (unsigned long long)stats[3], stats[3] ? (double)stats[2] / (double)stats[3] : 0.0, stats[5] ? (double)stats[4] / (double)stats[5] : 0.0,
           (unsigned long long)stats[6], (unsigned long long)stats[1], stats[1] ? (double)stats[7] / (double)stats[1] : 0.0);
    return 0;
}
