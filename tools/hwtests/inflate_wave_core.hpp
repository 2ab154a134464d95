// inflate_wave_core.hpp -- the decoding helpers of tools/hwtests/inflate_wave.hip, in a header of their own so that
// tools/hwtests/inflate_wave_host_check.cpp can compile them for the host (IW_DEV empty) and hold them -- and a lane-by-lane
// emulation of the kernel's stitch -- against zlib without a GPU.
#pragma once
#include <stdint.h>
#ifndef IW_DEV
#define IW_DEV __device__
#define IW_INLINE __device__ __forceinline__
#define IW_CONST __constant__
#define IW_BREV(x) __brev(x)
#endif

constexpr int LIT_BITS = 8, DIST_BITS = 5;
#ifndef IW_WINDOW
#define IW_WINDOW 1024
#endif
#ifndef IW_MIN_PIECE
#define IW_MIN_PIECE 2048
#endif
constexpr uint32_t WINDOW = IW_WINDOW, MIN_PIECE = IW_MIN_PIECE, NONE = 0xFFFFFFFFu;    // (-DIW_WINDOW=..: inflate_wave_host_check sweeps them)
constexpr uint32_t MATCH_CAP = 22016;                    // a member's matches: at most 65536 / 3

IW_CONST uint8_t c_clen_order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

struct Canon { uint16_t limit[16]; int16_t off[16]; uint16_t next[16]; };
struct Tabs {
    uint16_t lit_tab[1 << LIT_BITS];
    uint16_t dist_tab[1 << DIST_BITS];
    uint16_t lit_sorted[288];
    uint16_t dist_sorted[32];
    uint8_t lens[352];                                   // [0, 19): the code-length code; [32, 32 + 286 + 30): both alphabets
    Canon lit_cn, dist_cn;
};

enum { ST_OK = 0, ST_HEADER = 1, ST_NO_MEETING = 2, ST_FALSE_EOB = 3, ST_UNDECODABLE = 4, ST_CAPACITY = 5, ST_LENGTH = 6, ST_LANES = 7 };

IW_INLINE unsigned long long peek(const uint8_t* base, uint32_t pos)      // >= 57 bits from bit `pos`
{
    unsigned long long v;
    __builtin_memcpy(&v, base + (pos >> 3), 8);
    return v >> (pos & 7u);
}

// lens[0 .. n) -> limits, offsets and the symbols sorted by code (as k_inflate.hip); false: over-subscribed
IW_DEV bool build_code(const uint8_t* lens, int n, Canon& cn, uint16_t* sorted)
{
    for (int l = 0; l < 16; l++) cn.next[l] = 0;
    for (int i = 0; i < n; i++) cn.next[lens[i]]++;
    uint32_t code = 0, idx = 0, left = 1u << 15, prev = 0;
    bool ok = true;
    cn.limit[0] = 0; cn.off[0] = 0; cn.next[0] = 0;
    for (int l = 1; l < 16; l++) {
        const uint32_t cnt = cn.next[l];
        code = (code + prev) << 1;
        cn.limit[l] = (uint16_t)((code + cnt) << (15 - l));
        cn.off[l] = (int16_t)((int)idx - (int)code);
        cn.next[l] = (uint16_t)idx;
        idx += cnt; prev = cnt;
        const uint32_t need = cnt << (15 - l);
        if (need > left) ok = false; else left -= need;
    }
    if (!ok) return false;
    for (int s = 0; s < n; s++) {
        const int l = lens[s];
        if (l) sorted[cn.next[l]++] = (uint16_t)s;
    }
    return true;
}

IW_DEV void build_table(int bits, const Canon& cn, const uint16_t* sorted, uint16_t* table)
{
    for (int i = 0; i < (1 << bits); i++) table[i] = 0;
    uint32_t idx = 0;
    for (int l = 1; l <= bits; l++) {
        const uint32_t end = cn.next[l];
        const int off = cn.off[l];
        for (; idx < end; idx++) {
            const uint32_t e = (uint32_t)sorted[idx] | ((uint32_t)l << 9);
            for (uint32_t k = IW_BREV((uint32_t)((int)idx - off)) >> (32 - l); k < (1u << bits); k += 1u << l) table[k] = (uint16_t)e;
        }
    }
}

// Entry k of a primary table of `bits` bits WITHOUT the other entries: the canonical search on the index itself (k = the next bits of the
// stream, first bit lowest).  A code of at most `bits` bits is met iff c < limit[bits] -- exact on the padded value, as that limit has
// its low 15 - bits bits clear --, its length is the number of limits[0 .. bits) <= c.  So the 64 lanes of a wave fill a table side by
// side (four entries each for the literals), where build_table() is one lane's loop.  Same entries as build_table().
IW_INLINE uint16_t table_entry(uint32_t k, int bits, const Canon& cn, const uint16_t* sorted)
{
    const uint32_t c = IW_BREV(k) >> 17;                                   // the index as the top bits of a 15-bit code, zeros behind
    if (c >= cn.limit[bits]) return 0;
    int l = 0;
    for (int j = 0; j < bits; j++) l += cn.limit[j] <= c ? 1 : 0;
    return (uint16_t)((uint32_t)sorted[(int)cn.off[l] + (int)(c >> (15 - l))] | ((uint32_t)l << 9));
}

// one code: the primary table (primary_bits > 0), else the canonical search; -1: no such code.  len = bits of the code.
IW_INLINE int code_at(unsigned long long bits, const uint16_t* table, int primary_bits, const Canon& cn,
                                       const uint16_t* sorted, int& len)
{
    if (primary_bits) {
        const uint32_t e = table[(uint32_t)bits & ((1u << primary_bits) - 1u)];
        if (e) { len = (int)(e >> 9); return (int)(e & 0x1FFu); }
    }
    const uint32_t c = IW_BREV((uint32_t)bits) >> 17;
    int l = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) l += cn.limit[k] <= c ? 1 : 0;          // limit[0] = 0 is the "1 +"
    if (l > 15) return -1;
    len = l;
    return (int)sorted[(int)cn.off[l] + (int)(c >> (15 - l))];
}

struct Sym { uint32_t kind, a, b, used; };               // kind 0 literal (a = byte), 1 match (a = length, b = distance), 2 end of block, 3 nothing decodable

// the literal / length symbol at bit `pos` with everything that belongs to it (48 bits at most: one peek)
IW_INLINE Sym step(const uint8_t* base, uint32_t pos, uint32_t nbits, const Tabs& T)
{
    const unsigned long long bits = peek(base, pos);
    int l = 0;
    const int s = code_at(bits, T.lit_tab, LIT_BITS, T.lit_cn, T.lit_sorted, l);
    if (s < 0 || s > 285 || pos + (uint32_t)l > nbits) return {3u, 0u, 0u, 1u};
    uint32_t used = (uint32_t)l;
    if (s < 256) return {0u, (uint32_t)s, 0u, used};
    if (s == 256) return {2u, 0u, 0u, used};
    const uint32_t lc = (uint32_t)s - 257u;
    const uint32_t le = lc < 8u || lc == 28u ? 0u : (lc >> 2) - 1u;
    const uint32_t len = (lc < 8u ? 3u + lc : lc == 28u ? 258u : 3u + ((4u + (lc & 3u)) << le)) + ((uint32_t)(bits >> used) & ((1u << le) - 1u));
    used += le;
    int dl = 0;
    const int ds = code_at(bits >> used, T.dist_tab, DIST_BITS, T.dist_cn, T.dist_sorted, dl);
    if (ds < 0 || ds > 29) return {3u, 0u, 0u, 1u};
    used += (uint32_t)dl;
    const uint32_t de = ds < 4 ? 0u : ((uint32_t)ds >> 1) - 1u;
    const uint32_t dist = (ds < 4 ? (uint32_t)ds + 1u : 1u + ((2u + ((uint32_t)ds & 1u)) << de)) + ((uint32_t)(bits >> used) & ((1u << de) - 1u));
    used += de;
    if (pos + used > nbits) return {3u, 0u, 0u, 1u};
    return {1u, len, dist, used};
}

// lane 0: the header of the block at bit `pos`; -> type (0 stored, 1 fixed, 2 dynamic; 3 = damaged), pos behind the header
// codes = false: the code lengths only (T.lens + 32: hlit literal / length lengths, then hdist distance lengths); the caller builds the
// two codes (the kernel with all lanes: build_code_wave)
IW_DEV int block_header(const uint8_t* base, uint32_t& pos, uint32_t nbits, Tabs& T, bool& last, int& hlit, int& hdist, bool codes)
{
    if (pos + 3 > nbits) return 3;
    unsigned long long bits = peek(base, pos);
    last = (bits & 1u) != 0;
    const uint32_t type = (uint32_t)(bits >> 1) & 3u;
    pos += 3;
    if (type == 0) return 0;
    if (type == 3) return 3;
    uint8_t* ll = T.lens + 32;
    hlit = 288; hdist = 30;
    if (type == 1) {
        for (int i = 0; i < 288; i++) ll[i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8;
        for (int i = 0; i < 30; i++) ll[288 + i] = 5;
    } else {
        bits = peek(base, pos);
        hlit = (int)(bits & 31u) + 257; hdist = (int)((bits >> 5) & 31u) + 1;
        const int hclen = (int)((bits >> 10) & 15u) + 4;
        pos += 14;
        if (hlit > 286 || hdist > 30) return 3;
        for (int i = 0; i < 19; i++) T.lens[i] = 0;
        for (int i = 0; i < hclen; i++) { T.lens[c_clen_order[i]] = (uint8_t)(peek(base, pos) & 7u); pos += 3; }
        if (!build_code(T.lens, 19, T.dist_cn, T.dist_sorted)) return 3;
        int n = 0, prev = 0;
        while (n < hlit + hdist) {
            if (pos + 7 > nbits + 64) return 3;
            bits = peek(base, pos);
            int l = 0;
            const int sym = code_at(bits, nullptr, 0, T.dist_cn, T.dist_sorted, l);
            if (sym < 0 || sym > 18 || l > 7) return 3;
            pos += (uint32_t)l; bits >>= l;
            int rep = 1, val = sym;
            if (sym == 16) { if (n == 0) return 3; val = prev; rep = 3 + (int)(bits & 3u); pos += 2; }
            else if (sym == 17) { val = 0; rep = 3 + (int)(bits & 7u); pos += 3; }
            else if (sym == 18) { val = 0; rep = 11 + (int)(bits & 127u); pos += 7; }
            if (n + rep > hlit + hdist) return 3;
            for (int i = 0; i < rep; i++) ll[n + i] = (uint8_t)val;
            n += rep; prev = val;
        }
        if (ll[256] == 0) return 3;
    }
    if (codes && (!build_code(ll, hlit, T.lit_cn, T.lit_sorted) || !build_code(ll + hlit, hdist, T.dist_cn, T.dist_sorted))) return 3;
    // (the two primary tables are filled by the caller: entry by entry, table_entry(), by all the lanes of the wave)
    return (int)type;
}

