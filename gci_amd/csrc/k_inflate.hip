// k_inflate.hip -- N1 on the GPU: BGZF members inflated by the device and the BAM record walk without its serial chain
// (what pysam / htslib do for the reference at GCI.py:150-151).
//
// gci_bgzf_inflate_device.  DEFLATE is a serial bit stream: the parallelism of a BGZF file is ACROSS its members (gzip
// members of at most 64 KiB, RFC 1951 / 1952: 56 k for a chr19 40x HiFi file, millions for a genome).  A first version gave
// every member a whole wave (window and tables in LDS, every lane holding the same decoder state): correct, and bound by
// instruction issue on one lane's worth of work -- ~50 instructions per symbol, one wave per SIMD because of the 32 KiB
// window -- i.e. no faster than 32 host threads (5.9 GB/s).  This version gives every member ONE LANE:
//   * decoder state (bit buffer, positions) in registers; the lane's decode tables in LDS (8-bit primary table for
//     literals / lengths -- the block's code lengths share its space until the codes are built --, the symbols sorted by
//     code and 16 "limits" per alphabet for the longer codes and the distances: 1.3 KB per member, L members per
//     workgroup of one wave);
//   * no window: the output buffer in HBM is the window (a member's matches only reach back into its own output; a wave's
//     memory operations are issued in order, so a lane reads back what it has just written);
//   * the compressed bytes as naturally aligned 16-byte blocks, moved into the bit buffer a dword at a time; literals
//     gathered in a register and stored eight at a time; matches copied in 8-byte pieces (any address).
// Lanes diverge (literal / match, code lengths), but every wave instruction now serves L members, and a CU holds 112 of
// them (L = 8: 14 workgroups).  Stored, fixed and dynamic blocks.  Every member's output length is checked here, its CRC-32
// by k_bgzf_crc: one wave per member over the finished output, every lane the CRC register of its share, the shares
// concatenated with the GF(2) rule crc(A||B) = crc(A) * x^(8|B|) + crc(B) (as htslib verifies every block).
//
// gci_bam_record_offsets_device: the offsets of the records of an inflated BAM stream WITHOUT walking the block_size
// chain serially (137 k dependent loads of ~1 us each at chr19): every byte position is tested for "a record could
// start here" (block_size, refID, pos, l_read_name, l_seq, next_refID, next_pos consistent with the format -- SEQ / QUAL
// bytes never pass), each candidate's successor (offset + 4 + block_size) is looked up among the candidates, and the
// candidates reachable from the first record are found by pointer doubling.  The chain must end exactly at the end of
// the stream (or, for a chunk, in a record that runs past it); otherwise -- a record the strict test rejects -- the call
// reports GCI_E_MALFORMED and the caller takes the host path.
#include "gci_ctx.hpp"
#include <stdlib.h>

namespace {

#define LIT_BITS 8

__constant__ uint8_t c_clen_order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

constexpr uint32_t CRC_POLY = 0xEDB88320u;      // reflected: bit 31 holds x^0

constexpr uint32_t gf_mul_c(uint32_t a, uint32_t b)
{
    uint32_t r = 0;
    for (int i = 0; i < 32; i++) {
        r ^= (a & 0x80000000u) ? b : 0u;
        a <<= 1;
        b = (b >> 1) ^ ((b & 1u) ? CRC_POLY : 0u);
    }
    return r;
}
struct XPow8 { uint32_t v[32]; };
constexpr XPow8 make_xpow8()                       // v[k] = x^(8 * 2^k) mod P, reflected (x^8 = 0x00800000)
{
    XPow8 t{};
    uint32_t p = 0x00800000u;
    for (int k = 0; k < 32; k++) { t.v[k] = p; p = gf_mul_c(p, p); }
    return t;
}
__constant__ XPow8 c_xpow8 = make_xpow8();

__device__ __forceinline__ uint32_t gf_mul(uint32_t a, uint32_t b)   // a * b mod P over GF(2), reflected representation
{
    uint32_t r = 0;
    for (int i = 0; i < 32; i++) {
        r ^= (a & 0x80000000u) ? b : 0u;
        a <<= 1;
        b = (b >> 1) ^ ((b & 1u) ? CRC_POLY : 0u);
    }
    return r;
}

// Canonical Huffman code of one alphabet (RFC 1951 3.2.2), without a loop over the code lengths: with c = the next 15 bits
// read as a code (first bit highest), the codes of length l are exactly limit[l-1] <= c < limit[l], limit[l] = (first code
// of length l + their number) << (15 - l) -- non-decreasing in l -- so the length is 1 + the number of limits <= c (16: no
// such code), and the symbol is sorted[off[l] + (c >> (15 - l))], off[l] = index of the first symbol of length l in
// `sorted` - its code.  next[l]: one past the last symbol of length l in `sorted`.
// The literal / length alphabet also has a primary table indexed by the next LIT_BITS bits of the stream (codes are packed
// from their most significant bit, i.e. bit reversed in the LSB-first bit buffer): entry = symbol | length << 9, 0 = the
// code is longer (or unused).
struct alignas(16) Canon { uint16_t limit[16]; int16_t off[16]; uint16_t next[16]; };

// What one member's decoder keeps in LDS: 1.3 KB, which is what bounds the members a CU decodes at a time.  The code
// lengths of a block are only needed until its codes are built and share the primary table's space.
#define INF_WAVES 4
#define INF_DIST_BITS 5                // primary table of the distance codes: 5 bits (-2 % against none; 6 and 7 cost a workgroup per CU)
struct alignas(16) LaneTabs {
    union {
        uint16_t lit_tab[1 << LIT_BITS];
        uint8_t lens[352];          // [0, 19): the code-length code; [32, 32 + 286 + 30): both alphabets
    };
    uint16_t lit_sorted[288];
    uint16_t dist_sorted[32];
    uint16_t dist_tab[1 << INF_DIST_BITS];     // primary table of the distance codes (symbol | length << 9), like lit_tab
    Canon lit_cn, dist_cn;
};
static_assert(sizeof(uint16_t) << LIT_BITS >= 352, "the code lengths must fit under the primary table");

__device__ __forceinline__ uint32_t bit_reverse(uint32_t v, int n) { return __brev(v) >> (32 - n); }

// lens[0 .. n) -> limits, offsets and the symbols sorted by code; false: over-subscribed code
template <typename SortedPtr>
__device__ bool build_code(const uint8_t* lens, int n, Canon& cn, SortedPtr sorted)
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
        const uint32_t need = cnt << (15 - l);                           // sum count[l] * 2^(15 - l) must not exceed 2^15
        if (need > left) ok = false; else left -= need;
    }
    if (!ok) return false;
    for (int s = 0; s < n; s++) {
        const int l = lens[s];
        if (l) sorted[cn.next[l]++] = (uint16_t)s;
    }
    return true;
}

// the primary table of the codes of at most LIT_BITS bits, from the sorted symbols (the code lengths are gone by now)
template <int BITS, typename SortedPtr>
__device__ void build_table(const Canon& cn, SortedPtr sorted, uint16_t* table)
{
    for (int i = 0; i < (1 << BITS); i++) table[i] = 0;
    uint32_t idx = 0;
    for (int l = 1; l <= BITS; l++) {
        const uint32_t end = cn.next[l];
        const int off = cn.off[l];
        for (; idx < end; idx++) {
            const uint32_t e = (uint32_t)sorted[idx] | ((uint32_t)l << 9);
            for (uint32_t k = bit_reverse((uint32_t)((int)idx - off), l); k < (1u << BITS); k += 1u << l) table[k] = (uint16_t)e;
        }
    }
}

// The compressed bytes arrive as naturally aligned 16-byte loads, one block AHEAD of the one being consumed (the load
// issued when a block is taken up is needed 16 payload bytes later), and move into the bit buffer a dword at a time.
// 16 bytes in the global address space (a generic pointer kept in a struct becomes flat loads, which also count as LDS traffic)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) u32x4 gu4;
__device__ __forceinline__ uint4 ld_block(const gu4* p) { const u32x4 v = *p; return make_uint4(v.x, v.y, v.z, v.w); }
struct Bits {
    unsigned long long bb;      // bit buffer, next bit = bit 0
    int bn;                     // valid bits
    uint4 res;                  // the block being consumed (lowest dword first)
    int rn;                     // dwords left in res
    uint4 nxt;                  // the block after it
    const gu4* q;               // the block after nxt
    const gu4* qlim;            // the block holding the member's last byte: nothing is loaded beyond it
    uint32_t taken;             // dwords moved out of res so far
};

__device__ __forceinline__ void next_dword(Bits& B)
{
    B.res.x = B.res.y; B.res.y = B.res.z; B.res.z = B.res.w;
    B.rn--; B.taken++;
    if (B.rn == 0) {
        B.res = B.nxt; B.rn = 4;
        B.nxt = ld_block(B.q);
        B.q = B.q < B.qlim ? B.q + 1 : B.q;
    }
}

// at least 32 valid bits (bits past the payload are caught by the position check at the end of the member)
__device__ __forceinline__ void need32(Bits& B)
{
    if (B.bn < 32) {
        B.bb |= (unsigned long long)B.res.x << B.bn;
        B.bn += 32;
        next_dword(B);
    }
}

// eight bytes at any address (the target handles unaligned global accesses)
__device__ __forceinline__ unsigned long long ld8(const uint8_t* p) { unsigned long long v; __builtin_memcpy(&v, p, 8); return v; }
__device__ __forceinline__ void st8(uint8_t* p, unsigned long long v) { __builtin_memcpy(p, &v, 8); }

__device__ __forceinline__ uint32_t take(Bits& B, int n)
{
    const uint32_t v = (uint32_t)(B.bb & ((1ull << n) - 1ull));
    B.bb >>= n; B.bn -= n;
    return v;
}

// one symbol: the primary table (PRIMARY > 0), else the limits of the lengths PRIMARY + 1 .. MAXL (16-byte LDS reads)
template <int PRIMARY, int MAXL, typename SortedPtr>
__device__ __forceinline__ int decode(Bits& B, const uint16_t* table, const Canon& cn, SortedPtr sorted)
{
    if (PRIMARY) {
        const uint32_t e = table[(uint32_t)B.bb & ((1u << PRIMARY) - 1u)];
        if (e) { const int l = (int)(e >> 9); B.bb >>= l; B.bn -= l; return (int)(e & 0x1FFu); }
    }
    uint32_t w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (PRIMARY + 1 < 8) { const uint4 v = *reinterpret_cast<const uint4*>(&cn.limit[0]); w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w; }
    if (MAXL >= 8) { const uint4 v = *reinterpret_cast<const uint4*>(&cn.limit[8]); w[4] = v.x; w[5] = v.y; w[6] = v.z; w[7] = v.w; }
    const uint32_t c = __brev((uint32_t)B.bb) >> 17;
    // the length = the number of limits <= c among limit[0 .. 15] (limit[0] = 0 is the "1 +"; the limits of the lengths the primary
    // table covers are <= c for a code that is not in it, and a word that was not loaded is 0: counted as it should be).  Two limits
    // per word: with c < 2^15 and limit <= 2^15, (c + 2^15) - limit has bit 15 set iff c >= limit and never borrows from the half
    // above it -- one subtraction, one AND and one population count per pair instead of two compares and their bookkeeping.
    const uint32_t c2 = (c | (c << 16)) + 0x80008000u;
    int l = 0;
#pragma unroll
    for (int k = 0; k < (MAXL >= 8 ? 8 : 4); k++) l += __builtin_popcount((c2 - w[k]) & 0x80008000u);
    if (l > MAXL) return -1;
    const int at = (int)cn.off[l] + (int)(c >> (15 - l));
    B.bb >>= l; B.bn -= l;
    return (int)sorted[at];
}

}  // namespace

// One lane = one member; INF_LANES members per workgroup (one wave; its other lanes leave at once).
template <int INF_LANES>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(INF_WAVES, INF_WAVES)))
void k_bgzf_inflate(const uint8_t* __restrict__ raw, const uint64_t* __restrict__ member_pos,
                    const uint64_t* __restrict__ out_off, uint32_t n_members, uint8_t* __restrict__ out,
                    uint64_t out_cap, unsigned long long* __restrict__ status, const uint32_t* __restrict__ done_by_wave)
{
    __shared__ LaneTabs tabs[INF_LANES];
    const int lane = threadIdx.x;
    if (lane >= INF_LANES) return;
    const uint32_t m = blockIdx.x * INF_LANES + lane;
    if (m >= n_members) return;
    if (done_by_wave && done_by_wave[m] == 0u) return;                        // k_inflate_wave.hip decoded this one
    LaneTabs& T = tabs[lane];
    uint16_t* const lit_sorted = T.lit_sorted;
    const uint64_t pos = member_pos[m], pos_next = member_pos[m + 1];
    const uint64_t o0 = out_off[m];
    const uint32_t isize = (uint32_t)(out_off[m + 1] - o0);
    bool bad = pos_next < pos + 26 || o0 + isize > out_cap || isize > 65536u;
    uint32_t pay_len = 0, head = 0;
    Bits B;
    B.bb = 0; B.bn = 0; B.rn = 4; B.taken = 0;
    if (!bad) {
        const uint32_t xlen = (uint32_t)raw[pos + 10] | ((uint32_t)raw[pos + 11] << 8);
        const uint64_t pay0 = pos + 12 + xlen;
        bad = pay0 + 8 > pos_next || raw[pos] != 0x1f || raw[pos + 1] != 0x8b || raw[pos + 2] != 8;
        if (!bad) {
            pay_len = (uint32_t)(pos_next - 8 - pay0);
            const uintptr_t a0 = (uintptr_t)(raw + pay0);
            head = (uint32_t)(a0 & 15u);
            const gu4* base = (const gu4*)(a0 - head);
            B.qlim = (const gu4*)((uintptr_t)(raw + pos_next - 1) & ~(uintptr_t)15);
            B.res = ld_block(base);
            B.q = base < B.qlim ? base + 1 : base;
            B.nxt = ld_block(B.q);
            B.q = B.q < B.qlim ? B.q + 1 : B.q;
            for (uint32_t k = 0; k < (head >> 2); k++) next_dword(B);          // the bytes in front of the payload
            need32(B);
            B.bb >>= (head & 3u) * 8; B.bn -= (int)(head & 3u) * 8;
        }
    }
    uint8_t* __restrict__ dst = out + o0;
    uint8_t* const dst_end = dst + isize;
    // output: literals gather in a register and leave as one 8-byte store (any address).  A partial group is stored as 8
    // bytes as well where that stays inside the member's output: what lies behind `op` is overwritten before it is read.
    unsigned long long ob = 0;
    uint32_t oc = 0, op = 0;
    auto flush = [&]() __attribute__((always_inline)) {
        uint8_t* at = dst + op - oc;
        if (oc && at + 8 <= dst_end) st8(at, ob);
        else for (uint32_t i = 0; i < oc; i++) at[i] = (uint8_t)(ob >> (8 * i));
        ob = 0; oc = 0;
    };
    auto emit = [&](uint32_t byte) __attribute__((always_inline)) {
        ob |= (unsigned long long)byte << (8 * oc);
        oc++; op++;
        if (oc == 8) { st8(dst + op - 8, ob); ob = 0; oc = 0; }
    };
    for (bool last = false; !last && !bad;) {
        // every block header must lie inside the payload (a damaged stream of empty non-final blocks would never end otherwise:
        // everything else in a block is bounded by the member's output length)
        if (32ll * B.taken - 8ll * head - B.bn + 3 > 8ll * pay_len) { bad = true; break; }
        need32(B);
        last = take(B, 1) != 0;
        const uint32_t type = take(B, 2);
        if (type == 0) {                                                     // stored
            take(B, B.bn & 7);                                               // to the byte boundary
            need32(B);
            const uint32_t len = take(B, 16), nlen = take(B, 16);
            if ((len ^ 0xFFFFu) != nlen || op + len > isize) { bad = true; break; }
            for (uint32_t i = 0; i < len; i++) { need32(B); emit(take(B, 8)); }
            continue;
        }
        if (type == 3) { bad = true; break; }
        uint8_t* ll = T.lens + 32;
        if (type == 1) {                                                     // fixed code (RFC 1951 3.2.6)
            for (int i = 0; i < 288; i++) ll[i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8;
            for (int i = 0; i < 30; i++) ll[288 + i] = 5;
            if (!build_code(ll, 288, T.lit_cn, lit_sorted) || !build_code(ll + 288, 30, T.dist_cn, T.dist_sorted)) { bad = true; break; }
        } else {                                                             // dynamic code (3.2.7)
            need32(B);
            const int hlit = (int)take(B, 5) + 257, hdist = (int)take(B, 5) + 1, hclen = (int)take(B, 4) + 4;
            if (hlit > 286 || hdist > 30) { bad = true; break; }
            for (int i = 0; i < 19; i++) T.lens[i] = 0;
            for (int i = 0; i < hclen; i++) { need32(B); T.lens[c_clen_order[i]] = (uint8_t)take(B, 3); }
            // the code-length code: 19 symbols of at most 7 bits, decoded bit by bit (dist_cn / dist_sorted are free until below)
            if (!build_code(T.lens, 19, T.dist_cn, T.dist_sorted)) { bad = true; break; }
            int n = 0, prev = 0;
            while (n < hlit + hdist) {
                need32(B);
                const int sym = decode<0, 7>(B, nullptr, T.dist_cn, T.dist_sorted);
                if (sym < 0) { bad = true; break; }
                int rep = 1, val = sym;
                if (sym == 16) { if (n == 0) { bad = true; break; } val = prev; rep = 3 + (int)take(B, 2); }
                else if (sym == 17) { val = 0; rep = 3 + (int)take(B, 3); }
                else if (sym == 18) { val = 0; rep = 11 + (int)take(B, 7); }
                if (n + rep > hlit + hdist) { bad = true; break; }
                for (int i = 0; i < rep; i++) ll[n + i] = (uint8_t)val;
                n += rep; prev = val;
            }
            if (bad) break;
            if (ll[256] == 0) { bad = true; break; }                          // no end-of-block code
            if (!build_code(ll, hlit, T.lit_cn, lit_sorted) || !build_code(ll + hlit, hdist, T.dist_cn, T.dist_sorted)) { bad = true; break; }
        }
        build_table<LIT_BITS>(T.lit_cn, lit_sorted, T.lit_tab);
        build_table<INF_DIST_BITS>(T.dist_cn, T.dist_sorted, T.dist_tab);
        // ---- symbols of the block: runs of literals (the lanes of the wave meet again at their next match) -------------------
        for (;;) {
            int sym;
            for (;;) {
                need32(B);
                sym = decode<LIT_BITS, 15>(B, T.lit_tab, T.lit_cn, lit_sorted);
                if (sym < 0 || sym >= 256) break;
                if (op >= isize) { sym = -1; break; }
                emit((uint32_t)sym);
            }
            if (sym < 0 || sym > 285) { bad = true; break; }
            if (sym == 256) break;
            // length and distance codes (RFC 1951 3.2.5) by arithmetic: a table in constant memory is a vector-memory load
            // and a wait per look-up here
            const uint32_t lc = (uint32_t)sym - 257u;
            const uint32_t le = lc < 8u || lc == 28u ? 0u : (lc >> 2) - 1u;
            uint32_t len = (lc < 8u ? 3u + lc : lc == 28u ? 258u : 3u + ((4u + (lc & 3u)) << le)) + take(B, (int)le);
            need32(B);
            // (distance symbol 0 with a code of l bits is the entry l << 9: never 0, so "0 = longer code" stays unambiguous)
            const int ds = decode<INF_DIST_BITS, 15>(B, T.dist_tab, T.dist_cn, T.dist_sorted);
            if (ds < 0 || ds > 29) { bad = true; break; }
            const uint32_t de = ds < 4 ? 0u : ((uint32_t)ds >> 1) - 1u;
            const uint32_t dist = (ds < 4 ? (uint32_t)ds + 1u : 1u + ((2u + ((uint32_t)ds & 1u)) << de)) + take(B, (int)de);
            if (dist > op || op + len > isize) { bad = true; break; }
            // the copy reads this member's own output back from memory (a lane's stores and loads stay in order)
            flush();
            const uint8_t* src = dst + op - dist;
            uint8_t* to = dst + op;
            op += len;
            if (dist >= 8) {
                // up to four 8-byte pieces per round trip: as many as lie wholly in front of what the round itself writes
                while (len) {
                    const uint32_t nb = min(min(4u, dist >> 3), (len + 7u) >> 3);
                    unsigned long long v[4];
#pragma unroll
                    for (uint32_t i = 0; i < 4; i++) v[i] = i < nb ? ld8(src + 8 * i) : 0ull;
#pragma unroll
                    for (uint32_t i = 0; i < 4; i++) {
                        if (i >= nb) continue;
                        uint8_t* t = to + 8 * i;
                        if (t + 8 <= dst_end) st8(t, v[i]);
                        else for (uint32_t k = 0; 8 * i + k < len; k++) t[k] = (uint8_t)(v[i] >> (8 * k));
                    }
                    const uint32_t adv = min(len, 8u * nb);
                    src += adv; to += adv; len -= adv;
                }
            } else {
                // a short period: its bytes once, made periodic over eight bytes, stored in steps of whole periods
                unsigned long long pat = 0;
                if (op - len >= 8) pat = ld8(to - 8) >> (8 * (8 - dist));
                else {
#pragma unroll
                    for (int i = 0; i < 7; i++) pat |= (unsigned long long)((uint32_t)i < dist ? src[i] : (uint8_t)0) << (8 * i);
                }
                pat &= ~0ull >> (8 * (8 - dist));
                if (dist < 8) pat |= pat << (8 * dist);
                if (dist < 4) pat |= pat << (16 * dist);
                if (dist < 2) pat |= pat << 32;
                const uint32_t step = dist * (8u / dist);
                while (len) {
                    const uint32_t adv = min(len, step);
                    if (to + 8 <= dst_end) st8(to, pat);
                    else for (uint32_t k = 0; k < adv; k++) to[k] = (uint8_t)(pat >> (8 * k));
                    to += adv; len -= adv;
                }
            }
        }
    }
    flush();
    if (!bad && op != isize) bad = true;
    if (!bad) {                                                                // the stream may not run past the payload
        const long long bits = 32ll * B.taken - 8ll * head - B.bn;
        if (bits > 8ll * pay_len) bad = true;
    }
    if (bad) atomicMin(status, ((unsigned long long)m << 8) | (unsigned long long)(uint8_t)(-GCI_E_MALFORMED));
}

// CRC-32 of every member's output against its trailer: one wave per member, CRC_PER_WAVE members one after the other.
//
// The wave reads the member as rows of 1 KB -- lane l the 16 bytes at 16 l of every row: coalesced, where a contiguous share per lane
// made every load touch 64 cache lines for 16 bytes each (1 TB/s) -- and CRC's linearity puts the pieces together: lane l's register
// runs over ITS pieces as if the 1008 bytes between two of them were zeros, i.e. absorbing a piece and skipping to the next one is
//   reg' = sum over the 16 bytes b_j of W[15 - j][b_j]     (the first four bytes xor-ed with reg, as in slicing-by-16),
//   W[k][b] = T[k][b] * x^(8 * 1008),  T[k][b] = the register after byte b and k zero bytes;
// the lane's last piece is absorbed with T (nothing is skipped behind it), the register then multiplied by x^(8 * bytes behind
// the piece) from the powers x^(8 * 2^k), and the 64 products xor-ed.  The register starts at all ones where the member starts;
// the fewer than 16 bytes behind the last whole piece are one lane's, byte by byte, with no bytes behind them.  The 32 tables are
// made once per context (k_crc_tables) and copied into LDS by every workgroup.
#define CRC_ROW 1024u
#define CRC_PER_WAVE 8

__global__ __launch_bounds__(256) void k_crc_tables(uint32_t* __restrict__ tabs)          // tabs[0..16): T, [16..32): W
{
    __shared__ uint32_t t0[256];
    const int i = threadIdx.x;
    uint32_t c = (uint32_t)i;
    for (int k = 0; k < 8; k++) c = (c >> 1) ^ ((c & 1u) ? CRC_POLY : 0u);
    t0[i] = c;
    __syncthreads();
    uint32_t skip = 0x80000000u;                                               // x^(8 * (CRC_ROW - 16))
    for (uint32_t e = CRC_ROW - 16u, k = 0; e; e >>= 1, k++) if (e & 1u) skip = gf_mul(skip, c_xpow8.v[k]);
    for (int k = 0; k < 16; k++) {
        tabs[k * 256 + i] = c;
        tabs[(16 + k) * 256 + i] = gf_mul(c, skip);
        c = (c >> 8) ^ t0[c & 0xFFu];
    }
}

__global__ __launch_bounds__(BLOCK) void k_bgzf_crc(const uint8_t* __restrict__ raw, const uint64_t* __restrict__ member_pos,
                                                    const uint64_t* __restrict__ out_off, uint32_t n_members, const uint8_t* __restrict__ out,
                                                    const uint32_t* __restrict__ tabs, unsigned long long* __restrict__ status)
{
    __shared__ __attribute__((aligned(16))) uint32_t T[32][256];
    for (int i = threadIdx.x; i < 32 * 256 / 4; i += BLOCK)
        reinterpret_cast<uint4*>(&T[0][0])[i] = reinterpret_cast<const uint4*>(tabs)[i];
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
    for (uint32_t m = wave * CRC_PER_WAVE; m < min(n_members, (wave + 1) * CRC_PER_WAVE); m++) {
        const uint64_t o0 = out_off[m];
        const uint32_t n = (uint32_t)(out_off[m + 1] - o0);
        const uint8_t* p0 = out + o0;
        const uint32_t full = n >> 4;                                         // whole 16-byte pieces; piece j is lane j % 64's
        const uint32_t count = full > lane ? (full - lane + 63u) >> 6 : 0u;
        uint32_t c = (lane == 0 && full) ? 0xFFFFFFFFu : 0u;
        const uint8_t* q = p0 + 16u * lane;
        for (uint32_t i = 0; i < count; i++, q += CRC_ROW) {
            uint4 v;
            __builtin_memcpy(&v, q, 16);                                      // (any address: the member starts where the one before it ended)
            const uint32_t x = v.x ^ c;
            const uint32_t (*S)[256] = i + 1 < count ? &T[16] : &T[0];        // W but for the lane's last piece
            c = S[15][x & 0xFFu] ^ S[14][(x >> 8) & 0xFFu] ^ S[13][(x >> 16) & 0xFFu] ^ S[12][x >> 24]
              ^ S[11][v.y & 0xFFu] ^ S[10][(v.y >> 8) & 0xFFu] ^ S[9][(v.y >> 16) & 0xFFu] ^ S[8][v.y >> 24]
              ^ S[7][v.z & 0xFFu] ^ S[6][(v.z >> 8) & 0xFFu] ^ S[5][(v.z >> 16) & 0xFFu] ^ S[4][v.z >> 24]
              ^ S[3][v.w & 0xFFu] ^ S[2][(v.w >> 8) & 0xFFu] ^ S[1][(v.w >> 16) & 0xFFu] ^ S[0][v.w >> 24];
        }
        // the bytes behind the lane's last piece
        const uint32_t behind = count ? n - 16u * ((count - 1u) * 64u + lane + 1u) : 0u;
        uint32_t xp = 0x80000000u;                                            // x^0, reflected
        for (uint32_t e = behind, k = 0; e; e >>= 1, k++) if (e & 1u) xp = gf_mul(xp, c_xpow8.v[k]);
        c = gf_mul(c, xp);
        if (lane == 0 && (n & 15u)) {                                         // the last n % 16 bytes: nothing behind them
            uint32_t r = full ? 0u : 0xFFFFFFFFu;
            for (uint32_t k = 16u * full; k < n; k++) r = T[0][(r ^ p0[k]) & 0xFFu] ^ (r >> 8);
            c ^= r;
        }
        if (lane == 0 && n == 0) c = 0xFFFFFFFFu;
        for (int d = 1; d < 64; d <<= 1) c ^= (uint32_t)__shfl_xor((int)c, d, 64);
        if (lane == 0) {
            const uint32_t got = c ^ 0xFFFFFFFFu;
            const uint64_t tp = member_pos[m + 1] - 8;
            const uint32_t want = (uint32_t)raw[tp] | ((uint32_t)raw[tp + 1] << 8) | ((uint32_t)raw[tp + 2] << 16) | ((uint32_t)raw[tp + 3] << 24);
            if (got != want) atomicMin(status, ((unsigned long long)m << 8) | (unsigned long long)(uint8_t)(-GCI_E_MALFORMED));
        }
    }
}

int gci_inflate_wave_run(gci_ctx* ctx, const uint8_t* d_raw, const uint64_t* d_member_pos, const uint64_t* d_out_off, uint32_t n_members,
                         uint8_t* d_out, uint64_t out_cap, uint32_t* d_wstatus);          // k_inflate_wave.hip

// d_raw must be readable up to 8 bytes past its last member (the decoder loads whole aligned dwords).
extern "C" int gci_bgzf_inflate_device(gci_ctx* ctx, const uint8_t* d_raw, const uint64_t* d_member_pos, const uint64_t* d_out_off,
                                       uint32_t n_members, uint8_t* d_out, uint64_t out_cap, int check_crc, uint64_t* d_status)
{
    if (!ctx || !d_status || (n_members && (!d_raw || !d_member_pos || !d_out_off || !d_out))) return GCI_E_INVALID;
    HIPCHK(hipMemsetAsync(d_status, 0xFF, 8, ctx->stream));
    if (n_members) {
        // GCI_INFLATE=lane: every member by the lane-per-member kernel (round 2 - 4); default: a wave per member (k_inflate_wave.hip),
        // the members it hands back -- its status word != 0 -- by the lane-per-member kernel behind it
        static const bool wave = [] { const char* e = getenv("GCI_INFLATE"); return !(e && !strcmp(e, "lane")); }();
        const uint32_t* d_done = nullptr;
        ctx->inflate_last_n = 0;
        if (wave) {
            int st = gci_ensure(ctx, ctx->inflate_wstatus, (size_t)(n_members > 2048u && n_members < 262144u ? 262144u : n_members) * sizeof(uint32_t));
            if (st) return st;
            HIPCHK(hipMemsetAsync(ctx->inflate_wstatus.p, 0xFF, (size_t)n_members * sizeof(uint32_t), ctx->stream));
            st = gci_inflate_wave_run(ctx, d_raw, d_member_pos, d_out_off, n_members, d_out, out_cap, (uint32_t*)ctx->inflate_wstatus.p);
            if (st) return st;
            d_done = (const uint32_t*)ctx->inflate_wstatus.p;
            ctx->inflate_last_n = n_members;
        }
        // members per wave: fewer = more waves per SIMD to overlap the memory round trips, more = fewer instructions issued
        static const int lanes = [] { const char* e = getenv("GCI_INFLATE_LANES"); return e ? atoi(e) : 8; }();
        auto launch = [&](auto kern, int per) {
            hipLaunchKernelGGL(kern, dim3((n_members + per - 1) / per), dim3(64), 0, ctx->stream, d_raw, d_member_pos, d_out_off, n_members,
                               d_out, out_cap, (unsigned long long*)d_status, d_done);
        };
        if (lanes == 4) launch(k_bgzf_inflate<4>, 4);
        else if (lanes == 16) launch(k_bgzf_inflate<16>, 16);
        else if (lanes == 32) launch(k_bgzf_inflate<32>, 32);
        else launch(k_bgzf_inflate<8>, 8);
        LAUNCHCHK("k_bgzf_inflate");
        if (check_crc) {
            if (!ctx->crc_tabs_ready) {                                      // (ready only once the launch went through)
                const int st = gci_ensure(ctx, ctx->crc_tabs, 32 * 256 * sizeof(uint32_t));
                if (st) return st;
                hipLaunchKernelGGL(k_crc_tables, dim3(1), dim3(256), 0, ctx->stream, (uint32_t*)ctx->crc_tabs.p);
                LAUNCHCHK("k_crc_tables");
                ctx->crc_tabs_ready = true;
            }
            const uint32_t per_block = (BLOCK / 64) * CRC_PER_WAVE;
            hipLaunchKernelGGL(k_bgzf_crc, dim3((n_members + per_block - 1) / per_block), dim3(BLOCK), 0, ctx->stream, d_raw, d_member_pos,
                               d_out_off, n_members, (const uint8_t*)d_out, (const uint32_t*)ctx->crc_tabs.p, (unsigned long long*)d_status);
            LAUNCHCHK("k_bgzf_crc");
        }
    }
    return GCI_OK;
}

// How many members the device decodes AT A TIME (CUs x resident workgroups x members per workgroup).  Every member takes about
// as long as every other (~37 ms with the CUs full, whatever the launch size), so a launch is a whole number of such rounds:
// 65 536 members -- a run of 4 GiB -- are 2.3 rounds and take the time of 3.  A caller that cuts a file into runs makes them
// whole multiples of this (pipeline._bam_join_input_gpu).
extern "C" uint32_t gci_bgzf_inflate_round(gci_ctx* ctx)
{
    if (!ctx) return 0;
    int per_cu = 0, cus = 0, dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_bgzf_inflate<8>, 64, 0) != hipSuccess) return 0;
    return (uint32_t)(cus > 0 && per_cu > 0 ? cus * per_cu * 8 : 0);
}

// =====================================================================================================================
// record offsets without the serial chain
// =====================================================================================================================
namespace {

// "A BAM record could start at p": every field a well-formed record constrains (SAM spec 4.2) -- block_size >= 32, refID and
// next_refID in [-1, n_ref), pos and next_pos >= -1, l_read_name >= 1, l_seq >= 0, and the fixed part + name + CIGAR + SEQ +
// QUAL fit in block_size.
// One thread looks at 16 consecutive byte positions of the stream: their 52 bytes as four aligned 16-byte loads (`s_al` is the
// stream's address rounded down to 16 bytes, `delta` what was cut off), every field re-aligned in registers.  refID is
// tested first: SEQ / QUAL bytes and text almost never form a value in [-1, n_ref), so the full test runs for few positions.
__device__ __forceinline__ uint4 cand_load(const uint8_t* __restrict__ s_al, uint64_t q, uint64_t n_phys)
{
    if (q + 16 <= n_phys) return *reinterpret_cast<const uint4*>(s_al + q);
    uint32_t w[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    for (int k = 0; k < 16; k++)
        if (q + k < n_phys) w[k >> 2] = (w[k >> 2] & ~(0xFFu << (8 * (k & 3)))) | ((uint32_t)s_al[q + k] << (8 * (k & 3)));
    return make_uint4(w[0], w[1], w[2], w[3]);
}

__global__ __launch_bounds__(BLOCK) void k_rec_candidates(const uint8_t* __restrict__ s_al, uint32_t delta, uint64_t lo, uint64_t n, int32_t n_ref,
                                                          const uint32_t* __restrict__ tile_off, uint32_t* __restrict__ tile_count,
                                                          uint64_t* __restrict__ cand)
{
    __shared__ uint32_t wtot[BLOCK / 64];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const uint64_t n_phys = n + delta;
    const uint64_t q0 = (uint64_t)blockIdx.x * TILE + (uint64_t)t * 16;      // physical offset of this thread's first position
    uint32_t d[17];
    {
        const uint4 v0 = cand_load(s_al, q0, n_phys), v1 = cand_load(s_al, q0 + 16, n_phys), v2 = cand_load(s_al, q0 + 32, n_phys),
                    v3 = cand_load(s_al, q0 + 48, n_phys);
        const uint32_t tmp[16] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, v2.x, v2.y, v2.z, v2.w, v3.x, v3.y, v3.z, v3.w};
#pragma unroll
        for (int k = 0; k < 16; k++) d[k] = tmp[k];
        d[16] = 0xFFFFFFFFu;
    }
    uint32_t mask = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        // dword k of the record that would start at byte i of the window
#define FIELD(k) ((int32_t)__builtin_amdgcn_alignbyte(d[((i) >> 2) + (k) + 1], d[((i) >> 2) + (k)], (i) & 3))
        const int32_t ref_id = FIELD(1);
        if (ref_id < -1 || ref_id >= n_ref) continue;
        const uint64_t q = q0 + i;
        if (q < lo + delta || q + 36 > n_phys) continue;
        const int32_t block_size = FIELD(0), pos = FIELD(2), l_seq = FIELD(5), next_ref = FIELD(6), next_pos = FIELD(7);
        const uint32_t l_read_name = (uint32_t)FIELD(3) & 0xFFu, n_cigar = (uint32_t)FIELD(4) & 0xFFFFu;
#undef FIELD
        if (block_size < 32 || pos < -1 || l_read_name < 1 || l_seq < 0 || next_ref < -1 || next_ref >= n_ref || next_pos < -1) continue;
        const uint64_t need = 32ull + l_read_name + 4ull * n_cigar + (((uint64_t)(uint32_t)l_seq + 1) >> 1) + (uint64_t)(uint32_t)l_seq;
        if (need <= (uint64_t)(uint32_t)block_size) mask |= 1u << i;
    }
    const uint32_t cnt = (uint32_t)__builtin_popcount(mask);
    const uint32_t inc = wave_inclusive<uint32_t>(cnt, lane);
    if (lane == 63) wtot[wave] = inc;
    __syncthreads();
    if (!cand) {                                                             // counting pass
        if (t == 0) tile_count[blockIdx.x] = wtot[0] + wtot[1] + wtot[2] + wtot[3];
        return;
    }
    uint32_t w = tile_off[blockIdx.x] + inc - cnt;
    for (int k = 0; k < wave; k++) w += wtot[k];
    for (uint32_t m = mask; m; m &= m - 1) cand[w++] = q0 + (uint32_t)__builtin_ctz(m) - delta;
}

#define NODE_END 0xFFFFFFFEu          // the record ends exactly at the end of the stream
#define NODE_TAIL 0xFFFFFFFDu         // the record runs past the end of the stream (a chunk's partial last record)
#define NODE_HEAD 0xFFFFFFFCu         // the record is complete and fewer than 36 bytes follow it (a chunk's partial record head)
#define NODE_NONE 0xFFFFFFFFu         // its successor is not a candidate: not on the chain (or the chain is broken there)
__global__ __launch_bounds__(BLOCK) void k_rec_link(const uint8_t* __restrict__ s, uint64_t n, const uint64_t* __restrict__ cand, uint32_t nc,
                                                    uint32_t* __restrict__ next)
{
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= nc) return;
    const uint64_t p = cand[i];
    int32_t bs;
    __builtin_memcpy(&bs, s + p, 4);                                         // (a byte-wise copy: p has any alignment)
    const uint64_t q = p + 4 + (uint64_t)(uint32_t)bs;
    uint32_t r = NODE_NONE;
    if (q == n) r = NODE_END;
    else if (q > n) r = NODE_TAIL;
    else {
        uint32_t lo = i + 1, hi = nc;                                        // first candidate >= q
        while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (cand[mid] < q) lo = mid + 1; else hi = mid; }
        if (lo < nc && cand[lo] == q) r = lo;
        else if (q + 36 > n) r = NODE_HEAD;                                   // fewer than 36 bytes left: a partial record head
    }
    next[i] = r;
}

// jump_out[i] = jump_in[jump_in[i]] (terminal values stay)
__global__ __launch_bounds__(BLOCK) void k_rec_double(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint32_t nc)
{
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= nc) return;
    const uint32_t a = in[i];
    out[i] = a < nc ? in[a] : a;
}

// marked nodes mark their 2^k-th successor
__global__ __launch_bounds__(BLOCK) void k_rec_mark(const uint32_t* __restrict__ jump, uint32_t* __restrict__ mark, uint32_t nc)
{
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= nc || !mark[i]) return;
    const uint32_t a = jump[i];
    if (a < nc) mark[a] = 1u;
}

__global__ __launch_bounds__(BLOCK) void k_rec_emit(const uint8_t* __restrict__ s, const uint64_t* __restrict__ cand,
                                                    const uint32_t* __restrict__ mark, const uint32_t* __restrict__ pos,
                                                    const uint32_t* __restrict__ next, uint32_t nc, uint64_t n, uint64_t* __restrict__ offs,
                                                    uint64_t cap, uint64_t* __restrict__ result)
{
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= nc || !mark[i]) return;
    const uint32_t nx = next[i];
    // the last record of the chain decides how the walk ended: result[1] = bytes consumed, result[2] = 0 ok / 1 broken
    if (nx == NODE_TAIL) { result[1] = cand[i]; return; }                    // partial last record: not emitted
    if (pos[i] < cap) offs[pos[i]] = cand[i];
    if (nx == NODE_END) result[1] = n;
    else if (nx == NODE_HEAD) { int32_t bs; __builtin_memcpy(&bs, s + cand[i], 4); result[1] = cand[i] + 4 + (uint64_t)(uint32_t)bs; }
    else if (nx == NODE_NONE) { result[1] = cand[i]; result[2] = 1; }
}

__global__ __launch_bounds__(BLOCK) void k_rec_flags(const uint32_t* __restrict__ mark, const uint32_t* __restrict__ next, uint32_t nc,
                                                     uint32_t* __restrict__ flag)
{
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i < nc) flag[i] = (mark[i] && next[i] != NODE_TAIL) ? 1u : 0u;
}

}  // namespace

// d_result (device, 3 x uint64): [0] = number of records, [1] = bytes consumed (n_bytes when the last record ends there, else
// the offset of the partial last record), [2] = 0 ok / 1 the chain broke at offset [1] (GCI_E_MALFORMED on the host side).
// d_offs: up to `cap` offsets; [0] may exceed cap (call again with more room).  first_record: offset of the first record
// (the end of the BAM header), which must pass the strict test itself.
extern "C" int gci_bam_record_offsets_device(gci_ctx* ctx, const uint8_t* d_stream, uint64_t n_bytes, uint64_t first_record, int32_t n_ref,
                                             uint64_t* d_offs, uint64_t cap, uint64_t* d_result)
{
    if (!ctx || !d_result || (n_bytes && !d_stream) || first_record > n_bytes || (cap && !d_offs)) return GCI_E_INVALID;
    hipStream_t st = ctx->stream;
    {
        const uint64_t init[3] = {0, first_record, 0};
        GCI_TRY(gci_upload_small(ctx, d_result, init, sizeof init));
    }
    if (n_bytes - first_record < 36) return GCI_OK;                            // no complete record head: everything is tail
    // the candidate test walks the stream in tiles of TILE bytes counted from its address rounded down to 16 bytes
    const uint32_t delta = (uint32_t)((uintptr_t)d_stream & 15u);
    const uint8_t* s_al = d_stream - delta;
    const uint64_t n_tiles64 = (n_bytes + delta + TILE - 1) / TILE;
    if (n_tiles64 > 0x7fffffffULL) return GCI_E_INVALID;
    const uint32_t n_tiles = (uint32_t)n_tiles64;
    GCI_TRY(gci_ensure(ctx, ctx->part_hist, (size_t)(n_tiles + 2) * 4));
    GCI_TRY(gci_ensure(ctx, ctx->part_blk, (size_t)(n_tiles / TILE + 2) * 4));
    uint32_t* d_tile = (uint32_t*)ctx->part_hist.p;
    hipLaunchKernelGGL(k_rec_candidates, dim3(n_tiles), dim3(BLOCK), 0, st, s_al, delta, first_record, n_bytes, n_ref, (const uint32_t*)nullptr,
                       d_tile, (uint64_t*)nullptr);
    LAUNCHCHK("k_rec_candidates(count)");
    int r = device_exclusive_scan<uint32_t, uint32_t>(ctx, d_tile, d_tile, (uint32_t*)ctx->part_blk.p, (int64_t)n_tiles, true);
    if (r) return r;
    uint32_t nc = 0;
    HIPCHK(hipMemcpyAsync(&nc, d_tile + n_tiles, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (nc == 0) { const uint64_t res[3] = {0, first_record, 1}; return gci_upload_small(ctx, d_result, res, sizeof res); }
    // candidates (u64) | next, jump a, jump b, mark, flag, pos (u32 each)
    GCI_TRY(gci_ensure(ctx, ctx->part_a, (size_t)nc * 8 + 64));
    GCI_TRY(gci_ensure(ctx, ctx->part_b, (size_t)(nc + 1) * 4 * 6 + 64));
    uint64_t* d_cand = (uint64_t*)ctx->part_a.p;
    uint32_t* d_next = (uint32_t*)ctx->part_b.p;
    uint32_t* d_ja = d_next + (nc + 1);
    uint32_t* d_jb = d_ja + (nc + 1);
    uint32_t* d_mark = d_jb + (nc + 1);
    uint32_t* d_flag = d_mark + (nc + 1);
    uint32_t* d_pos = d_flag + (nc + 1);
    hipLaunchKernelGGL(k_rec_candidates, dim3(n_tiles), dim3(BLOCK), 0, st, s_al, delta, first_record, n_bytes, n_ref, (const uint32_t*)d_tile,
                       (uint32_t*)nullptr, d_cand);
    LAUNCHCHK("k_rec_candidates(write)");
    const dim3 grid((nc + BLOCK - 1) / BLOCK);
    hipLaunchKernelGGL(k_rec_link, grid, dim3(BLOCK), 0, st, d_stream, n_bytes, (const uint64_t*)d_cand, nc, d_next);
    LAUNCHCHK("k_rec_link");
    // The first record must be the first candidate.  Marks spread from it by pointer doubling: the 2^k-step successor
    // tables are built bottom up and kept on the device one after the other would need K tables; instead the marks are
    // spread bottom up as well -- after round k every node within 2^(k+1) - 1 steps of a marked node is marked, because
    // round k marks the 2^k-th successors of ALL nodes marked so far (distances 0 .. 2^k - 1 from node 0 become
    // 0 .. 2^(k+1) - 1).
    HIPCHK(hipMemsetAsync(d_mark, 0, (size_t)nc * 4, st));
    {
        uint64_t first_cand = 0;
        HIPCHK(hipMemcpyAsync(&first_cand, d_cand, 8, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        if (first_cand != first_record) { const uint64_t res[3] = {0, first_record, 1}; return gci_upload_small(ctx, d_result, res, sizeof res); }
        const uint32_t one = 1;
        GCI_TRY(gci_upload_small(ctx, d_mark, &one, 4));
    }
    HIPCHK(hipMemcpyAsync(d_ja, d_next, (size_t)nc * 4, hipMemcpyDeviceToDevice, st));
    uint32_t* jin = d_ja;
    uint32_t* jout = d_jb;
    for (uint64_t reach = 1; reach < (uint64_t)nc; reach <<= 1) {
        hipLaunchKernelGGL(k_rec_mark, grid, dim3(BLOCK), 0, st, (const uint32_t*)jin, d_mark, nc);
        hipLaunchKernelGGL(k_rec_double, grid, dim3(BLOCK), 0, st, (const uint32_t*)jin, jout, nc);
        uint32_t* t = jin; jin = jout; jout = t;
    }
    hipLaunchKernelGGL(k_rec_mark, grid, dim3(BLOCK), 0, st, (const uint32_t*)jin, d_mark, nc);
    LAUNCHCHK("k_rec_mark");
    hipLaunchKernelGGL(k_rec_flags, grid, dim3(BLOCK), 0, st, (const uint32_t*)d_mark, (const uint32_t*)d_next, nc, d_flag);
    LAUNCHCHK("k_rec_flags");
    GCI_TRY(gci_ensure(ctx, ctx->part_blk, (size_t)(nc / TILE + 2) * 4));
    r = device_exclusive_scan<uint32_t, uint32_t>(ctx, d_flag, d_pos, (uint32_t*)ctx->part_blk.p, (int64_t)nc, true);
    if (r) return r;
    hipLaunchKernelGGL(k_rec_emit, grid, dim3(BLOCK), 0, st, d_stream, (const uint64_t*)d_cand, (const uint32_t*)d_mark, (const uint32_t*)d_pos,
                       (const uint32_t*)d_next, nc, n_bytes, d_offs, cap, d_result);
    LAUNCHCHK("k_rec_emit");
    // the count: d_pos[nc] (u32) -> d_result[0] (u64)
    uint32_t n_rec = 0;
    HIPCHK(hipMemcpyAsync(&n_rec, d_pos + nc, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    const uint64_t n64 = n_rec;
    GCI_TRY(gci_upload_small(ctx, d_result, &n64, 8));
    return GCI_OK;
}
