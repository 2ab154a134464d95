// k_inflate_wave.hip -- N1 on the GPU, second generation: a WAVE per BGZF member instead of a lane per member
// (what pysam / htslib do for the reference at GCI.py:150-151; k_inflate.hip keeps the lane-per-member decoder as the
// fall-back for the members this one hands back, and the CRC-32 check over the finished output).
//
// DEFLATE is a serial bit stream, but a Huffman decoder started at a wrong bit offset falls into step with the true symbol
// sequence after a few symbols (measured on HiFi BAM members: 64 bits in the median, 413 at the 99th percentile --
// profiles/r04_inflate_resync.txt).  So the body of a block is cut into pieces of PIECE bits, one per lane:
//
//   k_inflate_symbols (kernel A; one wave = one workgroup per member, 20 - 28 KB of LDS each):
//     per block   the header's code lengths by lane 0 out of LDS, the two canonical codes and their primary tables by the
//                 whole wave (ballots per code length; every lane its table entries);
//     per chunk   64 pieces of the payload staged in LDS TRANSPOSED (8-byte unit u of piece k at [u][k]: whatever unit a lane
//                 stands on, lane k reads bank pair k -- no conflicts although the lanes advance at their own pace);
//       pass 1    lane k decodes from the start of piece k as if a symbol began there, counts symbols and output bytes and
//                 notes the symbol starts it visits in the first WINDOW bits as a bitmap;
//       stitch    lane k runs on into piece k + 1 until it stands on a position lane k + 1 noted: from there lane k + 1's
//                 sequence is the true one.  Lane k + 1 re-reads its first symbols up to there to know what to leave out.
//                 (Same position + same tables = same sequence: a stitched result is exact, not probable.)
//       pass 2    prefix sums over the lanes give every lane its place; it decodes its true range again and writes one
//                 32-bit entry per symbol -- literal byte, or (length, distance) -- to the member's SYMBOL STREAM in HBM.
//     A lane's piece does not stop at an end-of-block code (on a wrong path it is a false one); the first lane whose TRUE range
//     holds one ends the block, the lanes behind it are void, the next block starts a new chunk behind it.
//   k_inflate_copy (kernel B; one workgroup of 16 waves per member, the member's output as 65 536 16-bit cells in LDS):
//     place     a scan over the symbol stream gives every symbol its output offset; a literal's cell is its byte, a match's cells are
//               POINTERS (0x8000 | distance - 1) to the cell they copy;
//     resolve   pointer jumping: a cell whose target is a byte takes it, a cell whose target is a pointer adds the two distances.
//               No barriers: every value a wave can observe in another wave's cell is either its final byte or a pointer further
//               along the same chain.  A wave owns 4096 cells and iterates over its 64-cell segments that still hold pointers;
//     write     the wave's 4096 bytes leave as 16-byte stores.
//   The copies are the serial part of an inflate (12 800 matches per member of a HiFi BAM, mean length 5): resolved cell by cell
//   they are log2(chain depth) rounds of LDS traffic and no round trips to memory.
//
// A member that does not stitch (no meeting point within WINDOW bits, end-of-block codes on wrong paths, a header the quick
// parser rejects) is marked in its status word and left to k_bgzf_inflate (0.2 % of the members of a HiFi BAM).
#include "gci_ctx.hpp"
#include <stdlib.h>

#ifndef IW_PIECE_LOG2
#define IW_PIECE_LOG2 10                 // bits per lane and chunk: 1024 (8 KB of payload per wave in LDS)
#endif
#ifndef IW_WINDOW
#define IW_WINDOW 1024                   // bits of a piece whose symbol starts are noted (a lane that finds no meeting point there takes the piece over)
#endif
#ifndef IW_LIT_BITS
#define IW_LIT_BITS 9
#endif
#ifndef IW_DIST_BITS
#define IW_DIST_BITS 8
#endif

namespace iw {

constexpr uint32_t PIECE = 1u << IW_PIECE_LOG2;                 // bits
constexpr uint32_t PU_LOG2 = IW_PIECE_LOG2 - 6, PU = 1u << PU_LOG2;   // 8-byte units per piece
constexpr uint32_t CHUNK_U = 64u * PU, TAIL_U = 4;              // units per chunk; units kept behind it (a symbol that begins in the chunk ends there)
constexpr uint32_t WINDOW = IW_WINDOW, NW = WINDOW / 32;
static_assert(WINDOW <= PIECE, "a window is part of its piece");
constexpr int LIT_BITS = IW_LIT_BITS, DIST_BITS = IW_DIST_BITS;
// codes longer than the primary table's index: their 15-bit values (first bit highest) lie at the top of the code space, from
// limit[BITS] on -- a direct table over that range.  A HiFi BAM's blocks need ~300 / ~130 entries; beyond the table: the search.
constexpr uint32_t LIT_TAIL = 1024, DIST_TAIL = 256;
constexpr uint32_t HDR_U = CHUNK_U < 128u ? CHUNK_U : 128u;    // units staged for a block's header (8192 bits; the longest header has 4498)
constexpr uint32_t NONE = 0xFFFFFFFFu;
constexpr uint32_t SYM_STRIDE = 65536;                          // entries of symbol stream per member (one per output byte at most)

enum { ST_OK = 0, ST_HEADER = 1, ST_NO_MEETING = 2, ST_FALSE_EOB = 3, ST_UNDECODABLE = 4, ST_LENGTH = 5, ST_LANES = 6 };

__constant__ uint8_t c_clen_order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

// (k_inflate.hip explains limit / off / next: the canonical code without a loop over the lengths)
struct alignas(16) Canon { uint16_t limit[16]; int16_t off[16]; uint16_t next[16]; };

struct alignas(16) Lds {
    unsigned long long pay[CHUNK_U + TAIL_U];      // the chunk, transposed: unit u of piece k at [(u << 6) | k]; the tail linear behind it
    uint32_t note[64 * NW];                        // word w of lane k's bitmap at [(w << 6) | k]
    uint16_t lit_tab[1 << LIT_BITS];               // symbol | length << 9; 0 = a longer code
    uint16_t lit_tail[LIT_TAIL];                   // ... of the 15-bit code value c: [c - limit[LIT_BITS]]; 0 = no such code
    uint16_t dist_tab[1 << DIST_BITS];
    uint16_t dist_tail[DIST_TAIL];
    uint16_t lit_sorted[288];
    uint16_t dist_sorted[32];
    uint16_t cl_tab[128];                          // the code-length code (at most 7 bits)
    Canon lit_cn, dist_cn;
    uint8_t lens[352];                             // [0, 19): the code-length code; [32, 32 + 286 + 30): both alphabets
    uint32_t hdr[8];
};

__device__ __forceinline__ uint32_t pay_index(uint32_t U) { return U < CHUNK_U ? ((U & (PU - 1u)) << 6) | (U >> PU_LOG2) : (U < CHUNK_U + TAIL_U ? U : CHUNK_U + TAIL_U - 1u); }

// 64 bits from bit p of the chunk
__device__ __forceinline__ unsigned long long peek(const Lds& S, uint32_t p)
{
    const uint32_t U = p >> 6, s = p & 63u;
    const unsigned long long lo = S.pay[pay_index(U)], hi = S.pay[pay_index(U + 1u)];
    return s ? (lo >> s) | (hi << (64u - s)) : lo;
}

// A lane's place in the chunk: the unit it stands in and the two behind it in registers -- the bits of a symbol come out of
// them without a look into LDS, and the unit fetched when the lane moves on is needed a whole unit later.
struct Reader {
    uint32_t p, U;
    unsigned long long cur, nxt, nx2;
    __device__ __forceinline__ void seek(const Lds& S, uint32_t pos)
    {
        p = pos; U = pos >> 6;
        cur = S.pay[pay_index(U)]; nxt = S.pay[pay_index(U + 1u)]; nx2 = S.pay[pay_index(U + 2u)];
    }
    __device__ __forceinline__ unsigned long long bits() const
    {
        const uint32_t s = p & 63u;
        return s ? (cur >> s) | (nxt << (64u - s)) : cur;
    }
    __device__ __forceinline__ void advance(const Lds& S, uint32_t used)      // used < 64
    {
        p += used;
        if ((p >> 6) != U) { U++; cur = nxt; nxt = nx2; nx2 = S.pay[pay_index(U + 2u)]; }
    }
};

// the search over the limits of all lengths (two per word: k_inflate.hip decode()); -1: no such code
__device__ __forceinline__ int code_search(uint32_t bits, const Canon& cn, const uint16_t* sorted, int& len)
{
    const uint32_t c = __brev(bits) >> 17;
    const uint4 a = *reinterpret_cast<const uint4*>(&cn.limit[0]), b = *reinterpret_cast<const uint4*>(&cn.limit[8]);
    const uint32_t c2 = (c | (c << 16)) + 0x80008000u;
    const int l = __builtin_popcount((c2 - a.x) & 0x80008000u) + __builtin_popcount((c2 - a.y) & 0x80008000u)
                + __builtin_popcount((c2 - a.z) & 0x80008000u) + __builtin_popcount((c2 - a.w) & 0x80008000u)
                + __builtin_popcount((c2 - b.x) & 0x80008000u) + __builtin_popcount((c2 - b.y) & 0x80008000u)
                + __builtin_popcount((c2 - b.z) & 0x80008000u) + __builtin_popcount((c2 - b.w) & 0x80008000u);
    if (l > 15) return -1;
    len = l;
    return (int)sorted[(int)cn.off[l] + (int)(c >> (15 - l))];
}

// what the wave keeps in registers of a block's two codes (uniform)
struct Codes { uint32_t lit_lim, dist_lim; };

// entry (symbol | length << 9, 0 = none) of the code the bits begin with: primary table, tail table, search
template <int BITS, uint32_t TAIL>
__device__ __forceinline__ uint32_t code_entry(uint32_t b, const uint16_t* tab, const uint16_t* tail, uint32_t lim, const Canon& cn, const uint16_t* sorted)
{
    uint32_t e = tab[b & ((1u << BITS) - 1u)];
    if (e == 0u) {
        const uint32_t t = (__brev(b) >> 17) - lim;                           // (lim <= the code value of anything the primary table does not hold)
        if (t < TAIL) e = tail[t];
        else {
            int l = 0;
            const int s = code_search(b, cn, sorted, l);
            e = s < 0 ? 0u : (uint32_t)s | ((uint32_t)l << 9);
        }
    }
    return e;
}

// kind: 0 literal (a = byte), 1 match (a = length, b = distance), 2 end of block, 3 nothing decodable (used = 1: a wrong path moves on)
struct Sym { uint32_t kind, a, b, used; };

// the literal / length symbol the reader stands on with everything that belongs to it (48 bits at most); nb = end of the stream
__device__ __forceinline__ Sym step(const Lds& S, const Codes& C, const Reader& R, uint32_t nb)
{
    const unsigned long long bits = R.bits();
    const uint32_t b = (uint32_t)bits;
    const uint32_t e = code_entry<LIT_BITS, LIT_TAIL>(b, S.lit_tab, S.lit_tail, C.lit_lim, S.lit_cn, S.lit_sorted);
    const uint32_t l = e >> 9, s = e & 0x1FFu;
    // the length code's base and extra bits by arithmetic (RFC 1951 3.2.5); for a literal the values are not used
    const uint32_t lc = s - 257u;
    const uint32_t le = lc < 8u || lc >= 28u ? 0u : (lc >> 2) - 1u;
    const uint32_t len = (lc < 8u ? 3u + lc : lc == 28u ? 258u : 3u + ((4u + (lc & 3u)) << le)) + ((b >> l) & ((1u << le) - 1u));
    const uint32_t used1 = l + le;
    const uint32_t b2 = (uint32_t)(bits >> used1);
    const uint32_t e2 = code_entry<DIST_BITS, DIST_TAIL>(b2, S.dist_tab, S.dist_tail, C.dist_lim, S.dist_cn, S.dist_sorted);
    const uint32_t dl = e2 >> 9, ds = e2 & 0x1FFu;
    const uint32_t de = ds < 4u ? 0u : (ds >> 1) - 1u;
    const uint32_t dist = (ds < 4u ? ds + 1u : 1u + ((2u + (ds & 1u)) << de)) + ((b2 >> dl) & ((1u << de) - 1u));
    Sym r;
    const bool is_match = s > 256u;
    r.used = is_match ? used1 + dl + de : l;
    r.kind = is_match ? 1u : s == 256u ? 2u : 0u;
    r.a = is_match ? len : s;
    r.b = dist;
    if (e == 0u || s > 285u || (is_match && (e2 == 0u || ds > 29u)) || R.p + r.used > nb) { r.kind = 3u; r.used = 1u; }
    return r;
}

// lens[0 .. n) -> the canonical code, by one lane (the 19 symbols of the code-length code)
__device__ bool build_code(const uint8_t* lens, int n, Canon& cn, uint16_t* sorted)
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

// ... and by the whole wave: the counts per length and every symbol's place among those of its length by ballots (lane s + 64 c
// holds symbol s of chunk c), the fifteen-step prefix over the lengths by lane 0.  Same Canon, same `sorted`.
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
        if (lane == 0) cn.next[L] = (uint16_t)at;
    }
    __syncthreads();
    return *s_ok != 0u;
}

// the entry of the 15-bit code value c (first bit highest, zeros behind a shorter code): the canonical search on the value itself
__device__ __forceinline__ uint16_t entry_of_value(uint32_t c, const Canon& cn, const uint16_t* sorted)
{
    int l = 0;
#pragma unroll
    for (int j = 0; j < 16; j++) l += cn.limit[j] <= c ? 1 : 0;
    if (l > 15) return 0;
    return (uint16_t)((uint32_t)sorted[(int)cn.off[l] + (int)(c >> (15 - l))] | ((uint32_t)l << 9));
}
// entry k of a primary table of `bits` bits (k = the next bits of the stream, first bit lowest): 0 when the code is longer
__device__ __forceinline__ uint16_t table_entry(uint32_t k, int bits, const Canon& cn, const uint16_t* sorted)
{
    const uint32_t c = __brev(k) >> 17;
    if (c >= cn.limit[bits]) return 0;
    return entry_of_value(c, cn, sorted);
}

// the primary and the tail table of one code, every lane its entries
__device__ __forceinline__ void fill_tables(uint16_t* tab, int bits, uint16_t* tail, uint32_t n_tail, const Canon& cn, const uint16_t* sorted, int lane)
{
    for (uint32_t k = lane; k < (1u << bits); k += 64) tab[k] = table_entry(k, bits, cn, sorted);
    const uint32_t lim = cn.limit[bits];
    const uint32_t n = 32768u - lim < n_tail ? 32768u - lim : n_tail;
    for (uint32_t t = lane; t < n; t += 64) tail[t] = entry_of_value(lim + t, cn, sorted);
}

// lane 0: the code lengths of a dynamic block behind its 14 + 3 hclen bits, through the 7-bit table of the code-length code
__device__ bool read_code_lengths(Lds& S, uint32_t& p, int total)
{
    uint8_t* ll = S.lens + 32;
    int n = 0, prev = 0;
    unsigned long long bb = 0;
    int bn = 0;
    while (n < total) {
        if (bn < 14) {
            if (p + 128u > 64u * HDR_U) return false;               // (a header longer than what is staged for it: not ours)
            bb = peek(S, p); bn = 64;
        }
        const uint32_t e = S.cl_tab[(uint32_t)bb & 127u];
        if (e == 0u) return false;
        const int l = (int)(e >> 9), sym = (int)(e & 31u);
        bb >>= l; bn -= l; p += (uint32_t)l;
        if (sym < 16) { ll[n++] = (uint8_t)sym; prev = sym; continue; }
        int rep, val = 0, xb;
        if (sym == 16) { if (n == 0) return false; val = prev; rep = 3 + (int)(bb & 3u); xb = 2; }
        else if (sym == 17) { rep = 3 + (int)(bb & 7u); xb = 3; }
        else { rep = 11 + (int)(bb & 127u); xb = 7; }
        bb >>= xb; bn -= xb; p += (uint32_t)xb;
        if (n + rep > total) return false;
        for (int i = 0; i < rep; i++) ll[n + i] = (uint8_t)val;
        n += rep; prev = val;
    }
    return true;
}

__device__ __forceinline__ uint32_t wave_excl_sum(uint32_t v, int lane, uint32_t& total)
{
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)inc, d, 64); if (lane >= d) inc += o; }
    total = (uint32_t)__shfl((int)inc, 63, 64);
    return inc - v;
}

// units [U0, U0 + n) of the member's payload into S.pay (n <= CHUNK_U + TAIL_U); units from U_end on read as zero
__device__ __forceinline__ void stage(Lds& S, const unsigned long long* __restrict__ g8, uint32_t U0, uint32_t U_end, uint32_t n, int lane)
{
#pragma unroll 4
    for (uint32_t i = (uint32_t)lane; i < n; i += 64) {
        const uint32_t U = U0 + i;
        S.pay[pay_index(i)] = U < U_end ? g8[U] : 0ull;
    }
}

}  // namespace iw

using namespace iw;

// ---- kernel A ------------------------------------------------------------------------------------------------------------
// members [m0, m0 + n_batch) of the run; sym: n_batch x SYM_STRIDE entries; n_sym, wstatus: one word per member of the RUN
// entry: bit 31 set = literal (low 8 bits); else (length - 3) << 15 | (distance - 1)
extern "C" __global__ __launch_bounds__(64) void k_inflate_symbols(const uint8_t* __restrict__ raw, const uint64_t* __restrict__ member_pos,
                                                                   const uint64_t* __restrict__ out_off, uint64_t out_cap, uint32_t m0, uint32_t n_batch,
                                                                   uint32_t* __restrict__ sym, uint32_t* __restrict__ n_sym,
                                                                   uint32_t* __restrict__ wstatus, uint32_t cut)
{
    // (cut: measurements only -- 1 leaves a block behind its tables, 2 behind pass 1, 3 behind the stitch and the recount; 0 = the kernel)
    __shared__ Lds S;
    const int lane = threadIdx.x;
    for (uint32_t mb = blockIdx.x; mb < n_batch; mb += gridDim.x) {
        const uint32_t m = m0 + mb;
        const uint64_t pos0 = member_pos[m], pos1 = member_pos[m + 1];
        const uint32_t isize = (uint32_t)(out_off[m + 1] - out_off[m]);
        uint32_t* const msym = sym + (size_t)mb * SYM_STRIDE;
        uint32_t st = ST_OK;
        __syncthreads();
        if (pos1 < pos0 + 26 || isize > 65536u || out_off[m + 1] > out_cap || raw[pos0] != 0x1f || raw[pos0 + 1] != 0x8b || raw[pos0 + 2] != 8) {
            if (lane == 0) { wstatus[m] = ST_HEADER; n_sym[m] = 0; }
            continue;
        }
        const uint32_t xlen = (uint32_t)raw[pos0 + 10] | ((uint32_t)raw[pos0 + 11] << 8);
        if (pos0 + 12 + xlen + 8 > pos1) { if (lane == 0) { wstatus[m] = ST_HEADER; n_sym[m] = 0; } continue; }
        const uintptr_t a0 = (uintptr_t)(raw + pos0 + 12 + xlen);
        const unsigned long long* const g8 = (const unsigned long long*)(a0 & ~(uintptr_t)7);
        const uint32_t bias = 8u * (uint32_t)(a0 & 7u);
        const uint32_t nbits = bias + 8u * (uint32_t)(pos1 - 8 - (pos0 + 12 + xlen));          // end of the DEFLATE stream (bits from g8)
        const uint32_t U_end = (nbits + 63u) >> 6;
        uint32_t tpos = bias, out_pos = 0, ns = 0;                          // (uniform over the wave)
        for (bool last = false; !last && st == ST_OK;) {
            // ---- the block's header: 3 + 14 + 57 bits and the code lengths -- 128 units hold any header this kernel takes -----------
            uint32_t cbU = tpos >> 6;
            __syncthreads();
            stage(S, g8, cbU, U_end, HDR_U, lane);
            __syncthreads();
            uint32_t type, body0 = 0;
            int hlit = 288, hdist = 30;
            {
                uint32_t p = tpos - 64u * cbU;
                const uint32_t nb = nbits - 64u * cbU;
                const unsigned long long h = peek(S, p);                  // (every lane reads the same bits)
                last = (h & 1u) != 0;
                type = p + 3u > nb ? 3u : (uint32_t)(h >> 1) & 3u;
                p += 3;
                if (type == 1u) {
                    for (int i = lane; i < 288; i += 64) S.lens[32 + i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8;
                    if (lane < 30) S.lens[32 + 288 + lane] = 5;
                } else if (type == 2u) {
                    hlit = (int)((h >> 3) & 31u) + 257; hdist = (int)((h >> 8) & 31u) + 1;
                    const int hclen = (int)((h >> 13) & 15u) + 4;
                    p += 14;
                    if (hlit > 286 || hdist > 30) type = 3u;
                    else {
                        const unsigned long long c3 = peek(S, p);         // 19 x 3 = 57 bits at most
                        if (lane < 19) S.lens[lane] = 0;
                        __syncthreads();
                        if (lane < hclen) S.lens[c_clen_order[lane]] = (uint8_t)((c3 >> (3 * lane)) & 7u);
                        p += 3u * (uint32_t)hclen;
                        __syncthreads();
                        if (lane == 0) S.hdr[4] = build_code(S.lens, 19, S.dist_cn, S.dist_sorted) ? 1u : 0u;
                        __syncthreads();
                        if (S.hdr[4] == 0u) type = 3u;
                        else {
                            for (uint32_t k = lane; k < 128u; k += 64) S.cl_tab[k] = table_entry(k, 7, S.dist_cn, S.dist_sorted);
                            __syncthreads();
                            if (lane == 0) {
                                uint32_t q = p;
                                const bool ok = read_code_lengths(S, q, hlit + hdist);
                                S.hdr[0] = ok && q <= nb && S.lens[32 + 256] != 0 ? 1u : 0u; S.hdr[1] = q;
                            }
                            __syncthreads();
                            if (S.hdr[0] == 0u) type = 3u;
                            p = S.hdr[1];
                        }
                    }
                }
                body0 = p + 64u * cbU;
            }
            if (type == 3u) { st = ST_HEADER; break; }
            if (type == 0u) {                                             // stored: its bytes as literals
                const uint8_t* const gb = (const uint8_t*)g8;
                const uint32_t byte = (body0 + 7u) >> 3;
                if (8u * (byte + 4u) > nbits) { st = ST_HEADER; break; }
                const uint32_t len = (uint32_t)gb[byte] | ((uint32_t)gb[byte + 1] << 8), nlen = (uint32_t)gb[byte + 2] | ((uint32_t)gb[byte + 3] << 8);
                if ((len ^ 0xFFFFu) != nlen || out_pos + len > isize || 8u * (byte + 4u + len) > nbits) { st = ST_HEADER; break; }
                for (uint32_t i = lane; i < len; i += 64) msym[ns + i] = 0x80000000u | (uint32_t)gb[byte + 4u + i];
                out_pos += len; ns += len;
                tpos = 8u * (byte + 4u + len);
                continue;
            }
            Codes C;
            {
                __syncthreads();
                const bool ok_l = build_code_wave(S.lens + 32, hlit, S.lit_cn, S.lit_sorted, lane, &S.hdr[4]);
                const bool ok_d = build_code_wave(S.lens + 32 + hlit, hdist, S.dist_cn, S.dist_sorted, lane, &S.hdr[4]);
                if (!ok_l || !ok_d) { st = ST_HEADER; break; }
                fill_tables(S.lit_tab, LIT_BITS, S.lit_tail, LIT_TAIL, S.lit_cn, S.lit_sorted, lane);
                fill_tables(S.dist_tab, DIST_BITS, S.dist_tail, DIST_TAIL, S.dist_cn, S.dist_sorted, lane);
                C.lit_lim = S.lit_cn.limit[LIT_BITS]; C.dist_lim = S.dist_cn.limit[DIST_BITS];
            }
            if (cut == 1u) { st = ST_LANES; break; }
            // ---- the block's body, chunk by chunk ----------------------------------------------------------------------------
            uint32_t cpos = body0;                                        // a symbol begins here
            for (bool block_done = false; !block_done && st == ST_OK;) {
                cbU = cpos >> 6;
                __syncthreads();
                stage(S, g8, cbU, U_end, CHUNK_U + TAIL_U, lane);
                for (uint32_t i = lane; i < 64u * NW; i += 64) S.note[i] = 0u;
                __syncthreads();
                const uint32_t nb = nbits - 64u * cbU;                    // end of the stream, in bits of the chunk
                // ---- pass 1: every lane over its own piece -----------------------------------------------------------------
                const uint32_t start = lane == 0 ? cpos - 64u * cbU : (uint32_t)lane * PIECE;
                const uint32_t bound = ((uint32_t)lane + 1u) * PIECE;
                const bool active = start < nb;
                uint32_t ob = 0, os = 0;
                // end-of-block codes the lane passes: on a wrong path they are false ones (a fixed code has one in 128 symbols), so a
                // lane does not stop at them.  Those in the window (where the lane's true path may begin) as a list of positions, the
                // first one behind the window -- on the true path if the lane is on it at all -- with the counts in front of it.
                uint32_t ew0 = NONE, ew1 = NONE, ew2 = NONE, ew3 = NONE, n_ew = 0;
                uint32_t eob_at = NONE, eob_end = 0, eob_ob = 0, eob_os = 0;
                Reader R;
                R.seek(S, active ? start : 0u);
                if (active) {
                    const uint32_t pend = bound < nb ? bound : nb;
                    while (R.p < pend) {
                        const uint32_t rel = R.p - (uint32_t)lane * PIECE;
                        if (rel < WINDOW) atomicOr(&S.note[((rel >> 5) << 6) | (uint32_t)lane], 1u << (rel & 31u));
                        const Sym s = step(S, C, R, nb);
                        if (s.kind == 2u) {
                            if (rel < WINDOW) {
                                if (n_ew == 0u) ew0 = R.p; else if (n_ew == 1u) ew1 = R.p; else if (n_ew == 2u) ew2 = R.p; else if (n_ew == 3u) ew3 = R.p;
                                n_ew++;
                            } else if (eob_at == NONE) { eob_at = R.p; eob_end = R.p + s.used; eob_ob = ob; eob_os = os; }
                        }
                        R.advance(S, s.used);
                        ob += s.kind == 0u ? 1u : s.kind == 1u ? s.a : 0u;
                        os += s.kind < 2u ? 1u : 0u;
                    }
                }
                __syncthreads();
                if (cut == 2u) { st = __ballot(ob == NONE) ? ST_LANES : ST_LENGTH; break; }
                // ---- stitch: on behind the piece until a position the lane of THAT piece noted ---------------------------------
                // (normally within a few symbols in the neighbour's window; a lane that finds none there takes the neighbour's piece
                // over -- decodes it to its end -- and looks in the window of the piece after it: the neighbour is void then)
                uint32_t meet = NONE, meet_lane = 64u, xb = 0, xs = 0, x_eob_at = 0, x_eob_end = 0, x_end = 0;
                bool x_eob = false, x_fail = false;
                if (active) {
                    for (;;) {
                        const uint32_t q = R.p;
                        if (q >= 64u * PIECE) { x_end = q; break; }                        // the end of the chunk: the block goes on behind it
                        if (q >= nb) { x_fail = true; break; }                              // the stream ends without an end-of-block code
                        const uint32_t j = q >> IW_PIECE_LOG2, rel = q & (PIECE - 1u);
                        if (rel < WINDOW && ((S.note[((rel >> 5) << 6) | j] >> (rel & 31u)) & 1u)) { meet = q; meet_lane = j; break; }
                        const Sym s = step(S, C, R, nb);
                        if (s.kind == 3u) { x_fail = true; break; }
                        if (s.kind == 2u) { x_eob = true; x_eob_at = q; x_eob_end = q + s.used; break; }
                        R.advance(S, s.used);
                        xs += 1u;
                        xb += s.kind == 0u ? 1u : s.a;
                    }
                }
                // ---- the chain of lanes on the true path: lane 0, the lane it met, ... (uniform; a step per live lane) ------------
                uint32_t from = NONE;
                bool live = false, own_eob = false, block_ends = false, chain_bad = false;
                uint32_t own_at = NONE;                                                      // the lane's own end-of-block code, if it ends the block
                int E = 0;
                {
                    uint32_t cur = 0, from_cur = cpos - 64u * cbU;
                    for (;;) {
                        if ((uint32_t)lane == cur) {
                            live = true; from = from_cur;
                            // its first end-of-block code at or behind `from`
                            uint32_t c = NONE;
                            if (ew0 != NONE && ew0 >= from) c = ew0; else if (ew1 != NONE && ew1 >= from) c = ew1;
                            else if (ew2 != NONE && ew2 >= from) c = ew2; else if (ew3 != NONE && ew3 >= from) c = ew3;
                            else if (n_ew > 4u) c = NONE - 1u;                               // (more of them than the list holds: not ours)
                            else c = eob_at;
                            own_at = c;
                        }
                        const uint32_t c_u = (uint32_t)__shfl((int)own_at, (int)cur, 64);
                        if (c_u == NONE - 1u || !(bool)__shfl((int)active, (int)cur, 64)) { chain_bad = true; break; }
                        if (c_u != NONE) { E = (int)cur; block_ends = true; if ((uint32_t)lane == cur) own_eob = true; break; }
                        if ((bool)__shfl((int)x_fail, (int)cur, 64)) { chain_bad = true; break; }
                        if ((bool)__shfl((int)x_eob, (int)cur, 64)) { E = (int)cur; block_ends = true; break; }
                        const uint32_t nl = (uint32_t)__shfl((int)meet_lane, (int)cur, 64);
                        if (nl >= 64u) { E = (int)cur; break; }                              // ran to the end of the chunk
                        from_cur = (uint32_t)__shfl((int)meet, (int)cur, 64);
                        cur = nl;
                    }
                }
                if (chain_bad) { st = ST_NO_MEETING; break; }
                // what a live lane decoded in front of `from` does not count: the same symbols again, counted; a lane that ends the block
                // inside its window counts on to its end-of-block code
                uint32_t sb = 0, ss = 0, tb = 0, ts = 0;
                bool undec = false;
                const bool eob_in_window = own_eob && own_at != eob_at;
                if (live && (lane >= 1 || eob_in_window)) {
                    R.seek(S, start);
                    while (R.p < from) {
                        const Sym s = step(S, C, R, nb);
                        R.advance(S, s.used);
                        sb += s.kind == 0u ? 1u : s.kind == 1u ? s.a : 0u;
                        ss += s.kind < 2u ? 1u : 0u;
                    }
                    if (R.p != from) undec = true;
                    if (eob_in_window) {
                        while (R.p < own_at) {
                            const Sym s = step(S, C, R, nb);
                            R.advance(S, s.used);
                            if (s.kind >= 2u) undec = true;
                            tb += s.kind == 0u ? 1u : s.a;
                            ts += 1u;
                        }
                        if (R.p != own_at) undec = true;
                        const Sym s = step(S, C, R, nb);
                        if (s.kind != 2u) undec = true;
                        eob_end = R.p + s.used;
                    }
                }
                if (__ballot(undec)) { st = ST_UNDECODABLE; break; }
                if (cut == 3u) { st = __ballot(sb == NONE) ? ST_LANES : ST_LENGTH; break; }
                // ---- every live lane's share and its place -------------------------------------------------------------------------
                uint32_t cb = 0, cs = 0, stop = 0;
                if (live) {
                    if (own_eob) {
                        if (eob_in_window) { cb = tb; cs = ts; } else { cb = eob_ob - sb; cs = eob_os - ss; }
                        stop = own_at;
                    } else {
                        cb = ob - sb + xb; cs = os - ss + xs;
                        stop = x_eob ? x_eob_at : meet != NONE ? meet : x_end;
                    }
                }
                uint32_t tot_b = 0, tot_s = 0;
                const uint32_t off_b = wave_excl_sum(cb, lane, tot_b), off_s = wave_excl_sum(cs, lane, tot_s);
                if (out_pos + tot_b > isize) { st = ST_LENGTH; break; }
                // ---- pass 2: the true range again, one entry per symbol; four entries leave as one 16-byte store ------------------------
                bool w_bad = false;
                if (live) {
                    R.seek(S, from);
                    uint32_t o = out_pos + off_b, i = 0;
                    uint32_t* const w = msym + ns + off_s;
                    uint32_t e0 = 0, e1 = 0, e2 = 0, e3 = 0;
                    while (i < cs) {
                        const Sym s = step(S, C, R, nb);
                        if (s.kind >= 2u) { w_bad = true; break; }
                        R.advance(S, s.used);
                        if (s.kind == 1u && s.b > o) { w_bad = true; break; }
                        const uint32_t e = s.kind == 0u ? 0x80000000u | s.a : ((s.a - 3u) << 15) | (s.b - 1u);
                        o += s.kind == 0u ? 1u : s.a;
                        const uint32_t k = i & 3u;                          // (i is the same in every lane that is still at work)
                        if (k == 0u) e0 = e; else if (k == 1u) e1 = e; else if (k == 2u) e2 = e; else e3 = e;
                        i++;
                        if (k == 3u) { const uint4 v = make_uint4(e0, e1, e2, e3); __builtin_memcpy(w + i - 4u, &v, 16); }
                    }
                    if (!w_bad) {
                        const uint32_t k = i & 3u, at = i - k;
                        if (k > 0u) w[at] = e0;
                        if (k > 1u) w[at + 1u] = e1;
                        if (k > 2u) w[at + 2u] = e2;
                    }
                    if (R.p != stop || o != out_pos + off_b + cb) w_bad = true;
                }
                if (__ballot(w_bad)) { st = ST_UNDECODABLE; break; }
                out_pos += tot_b; ns += tot_s;
                if (block_ends) {
                    tpos = 64u * cbU + (uint32_t)__shfl((int)(own_eob ? eob_end : x_eob_end), E, 64);
                    block_done = true;
                } else cpos = 64u * cbU + (uint32_t)__shfl((int)x_end, E, 64);
            }
        }
        if (st == ST_OK && out_pos != isize) st = ST_LENGTH;
        if (st == ST_OK && tpos > nbits) st = ST_LENGTH;
        if (lane == 0) { wstatus[m] = st; n_sym[m] = st == ST_OK ? ns : 0u; }
    }
}

// ---- kernel B ------------------------------------------------------------------------------------------------------------
constexpr int CP_THREADS = 1024, CP_WAVES = CP_THREADS / 64;
constexpr uint32_t CP_UNIT = 256;                               // cells a wave handles at a time (four per lane); unit u is wave u % 16's

extern "C" __global__ __launch_bounds__(CP_THREADS) void k_inflate_copy(const uint32_t* __restrict__ sym, const uint32_t* __restrict__ n_sym,
                                                                        uint32_t* __restrict__ wstatus, const uint64_t* __restrict__ out_off,
                                                                        uint32_t m0, uint8_t* __restrict__ out, uint32_t cut)
{
    __shared__ uint16_t W[65536];
    __shared__ uint32_t wsum[2][CP_WAVES];
    __shared__ uint32_t s_bad;
    const uint32_t mb = blockIdx.x, m = m0 + mb;
    if (wstatus[m] != ST_OK) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t ns = n_sym[m];
    const uint64_t o0 = out_off[m];
    const uint32_t isize = (uint32_t)(out_off[m + 1] - o0);
    const uint32_t* const msym = sym + (size_t)mb * SYM_STRIDE;
    if (tid == 0) s_bad = 0u;
    // ---- place: every symbol's output offset by a scan (four symbols per thread and tile), its cells ---------------------------------
    uint32_t run = 0;
    bool bad = false;
    auto load4 = [&](uint32_t i0) -> uint4 {
        if (i0 + 4u <= ns) { uint4 v; __builtin_memcpy(&v, msym + i0, 16); return v; }
        uint4 v = make_uint4(0x80000000u, 0x80000000u, 0x80000000u, 0x80000000u);
        if (i0 < ns) v.x = msym[i0];
        if (i0 + 1u < ns) v.y = msym[i0 + 1u];
        if (i0 + 2u < ns) v.z = msym[i0 + 2u];
        return v;
    };
    uint4 nxt = load4(4u * (uint32_t)tid);
    uint32_t par = 0;
    for (uint32_t t0 = 0; t0 < ns; t0 += 4u * CP_THREADS, par ^= 1u) {
        const uint32_t i0 = t0 + 4u * (uint32_t)tid;
        const uint4 ev = nxt;
        if (t0 + 4u * CP_THREADS < ns) nxt = load4(i0 + 4u * CP_THREADS);          // the next tile's entries while this one is placed
        const uint32_t e[4] = {ev.x, ev.y, ev.z, ev.w};
        uint32_t len[4], tl = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) { len[k] = i0 + (uint32_t)k < ns ? ((e[k] >> 31) ? 1u : ((e[k] >> 15) & 0xFFu) + 3u) : 0u; tl += len[k]; }
        uint32_t inc = tl;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)inc, d, 64); if (lane >= d) inc += o; }
        if (lane == 63) wsum[par][wave] = inc;
        __syncthreads();
        uint32_t woff = 0, total = 0;
#pragma unroll
        for (int w = 0; w < CP_WAVES; w++) { const uint32_t v = wsum[par][w]; if (w < wave) woff += v; total += v; }
        if (run + total > isize) { bad = true; break; }                   // (uniform)
        uint32_t dst = run + woff + inc - tl;
        // literals and the first cells of a match by its own lane; what lies behind the eighth cell by the whole wave, match by match
        uint32_t long_dst = 0, long_len = 0, long_dist = 0;               // (at most one of a thread's four is taken over: the others stay inline)
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (len[k] == 0u) continue;
            if (e[k] >> 31) W[dst] = (uint16_t)(e[k] & 0xFFu);
            else {
                const uint32_t dist = (e[k] & 0x7FFFu) + 1u;
                if (dist > dst) bad = true;
                else {
                    // cell x of the match copies cell x - dist; behind the first `dist` cells that is a cell of the match itself:
                    // point past it, to the cell in front of the match that holds the same byte (x mod dist - dist)
                    const bool hand_over = len[k] > 8u && dist >= len[k] && long_len == 0u;
                    const uint32_t n_in = hand_over ? 8u : len[k];
                    uint32_t r = 0, D = dist;
                    for (uint32_t x = 0; x < n_in; x++) {
                        W[dst + x] = (uint16_t)(0x8000u | ((D <= 0x8000u ? D : dist) - 1u));
                        r++;
                        if (r == dist) { r = 0; D += dist; }
                    }
                    if (hand_over) { long_dst = dst + 8u; long_len = len[k] - 8u; long_dist = dist; }
                }
            }
            dst += len[k];
        }
        for (unsigned long long todo = __ballot(long_len != 0u); todo; todo &= todo - 1ull) {
            const int src = __ffsll((long long)todo) - 1;
            const uint32_t d0 = (uint32_t)__shfl((int)long_dst, src, 64), n = (uint32_t)__shfl((int)long_len, src, 64);
            const uint16_t v = (uint16_t)(0x8000u | ((uint32_t)__shfl((int)long_dist, src, 64) - 1u));
            for (uint32_t x = (uint32_t)lane; x < n; x += 64) W[d0 + x] = v;
        }
        run += total;
    }
    if (bad) atomicOr(&s_bad, 1u);
    __syncthreads();
    if (s_bad || run != isize) { if (tid == 0) wstatus[m] = ST_UNDECODABLE; return; }
    if (cut == 1u) { if (W[tid] == 0xFFFFu) wstatus[m] = ST_LANES; return; }
    // ---- resolve: pointer jumping; unit u (256 cells) is wave u % 16's, four cells per lane in flight -----------------------------------
    volatile uint16_t* const Wv = W;
    const uint32_t n_units = (isize + CP_UNIT - 1u) / CP_UNIT;
    uint32_t pending = 0;                                               // bit j: unit wave + 16 j
    for (uint32_t j = 0; j < 16u; j++) if ((uint32_t)wave + 16u * j < n_units) pending |= 1u << j;
    while (pending) {
        uint32_t next = 0;
        for (uint32_t rest = pending; rest; rest &= rest - 1u) {
            const uint32_t j = (uint32_t)__ffs((int)rest) - 1u;
            const uint32_t i0 = ((uint32_t)wave + 16u * j) * CP_UNIT + (uint32_t)lane;
            uint32_t v[4], u[4];
#pragma unroll
            for (int k = 0; k < 4; k++) { const uint32_t i = i0 + 64u * (uint32_t)k; v[k] = i < isize ? (uint32_t)Wv[i] : 0u; }
#pragma unroll
            for (int k = 0; k < 4; k++) { const uint32_t i = i0 + 64u * (uint32_t)k; u[k] = (v[k] & 0x8000u) ? (uint32_t)Wv[i - (v[k] & 0x7FFFu) - 1u] : 0u; }
            bool open = false;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (!(v[k] & 0x8000u)) continue;
                const uint32_t i = i0 + 64u * (uint32_t)k;
                if (!(u[k] & 0x8000u)) Wv[i] = (uint16_t)u[k];
                else {
                    open = true;
                    const uint32_t d2 = (v[k] & 0x7FFFu) + (u[k] & 0x7FFFu) + 2u;
                    if (d2 <= 0x8000u) Wv[i] = (uint16_t)(0x8000u | (d2 - 1u));
                }
            }
            if (__ballot(open)) next |= 1u << j;
        }
        pending = next;
    }
    if (cut == 2u) { if (W[tid] == 0xFFFFu) wstatus[m] = ST_LANES; return; }
    // ---- write: the wave's units, 16 bytes per lane and step (a unit = 16 lanes' worth: four units per step) -----------------------------
    uint8_t* const dst = out + o0;
    for (uint32_t j = 0; j < 16u; j += 4u) {
        const uint32_t unit = (uint32_t)wave + 16u * (j + ((uint32_t)lane >> 4));
        const uint32_t i = unit * CP_UNIT + 16u * ((uint32_t)lane & 15u);
        if (i >= isize) continue;
        const uint4 a = *reinterpret_cast<const uint4*>(&W[i]), b = *reinterpret_cast<const uint4*>(&W[i + 8]);
        uint4 v;
        v.x = __builtin_amdgcn_perm(a.y, a.x, 0x06040200u);
        v.y = __builtin_amdgcn_perm(a.w, a.z, 0x06040200u);
        v.z = __builtin_amdgcn_perm(b.y, b.x, 0x06040200u);
        v.w = __builtin_amdgcn_perm(b.w, b.z, 0x06040200u);
        if (i + 16u <= isize) __builtin_memcpy(dst + i, &v, 16);
        else {
            const unsigned long long lo = ((unsigned long long)v.y << 32) | v.x, hi = ((unsigned long long)v.w << 32) | v.z;
            for (uint32_t x = 0; x < 16u && i + x < isize; x++) dst[i + x] = (uint8_t)(x < 8u ? lo >> (8u * x) : hi >> (8u * (x - 8u)));
        }
    }
}

// ---- host side -------------------------------------------------------------------------------------------------------------
// The members of a run in batches of `batch` members (the symbol stream of a batch: batch x 256 KB of scratch): A over the batch,
// B over the batch.  d_wstatus / d_nsym: one word per member of the run (scratch of the context).
int gci_inflate_wave_run(gci_ctx* ctx, const uint8_t* d_raw, const uint64_t* d_member_pos, const uint64_t* d_out_off, uint32_t n_members,
                         uint8_t* d_out, uint64_t out_cap, uint32_t* d_wstatus)
{
    static const uint32_t batch_max = [] { const char* e = getenv("GCI_INFLATE_BATCH"); const int v = e ? atoi(e) : 8192; return (uint32_t)(v < 64 ? 64 : v); }();
    static const int waves_per_cu = [] { const char* e = getenv("GCI_INFLATE_WAVES"); return e ? atoi(e) : 0; }();
    static const uint32_t cut_a = [] { const char* e = getenv("GCI_IW_CUT_A"); return (uint32_t)(e ? atoi(e) : 0); }();   // (measurements)
    static const uint32_t cut_b = [] { const char* e = getenv("GCI_IW_CUT_B"); return (uint32_t)(e ? atoi(e) : 0); }();
    const uint32_t batch = n_members < batch_max ? n_members : batch_max;
    int st = gci_ensure(ctx, ctx->inflate_sym, (size_t)batch * SYM_STRIDE * sizeof(uint32_t));
    if (st) return st;
    st = gci_ensure(ctx, ctx->inflate_nsym, (size_t)n_members * sizeof(uint32_t));
    if (st) return st;
    int cus = 0, per_cu = 0;
    HIPCHK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device));
    HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_inflate_symbols, 64, 0));
    if (waves_per_cu > 0 && waves_per_cu < per_cu) per_cu = waves_per_cu;
    const uint32_t resident = (uint32_t)(cus > 0 && per_cu > 0 ? cus * per_cu : 1024);
    for (uint32_t m0 = 0; m0 < n_members; m0 += batch) {
        const uint32_t nb = n_members - m0 < batch ? n_members - m0 : batch;
        hipLaunchKernelGGL(k_inflate_symbols, dim3(nb < resident ? nb : resident), dim3(64), 0, ctx->stream, d_raw, d_member_pos, d_out_off, out_cap, m0, nb,
                           (uint32_t*)ctx->inflate_sym.p, (uint32_t*)ctx->inflate_nsym.p, d_wstatus, cut_a);
        LAUNCHCHK("k_inflate_symbols");
        hipLaunchKernelGGL(k_inflate_copy, dim3(nb), dim3(CP_THREADS), 0, ctx->stream, (const uint32_t*)ctx->inflate_sym.p,
                           (const uint32_t*)ctx->inflate_nsym.p, d_wstatus, d_out_off, m0, d_out, cut_b);
        LAUNCHCHK("k_inflate_copy");
    }
    return GCI_OK;
}

extern "C" int gci_bgzf_inflate_last_stats(gci_ctx* ctx, uint32_t h_counts[8])
{
    if (!ctx || !h_counts) return GCI_E_INVALID;
    for (int k = 0; k < 8; k++) h_counts[k] = 0;
    const uint32_t n = ctx->inflate_last_n;
    if (!n) return GCI_OK;
    std::vector<uint32_t> h(n);
    HIPCHK(hipMemcpyAsync(h.data(), ctx->inflate_wstatus.p, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    for (uint32_t v : h) h_counts[v < 7u ? v : 7u]++;
    return GCI_OK;
}
