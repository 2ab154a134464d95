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
//       pass 1    lane k decodes from the start of piece k as if a symbol began there and LISTS its symbols (32-bit entries: literal
//                 byte, or length and distance) in scratch memory, entry i of lane k at [i][k]; behind every 64-bit boundary of its
//                 piece it notes where it stood first and how many symbols it had listed by then (a checkpoint);
//       stitch    lane k runs on into piece k + 1 until, behind a boundary, it stands where lane k + 1 stood: from there lane
//                 k + 1's sequence is the true one, and its checkpoint says which of its listed symbols to leave out.
//                 (Same position + same tables = same sequence: a stitched result is exact, not probable.)
//       gather    prefix sums over the lanes give every lane's share of its list its place in the member's SYMBOL STREAM in HBM.
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
#include <stdio.h>
#include <stdlib.h>

#ifndef IW_PIECE_LOG2
#define IW_PIECE_LOG2 9                  // bits per lane and chunk: 512 (4 KB of payload per wave in LDS)
#endif
#ifndef IW_GRAIN_LOG2
#define IW_GRAIN_LOG2 6                  // a lane notes where it stands first behind every 64-bit boundary of its piece
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
constexpr uint32_t CHUNK_U = 64u * PU, TAIL_U = PU;             // units per chunk; units kept behind it (a symbol that begins in the chunk ends there)
constexpr uint32_t GRAIN_LOG2 = IW_GRAIN_LOG2, NCK = PIECE >> GRAIN_LOG2;   // checkpoints per piece (a symbol is shorter than a grain: no grain is skipped)
static_assert((1u << GRAIN_LOG2) >= 48u, "a symbol with its extra bits must be shorter than a grain");
constexpr int LIT_BITS = IW_LIT_BITS, DIST_BITS = IW_DIST_BITS;
// codes longer than the primary table's index: their 15-bit values (first bit highest) lie at the top of the code space, from
// limit[BITS] on -- a direct table over that range.  A HiFi BAM's blocks need ~300 / ~130 entries; in a block that needs more
// than the tables hold (Codes::search_*, uniform) the lanes that stand on such a code search the limits.
constexpr uint32_t LIT_TAIL = 512, DIST_TAIL = 256;
constexpr uint32_t HDR_U = CHUNK_U < 128u ? CHUNK_U : 128u;    // units staged for a block's header (8192 bits; the longest header has 4498)
constexpr uint32_t MAXS = PIECE < 512u ? 256u : PIECE / 2u;      // symbols a lane may list per chunk (its piece and what it runs on into)
constexpr uint32_t NONE = 0xFFFFFFFFu;
constexpr uint32_t SYM_STRIDE = 65536;                          // entries of symbol stream per member (one per output byte at most)

enum { ST_OK = 0, ST_HEADER = 1, ST_NO_MEETING = 2, ST_FALSE_EOB = 3, ST_UNDECODABLE = 4, ST_LENGTH = 5, ST_LANES = 6 };

__constant__ uint8_t c_clen_order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

// (k_inflate.hip explains limit / off / next: the canonical code without a loop over the lengths)
struct alignas(16) Canon { uint16_t limit[16]; int16_t off[16]; uint16_t next[16]; };

struct alignas(16) Lds {
    unsigned long long pay[CHUNK_U + TAIL_U];      // the chunk, transposed: unit u of piece k at [(u << 6) | k]; the tail linear behind it
    uint32_t ckpt[64 * (NCK + 1)];                 // lane k, grain c at [(c << 6) | k]: (where it stood first in the grain) << 16 | symbols listed by then; row NCK: writes that are none
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

__device__ __forceinline__ uint32_t pay_index(uint32_t U) { return U < CHUNK_U ? ((U & (PU - 1u)) << 6) | (U >> PU_LOG2) : U; }

// 64 bits from bit p of the chunk
__device__ __forceinline__ unsigned long long peek(const Lds& S, uint32_t p)
{
    const uint32_t U = p >> 6, s = p & 63u;
    const unsigned long long lo = S.pay[pay_index(U)], hi = S.pay[pay_index(U + 1u)];
    return s ? (lo >> s) | (hi << (64u - s)) : lo;
}

// A lane's place in the chunk: the unit it stands in and the two behind it in registers -- the bits of a symbol come out of
// them without a look into LDS; the unit after those is fetched at the top of every step and taken up when the lane moves on.
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
    __device__ __forceinline__ unsigned long long ahead(const Lds& S) const { return S.pay[pay_index(U + 3u)]; }
    __device__ __forceinline__ void advance(uint32_t used, unsigned long long n3)      // used < 64; n3 = ahead()
    {
        p += used;
        const bool cross = (p >> 6) != U;
        cur = cross ? nxt : cur; nxt = cross ? nx2 : nxt; nx2 = cross ? n3 : nx2;
        U += cross ? 1u : 0u;
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
struct Codes { uint32_t lit_lim, dist_lim; bool search_l, search_d; };

// entry (symbol | length << 9, 0 = none) of the code the bits begin with: the primary table and the tail table, both looked up at
// once -- no branch, one wait
template <int BITS, uint32_t TAIL>
__device__ __forceinline__ uint32_t code_entry(uint32_t b, const uint16_t* tab, const uint16_t* tail, uint32_t lim, bool search, const Canon& cn,
                                               const uint16_t* sorted)
{
    const uint32_t t = (__brev(b) >> 17) - lim;                               // (wraps to a huge value for a code in front of the tail)
    const uint32_t e_p = tab[b & ((1u << BITS) - 1u)], e_t = tail[t < TAIL ? t : TAIL - 1u];
    uint32_t e = e_p ? e_p : t < TAIL ? e_t : 0u;
    if (search) {                                                             // (uniform, and false for the blocks of a BAM)
        if (e_p == 0u && t >= TAIL && t < 0x8000u) {
            int l = 0;
            const int s = code_search(b, cn, sorted, l);
            e = s < 0 ? 0u : (uint32_t)s | ((uint32_t)l << 9);
        }
    }
    return e;
}

// kind: 0 literal, 1 match (entry = what the symbol stream holds for them), 2 end of block, 3 nothing decodable (used = 1: a wrong path moves on)
struct Sym { uint32_t kind, used, entry; };

// the literal / length symbol the reader stands on with everything that belongs to it (48 bits at most); nb = end of the stream
__device__ __forceinline__ Sym step(const Lds& S, const Codes& C, const Reader& R, uint32_t nb)
{
    const unsigned long long bits = R.bits();
    const uint32_t b = (uint32_t)bits;
    const uint32_t e = code_entry<LIT_BITS, LIT_TAIL>(b, S.lit_tab, S.lit_tail, C.lit_lim, C.search_l, S.lit_cn, S.lit_sorted);
    const uint32_t l = e >> 9, s = e & 0x1FFu;
    // the length code's base and extra bits by arithmetic (RFC 1951 3.2.5); for a literal the values are not used
    const uint32_t lc = s - 257u;
    const uint32_t le = lc < 8u || lc >= 28u ? 0u : (lc >> 2) - 1u;
    const uint32_t len3 = (lc < 8u ? lc : lc == 28u ? 255u : ((4u + (lc & 3u)) << le)) + ((b >> l) & ((1u << le) - 1u));
    const uint32_t used1 = l + le;
    const uint32_t b2 = (uint32_t)(bits >> used1);
    const uint32_t e2 = code_entry<DIST_BITS, DIST_TAIL>(b2, S.dist_tab, S.dist_tail, C.dist_lim, C.search_d, S.dist_cn, S.dist_sorted);
    const uint32_t dl = e2 >> 9, ds = e2 & 0x1FFu;
    const uint32_t de = ds < 4u ? 0u : ((ds >> 1) - 1u) & 15u;
    const uint32_t dist1 = (ds < 4u ? ds : ((2u + (ds & 1u)) << de)) + ((b2 >> dl) & ((1u << de) - 1u));
    const bool is_match = s > 256u;
    const uint32_t used = is_match ? used1 + dl + de : l;
    const bool bad = e == 0u || s > 285u || (is_match && (e2 == 0u || ds > 29u)) || R.p + used > nb;
    Sym r;
    r.used = bad ? 1u : used;
    r.kind = bad ? 3u : is_match ? 1u : s == 256u ? 2u : 0u;
    r.entry = is_match ? (len3 << 15) | dist1 : 0x80000000u | s;
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
__device__ __forceinline__ bool fill_tables(uint16_t* tab, int bits, uint16_t* tail, uint32_t n_tail, const Canon& cn, const uint16_t* sorted, int lane)
{
    for (uint32_t k = lane; k < (1u << bits); k += 64) tab[k] = table_entry(k, bits, cn, sorted);
    const uint32_t lim = cn.limit[bits], top = cn.limit[15];                   // codes exist below `top` only
    const uint32_t want = top > lim ? top - lim : 0u;
    for (uint32_t t = lane; t < n_tail; t += 64) tail[t] = t < want ? entry_of_value(lim + t, cn, sorted) : (uint16_t)0;
    return want <= n_tail;                                                     // false: codes the tables do not hold
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
// members [m0, m0 + n_batch) of the run; sym: n_batch x SYM_STRIDE entries; n_sym, wstatus: one word per member of the RUN;
// lists: per workgroup 64 x MAXS entries, entry i of lane k at [i * 64 + k] (a step's stores are one 256-byte row)
// entry: bit 31 set = literal (low 8 bits); else (length - 3) << 15 | (distance - 1)
// MEASURE = false (what runs): the phase counters and the cuts do not exist -- they cost registers both kernels are short of.
template <bool MEASURE>
__global__ __launch_bounds__(64, 4) void k_inflate_symbols(const uint8_t* __restrict__ raw, const uint64_t* __restrict__ member_pos,
                                                                   const uint64_t* __restrict__ out_off, uint64_t out_cap, uint32_t m0, uint32_t n_batch,
                                                                   uint32_t* __restrict__ sym, uint32_t* __restrict__ n_sym,
                                                                   uint32_t* __restrict__ wstatus, uint32_t* __restrict__ lists, uint32_t* __restrict__ next,
                                                                   uint32_t cut_arg, unsigned long long* __restrict__ prof_arg)
{
    unsigned long long* const prof = MEASURE ? prof_arg : nullptr;
    const uint32_t cut = MEASURE ? cut_arg : 0u;
    // (prof: measurements only -- cycles per phase, summed over the waves: GCI_IW_PROF=1, tools/hwtests/inflate_product.py)
    unsigned long long t_hdr = 0, t_stage = 0, t_p1 = 0, t_st = 0, t_ch = 0, t_ga = 0, n_chunks = 0, t_all0 = prof ? __builtin_amdgcn_s_memtime() : 0;
#define IW_T(acc, t0) do { if (prof) { const unsigned long long _n = __builtin_amdgcn_s_memtime(); acc += _n - t0; t0 = _n; } } while (0)
    // (cut: measurements only -- 1 leaves a block behind its tables, 2 behind pass 1, 3 behind the stitch; 0 = the kernel)
    __shared__ Lds S;
    const int lane = threadIdx.x;
    uint32_t* const list = lists + (size_t)blockIdx.x * (64u * (MAXS + 1u)) + (uint32_t)lane;     // entry i at list[64 i]; row MAXS: writes that are none
    // the members of the batch are handed out one at a time (`next`: a counter the host zeroes): a member takes as long as it takes
    for (;;) {
        __syncthreads();
        if (lane == 0) S.hdr[7] = atomicAdd(next, 1u);
        __syncthreads();
        const uint32_t mb = S.hdr[7];
        if (mb >= n_batch) break;
        const uint32_t m = m0 + mb;
        const uint64_t pos0 = member_pos[m], pos1 = member_pos[m + 1];
        const uint32_t isize = (uint32_t)(out_off[m + 1] - out_off[m]);
        uint32_t* const msym = sym + (size_t)mb * SYM_STRIDE;
        uint32_t st = ST_OK;
        if (pos1 < pos0 + 26 || isize > 65536u || out_off[m + 1] > out_cap || raw[pos0] != 0x1f || raw[pos0 + 1] != 0x8b || raw[pos0 + 2] != 8) {
            if (lane == 0) { wstatus[m] = ST_HEADER; n_sym[m] = 0; }
            continue;
        }
        const uint32_t xlen = (uint32_t)raw[pos0 + 10] | ((uint32_t)raw[pos0 + 11] << 8);
        if (pos0 + 12 + xlen + 8 > pos1) { if (lane == 0) { wstatus[m] = ST_HEADER; n_sym[m] = 0; } continue; }
        // the payload as 8-byte units of global memory from the aligned address in front of it (derived from `raw`, so that the loads
        // stay global ones: a pointer made out of an integer would be a flat one)
        const uint32_t mis = (uint32_t)(((uintptr_t)raw + pos0 + 12 + xlen) & 7u);
        const unsigned long long* const g8 = (const unsigned long long*)(raw + (pos0 + 12 + xlen - mis));
        const uint32_t bias = 8u * mis;
        const uint32_t nbits = bias + 8u * (uint32_t)(pos1 - 8 - (pos0 + 12 + xlen));          // end of the DEFLATE stream (bits from g8)
        const uint32_t U_end = (nbits + 63u) >> 6;
        uint32_t tpos = bias, ns = 0;                                       // (uniform over the wave)
        for (bool last = false; !last && st == ST_OK;) {
            // ---- the block's header: 3 + 14 + 57 bits and the code lengths -- 128 units hold any header this kernel takes -----------
            unsigned long long tt = prof ? __builtin_amdgcn_s_memtime() : 0;
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
            if (type == 3u) { st = 15u; break; }
            if (type == 0u) {                                             // stored: its bytes as literals
                const uint8_t* const gb = (const uint8_t*)g8;
                const uint32_t byte = (body0 + 7u) >> 3;
                if (8u * (byte + 4u) > nbits) { st = ST_HEADER; break; }
                const uint32_t len = (uint32_t)gb[byte] | ((uint32_t)gb[byte + 1] << 8), nlen = (uint32_t)gb[byte + 2] | ((uint32_t)gb[byte + 3] << 8);
                if ((len ^ 0xFFFFu) != nlen || ns + len > isize || 8u * (byte + 4u + len) > nbits) { st = ST_HEADER; break; }
                for (uint32_t i = lane; i < len; i += 64) msym[ns + i] = 0x80000000u | (uint32_t)gb[byte + 4u + i];
                ns += len;
                tpos = 8u * (byte + 4u + len);
                continue;
            }
            Codes C;
            {
                __syncthreads();
                const bool ok_l = build_code_wave(S.lens + 32, hlit, S.lit_cn, S.lit_sorted, lane, &S.hdr[4]);
                const bool ok_d = build_code_wave(S.lens + 32 + hlit, hdist, S.dist_cn, S.dist_sorted, lane, &S.hdr[4]);
                if (!ok_l || !ok_d) { st = 12u; break; }
                const bool fits_l = fill_tables(S.lit_tab, LIT_BITS, S.lit_tail, LIT_TAIL, S.lit_cn, S.lit_sorted, lane);
                const bool fits_d = fill_tables(S.dist_tab, DIST_BITS, S.dist_tail, DIST_TAIL, S.dist_cn, S.dist_sorted, lane);
                // (the same in every lane: said so, and the four live in scalar registers -- a branch on them costs no exec-mask work)
                C.search_l = __builtin_amdgcn_readfirstlane(fits_l ? 0 : 1) != 0;             // long codes beyond the tail tables: searched
                C.search_d = __builtin_amdgcn_readfirstlane(fits_d ? 0 : 1) != 0;
                C.lit_lim = (uint32_t)__builtin_amdgcn_readfirstlane((int)S.lit_cn.limit[LIT_BITS]);
                C.dist_lim = (uint32_t)__builtin_amdgcn_readfirstlane((int)S.dist_cn.limit[DIST_BITS]);
            }
            if (cut == 1u) { st = ST_LANES; break; }
            IW_T(t_hdr, tt);
            // ---- the block's body, chunk by chunk ----------------------------------------------------------------------------
            uint32_t cpos = body0;                                        // a symbol begins here
            for (bool block_done = false; !block_done && st == ST_OK;) {
                cbU = cpos >> 6;
                __syncthreads();
                stage(S, g8, cbU, U_end, CHUNK_U + TAIL_U, lane);
                for (uint32_t i = lane; i < 64u * NCK; i += 64) S.ckpt[i] = NONE;
                __syncthreads();
                const uint32_t nb = nbits - 64u * cbU;                    // end of the stream, in bits of the chunk
                if (cut == 4u) { st = ST_LANES; break; }
                IW_T(t_stage, tt); n_chunks++;
                // ---- pass 1: every lane over its own piece, its symbols into its list -------------------------------------------
                const uint32_t start = lane == 0 ? cpos - 64u * cbU : (uint32_t)lane * PIECE;
                const uint32_t bound = ((uint32_t)lane + 1u) * PIECE;
                const bool active = start < nb;
                uint32_t os = 0;                                          // entries in the list
                // End-of-block codes the lane passes: on a wrong path they mean nothing (a fixed code has one in 128 symbols), so the lane
                // goes on; the first four with the symbols listed in front of them (position << 16 | count).
                uint32_t eb0 = NONE, eb1 = NONE, eb2 = NONE, eb3 = NONE, n_eb = 0;
                // The first step of the piece that met bits no code matches (kind 3: it moves on by one bit and lists nothing).  On a wrong
                // path that means nothing; on the lane's TRUE path -- decided below -- the member is damaged, and is handed to the lane
                // decoder, which says so (with the CRC check off, such a member whose lengths still added up was accepted).
                uint32_t bad_rel = NONE;
                Reader R;
                R.seek(S, active ? start : 0u);
                if (active) {
                    const uint32_t pend = bound < nb ? bound : nb;
                    uint32_t grain = NONE;                                // the grain the lane stood in last
                    while (R.p < pend) {
                        const unsigned long long n3 = R.ahead(S);
                        const uint32_t rel = R.p - (uint32_t)lane * PIECE, g = rel >> GRAIN_LOG2;
                        S.ckpt[((g != grain ? g : NCK) << 6) | (uint32_t)lane] = (rel << 16) | os;     // (no branch: a step in a grain it has noted writes the spare row)
                        grain = g;
                        const Sym s = step(S, C, R, nb);
                        if (__ballot(s.kind == 3u)) { if (s.kind == 3u && bad_rel == NONE) bad_rel = rel; }      // (rare: the whole wave skips this)
                        if (__ballot(s.kind == 2u)) {                      // (rare: the whole wave skips this)
                            if (s.kind == 2u) {
                                const uint32_t v = (rel << 16) | os;
                                if (n_eb == 0u) eb0 = v; else if (n_eb == 1u) eb1 = v; else if (n_eb == 2u) eb2 = v; else if (n_eb == 3u) eb3 = v;
                                n_eb++;
                            }
                        }
                        // (no branch: what is no symbol, or lies beyond the list, goes to the spare row; a share that reaches beyond the list
                        // is caught below)
                        list[64u * (s.kind < 2u && os < MAXS ? os : MAXS)] = s.entry;
                        os += s.kind < 2u ? 1u : 0u;
                        R.advance(s.used, n3);
                    }
                }
                __syncthreads();
                IW_T(t_p1, tt);
                if (cut == 2u || cut == 6u) { st = __ballot(os == NONE) ? ST_LANES : ST_LENGTH; break; }
                // ---- stitch: on behind the piece until, first behind a boundary, the lane stands where the lane of THAT piece stood ------
                // (normally within a few symbols; a lane that never falls into step with its neighbour inside the neighbour's piece
                // has decoded that piece itself by then -- the neighbour is void -- and goes on into the piece after it)
                uint32_t meet = NONE, meet_lane = 64u, meet_cnt = 0, xs = 0, x_eob_end = 0, x_end = 0;
                bool x_eob = false, x_fail = false;
                if (active) {
                    uint32_t grain = NONE;
                    for (;;) {
                        const uint32_t q = R.p;
                        if (q >= 64u * PIECE) { x_end = q; break; }                        // the end of the chunk: the block goes on behind it
                        if (q >= nb) { x_fail = true; break; }                              // the stream ends without an end-of-block code
                        const uint32_t g = q >> GRAIN_LOG2;                                 // (grains of the chunk: piece j's are j NCK ...)
                        if (g != grain) {
                            grain = g;
                            const uint32_t j = q >> IW_PIECE_LOG2, rel = q & (PIECE - 1u);
                            const uint32_t c = S.ckpt[((rel >> GRAIN_LOG2) << 6) | j];
                            if (c != NONE && (c >> 16) == rel) { meet = q; meet_lane = j; meet_cnt = c & 0xFFFFu; break; }
                        }
                        const unsigned long long n3 = R.ahead(S);
                        const Sym s = step(S, C, R, nb);
                        if (s.kind == 3u) { x_fail = true; break; }
                        if (s.kind == 2u) { x_eob = true; x_eob_end = q + s.used; break; }
                        if (os + xs < MAXS) list[64u * (os + xs)] = s.entry;
                        xs++;
                        R.advance(s.used, n3);
                    }
                }
                IW_T(t_st, tt);
                // ---- the lanes on the true path: lane 0, the lane it met, the lane THAT one met, ... up to the first one that ends the block
                // or runs out of the chunk.  Nearly always every lane met its neighbour: then each lane's beginning is what the lane in
                // front of it found, and all of it is decided side by side; a lane that took a piece over (or has more end-of-block codes
                // than it lists) sends the wave down the chain one lane at a time.
                uint32_t ss = 0, cs = 0;                                                     // the lane's share of its list: [ss, ss + cs)
                bool live = false, own_eob = false, block_ends = false, chain_bad = false, undecodable = false;
                uint32_t own_end = 0;                                                        // behind the lane's own end-of-block code, if it ends the block
                int E = 0;
                const uint32_t base = (uint32_t)lane * PIECE;
                // the lane's first end-of-block code at or behind a position of its piece (both relative to the piece); NONE - 1: look again
                auto own_code = [&](uint32_t from_rel) -> uint32_t {
                    uint32_t own = NONE;
                    if (eb3 != NONE && (eb3 >> 16) >= from_rel) own = eb3;
                    if (eb2 != NONE && (eb2 >> 16) >= from_rel) own = eb2;
                    if (eb1 != NONE && (eb1 >> 16) >= from_rel) own = eb1;
                    if (eb0 != NONE && (eb0 >> 16) >= from_rel) own = eb0;
                    return own == NONE && n_eb > 4u ? NONE - 1u : own;
                };
                bool fast = false;
                {
                    // (the shuffles by every lane: a lane switched off for them would hand its neighbour a zero)
                    const uint32_t up_meet = (uint32_t)__shfl_up((int)meet, 1, 64), up_cnt = (uint32_t)__shfl_up((int)meet_cnt, 1, 64);
                    const uint32_t from_f = lane == 0 ? cpos - 64u * cbU : up_meet;
                    const uint32_t ss_f = lane == 0 ? 0u : up_cnt;
                    // (for a lane whose neighbour in front found nothing these are NONE / garbage: such a lane lies behind E or the fast way is left)
                    const uint32_t own = active && from_f != NONE && from_f >= base ? own_code(from_f - base) : NONE;
                    const bool ends = own != NONE || x_eob || (active && !x_eob && !x_fail && meet_lane >= 64u) || !active;   // ... the chain, one way or the other
                    const unsigned long long ends_mask = __ballot(ends);
                    const int e = ends_mask ? __ffsll((long long)ends_mask) - 1 : 63;
                    const unsigned long long upto = e >= 63 ? ~0ull : ((1ull << (e + 1)) - 1ull), below = upto >> 1;
                    const bool odd = (lane < e && (meet_lane != (uint32_t)lane + 1u || x_fail)) || (lane <= e && (!active || own == NONE - 1u));
                    if (cut != 7u && ends_mask && (__ballot(odd) & upto) == 0ull && !(bool)__shfl((int)(x_fail && own == NONE), e, 64)) {
                        fast = true; E = e; (void)below;
                        live = lane <= e; ss = ss_f;
                        if (live) {
                            if (lane == e && own != NONE) {
                                own_eob = true; cs = (own & 0xFFFFu) - ss;
                                R.seek(S, base + (own >> 16));
                                const Sym y = step(S, C, R, nb);
                                own_end = base + (own >> 16) + y.used;
                            } else cs = os + xs - ss;
                            // an undecodable step inside what this lane contributes: from where its true path enters the piece to its
                            // end-of-block code (the last lane) or the end of the piece
                            undecodable = bad_rel != NONE && bad_rel >= from_f - base && (!own_eob || bad_rel < (own >> 16));
                        }
                        block_ends = (bool)__shfl((int)(own != NONE || x_eob), e, 64);
                    }
                }
                if (!fast) {
                    uint32_t cur = 0, from_cur = cpos - 64u * cbU, ss_cur = 0;
                    for (;;) {
                        if ((uint32_t)lane == cur) {
                            live = true; ss = ss_cur;
                            uint32_t own = own_code(from_cur - base);
                            if (own == NONE - 1u) {
                                // more of them than the list holds and none of the listed ones behind `from`: the lane's piece once
                                // more from there, alone (one lane in thousands of chunks)
                                own = NONE;
                                R.seek(S, from_cur);
                                const uint32_t pend = bound < nb ? bound : nb;
                                uint32_t cnt = ss_cur;
                                while (R.p < pend) {
                                    const unsigned long long n3 = R.ahead(S);
                                    const Sym y = step(S, C, R, nb);
                                    if (y.kind == 2u) { own = ((R.p - base) << 16) | cnt; break; }
                                    cnt += y.kind < 2u ? 1u : 0u;
                                    R.advance(y.used, n3);
                                }
                            }
                            if (own != NONE) {
                                own_eob = true; cs = (own & 0xFFFFu) - ss;
                                R.seek(S, base + (own >> 16));
                                const Sym y = step(S, C, R, nb);
                                own_end = base + (own >> 16) + y.used;
                            } else cs = os + xs - ss;
                            undecodable = bad_rel != NONE && bad_rel >= from_cur - base && (!own_eob || bad_rel < (own >> 16));
                        }
                        if (!(bool)__shfl((int)active, (int)cur, 64)) { chain_bad = true; st = 10u; break; }
                        if ((bool)__shfl((int)own_eob, (int)cur, 64)) { E = (int)cur; block_ends = true; break; }
                        if ((bool)__shfl((int)x_fail, (int)cur, 64)) { chain_bad = true; st = 11u; break; }
                        if ((bool)__shfl((int)x_eob, (int)cur, 64)) { E = (int)cur; block_ends = true; break; }
                        const uint32_t nl = (uint32_t)__shfl((int)meet_lane, (int)cur, 64);
                        if (nl >= 64u) { E = (int)cur; break; }                              // ran to the end of the chunk
                        from_cur = (uint32_t)__shfl((int)meet, (int)cur, 64);
                        ss_cur = (uint32_t)__shfl((int)meet_cnt, (int)cur, 64);
                        cur = nl;
                    }
                }
                if (chain_bad) break;
                if (__ballot(live && undecodable)) { st = ST_UNDECODABLE; break; }
                if (cut == 3u) { st = __ballot(ss == NONE - 3u) ? ST_LANES : ST_LENGTH; break; }
                if (__ballot(live && (ss + cs > MAXS || ss + cs < ss))) { st = 9u; break; }   // (a share the list did not hold)
                IW_T(t_ch, tt);
                uint32_t tot_s = 0;
                const uint32_t off_s = wave_excl_sum(cs, lane, tot_s);
                if (ns + tot_s > isize) { st = ST_LENGTH; break; }                            // (a symbol is at least a byte)
                // ---- the shares into the member's symbol stream, four entries per store ------------------------------------------------
                if (live) {
                    uint32_t* const w = msym + ns + off_s;
                    const uint32_t* const r = list + 64u * ss;
                    uint32_t i = 0;
                    for (; i + 4u <= cs; i += 4u) {
                        const uint4 v = make_uint4(r[64u * i], r[64u * (i + 1u)], r[64u * (i + 2u)], r[64u * (i + 3u)]);
                        __builtin_memcpy(w + i, &v, 16);
                    }
                    for (; i < cs; i++) w[i] = r[64u * i];
                }
                ns += tot_s;
                IW_T(t_ga, tt);
                if (block_ends) {
                    tpos = 64u * cbU + (uint32_t)__shfl((int)(own_eob ? own_end : x_eob_end), E, 64);
                    block_done = true;
                } else cpos = 64u * cbU + (uint32_t)__shfl((int)x_end, E, 64);
            }
        }
        if (st == ST_OK && tpos > nbits) st = ST_LENGTH;
        if (lane == 0) { wstatus[m] = st; n_sym[m] = st == ST_OK ? ns : 0u; }
    }
    if (prof && lane == 0) {
        atomicAdd(&prof[0], t_hdr); atomicAdd(&prof[1], t_stage); atomicAdd(&prof[2], t_p1); atomicAdd(&prof[3], t_st); atomicAdd(&prof[4], t_ch);
        atomicAdd(&prof[5], t_ga); atomicAdd(&prof[6], __builtin_amdgcn_s_memtime() - t_all0); atomicAdd(&prof[7], n_chunks);
    }
#undef IW_T
}

// ---- kernel B ------------------------------------------------------------------------------------------------------------
#ifndef IW_CP_THREADS
#define IW_CP_THREADS 1024
#endif
constexpr int CP_THREADS = IW_CP_THREADS, CP_WAVES = CP_THREADS / 64;
constexpr uint32_t CP_UNIT = 256;                               // cells a wave handles at a time (four per lane); unit u is wave u % CP_WAVES's
constexpr uint32_t CP_UPW = 256u / (uint32_t)CP_WAVES;          // units per wave (65 536 cells = 256 units)
static_assert(CP_UPW <= 32u && CP_UPW % 4u == 0u, "a wave's units are a 32-bit mask, written four at a time");

// What the workgroup needs to know of a member before it can start on it, and the first tile of its symbols: fetched while the
// member in front of it is being resolved (IW_CP_PREFETCH; one workgroup per CU means nothing else hides these round trips --
// status, then count and offsets, then the symbols: three of them in a row at the head of every member).
#ifndef IW_CP_PREFETCH
#define IW_CP_PREFETCH 1
#endif
#ifndef IW_CP_RES_UNITS
#define IW_CP_RES_UNITS 1
#endif
struct CpMeta { uint32_t status, ns, isize; uint64_t o0; };
__device__ __forceinline__ CpMeta cp_meta(uint32_t mb, uint32_t n_batch, uint32_t m0, const uint32_t* __restrict__ n_sym,
                                          const uint32_t* __restrict__ wstatus, const uint64_t* __restrict__ out_off)
{
    CpMeta c{0xFFFFFFFFu, 0u, 0u, 0ull};
    if (mb < n_batch) {
        const uint32_t m = m0 + mb;
        c.status = wstatus[m]; c.ns = n_sym[m]; c.o0 = out_off[m]; c.isize = (uint32_t)(out_off[m + 1] - c.o0);
    }
    return c;
}
// IW_CP_PF tiles of a member's symbols are in registers ahead of the tile being placed (a tile = four symbols per thread): the
// place phase took a round trip to memory per tile with one tile ahead -- its tiles are too short to hide one.
#ifndef IW_CP_PF
#define IW_CP_PF 2
#endif
#ifndef IW_CP_FILL_LEAN
#define IW_CP_FILL_LEAN 0                // (see the fill loop: a variant waiting for its measurement)
#endif
#ifndef IW_CP_FILL_STEPS
#define IW_CP_FILL_STEPS 4
#endif
#ifndef IW_CP_CUT
#define IW_CP_CUT 0                      // (measurements: 1 = the copy kernel leaves a member behind its place phase, 2 = behind the resolve phase)
#endif
struct CpTiles { uint4 t[IW_CP_PF]; };
__device__ __forceinline__ uint4 cp_load4(const uint32_t* __restrict__ msym, uint32_t i0, uint32_t ns)
{
    if (i0 + 4u <= ns) { uint4 v; __builtin_memcpy(&v, msym + i0, 16); return v; }
    uint4 v = make_uint4(0x80000000u, 0x80000000u, 0x80000000u, 0x80000000u);
    if (i0 < ns) v.x = msym[i0];
    if (i0 + 1u < ns) v.y = msym[i0 + 1u];
    if (i0 + 2u < ns) v.z = msym[i0 + 2u];
    return v;
}
__device__ __forceinline__ void cp_first_tiles(CpTiles& b, const uint32_t* __restrict__ msym, uint32_t tid, uint32_t ns)
{
#pragma unroll
    for (int k = 0; k < IW_CP_PF; k++) b.t[k] = cp_load4(msym, 4u * tid + (uint32_t)k * 4u * (uint32_t)IW_CP_THREADS, ns);
}

// One sweep over a unit of 256 cells, four per lane (cells i0 + 64 k): a cell whose target holds a byte takes it, one whose target is a
// pointer adds the two distances when the sum still fits.  -> some cell of this lane still points at a pointer.
template <bool WHOLE>
__device__ __forceinline__ bool resolve_unit(uint16_t* __restrict__ W, uint32_t i0, uint32_t isize)
{
    // (written for the instruction count -- the kernel is bound by it: a pointer is 0x8000 | (distance - 1), so its distance is
    //  max(v, 0x7FFF) - 0x7FFF, which is 0 for a byte; two pointers in a row add up to v + u - 0x7FFF, which fits while it is <= 0xFFFF)
    uint32_t v[4], u[4];
#pragma unroll
    for (int k = 0; k < 4; k++) { const uint32_t i = i0 + 64u * (uint32_t)k; v[k] = WHOLE || i < isize ? (uint32_t)W[i] : 0u; }
#pragma unroll
    for (int k = 0; k < 4; k++) {                                        // (a cell that holds a byte reads itself: no branch around the look-up)
        const uint32_t i = i0 + 64u * (uint32_t)k;
        const uint32_t back = max(v[k], 0x7FFFu) - 0x7FFFu;
        u[k] = (uint32_t)W[WHOLE || i < isize ? i - back : 0u];
    }
    uint32_t both = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t i = i0 + 64u * (uint32_t)k;
        const uint32_t sum = v[k] + u[k] - 0x7FFFu;
        const uint32_t nv = u[k] < 0x8000u ? u[k] : sum <= 0xFFFFu ? sum : v[k];
        if (v[k] >= 0x8000u && nv != v[k]) W[i] = (uint16_t)nv;
        both |= v[k] & u[k];
    }
    return (both & 0x8000u) != 0u;
}

// one member (member mb of the batch) by the whole workgroup; mb_next: the member this workgroup takes after it
template <bool MEASURE>
__device__ __forceinline__ void copy_member(uint16_t* __restrict__ W, uint32_t* __restrict__ SB, uint32_t (*wsum)[CP_WAVES], uint32_t& s_bad, uint32_t mb,
                                            const uint32_t* __restrict__ sym, const uint32_t* __restrict__ n_sym,
                                            uint32_t* __restrict__ wstatus, const uint64_t* __restrict__ out_off,
                                            uint32_t m0, uint8_t* __restrict__ out, uint32_t cut_arg,
                                            unsigned long long* __restrict__ prof_arg, CpMeta& meta, CpTiles& first, uint32_t mb_next, uint32_t n_batch)
{
    unsigned long long* const prof = MEASURE ? prof_arg : nullptr;
    const uint32_t cut = MEASURE ? cut_arg : (uint32_t)IW_CP_CUT;
    unsigned long long tb = prof ? __builtin_amdgcn_s_memtime() : 0, t_place = 0, t_res = 0, n_ur = 0, n_rounds = 0;
    unsigned long long t_scan = 0, t_cells = 0, t_long = 0;
    const unsigned long long t_begin = tb;
    const uint32_t m = m0 + mb;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);          // (uniform: the loops over a wave's units and tiles are scalar code)
    const CpMeta me = meta;
    CpTiles buf = first;
    meta = cp_meta(mb_next, n_batch, m0, n_sym, wstatus, out_off);         // (in flight from here on; looked at behind the place phase)
    if (me.status != ST_OK) {
        if (meta.status == ST_OK) cp_first_tiles(first, sym + (size_t)mb_next * SYM_STRIDE, (uint32_t)tid, meta.ns);
        return;
    }
    const uint32_t ns = me.ns;
    const uint64_t o0 = me.o0;
    const uint32_t isize = me.isize;
    const uint32_t* const msym = sym + (size_t)mb * SYM_STRIDE;
    if (tid == 0) s_bad = 0u;
    SB[tid] = 0u; SB[tid + CP_THREADS] = 0u;                               // (in front of the first barrier of the place phase)
    if (tid < 2) SB[2048 + tid] = 0u;                                      // (the word behind the last one: read, never set)
    static_assert(2 * CP_THREADS == 2048, "two words of the start bitmap per thread");
    // ---- place: every symbol's output offset by a scan (four symbols per thread and tile), its cells ---------------------------------
    uint32_t run = 0;
    bool bad = false;
    auto load4 = [&](uint32_t i0) -> uint4 { return cp_load4(msym, i0, ns); };
    // the lengths of a thread's four symbols of a tile (0 behind the member's last symbol) and their sum
    auto lengths = [&](const uint4& ev, uint32_t i0, uint32_t (&len)[4]) -> uint32_t {
        const uint32_t e[4] = {ev.x, ev.y, ev.z, ev.w};
        uint32_t tl = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) { len[k] = i0 + (uint32_t)k < ns ? ((e[k] >> 31) ? 1u : ((e[k] >> 15) & 0xFFu) + 3u) : 0u; tl += len[k]; }
        return tl;
    };
    auto wave_scan = [&](uint32_t tl) -> uint32_t {
        uint32_t inc = tl;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)inc, d, 64); if (lane >= d) inc += o; }
        return inc;
    };
    // A symbol writes its FIRST cell only -- the byte, or the pointer of a match's first cell -- and sets that cell's bit in the start
    // bitmap; the other cells of the matches are filled in afterwards, cell by cell (below), from the nearest start at or in front of
    // them.  (Until round 5's last day every symbol wrote all its cells: per-symbol loops, lanes with matches of every length side by
    // side, and a serial pass over the long matches of a wave -- 5.2 ms per 1.83 GB where marks + fill take 2.0.)
    auto mark4 = [&](const uint4& ev, const uint32_t (&len)[4], uint32_t dst) {
        const uint32_t e[4] = {ev.x, ev.y, ev.z, ev.w};
        uint32_t word = 0xFFFFFFFFu, bits = 0u;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (len[k] == 0u) continue;
            if (e[k] >> 31) W[dst] = (uint16_t)(e[k] & 0xFFu);
            else {
                const uint32_t dist = (e[k] & 0x7FFFu) + 1u;
                if (dist > dst) bad = true;
                W[dst] = (uint16_t)(0x8000u | (dist - 1u));
            }
            if ((dst >> 5) != word) { if (bits) atomicOr(&SB[word], bits); word = dst >> 5; bits = 0u; }
            bits |= 1u << (dst & 31u);
            dst += len[k];
        }
        if (bits) atomicOr(&SB[word], bits);
    };
    {
        uint32_t par = 0;
        for (uint32_t t0 = 0; t0 < ns; t0 += 4u * CP_THREADS, par ^= 1u) {
            const uint32_t i0 = t0 + 4u * (uint32_t)tid;
            const uint4 ev = buf.t[0];
#pragma unroll
            for (int k = 0; k + 1 < IW_CP_PF; k++) buf.t[k] = buf.t[k + 1];
            if (t0 + (uint32_t)IW_CP_PF * 4u * CP_THREADS < ns) buf.t[IW_CP_PF - 1] = load4(i0 + (uint32_t)IW_CP_PF * 4u * CP_THREADS);   // (tile t + PF)
            uint32_t len[4];
            const uint32_t tl = lengths(ev, i0, len);
            const uint32_t inc = wave_scan(tl);
            if (lane == 63) wsum[par][wave] = inc;
            __syncthreads();
            uint32_t woff = 0, total = 0;
#pragma unroll
            for (int w = 0; w < CP_WAVES; w++) { const uint32_t v = wsum[par][w]; if (w < wave) woff += v; total += v; }
            if (run + total > isize) { bad = true; break; }               // (uniform)
            if (prof) { const unsigned long long n = __builtin_amdgcn_s_memtime(); t_scan += n - tb; tb = n; }
            if (cut != 4u) mark4(ev, len, run + woff + inc - tl);          // (4: the scans and their barriers alone)
            run += total;
        }
    }
    __syncthreads();                                                       // (every symbol's first cell and bit are there)
    if (run == isize && isize && cut != 4u && cut != 5u) {                 // (5: without the fill)
        // fill_cells: wave w takes cells [4096 w, 4096 (w + 1)), 64 per step, a cell per lane.  The symbol a cell belongs to begins at
        // the highest set bit at or below it -- in the step's own 64 bits, else where the last step's (or, for the wave's first step,
        // a look backwards) says; cell x of a match of distance d points d (x / d + 1) back (past the match, to the cell in front
        // of it that holds the same byte) as long as that fits the cell's 15 bits; a cell whose target holds the same byte as the
        // cell `dist` back is what makes an overlapping match (distance 1: a run of one byte) resolve in one hop instead of `length`.
        const unsigned long long le = lane == 63 ? ~0ull : (2ull << lane) - 1ull;
        uint32_t c0 = (uint32_t)wave * 4096u;
        uint32_t carry = 0u;
        if (wave && c0 < isize) {
            uint32_t p = (c0 >> 5) - 1u;
            while (SB[p] == 0u) p--;                                           // (word 0 holds bit 0: the member's first symbol)
            carry = 32u * p + 31u - (uint32_t)__clz((int)SB[p]);
        }
#if IW_CP_FILL_LEAN
        carry = (uint32_t)__builtin_amdgcn_readfirstlane((int)carry);
#endif
        const uint32_t c_end = min(isize, c0 + 4096u);
        constexpr uint32_t FS = IW_CP_FILL_STEPS;                              // steps (64 cells each) in flight: their LDS round trips overlap
        for (; c0 < c_end; c0 += 64u * FS) {
            unsigned long long m[FS];
            uint32_t st[FS], v[FS];
#pragma unroll
            for (uint32_t f = 0; f < FS; f++) {
                const uint32_t w0 = min((c0 >> 5) + 2u * f, 2048u);              // (behind the member's cells: the two zero words)
#if IW_CP_FILL_LEAN
                // (NOT MEASURED YET -- built on the round's last day without a device to run it on; tools/hwtests/build_iw_variants.sh
                //  lean="-DIW_CP_FILL_LEAN=1": the step's start bits as scalars, so that what depends on them alone -- the start
                //  carried into the next step -- is scalar code)
                m[f] = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)SB[w0 + 1u]) << 32) |
                       (uint32_t)__builtin_amdgcn_readfirstlane((int)SB[w0]);
#else
                m[f] = ((unsigned long long)SB[w0 + 1u] << 32) | SB[w0];
#endif
            }
#pragma unroll
            for (uint32_t f = 0; f < FS; f++) {
                const unsigned long long mj = m[f] & le;
                st[f] = mj ? c0 + 64u * f + 63u - (uint32_t)__clzll((long long)mj) : carry;
                if (m[f]) carry = c0 + 64u * f + 63u - (uint32_t)__clzll((long long)m[f]);
                v[f] = (uint32_t)W[st[f]];
            }
#pragma unroll
            for (uint32_t f = 0; f < FS; f++) {
                const uint32_t c = c0 + 64u * f + (uint32_t)lane;
                const uint32_t x = c - st[f], dist = (v[f] & 0x7FFFu) + 1u;
                uint32_t out16 = v[f];                                         // (cell x < dist of a match: the first cell's own pointer)
                if (__any(x >= dist)) {                                        // (some lane inside a match that overlaps itself: one step in four)
                    // x / dist: dist <= 257 here, and the quotient of two small integers survives the reciprocal's rounding
                    // ((x + 0.5) / dist is at least 0.5 / 257 away from an integer)
                    const uint32_t q = x < dist ? 0u : (uint32_t)(((float)x + 0.5f) * __builtin_amdgcn_rcpf((float)dist));
                    const uint32_t D = dist * (q + 1u);
                    out16 = 0x8000u | ((D <= 0x8000u ? D : dist) - 1u);
                }
                if (x != 0u && c < isize) {
                    if (!(v[f] & 0x8000u)) bad = true;                         // (a literal is one cell long)
                    W[c] = (uint16_t)out16;
                }
            }
        }
    }
    if (bad) atomicOr(&s_bad, 1u);
    // the next member's first symbols: on their way while this one's cells are resolved and written
    if (meta.status == ST_OK) cp_first_tiles(first, sym + (size_t)mb_next * SYM_STRIDE, (uint32_t)tid, meta.ns);
    __syncthreads();
    if (s_bad || run != isize) { if (tid == 0) wstatus[m] = s_bad ? 17u : run < isize ? 16u : 18u; return; }
    if (cut == 1u || cut == 4u || cut == 5u) { if (W[tid] == 0xFFFFu) wstatus[m] = ST_LANES; return; }
    if (prof) { const unsigned long long n = __builtin_amdgcn_s_memtime(); t_place = n - t_begin; tb = n; }
    // ---- resolve: pointer jumping; unit u (256 cells) is wave u % 16's, four stretches of 64 cells in flight per step ------------------------
    // (plain LDS accesses; the compiler may keep nothing of W in registers from one look at a cell to the next: the barrier below)
    // ---- resolve: pointer jumping; unit u (256 cells) is wave u % 16's, four cells per lane in flight -----------------------------------
    // (plain LDS accesses; the compiler may keep nothing of W in registers from one look at a unit to the next: the barrier below)
    const uint32_t n_units = (isize + CP_UNIT - 1u) / CP_UNIT;
    uint32_t pending = 0;                                               // bit j: unit wave + 16 j
    for (uint32_t j = 0; j < CP_UPW; j++) if ((uint32_t)wave + (uint32_t)CP_WAVES * j < n_units) pending |= 1u << j;
    while (pending) {
        uint32_t next = 0;
        n_rounds++;
        for (uint32_t rest = pending; rest; rest &= rest - 1u) {
            const uint32_t j = (uint32_t)__ffs((int)rest) - 1u;
            const uint32_t base = ((uint32_t)wave + (uint32_t)CP_WAVES * j) * CP_UNIT;
            n_ur++;
            __asm__ volatile("" ::: "memory");
            // (a member of 0xFF00 bytes is 255 whole units: the bounds of a cell are looked at for the member's last, partial unit only)
            const bool open = base + CP_UNIT <= isize ? resolve_unit<true>(W, base + (uint32_t)lane, isize)
                                                      : resolve_unit<false>(W, base + (uint32_t)lane, isize);
            if (__ballot(open)) next |= 1u << j;
        }
        pending = next;
        // (tried: what is open kept per stretch of 64 cells instead of per unit -- 2.4 x the sweeps, 1.8 x the time: a wave with little
        //  left sweeps it over and over while it waits for cells of other waves; a sleep behind a sweep that changed nothing -- 4 % slower
        //  whatever its length; two units per step in flight -- 6 - 10 % slower.  profiles/r05h_inflate_copy_place_ab.txt)
    }
    __asm__ volatile("" ::: "memory");
    if (cut == 2u) { if (W[tid] == 0xFFFFu) wstatus[m] = ST_LANES; return; }
    if (prof) { const unsigned long long n = __builtin_amdgcn_s_memtime(); t_res = n - tb; tb = n; }
    // ---- write: the wave's units, 16 bytes per lane and step (a unit = 16 lanes' worth: four units per step) -----------------------------
    uint8_t* const dst = out + o0;
    for (uint32_t j = 0; j < CP_UPW; j += 4u) {
        const uint32_t unit = (uint32_t)wave + (uint32_t)CP_WAVES * (j + ((uint32_t)lane >> 4));
        const uint32_t i = unit * CP_UNIT + 16u * ((uint32_t)lane & 15u);
        if (i >= isize) continue;
        const uint4 a = *reinterpret_cast<const uint4*>(&W[i]), b = *reinterpret_cast<const uint4*>(&W[i + 8]);
        uint4 v;
        v.x = __builtin_amdgcn_perm(a.y, a.x, 0x06040200u);
        v.y = __builtin_amdgcn_perm(a.w, a.z, 0x06040200u);
        v.z = __builtin_amdgcn_perm(b.y, b.x, 0x06040200u);
        v.w = __builtin_amdgcn_perm(b.w, b.z, 0x06040200u);
        if (i + 16u <= isize) __builtin_memcpy(dst + i, &v, 16);
        else {                                                             // (the member's last bytes: a loop the compiler leaves alone -- sixteen
#pragma clang loop unroll(disable)                                         //  predicated byte stores, four times over, kept 60 registers busy)
            for (uint32_t x = i; x < isize; x++) dst[x] = (uint8_t)W[x];
        }
    }
    if (prof && lane == 0) {
        atomicAdd(&prof[8], t_place); atomicAdd(&prof[9], t_res); atomicAdd(&prof[10], __builtin_amdgcn_s_memtime() - tb); atomicAdd(&prof[11], n_ur);
        atomicAdd(&prof[12], n_rounds); atomicMax(&prof[13], n_rounds); atomicAdd(&prof[14], 1ull);
        atomicAdd(&prof[15], t_scan); atomicAdd(&prof[16], t_cells); atomicAdd(&prof[17], t_long);
    }
}

// One workgroup per CU for the whole batch (the member's cells take the CU's LDS anyway): the members in strides of the grid, so that
// no CU waits for a workgroup to be dispatched between two members.
template <bool MEASURE>
__global__ __launch_bounds__(CP_THREADS) void k_inflate_copy(const uint32_t* __restrict__ sym, const uint32_t* __restrict__ n_sym,
                                                                        uint32_t* __restrict__ wstatus, const uint64_t* __restrict__ out_off,
                                                                        uint32_t m0, uint32_t n_batch, uint8_t* __restrict__ out, uint32_t cut,
                                                                        unsigned long long* __restrict__ prof)
{
    __shared__ uint16_t W[65536];
    __shared__ uint32_t wsum[2][CP_WAVES];
    __shared__ uint32_t SB[2048 + 2];                                      // one bit per cell, set where a symbol begins (+ two words that stay zero)
    __shared__ uint32_t s_bad;
    CpMeta meta = cp_meta(blockIdx.x, n_batch, m0, n_sym, wstatus, out_off);
    CpTiles first;
#pragma unroll
    for (int k = 0; k < IW_CP_PF; k++) first.t[k] = make_uint4(0u, 0u, 0u, 0u);
    if (meta.status == ST_OK) cp_first_tiles(first, sym + (size_t)blockIdx.x * SYM_STRIDE, threadIdx.x, meta.ns);
    for (uint32_t mb = blockIdx.x; mb < n_batch; mb += gridDim.x) {
#if IW_CP_PREFETCH
        copy_member<MEASURE>(W, SB, wsum, s_bad, mb, sym, n_sym, wstatus, out_off, m0, out, cut, prof, meta, first, mb + gridDim.x, n_batch);
#else
        meta = cp_meta(mb, n_batch, m0, n_sym, wstatus, out_off);
        if (meta.status == ST_OK) cp_first_tiles(first, sym + (size_t)mb * SYM_STRIDE, threadIdx.x, meta.ns);
        copy_member<MEASURE>(W, SB, wsum, s_bad, mb, sym, n_sym, wstatus, out_off, m0, out, cut, prof, meta, first, n_batch, n_batch);
#endif
        __syncthreads();                                                  // (every wave has written its bytes: the cells are the next member's)
    }
}

// ---- host side -------------------------------------------------------------------------------------------------------------
// The members of a run in batches of `batch` members (the symbol stream of a batch: batch x 256 KB of scratch): A over the batch,
// B over the batch.  d_wstatus / d_nsym: one word per member of the run (scratch of the context).
int gci_inflate_wave_run(gci_ctx* ctx, const uint8_t* d_raw, const uint64_t* d_member_pos, const uint64_t* d_out_off, uint32_t n_members,
                         uint8_t* d_out, uint64_t out_cap, uint32_t* d_wstatus)
{
    static const uint32_t batch_max = [] { const char* e = getenv("GCI_INFLATE_BATCH"); const int v = e ? atoi(e) : 16384; return (uint32_t)(v < 64 ? 64 : v); }();
    static const int waves_per_cu = [] { const char* e = getenv("GCI_INFLATE_WAVES"); return e ? atoi(e) : 0; }();
    static const uint32_t cut_a = [] { const char* e = getenv("GCI_IW_CUT_A"); return (uint32_t)(e ? atoi(e) : 0); }();   // (measurements)
    static const uint32_t cut_b = [] { const char* e = getenv("GCI_IW_CUT_B"); return (uint32_t)(e ? atoi(e) : 0); }();
    static const bool want_prof = [] { const char* e = getenv("GCI_IW_PROF"); return e && atoi(e) != 0; }();
    unsigned long long* d_prof = nullptr;
    if (want_prof) {
        const int stp = gci_ensure(ctx, ctx->inflate_prof, 24 * sizeof(unsigned long long));
        if (stp) return stp;
        d_prof = (unsigned long long*)ctx->inflate_prof.p;
        HIPCHK(hipMemsetAsync(d_prof, 0, 24 * sizeof(unsigned long long), ctx->stream));
    }
    // The members in batches (the symbol streams of a batch: batch x 256 KB of scratch), every other batch on a second stream with
    // scratch of its own: a wave takes a millisecond per member, so the last members of a batch leave most of the chip idle -- the
    // other stream's kernels move in as CUs fall free (and the copy kernel of one batch runs beside the decode of the next).
    static const bool copy_persistent = [] { const char* e = getenv("GCI_INFLATE_COPY_GRID"); return !(e && !strcmp(e, "members")); }();   // (A/B)
    // (two streams only when the HOST says its runtime has hardware queues to spare -- gci_bgzf_inflate_streams(ctx, 2), which
    // gci_amd calls when it placed GPU_MAX_HW_QUEUES=8 in the environment BEFORE the runtime started --: with the default four, the
    // second stream came to share a queue with the host's copy stream and every upload waited for an inflate.  The environment
    // string alone proves nothing: a runtime that was already running when the variable was set has four queues all the same.)
    static const int env_streams = [] { const char* e = getenv("GCI_INFLATE_STREAMS"); return e ? atoi(e) : 0; }();   // (A/B)
    const int n_streams = env_streams >= 1 ? (env_streams > 2 ? 2 : env_streams) : ctx->inflate_streams;
    const uint32_t batch = n_members < batch_max ? n_members : batch_max;
    const uint32_t n_batches = (n_members + batch - 1u) / batch;
    const bool two = n_streams == 2 && n_batches > 1u && !want_prof;
    // (scratch for a whole batch as soon as a call is not a small one: a file's first run is cut short, and growing the scratch behind
    // it -- synchronise, hipFree, hipMalloc -- stalled the second run of every file by 0.45 s)
    const uint32_t batch_alloc = n_members > 2048u ? batch_max : batch;
    int st = gci_ensure(ctx, ctx->inflate_sym, (size_t)batch_alloc * SYM_STRIDE * sizeof(uint32_t));
    if (st) return st;
    if (n_members > 2048u && n_streams == 2) {
        st = gci_ensure(ctx, ctx->inflate_sym2, (size_t)batch_alloc * SYM_STRIDE * sizeof(uint32_t));
        if (st) return st;
    }
    st = gci_ensure(ctx, ctx->inflate_nsym, (size_t)(n_members > 2048u && n_members < 262144u ? 262144u : n_members) * sizeof(uint32_t));
    if (st) return st;
    int cus = 0, per_cu = 0;
    HIPCHK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device));
    const bool measure = want_prof || cut_a || cut_b;
    HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_inflate_symbols<false>, 64, 0));
    if (waves_per_cu > 0 && waves_per_cu < per_cu) per_cu = waves_per_cu;
    const uint32_t resident = (uint32_t)(cus > 0 && per_cu > 0 ? cus * per_cu : 1024);
    st = gci_ensure(ctx, ctx->inflate_lists, (size_t)resident * 64u * (MAXS + 1u) * sizeof(uint32_t));
    if (st) return st;
    st = gci_ensure(ctx, ctx->inflate_next, (size_t)n_batches * sizeof(uint32_t));
    if (st) return st;
    HIPCHK(hipMemsetAsync(ctx->inflate_next.p, 0, (size_t)n_batches * sizeof(uint32_t), ctx->stream));
    if (two) {
        st = gci_ensure(ctx, ctx->inflate_sym2, (size_t)batch_alloc * SYM_STRIDE * sizeof(uint32_t));
        if (st) return st;
        st = gci_ensure(ctx, ctx->inflate_lists2, (size_t)resident * 64u * (MAXS + 1u) * sizeof(uint32_t));
        if (st) return st;
        if (!ctx->inflate_stream2) {
            HIPCHK(hipStreamCreateWithFlags(&ctx->inflate_stream2, hipStreamNonBlocking));
            HIPCHK(hipEventCreateWithFlags(&ctx->inflate_ev_in, hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&ctx->inflate_ev_out, hipEventDisableTiming));
        }
        HIPCHK(hipEventRecord(ctx->inflate_ev_in, ctx->stream));              // (the inputs, the status words, the counters: ready)
        HIPCHK(hipStreamWaitEvent(ctx->inflate_stream2, ctx->inflate_ev_in, 0));
    }
    uint32_t k = 0;
    for (uint32_t m0 = 0; m0 < n_members; m0 += batch, k++) {
        const uint32_t nb = n_members - m0 < batch ? n_members - m0 : batch;
        const bool second = two && (k & 1u);
        hipStream_t sm = second ? ctx->inflate_stream2 : ctx->stream;
        uint32_t* const symk = (uint32_t*)(second ? ctx->inflate_sym2.p : ctx->inflate_sym.p);
        uint32_t* const listk = (uint32_t*)(second ? ctx->inflate_lists2.p : ctx->inflate_lists.p);
        hipLaunchKernelGGL(measure ? k_inflate_symbols<true> : k_inflate_symbols<false>, dim3(nb < resident ? nb : resident), dim3(64), 0, sm, d_raw,
                           d_member_pos, d_out_off, out_cap, m0, nb, symk, (uint32_t*)ctx->inflate_nsym.p, d_wstatus, listk,
                           (uint32_t*)ctx->inflate_next.p + k, cut_a, d_prof);
        LAUNCHCHK("k_inflate_symbols");
        const uint32_t copy_grid = copy_persistent && cus > 0 ? (nb < (uint32_t)cus ? nb : (uint32_t)cus) : nb;
        hipLaunchKernelGGL(measure ? k_inflate_copy<true> : k_inflate_copy<false>, dim3(copy_grid), dim3(CP_THREADS), 0, sm, (const uint32_t*)symk,
                           (const uint32_t*)ctx->inflate_nsym.p, d_wstatus, d_out_off, m0, nb, d_out, cut_b, d_prof);
        LAUNCHCHK("k_inflate_copy");
    }
    if (two) {
        HIPCHK(hipEventRecord(ctx->inflate_ev_out, ctx->inflate_stream2));
        HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->inflate_ev_out, 0));       // (the lane decoder and the CRC check behind this: after both)
    }
    if (want_prof) {
        unsigned long long h[24];
        HIPCHK(hipMemcpyAsync(h, d_prof, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        const double a = (double)h[6] > 0 ? 100.0 / (double)h[6] : 0.0, b = (double)(h[8] + h[9] + h[10]) > 0 ? 100.0 / (double)(h[8] + h[9] + h[10]) : 0.0;
        fprintf(stderr, "iw prof: A %% of wave time: header %.1f stage %.1f pass1 %.1f stitch %.1f chain %.1f gather %.1f; chunks %llu (%.0f cycles each); "
                        "B %%: place %.1f (scan + barrier %.1f, cells %.1f, long matches %.1f) resolve %.1f write %.1f; unit-rounds per member %.1f, rounds per wave %.2f (max %llu)\n",
                a * h[0], a * h[1], a * h[2], a * h[3], a * h[4], a * h[5], h[7], h[7] ? (double)h[6] / (double)h[7] : 0.0, b * h[8], b * h[15], b * h[16], b * h[17], b * h[9], b * h[10],
                h[14] ? (double)h[11] / ((double)h[14] / 16.0) : 0.0, h[14] ? (double)h[12] / (double)h[14] : 0.0, h[13]);
    }
    return GCI_OK;
}

extern "C" int gci_bgzf_inflate_last_stats(gci_ctx* ctx, uint32_t h_counts[32])
{
    if (!ctx || !h_counts) return GCI_E_INVALID;
    for (int k = 0; k < 32; k++) h_counts[k] = 0;
    const uint32_t n = ctx->inflate_last_n;
    if (!n) return GCI_OK;
    std::vector<uint32_t> h(n);
    HIPCHK(hipMemcpyAsync(h.data(), ctx->inflate_wstatus.p, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    for (uint32_t v : h) h_counts[v < 31u ? v : 31u]++;
    return GCI_OK;
}

extern "C" int gci_bgzf_inflate_streams(gci_ctx* ctx, int n)
{
    if (!ctx || n < 1 || n > 2) return GCI_E_INVALID;
    ctx->inflate_streams = n;
    return GCI_OK;
}
