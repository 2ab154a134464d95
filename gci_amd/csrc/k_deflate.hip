// k_deflate.hip -- N2: the depth text as gzip members, written by the GPU straight from the track
// (write_depth, GCI.py:99-143: every base as f'{depth}\n' through gzip).
//
// The text of a depth track is runs of one short line repeated ("37\n37\n37\n..."): in DEFLATE terms w literals (the
// line) followed by matches of distance w.  So the encoder never materialises the text: a lane walks one tile
// (4096 bases) of the int32 track, collects its constant-depth runs, and emits per run the line's literals and
// ceil((n-1)w / 258) length/distance pairs with the FIXED Huffman code (RFC 1951, 3.2.6) -- about 13 bits per 258 bytes
// of text.  A tile is one deflate block closed by an empty stored block (byte alignment, as Z_SYNC_FLUSH does), so tile
// streams concatenate bytewise; 64 tiles -- one wave -- form one gzip member:
//     1f 8b 08 00 00000000 00 ff | tile 0 | ... | tile 63 | 03 00 | CRC-32 | ISIZE
// CRC-32 of text that is never written: CRCs are polynomials mod P over GF(2), crc(A||B) = crc(A) * x^(8|B|) + crc(B), so
// a run of n copies of a line is n - 1 such steps done by squaring (pairs (crc, x^(8 len)) multiply like 2x2 triangular
// matrices), a tile is the product of its runs and the member the ordered product of its 64 tiles (wave tree).
// Any multi-member gzip whose payload equals the reference's text is a valid .depth.gz; Python's gzip module checks
// every member's CRC and length when the consumers (utility/GCI_score.py:25-37) read it.
//
// Two passes over the track (4 B/base each): sizes + CRCs, then the bytes at their scanned offsets.
#include "gci_ctx.hpp"

namespace {

constexpr uint32_t CRC_POLY = 0xEDB88320u;      // reflected: bit 31 holds x^0, bit 0 holds x^31
constexpr uint32_t GF_ONE = 0x80000000u;        // the polynomial 1
constexpr int RUNS = 16;                        // runs a lane collects before the wave encodes them together
constexpr int MEMBER_TILES = 64;

__device__ __forceinline__ uint32_t gf_mul(uint32_t a, uint32_t b)          // a * b mod P
{
    uint32_t p = 0;
#pragma unroll 4
    for (int i = 0; i < 32; i++) {
        p ^= (a & (0x80000000u >> i)) ? b : 0u;                               // + b * x^i
        b = (b >> 1) ^ ((b & 1u) ? CRC_POLY : 0u);                            // b *= x
    }
    return p;
}

// (crc, x^(8 len)) of a byte string; strings concatenate as  (a.c, a.x) . (b.c, b.x) = (a.c * b.x + b.c, a.x * b.x)
struct CrcPair { uint32_t c, x; };
__device__ __forceinline__ CrcPair crc_cat(CrcPair a, CrcPair b) { return {gf_mul(a.c, b.x) ^ b.c, gf_mul(a.x, b.x)}; }

// one line of the text: decimal digits of v (v >= 0) and '\n', as bytes packed little-endian into 96 bits
struct Line { uint32_t lo, mid, hi; uint32_t w; };
__device__ __forceinline__ uint32_t line_byte(const Line& l, uint32_t k)
{
    const uint32_t word = k < 4 ? l.lo : k < 8 ? l.mid : l.hi;
    return (word >> (8u * (k & 3u))) & 0xFFu;
}
__device__ __forceinline__ Line make_line(uint32_t v)
{
    uint32_t nd = 1;
    for (uint32_t t = v; t >= 10u; t /= 10u) nd++;
    Line l{0u, 0u, 0u, nd + 1u};
    uint32_t t = v;
    for (uint32_t k = nd; k-- > 0;) {                                           // digit k (0 = most significant)
        const uint32_t dg = 0x30u + t % 10u;
        t /= 10u;
        const uint32_t sh = 8u * (k & 3u);
        if (k < 4) l.lo |= dg << sh; else if (k < 8) l.mid |= dg << sh; else l.hi |= dg << sh;
    }
    const uint32_t sh = 8u * (nd & 3u);
    if (nd < 4) l.lo |= 0x0Au << sh; else if (nd < 8) l.mid |= 0x0Au << sh; else l.hi |= 0x0Au << sh;
    return l;
}

__device__ __forceinline__ CrcPair crc_line(const Line& l)
{
    uint32_t c = 0xFFFFFFFFu, x = GF_ONE;
    for (uint32_t k = 0; k < l.w; k++) {
        c ^= line_byte(l, k);
        for (int b = 0; b < 8; b++) {
            c = (c >> 1) ^ ((c & 1u) ? CRC_POLY : 0u);
            x = (x >> 1) ^ ((x & 1u) ? CRC_POLY : 0u);                          // x^(8 w) alongside
        }
    }
    // c is the register of the string started at all-ones; the finalised CRC of a string S is reg(S) ^ ~0, and the
    // concatenation rule above holds for finalised CRCs
    return {c ^ 0xFFFFFFFFu, x};
}

// n >= 1 copies of a line of w bytes behind a string.  With X = x^(8 w) the CRC of n copies behind a string of CRC c is
// c * X^n + line_crc * G_n, G_n = 1 + X + ... + X^(n - 1) -- X^n and G_n depend on w and n only, and G_(a + b) = G_a * X^b + G_b.
// Round 3 put them together from a table of (X^(2^k), G_(2^k)) per line width: two products per set bit of n, three to apply
// them and one more for the string's own x^(8 len) -- ~16 products of 32 shift-and-add steps per run, and the size pass of
// the members (5.3 ms at genome scale, `profiles/r03t_bench_kernel_stats.csv`) was bound by exactly this arithmetic.  Round 5:
// n <= 4096 is two digits to the base 64, the table (made by the host when a context first deflates: 12.5 KB) holds
// (X^(j 64^k), G_(j 64^k)) for j <= 64, so X^n and G_n are ONE product each; the CRC of a line below 1024 is a table entry
// too; and the tile's x^(8 len) is made once, from the bits of its text length, behind the last run: four products per run.
constexpr uint32_t gf_mul_c(uint32_t a, uint32_t b)
{
    uint32_t p = 0;
    for (int i = 0; i < 32; i++) {
        p ^= (a & (0x80000000u >> i)) ? b : 0u;
        b = (b >> 1) ^ ((b & 1u) ? CRC_POLY : 0u);
    }
    return p;
}
constexpr int REP_W = 12;                                                       // line widths 0 .. 11
constexpr int REP_J = 65;                                                       // digits 0 .. 64 (n = 4096 = 64 * 64)
constexpr int LINE_TAB = 1024;                                                  // CRCs of the lines "0\n" .. "1023\n"
constexpr int POW_K = 17;                                                       // x^(8 2^k): a tile's text is < 2^17 bytes
static_assert(TILE <= 64 * 64, "a run is at most two digits to the base 64");
static_assert((uint64_t)TILE * (REP_W - 1) < (1ull << POW_K), "the text of a tile");
// layout of the table (uint32): [REP_W][2][REP_J] pairs (x, g) | LINE_TAB line CRCs | POW_K powers
constexpr size_t TAB_REP = 0, TAB_LINE = (size_t)REP_W * 2 * REP_J * 2, TAB_POW = TAB_LINE + LINE_TAB, TAB_WORDS = TAB_POW + POW_K;

struct CrcTab {
    const uint2* __restrict__ rep;
    const uint32_t* __restrict__ line;
    const uint32_t* __restrict__ pow8;
};

__device__ __forceinline__ uint32_t crc_append_lines(uint32_t front_c, uint32_t line_crc, uint32_t w, uint32_t n, const CrcTab& t)
{
    const uint2 lo = t.rep[(w * 2u + 0u) * REP_J + (n & 63u)], hi = t.rep[(w * 2u + 1u) * REP_J + (n >> 6)];
    const uint32_t xn = gf_mul(hi.x, lo.x);                                     // X^(64 a + b)
    const uint32_t gn = gf_mul(hi.y, lo.x) ^ lo.y;                              // G_(64 a) * X^b + G_b
    return gf_mul(front_c, xn) ^ gf_mul(line_crc, gn);
}

__device__ __forceinline__ uint32_t pow_x8(uint32_t len, const CrcTab& t)       // x^(8 len)
{
    uint32_t x = GF_ONE;
    for (uint32_t k = 0; len; len >>= 1, k++)
        if (len & 1u) x = gf_mul(x, t.pow8[k]);
    return x;
}

// ---- bit writer: DEFLATE packs bits LSB first; Huffman codes go in most-significant bit first, i.e. bit-reversed --------
struct BitOut {
    unsigned long long acc = 0;
    uint32_t nb = 0;            // bits in acc
    uint64_t total = 0;         // bits emitted so far
    uint8_t* out = nullptr;     // nullptr: count only
    __device__ __forceinline__ void put(uint32_t bits, uint32_t n)
    {
        acc |= (unsigned long long)bits << nb;
        nb += n;
        total += n;
        while (nb >= 8u) {
            if (out) *out++ = (uint8_t)acc;
            acc >>= 8;
            nb -= 8u;
        }
    }
    __device__ __forceinline__ void align()                                     // zero bits up to the next byte boundary
    {
        if (nb) put(0u, 8u - nb);
    }
};

__device__ __forceinline__ void put_literal(BitOut& o, uint32_t byte)           // bytes < 144: 8-bit code 0x30 + byte
{
    o.put(__brev(0x30u + byte) >> 24, 8u);
}

__device__ __forceinline__ void put_match(BitOut& o, uint32_t len, uint32_t dist)   // 3 <= len <= 258, 2 <= dist <= 12
{
    if (len == 258u) {
        o.put(__brev(0xC5u) >> 24, 8u);                                          // symbol 285: 8-bit code 0xC0 + 5
    } else {
        const uint32_t t = len - 3u;
        const uint32_t e = t < 8u ? 0u : (uint32_t)(31 - __clz((int)t)) - 2u;
        const uint32_t sym = 257u + 4u * e + (e ? (t >> e) : t);
        if (sym <= 279u) o.put(__brev(sym - 256u) >> 25, 7u);                    // 7-bit codes 0000000 .. 0010111
        else o.put(__brev(0xC0u + (sym - 280u)) >> 24, 8u);
        if (e) o.put(t & ((1u << e) - 1u), e);
    }
    uint32_t code, eb, ev;
    if (dist <= 4u) { code = dist - 1u; eb = 0; ev = 0; }
    else if (dist <= 8u) { code = 4u + ((dist - 5u) >> 1); eb = 1; ev = (dist - 5u) & 1u; }
    else { code = 6u + ((dist - 9u) >> 2); eb = 2; ev = (dist - 9u) & 3u; }
    o.put(__brev(code) >> 27, 5u);
    if (eb) o.put(ev, eb);
}

// the tokens of n copies of a line
__device__ __forceinline__ void put_run(BitOut& o, const Line& l, uint32_t n)
{
    const uint32_t w = l.w;
    for (uint32_t k = 0; k < w; k++) put_literal(o, line_byte(l, k));
    uint32_t rest = (n - 1u) * w;
    while (rest >= 3u) {
        uint32_t len = rest < 258u ? rest : 258u;
        if (rest - len != 0u && rest - len < 3u) len = rest - 3u;                // never leave 1 or 2 bytes behind
        put_match(o, len, w);
        rest -= len;
    }
    for (uint32_t k = 0; k < rest; k++) put_literal(o, line_byte(l, k));        // (n - 1) w < 3: n == 2, w == 2... as literals
}

// ---- the runs of every tile, found by a whole wave --------------------------------------------------------------------
// A lane that walks its tile alone reads 16 KiB in sixteen dependent round trips of addresses no other lane shares: the two
// encode passes took 50 ms at genome scale for 25 GB of reads (0.5 TB/s).  The walk is therefore done ONCE, by a wave per tile
// with coalesced 16-byte loads: a base starts a run where it differs from the base in front of it (the neighbour lane's last
// element through DPP), the starts are ranked with a wave scan and land in LDS, and the tile's runs -- {depth, length},
// ~20 of them in long-read data -- go to a list the encode passes read instead of the track.  A tile with more than
// RUN_MAX runs (short reads, pile-ups) keeps the lane's own walk.
constexpr int RUN_MAX = GCI_RUN_MAX;
constexpr uint32_t RUNS_WALK = GCI_RUNS_WALK;    // tile_nruns: not in the list, walk the track

// Where a tile's list lies.  by_track = false: entry 64 m + t (the lists k_depth_runs made for exactly these members).  by_track = true:
// the lists are those of a depth build (gci_build_opts.want_runs: k_tile_build wrote down the segments it had in registers), one
// per 4096-base tile of the LAYOUT -- a member that starts on a tile boundary finds its tile t at element / 4096 + t; one that
// does not has no lists (NO_LIST: its lanes walk the track).
constexpr uint64_t NO_LIST = ~0ull;
__device__ __forceinline__ uint64_t list_index(bool by_track, uint64_t n_lists, uint32_t m, uint32_t t, uint64_t elem0)
{
    if (!by_track) return (uint64_t)m * MEMBER_TILES + t;
    if (elem0 % TILE) return NO_LIST;
    const uint64_t i = elem0 / TILE + t;
    return i < n_lists ? i : NO_LIST;
}

__global__ __launch_bounds__(BLOCK) void k_depth_runs(const int32_t* __restrict__ depth, const uint64_t* __restrict__ member_elem,
                                                      const uint32_t* __restrict__ member_n, uint32_t n_members,
                                                      uint32_t* __restrict__ tile_nruns, int2* __restrict__ tile_runs,
                                                      bool by_track, uint64_t n_lists)
{
    __shared__ uint32_t starts[BLOCK / 64][RUN_MAX + 1];
    __shared__ int32_t vals[BLOCK / 64][RUN_MAX];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t g = (uint64_t)blockIdx.x * (BLOCK / 64) + wave;                       // tile = (member, tile of the member)
    const uint32_t m = (uint32_t)(g / MEMBER_TILES), t = (uint32_t)(g % MEMBER_TILES);
    if (m >= n_members) return;
    const uint32_t n_all = member_n[m], first = t * TILE;
    const uint32_t n = n_all > first ? min((uint32_t)TILE, n_all - first) : 0u;
    const uint64_t li = list_index(by_track, n_lists, m, t, member_elem[m]);
    if (li == NO_LIST) return;
    // behind a build that kept its lists only the tiles it left out (dense ones: more events than a wave has lanes) are looked at
    if (by_track && (n == 0 || tile_nruns[li] != RUNS_WALK)) return;
    if (n == 0) { if (lane == 0) tile_nruns[li] = 0u; return; }
    const int4* __restrict__ src = reinterpret_cast<const int4*>(depth + member_elem[m] + first);
    uint32_t total = 0;
    int32_t carry = 0;
#pragma unroll 4
    for (uint32_t k = 0; k < TILE / 256; k++) {
        const uint32_t e = 4u * (k * 64u + (uint32_t)lane);
        const int4 q = e < n ? src[k * 64u + lane] : make_int4(0, 0, 0, 0);              // (a tile's padding is readable: zeros)
        int32_t prev = __shfl_up(q.w, 1, 64);
        if (lane == 0) prev = carry;
        const bool s0 = e < n && (e == 0 || q.x != prev), s1 = e + 1 < n && q.y != q.x, s2 = e + 2 < n && q.z != q.y,
                   s3 = e + 3 < n && q.w != q.z;
        const uint32_t cnt = (uint32_t)s0 + (uint32_t)s1 + (uint32_t)s2 + (uint32_t)s3;
        const uint32_t inc = wave_inclusive<uint32_t>(cnt, lane);
        uint32_t r = total + inc - cnt;
        if (s0) { if (r < RUN_MAX) { starts[wave][r] = e; vals[wave][r] = q.x; } r++; }
        if (s1) { if (r < RUN_MAX) { starts[wave][r] = e + 1; vals[wave][r] = q.y; } r++; }
        if (s2) { if (r < RUN_MAX) { starts[wave][r] = e + 2; vals[wave][r] = q.z; } r++; }
        if (s3) { if (r < RUN_MAX) { starts[wave][r] = e + 3; vals[wave][r] = q.w; } r++; }
        total += (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
        carry = __builtin_amdgcn_readlane(q.w, 63);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (total > RUN_MAX) { if (lane == 0) tile_nruns[li] = RUNS_WALK; return; }
    if ((uint32_t)lane < total) {
        const uint32_t a = starts[wave][lane], b = (uint32_t)lane + 1 < total ? starts[wave][lane + 1] : n;
        tile_runs[li * RUN_MAX + lane] = make_int2(vals[wave][lane], (int)(b - a));
    }
    if (lane == 0) tile_nruns[li] = total;
}

// One wave = one member of up to 64 tiles; lane t = tile t.  PASS 1: tile_bytes[], member totals + CRC; PASS 2: bytes.
template <int PASS>
__global__ __launch_bounds__(64) void k_depth_deflate(const int32_t* __restrict__ depth, const uint64_t* __restrict__ member_elem,
                                                      const uint32_t* __restrict__ member_n, uint32_t n_members,
                                                      uint32_t* __restrict__ tile_bytes, uint32_t* __restrict__ member_bytes,
                                                      uint32_t* __restrict__ member_crc, uint32_t* __restrict__ member_isize,
                                                      const uint64_t* __restrict__ member_out, uint8_t* __restrict__ out, uint64_t cap,
                                                      const uint32_t* __restrict__ tile_nruns, const int2* __restrict__ tile_runs,
                                                      bool by_track, uint64_t n_lists, const uint32_t* __restrict__ crc_tab)
{
    __shared__ int32_t run_v[RUNS][64];
    __shared__ uint32_t run_n[RUNS][64];
    const uint32_t m = blockIdx.x;
    if (m >= n_members) return;
    const int lane = threadIdx.x;
    const uint64_t e0 = member_elem[m];
    const uint32_t n_all = member_n[m];
    const uint32_t first = (uint32_t)lane * TILE;
    const uint32_t n = n_all > first ? min((uint32_t)TILE, n_all - first) : 0u;     // elements of this lane's tile
    const int32_t* src = depth + e0 + first;

    BitOut o;
    uint64_t my_off = 0;
    if (PASS == 2) {
        // byte offset of this lane's tile stream inside the member: header + the tiles in front
        uint32_t s = tile_bytes[(size_t)m * MEMBER_TILES + lane], incl = s;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t v = (uint32_t)__shfl_up((int)incl, d, 64); if (lane >= d) incl += v; }
        my_off = member_out[m] + 10ull + (incl - s);
        const uint32_t total = (uint32_t)__shfl((int)incl, 63, 64) + 20u;
        if (member_out[m] + total > cap) return;                                      // caller sized the buffer from pass 1
        o.out = out + my_off;
        if (lane == 0) {
            uint8_t* h = out + member_out[m];
            h[0] = 0x1F; h[1] = 0x8B; h[2] = 8; h[3] = 0; h[4] = h[5] = h[6] = h[7] = 0; h[8] = 0; h[9] = 0xFF;
            uint8_t* t = h + total - 10u;                                             // 03 00 | crc | isize
            const uint32_t c = member_crc[m], z = member_isize[m];
            t[0] = 0x03; t[1] = 0x00;
            t[2] = (uint8_t)c; t[3] = (uint8_t)(c >> 8); t[4] = (uint8_t)(c >> 16); t[5] = (uint8_t)(c >> 24);
            t[6] = (uint8_t)z; t[7] = (uint8_t)(z >> 8); t[8] = (uint8_t)(z >> 16); t[9] = (uint8_t)(z >> 24);
        }
    }
    const CrcTab ct{reinterpret_cast<const uint2*>(crc_tab + TAB_REP), crc_tab + TAB_LINE, crc_tab + TAB_POW};
    uint32_t tile_c = 0u;                                                             // CRC of the tile's text so far
    uint32_t text_len = 0;
    if (n) o.put(2u, 3u);                                                             // BFINAL = 0, BTYPE = 01 (fixed codes)

    // walk: collect up to RUNS runs per lane, then all lanes encode their runs side by side
    uint32_t pos = 0;
    int32_t cur = 0;
    uint32_t cnt = 0;                                                                 // the open run
    bool done = n == 0;
    // the tile's runs from the list k_depth_runs made (the usual case), else the walk below
    const uint64_t li = tile_nruns ? list_index(by_track, n_lists, m, (uint32_t)lane, e0) : NO_LIST;
    const uint32_t listed = li != NO_LIST && n ? tile_nruns[li] : RUNS_WALK;
    const int2* __restrict__ my_runs = tile_runs + (li != NO_LIST ? li : 0ull) * RUN_MAX;
    uint32_t taken = 0;                                                               // elements the listed runs read so far cover
    while (__any(!done)) {
        int k = 0;
        if (listed != RUNS_WALK) {
            // a build's list may hold empty segments and neighbours of one depth (an interval ending where another begins): they
            // are put together here, so that either kind of list gives the same runs -- and the same bytes -- as the walk
            while (!done && k < RUNS) {                                               // (k < RUNS: the slot a closing run needs)
                if (pos == listed || taken == n) {
                    if (cnt) { run_v[k][lane] = cur; run_n[k][lane] = cnt; k++; cnt = 0; }
                    done = true;
                    break;
                }
                const int2 rr = my_runs[pos++];
                const uint32_t len = min((uint32_t)rr.y, n - taken);                  // (a member that ends inside the tile)
                if (len == 0u) continue;
                if (cnt && rr.x != cur) { run_v[k][lane] = cur; run_n[k][lane] = cnt; k++; cnt = 0; }
                cur = rr.x; cnt += len; taken += len;
            }
        }
        // one element into the open run; false: it would close a run and the buffer is full (the element stays unread)
        auto feed = [&](int32_t x) -> bool {
            if (cnt && x != cur) {
                if (k == RUNS) return false;
                run_v[k][lane] = cur; run_n[k][lane] = cnt; k++; cnt = 0;
            }
            cur = x; cnt++;
            return true;
        };
        while (listed == RUNS_WALK && !done && k < RUNS) {
            if (pos == n) {                                                           // close the last run
                if (cnt) { run_v[k][lane] = cur; run_n[k][lane] = cnt; k++; cnt = 0; }
                done = true;
                break;
            }
            if ((pos & 31u) == 0u && pos + 32u <= n) {
                // 32 elements per step, all eight 16-byte loads in flight together (the lanes of a wave read tiles 16 KB
                // apart: nothing coalesces, so the walk is bound by round trips -- one per step instead of eight)
                int4 q[8];
#pragma unroll
                for (int i = 0; i < 8; i++) q[i] = *reinterpret_cast<const int4*>(src + pos + 4 * i);
                bool full = false;
                uint32_t used = 0;
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const int32_t x[4] = {q[i].x, q[i].y, q[i].z, q[i].w};
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        if (!full) { if (feed(x[j])) used++; else full = true; }
                }
                pos += used;
                if (full) break;
            } else {
                if (!feed(src[pos])) break;
                pos++;
            }
        }
        const int kmax = k;
        // encode: run r of every lane in the same iteration (the heavy arithmetic stays converged)
        int wave_max = kmax;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) wave_max = max(wave_max, __shfl_xor(wave_max, d, 64));
        for (int r = 0; r < wave_max; r++) {
            if (r < kmax) {
                const uint32_t rn = run_n[r][lane];
                const uint32_t rv = (uint32_t)run_v[r][lane];
                const Line l = make_line(rv);
                if (PASS == 1) {
                    tile_c = crc_append_lines(tile_c, rv < (uint32_t)LINE_TAB ? ct.line[rv] : crc_line(l).c, l.w, rn, ct);
                    text_len += rn * l.w;
                }
                put_run(o, l, rn);
            }
        }
    }
    if (n) {
        o.put(0u, 7u);                                                                // end of block (symbol 256)
        o.put(0u, 3u);                                                                // empty stored block: BFINAL 0, BTYPE 00
        o.align();
        o.put(0x0000u, 16u);
        o.put(0xFFFFu, 16u);
    }
    if (PASS == 1) {
        const uint32_t bytes = (uint32_t)(o.total >> 3);
        tile_bytes[(size_t)m * MEMBER_TILES + lane] = bytes;
        uint32_t sum = bytes, len = text_len;
        CrcPair tile{tile_c, pow_x8(text_len, ct)};
        // ordered product of the 64 tiles (lane i absorbs lane i + d) and plain sums
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            CrcPair other{(uint32_t)__shfl_down((int)tile.c, d, 64), (uint32_t)__shfl_down((int)tile.x, d, 64)};
            const uint32_t os = (uint32_t)__shfl_down((int)sum, d, 64), ol = (uint32_t)__shfl_down((int)len, d, 64);
            if ((lane & (2 * d - 1)) == 0) { tile = crc_cat(tile, other); sum += os; len += ol; }
        }
        if (lane == 0) {
            member_bytes[m] = sum + 20u;
            member_crc[m] = tile.c;
            member_isize[m] = len;
        }
    }
}

// The CRC tables (layout above), made once per process on the host and copied into the context on its first deflate.
const uint32_t* host_crc_tab()
{
    static uint32_t tab[TAB_WORDS];
    static const bool made = [] {
        uint32_t xw = GF_ONE;                                                   // x^(8 w)
        for (int w = 0; w < REP_W; w++) {
            uint32_t base_x = xw, base_g = GF_ONE;                              // (X^m, G_m), m = 64^k
            for (int k = 0; k < 2; k++) {
                uint32_t x = GF_ONE, g = 0u;                                    // (X^(j m), G_(j m)), j = 0
                for (int j = 0; j < REP_J; j++) {
                    uint32_t* e = tab + TAB_REP + 2 * (((size_t)w * 2 + k) * REP_J + j);
                    e[0] = x; e[1] = g;
                    g = gf_mul_c(g, base_x) ^ base_g;                           // G_(a + m) = G_a X^m + G_m
                    x = gf_mul_c(x, base_x);
                }
                // m -> 64 m: entry j = 64 of this digit
                const uint32_t* e64 = tab + TAB_REP + 2 * (((size_t)w * 2 + k) * REP_J + 64);
                base_x = e64[0]; base_g = e64[1];
            }
            xw = gf_mul_c(xw, 0x00800000u);                                     // * x^8
        }
        for (uint32_t v = 0; v < (uint32_t)LINE_TAB; v++) {
            char txt[16];
            const int nc = snprintf(txt, sizeof txt, "%u\n", v);
            uint32_t c = 0xFFFFFFFFu;
            for (int i = 0; i < nc; i++) {
                c ^= (uint8_t)txt[i];
                for (int b = 0; b < 8; b++) c = (c >> 1) ^ ((c & 1u) ? CRC_POLY : 0u);
            }
            tab[TAB_LINE + v] = c ^ 0xFFFFFFFFu;
        }
        uint32_t x = 0x00800000u;                                               // x^8
        for (int k = 0; k < POW_K; k++) { tab[TAB_POW + k] = x; x = gf_mul_c(x, x); }
        return true;
    }();
    (void)made;
    return tab;
}

int ensure_crc_tab(gci_ctx* ctx)
{
    if (ctx->deflate_tab_ready) return GCI_OK;
    GCI_TRY(gci_ensure(ctx, ctx->deflate_tab, TAB_WORDS * sizeof(uint32_t)));
    if (hipMemcpyAsync(ctx->deflate_tab.p, host_crc_tab(), TAB_WORDS * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
        return GCI_E_HIP;
    ctx->deflate_tab_ready = true;
    return GCI_OK;
}

}  // namespace

// PASS 1.  d_member_elem[m]: element offset of member m's first base in the track (a multiple of 4); d_member_n[m]: its
// bases (<= 64 * 4096; a member never spans two contigs).  Writes d_tile_bytes[64 m + t], d_member_bytes / _crc / _isize.
extern "C" int gci_depth_deflate_size(gci_ctx* ctx, const int32_t* d_depth, const uint64_t* d_member_elem, const uint32_t* d_member_n,
                                      uint32_t n_members, uint32_t* d_tile_bytes, uint32_t* d_member_bytes, uint32_t* d_member_crc,
                                      uint32_t* d_member_isize)
{
    if (!ctx || (n_members && (!d_depth || !d_member_elem || !d_member_n || !d_tile_bytes || !d_member_bytes || !d_member_crc ||
                               !d_member_isize))) return GCI_E_INVALID;
    if (n_members == 0) return GCI_OK;
    const uint64_t n_tiles = (uint64_t)n_members * MEMBER_TILES;
    // the run lists: those the depth build of this very track kept (gci_build_opts.want_runs -- the track is then not read for
    // them; k_depth_runs only looks at the tiles the build left out), else made here from the track
    const bool from_build = ctx->build_runs_armed && ctx->build_runs_track != nullptr && ctx->build_runs_track == d_depth;
    ctx->build_runs_armed = false;                           // (said for one size call)
    if (!from_build) {
        GCI_TRY(gci_ensure(ctx, ctx->deflate_nruns, n_tiles * 4));
        GCI_TRY(gci_ensure(ctx, ctx->deflate_runs, n_tiles * RUN_MAX * sizeof(int2)));
    }
    uint32_t* nruns = (uint32_t*)(from_build ? ctx->build_nruns.p : ctx->deflate_nruns.p);
    int2* runs = (int2*)(from_build ? ctx->build_runs.p : ctx->deflate_runs.p);
    const uint64_t n_lists = from_build ? (uint64_t)ctx->n_tiles : n_tiles;
    GCI_TRY(ensure_crc_tab(ctx));
    ctx->deflate_members = n_members; ctx->deflate_key_depth = d_depth; ctx->deflate_key_elem = d_member_elem;
    ctx->deflate_from_build = from_build;
    hipLaunchKernelGGL(k_depth_runs, dim3((uint32_t)((n_tiles + BLOCK / 64 - 1) / (BLOCK / 64))), dim3(BLOCK), 0, ctx->stream, d_depth,
                       d_member_elem, d_member_n, n_members, nruns, runs, from_build, n_lists);
    LAUNCHCHK("k_depth_runs");
    hipLaunchKernelGGL(k_depth_deflate<1>, dim3(n_members), dim3(64), 0, ctx->stream, d_depth, d_member_elem, d_member_n, n_members,
                       d_tile_bytes, d_member_bytes, d_member_crc, d_member_isize, (const uint64_t*)nullptr, (uint8_t*)nullptr, 0ull,
                       (const uint32_t*)nruns, (const int2*)runs, from_build, n_lists, (const uint32_t*)ctx->deflate_tab.p);
    LAUNCHCHK("k_depth_deflate<1>");
    return GCI_OK;
}

// PASS 2.  d_member_out[m]: byte offset of member m in d_out (exclusive scan of d_member_bytes, done by the caller).
extern "C" int gci_depth_deflate_write(gci_ctx* ctx, const int32_t* d_depth, const uint64_t* d_member_elem, const uint32_t* d_member_n,
                                       uint32_t n_members, const uint32_t* d_tile_bytes, const uint32_t* d_member_crc,
                                       const uint32_t* d_member_isize, const uint64_t* d_member_out, uint8_t* d_out, uint64_t cap)
{
    if (!ctx || (n_members && (!d_depth || !d_member_elem || !d_member_n || !d_tile_bytes || !d_member_crc || !d_member_isize ||
                               !d_member_out || !d_out))) return GCI_E_INVALID;
    if (n_members == 0) return GCI_OK;
    GCI_TRY(ensure_crc_tab(ctx));
    // the run lists of the size call over the same track and members (else: the lanes walk the track)
    const bool same = ctx->deflate_members == n_members && ctx->deflate_key_depth == d_depth && ctx->deflate_key_elem == d_member_elem;
    const bool from_build = same && ctx->deflate_from_build && ctx->build_runs_track == d_depth;
    const bool listed = same && (from_build || !ctx->deflate_from_build);
    hipLaunchKernelGGL(k_depth_deflate<2>, dim3(n_members), dim3(64), 0, ctx->stream, d_depth, d_member_elem, d_member_n, n_members,
                       const_cast<uint32_t*>(d_tile_bytes), (uint32_t*)nullptr, const_cast<uint32_t*>(d_member_crc),
                       const_cast<uint32_t*>(d_member_isize), d_member_out, d_out, cap,
                       listed ? (const uint32_t*)(from_build ? ctx->build_nruns.p : ctx->deflate_nruns.p) : (const uint32_t*)nullptr,
                       (const int2*)(from_build ? ctx->build_runs.p : ctx->deflate_runs.p), from_build,
                       from_build ? (uint64_t)ctx->n_tiles : (uint64_t)n_members * MEMBER_TILES, (const uint32_t*)ctx->deflate_tab.p);
    LAUNCHCHK("k_depth_deflate<2>");
    if (from_build) { ctx->build_runs_track = nullptr; ctx->deflate_from_build = false; ctx->deflate_members = 0; }   // one shot
    return GCI_OK;
}

extern "C" int gci_depth_deflate_from_build(gci_ctx* ctx, const int32_t* d_depth)
{
    if (!ctx || !d_depth || ctx->build_runs_track == nullptr || ctx->build_runs_track != d_depth) return GCI_E_INVALID;
    ctx->build_runs_armed = true;
    return GCI_OK;
}
