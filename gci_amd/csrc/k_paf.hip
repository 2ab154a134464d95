// k_paf.hip -- K2: the PAF path of filter() (/root/reference/GCI.py:211-254, helpers :49-61 and :64-96) on the GPU.
//
//   text of every PAF file (one device buffer, files back to back)
//     -> line starts (universal newlines; count per 4096-byte tile, scan, write)
//     -> one lane per line: str.strip() + split('\t'), the twelve columns the reference reads, int() of eight of them,
//        target lookup, identity = nmatch / alnlen (IEEE f64), the mapq / identity filter (GCI.py:218-239)
//     -> the lines that pass ("hits") compacted in file order, files appended to one another (the reference never
//        resets its block table between files, GCI.py:214-215)
//     -> queries numbered by an open-addressing table on the 64-bit name hash (names confirmed on their bytes); the hits
//        of a query side by side (count, scan, scatter, then ordered by line)
//     -> per file i, one lane per query over its hits of files <= i (GCI.py:241-254): per target the union of the query
//        blocks (touching blocks merge), mean identity as a sequential f64 sum in file order, score = mean * covered / qlen
//        (qlen of the first block), best target by (score, target name), its interval the longest merged target block
//        (leftmost on ties) -> a compact gci_rec + where its name lies in the text.
// Same IEEE operations in the same order as the reference (the library is built with -fno-fast-math -ffp-contract=off).
// Errors are those of the native host filter (gci_paf_filter): the first line the reference would raise on.
#include "gci_ctx.hpp"
#include <stdlib.h>
#include <algorithm>
#include <string>
#include <vector>

namespace {

typedef gci_paf_hit PafHitD;             // include/gci_hip.h: it crosses the boundary in sharded runs

struct PafTargets {
    const int32_t* slot;      // open addressing on the name hash: target index or -1
    const uint64_t* hash;     // hash of target t
    const uint64_t* off;      // name bytes of target t: names[off[t] .. off[t + 1])
    const uint8_t* names;
    const int32_t* rank;      // position of target t in sorted(name) order: the tie break of GCI.py:252
    uint32_t mask;
};

__device__ __forceinline__ bool is_space(uint8_t c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == 0x0b || c == 0x0c; }

// a line starts at p: the first byte of the file, or behind '\n', or behind a '\r' that is not followed by '\n'
//
// k_paf_lines<false> counts the line starts of every 4096-byte tile, k_paf_lines<true> writes them at the scanned offsets.
// (Round 3 tested text[p - 1] and text[p] with byte loads, sixteen positions per lane: 0.5 TB/s for ONE sequential read.)  A lane
// takes sixteen bytes with one aligned 16-byte load -- tiles are laid over the text from the 16-byte boundary at or below the
// file's first byte --, the byte in front of them comes from the lane before it (the first lane of a wave reads it), and the
// sixteen tests run on registers.
template <bool WRITE>
__global__ __launch_bounds__(BLOCK) void k_paf_lines(const uint8_t* __restrict__ text, uint64_t lo, uint64_t hi,
                                                     uint32_t* __restrict__ tile_count, const uint32_t* __restrict__ tile_off,
                                                     uint64_t* __restrict__ starts)
{
    __shared__ uint32_t wtot[BLOCK / 64];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const uint32_t delta = (uint32_t)((uintptr_t)(text + lo) & 15u);
    const int64_t w0 = (int64_t)lo - (int64_t)delta + (int64_t)blockIdx.x * TILE + (int64_t)t * 16;     // first byte of this lane's window
    uint32_t d[4] = {0u, 0u, 0u, 0u};
    if (w0 >= 0 && (uint64_t)w0 + 16 <= hi) {
        const uint4 v = *reinterpret_cast<const uint4*>(text + w0);
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    } else {
        for (int i = 0; i < 16; i++) {
            const int64_t p = w0 + i;
            if (p >= (int64_t)lo && (uint64_t)p < hi) d[i >> 2] |= (uint32_t)text[p] << (8 * (i & 3));
        }
    }
    uint32_t prev = (uint32_t)__shfl_up((int)(d[3] >> 24), 1, 64);
    if (lane == 0) prev = w0 - 1 >= (int64_t)lo && (uint64_t)(w0 - 1) < hi ? text[w0 - 1] : 0u;
    uint32_t mask = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const uint32_t cur = (d[i >> 2] >> (8 * (i & 3))) & 0xFFu;
        const int64_t p = w0 + i;
        const bool in = p >= (int64_t)lo && (uint64_t)p < hi;
        const bool start = p == (int64_t)lo || prev == '\n' || (prev == '\r' && cur != '\n');
        if (in && start) mask |= 1u << i;
        prev = cur;
    }
    const uint32_t n = (uint32_t)__builtin_popcount(mask);
    if (!WRITE) {
        const uint32_t tot = wave_sum<uint32_t>(n);
        if (lane == 0) wtot[wave] = tot;
        __syncthreads();
        if (t == 0) tile_count[blockIdx.x] = wtot[0] + wtot[1] + wtot[2] + wtot[3];
    } else {
        const uint32_t inc = wave_inclusive<uint32_t>(n, lane);
        if (lane == 63) wtot[wave] = inc;
        __syncthreads();
        uint32_t w = tile_off[blockIdx.x] + inc - n;
        for (int k = 0; k < wave; k++) w += wtot[k];
        for (uint32_t m = mask; m; m &= m - 1) starts[w++] = (uint64_t)(w0 + (int64_t)__builtin_ctz(m));
    }
}

// Where the bytes of the text are read from: the text itself, or the copy of a stretch of it a workgroup holds in LDS
// (`origin` = the position of the copy's first byte).
struct GlobalSrc {
    static constexpr bool WIDE = false;
    const uint8_t* __restrict__ p;
    __device__ __forceinline__ uint8_t operator[](uint64_t i) const { return p[i]; }
    __device__ __forceinline__ unsigned long long load8(uint64_t) const { return 0; }
};
struct LdsSrc {
    static constexpr bool WIDE = true;                     // eight bytes at any address in one read (the copy has 16 bytes of slack)
    const uint8_t* l; uint64_t origin;
    __device__ __forceinline__ uint8_t operator[](uint64_t i) const { return l[(uint32_t)(i - origin)]; }
    __device__ __forceinline__ unsigned long long load8(uint64_t i) const
    {
        unsigned long long v;
        __builtin_memcpy(&v, l + (uint32_t)(i - origin), 8);
        return v;
    }
};

// Bytes of a source through an eight-byte window held in registers: a column of digits or a name costs one read of LDS per
// eight bytes instead of one per byte (the reads of a lane depend on each other, ~100 cycles apiece).  A narrow source reads
// through.
template <typename Src>
struct Window {
    const Src t;
    uint64_t base; unsigned long long w;
    __device__ __forceinline__ Window(const Src s, uint64_t at) : t(s), base(at), w(Src::WIDE ? s.load8(at) : 0ull) {}
    __device__ __forceinline__ uint8_t operator[](uint64_t i)
    {
        if (!Src::WIDE) return t[i];
        if (i - base >= 8) { base = i; w = t.load8(i); }                   // also when i < base (unsigned)
        return (uint8_t)(w >> (8 * (uint32_t)(i - base)));
    }
};

// Python's int() on a column: optional blanks, optional sign, digits with single underscores between them ('1_000' is
// 1000; '_1', '1_', '1__0' are not numbers); false = ValueError.  (A value beyond 63 bits is reported as malformed: the
// reference would go on with a big integer.)
template <typename Src>
__device__ __forceinline__ bool parse_int(const Src src, uint64_t a, uint64_t b, int64_t& v)
{
    Window<Src> text(src, a);
    while (a < b && is_space(text[a])) a++;
    if (Src::WIDE) { if (b > a && is_space(src[b - 1])) { do b--; while (b > a && is_space(src[b - 1])); } }
    else while (b > a && is_space(text[b - 1])) b--;
    bool neg = false;
    if (a < b && (text[a] == '+' || text[a] == '-')) { neg = text[a] == '-'; a++; }
    if (a >= b) return false;
    uint64_t x = 0;
    bool prev_digit = false;
    uint32_t nd = 0;                                   // eighteen digits cannot overflow: the test is for the ones behind them
    for (; a < b; a++) {
        const uint8_t c = text[a];
        if (c == '_') {
            if (!prev_digit || a + 1 >= b || src[a + 1] < '0' || src[a + 1] > '9') return false;
            prev_digit = false;
            continue;
        }
        if (c < '0' || c > '9') return false;
        prev_digit = true;
        if (++nd > 18 && x > (0x7fffffffffffffffULL - (uint64_t)(c - '0')) / 10) return false;
        x = x * 10 + (uint64_t)(c - '0');
    }
    v = neg ? -(int64_t)x : (int64_t)x;
    return true;
}

template <typename Src>
__device__ __forceinline__ uint64_t hash_src(const Src text, uint64_t at, uint32_t len)      // == gci_name_hash
{
    uint64_t acc = 0;
    for (uint32_t k = 0; k * 8 < len; k++) {
        uint64_t w = 0;
        if (Src::WIDE) {
            w = text.load8(at + k * 8);
            const uint32_t left = len - k * 8;
            if (left < 8) w &= (1ull << (8 * left)) - 1ull;
        } else {
            for (int b = 0; b < 8; b++) if (k * 8 + b < len) w |= (uint64_t)text[at + k * 8 + b] << (8 * b);
        }
        acc += gci_hash_word(w, k);
    }
    return gci_hash_finish(acc, len);
}

__device__ __forceinline__ bool bytes_equal(const uint8_t* __restrict__ a, const uint8_t* __restrict__ b, uint32_t n)
{
    for (uint32_t i = 0; i < n; i++) if (a[i] != b[i]) return false;
    return true;
}

// One line: str.strip() + split('\t'), the twelve columns, the target, the eight int()s, the filter (GCI.py:218-239).
// flag: 1 = the line passed the filter (h is valid), 0 = skipped.  An offending line reports (line number << 8 | -status) into
// *status with atomicMin: the FIRST such line of the file is what the reference raises on (its lines are read in order).
template <typename Src>
__device__ __forceinline__ uint32_t paf_line(const Src text, uint64_t a, uint64_t hi, unsigned long long line_no, const PafTargets& T,
                                             int map_qual, int mq_cutoff, double iden_percent, PafHitD& h,
                                             unsigned long long* __restrict__ status)
{
    // the end of the line is only looked for when fewer than thirteen columns turn up before it
    while (a < hi && text[a] != '\n' && text[a] != '\r' && is_space(text[a])) a++;               // lstrip
    uint64_t col[12], cend[12];
    int nc = 0;                                       // columns found so far - 1
    col[0] = a;
    uint64_t q = a;
    bool eol = false;
    if (Src::WIDE) {
        // eight bytes per step: the tabs and line ends among them by exact byte-wise zero detection (no false positives: the
        // per-byte add cannot carry into its neighbour)
        for (bool done = false; !done;) {
            if (q >= hi) { eol = true; break; }
            const unsigned long long x = text.load8(q);
            const uint32_t lim = hi - q < 8 ? (uint32_t)(hi - q) : 8u;
            auto zeros = [](unsigned long long v) __attribute__((always_inline)) {
                const unsigned long long k = 0x7f7f7f7f7f7f7f7fULL;
                return ~(((v & k) + k) | v | k);                         // 0x80 in every byte of v that is zero
            };
            unsigned long long m = zeros(x ^ 0x0909090909090909ULL) | zeros(x ^ 0x0a0a0a0a0a0a0a0aULL) | zeros(x ^ 0x0d0d0d0d0d0d0d0dULL);
            while (m) {
                const uint32_t b = (uint32_t)__builtin_ctzll(m) >> 3;
                m &= m - 1ull;
                if (b >= lim) break;
                if (((x >> (8 * b)) & 0xFFu) != '\t') { q += b; eol = true; done = true; break; }
                cend[nc] = q + b;
                if (nc == 11) { q += b; done = true; break; }            // column 11 ends at a tab: nothing behind it matters
                col[++nc] = q + b + 1;
            }
            if (!done) q += lim;
        }
    } else {
        for (;;) {
            if (q >= hi) { eol = true; break; }
            const uint8_t c = text[q];
            if (c == '\n' || c == '\r') { eol = true; break; }
            if (c == '\t') {
                cend[nc] = q;
                if (nc == 11) break;                      // column 11 ends at a tab: nothing behind it matters
                col[++nc] = q + 1;
            }
            q++;
        }
    }
    if (eol) {
        // str.strip(): blanks (tabs too) at the end of the line go before it is split
        uint64_t b = q;
        while (b > a && is_space(text[b - 1])) b--;
        while (nc > 0 && col[nc] > b) nc--;          // columns that were only trailing tabs
        cend[nc] = b;
        if (b == a) nc = 0;                           // ''.split('\t') == ['']: one empty column
    }
    auto fail = [&](int code) { atomicMin(status, (line_no << 8) | (unsigned long long)(uint8_t)(-code)); };
    if (nc < 5) { fail(GCI_E_MALFORMED); return 0; }                                            // col[5]: IndexError
    // target among the selected contigs (GCI.py:220)
    const uint32_t tlen = (uint32_t)(cend[5] - col[5]);
    const uint64_t th = hash_src(text, col[5], tlen);
    int32_t t = -1;
    for (uint32_t s = (uint32_t)(th ^ (th >> 29)) & T.mask;; s = (s + 1) & T.mask) {
        const int32_t c = T.slot[s];
        if (c < 0) break;
        if (T.hash[c] == th && T.off[c + 1] - T.off[c] == tlen) {
            bool same = true;
            Window<Src> name(text, col[5]);
            for (uint32_t i = 0; i < tlen; i++) same = same && T.names[T.off[c] + i] == name[col[5] + i];
            if (same) { t = c; break; }
        }
    }
    if (t < 0) return 0;
    if (nc < 11) { fail(GCI_E_MALFORMED); return 0; }                                           // IndexError further right
    int64_t qlen, qs, qe, ts, te, nmatch, alnlen, mapq;
    bool ok = parse_int(text, col[1], cend[1], qlen);
    ok = ok && parse_int(text, col[2], cend[2], qs);
    ok = ok && parse_int(text, col[3], cend[3], qe);
    ok = ok && parse_int(text, col[7], cend[7], ts);
    ok = ok && parse_int(text, col[8], cend[8], te);
    ok = ok && parse_int(text, col[9], cend[9], nmatch);
    ok = ok && parse_int(text, col[10], cend[10], alnlen);
    ok = ok && parse_int(text, col[11], cend[11], mapq);
    if (!ok) { fail(GCI_E_MALFORMED); return 0; }                                               // ValueError
    if (alnlen == 0) { fail(GCI_E_ZERO_DIV); return 0; }                                        // nmatch / alnlen
    const double identity = (double)nmatch / (double)alnlen;
    if (!(mapq >= map_qual && identity >= iden_percent)) return 0;
    h.qn_off = col[0]; h.qn_len = (uint32_t)(cend[0] - col[0]);
    h.qhash = hash_src(text, col[0], h.qn_len);
    h.qlen = qlen; h.qs = qs; h.qe = qe; h.ts = ts; h.te = te; h.identity = identity;
    h.t = t; h.hq = mapq >= mq_cutoff ? 1u : 0u; h.slot = 0;
    return 1;
}

// One lane per line, one workgroup per BLOCK consecutive lines.  Their bytes are one stretch of the text, [start of the first,
// start of the line behind the last): the workgroup copies it into LDS with aligned 16-byte loads -- ONE coalesced trip to memory
// -- and the lanes walk their lines there (round 3 walked ~150 bytes of global memory byte by byte per lane: 165 GB/s).  A
// stretch that does not fit (lines of kilobytes: cg:Z: tags) is walked in global memory as before.
#define PAF_LDS_BYTES 49152
__global__ __launch_bounds__(BLOCK) void k_paf_tokenise(const uint8_t* __restrict__ text, uint64_t hi, const uint64_t* __restrict__ starts,
                                                        uint32_t n_lines, uint64_t line_base, PafTargets T, int map_qual, int mq_cutoff,
                                                        double iden_percent, PafHitD* __restrict__ hit, uint32_t* __restrict__ flag,
                                                        unsigned long long* __restrict__ status)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t s_text[];
    const uint32_t i0 = blockIdx.x * BLOCK, i = i0 + threadIdx.x;
    const uint64_t s0 = starts[i0];
    const uint64_t s1 = i0 + BLOCK < n_lines ? starts[i0 + BLOCK] : hi;
    const uint32_t al = (uint32_t)((uintptr_t)(text + s0) & 15u);
    const uint64_t origin = s0 - al;                                   // (may lie in front of the text by < 16 bytes: never read)
    const bool in_lds = s1 - s0 + al <= PAF_LDS_BYTES;
    if (in_lds) {
        const uint32_t nv = (uint32_t)((s1 - s0 + al + 15) >> 4);
        for (uint32_t v = threadIdx.x; v < nv; v += BLOCK) {
            const uint64_t p = origin + 16ull * v;
            uint4 x = make_uint4(0u, 0u, 0u, 0u);
            if ((v > 0 || al == 0) && p + 16 <= s1) x = *reinterpret_cast<const uint4*>(text + p);
            else {
                uint32_t w[4] = {0u, 0u, 0u, 0u};
                for (int k = 0; k < 16; k++) if (p + k >= s0 && p + k < s1) w[k >> 2] |= (uint32_t)text[p + k] << (8 * (k & 3));
                x = make_uint4(w[0], w[1], w[2], w[3]);
            }
            *reinterpret_cast<uint4*>(s_text + 16u * v) = x;
        }
        __syncthreads();
    }
    if (i >= n_lines) return;
    PafHitD h;
    uint32_t f;
    // (a line ends in front of the next one's start: inside the stretch, so `s1` bounds the walk exactly as `hi` does)
    if (in_lds) f = paf_line(LdsSrc{s_text, origin}, starts[i], s1, line_base + i + 1, T, map_qual, mq_cutoff, iden_percent, h, status);
    else f = paf_line(GlobalSrc{text}, starts[i], hi, line_base + i + 1, T, map_qual, mq_cutoff, iden_percent, h, status);
    flag[i] = f;
    if (f) hit[i] = h;
}

__global__ __launch_bounds__(BLOCK) void k_paf_compact(const PafHitD* __restrict__ hit, const uint32_t* __restrict__ flag,
                                                       const uint32_t* __restrict__ pos, uint32_t n_lines, PafHitD* __restrict__ out)
{
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i < n_lines && flag[i]) out[pos[i]] = hit[i];
}

#define QEMPTY 0xFFFFFFFFu
// query table: slot = index of the hit that claimed it; per slot ONE 64-bit word -- the number of its hits (low half) and of its
// high-quality hits (high half) -- bumped by one returning atomic whose old value is the hit's place among the slot's hits: the
// scatter behind the scan needs no atomic of its own, and reads the slots and places as two dense arrays instead of one word out
// of every 88-byte hit.
__global__ __launch_bounds__(BLOCK) void k_paf_insert(const uint8_t* __restrict__ text, const PafHitD* __restrict__ hits, uint32_t first,
                                                      uint32_t n, uint32_t* __restrict__ table, uint32_t mask,
                                                      unsigned long long* __restrict__ count, uint32_t* __restrict__ slot_of,
                                                      uint32_t* __restrict__ place_of)
{
    const uint32_t i = first + blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const PafHitD h = hits[i];
    uint32_t s = (uint32_t)(h.qhash ^ (h.qhash >> 29)) & mask;
    for (;;) {
        uint32_t c = table[s];
        if (c == QEMPTY) {
            c = atomicCAS(table + s, QEMPTY, i);
            if (c == QEMPTY) break;
        }
        const PafHitD o = hits[c];
        if (o.qhash == h.qhash && o.qn_len == h.qn_len && bytes_equal(text + o.qn_off, text + h.qn_off, h.qn_len)) break;
        s = (s + 1) & mask;
    }
    slot_of[i] = s;
    place_of[i] = (uint32_t)atomicAdd(count + s, 1ull | ((unsigned long long)(h.hq ? 1u : 0u) << 32));
}

__global__ __launch_bounds__(BLOCK) void k_paf_scatter(const uint32_t* __restrict__ slot_of, const uint32_t* __restrict__ place_of, uint32_t n,
                                                       const uint32_t* __restrict__ start, uint32_t* __restrict__ order)
{
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    order[start[slot_of[i]] + place_of[i]] = i;
}

// union of closed-touching blocks [a, b), sorted in place: covered length and the longest merged block (leftmost on ties)
__device__ __forceinline__ void merge_span(int64_t* __restrict__ pa, int64_t* __restrict__ pb, uint32_t n, int64_t& covered, int64_t& bs,
                                           int64_t& be)
{
    for (uint32_t i = 1; i < n; i++) {                                   // insertion sort by (a, b): the lists are short
        const int64_t xa = pa[i], xb = pb[i];
        uint32_t j = i;
        while (j > 0 && (pa[j - 1] > xa || (pa[j - 1] == xa && pb[j - 1] > xb))) { pa[j] = pa[j - 1]; pb[j] = pb[j - 1]; j--; }
        pa[j] = xa; pb[j] = xb;
    }
    covered = 0;
    int64_t best = -1;
    bs = be = 0;
    int64_t lo = pa[0], hi = pb[0];
    for (uint32_t i = 1; i <= n; i++) {
        if (i < n && hi >= pa[i]) { hi = hi > pb[i] ? hi : pb[i]; continue; }
        covered += hi - lo;
        if (hi - lo > best) { best = hi - lo; bs = lo; be = hi; }
        if (i < n) { lo = pa[i]; hi = pb[i]; }
    }
}

// One lane per query (table slot) over its hits below `limit` (= the hits of files 0 .. i): GCI.py:241-254.
__global__ __launch_bounds__(BLOCK) void k_paf_score(const PafHitD* __restrict__ hits, const uint32_t* __restrict__ table, uint32_t n_slots,
                                                     const uint32_t* __restrict__ start, uint32_t* __restrict__ order, uint32_t limit,
                                                     const unsigned long long* __restrict__ hq, const int32_t* __restrict__ trank,
                                                     int64_t* __restrict__ pa, int64_t* __restrict__ pb, int sort_lists,
                                                     gci_rec* __restrict__ out, uint64_t* __restrict__ out_name_off,
                                                     uint32_t* __restrict__ n_out, unsigned long long* __restrict__ status)
{
    const uint32_t s = blockIdx.x * BLOCK + threadIdx.x;
    bool emit = false;
    gci_rec r;
    uint64_t name_off = 0;
    if (s < n_slots && table[s] != QEMPTY) {
        const uint32_t a0 = start[s], a1 = start[s + 1];
        if (sort_lists) {                                                // once: the scatter handed the slots out in any order
            for (uint32_t i = a0 + 1; i < a1; i++) {
                const uint32_t x = order[i];
                uint32_t j = i;
                while (j > a0 && order[j - 1] > x) { order[j] = order[j - 1]; j--; }
                order[j] = x;
            }
        }
        uint32_t m = a0;
        while (m < a1 && order[m] < limit) m++;                          // the query's hits in files 0 .. i
        if (m == a0 + 1) {
            // ONE line of this query so far -- nine queries in ten: its target, its target block, nothing to merge and nothing to
            // compare (the score only orders targets); the division by qlen of GCI.py:249 still raises on 0.  One read of the hit
            // instead of the dependent trips through the scratch lists below.
            const PafHitD x = hits[order[a0]];
            if (x.qlen == 0) atomicMin(status, ((unsigned long long)order[a0] << 8) | (unsigned)(-GCI_E_ZERO_DIV));
            else if ((x.te - x.ts > -1 && (x.ts > 0x7fffffffLL || x.ts < -0x80000000LL || x.te > 0x7fffffffLL || x.te < -0x80000000LL)) ||
                     x.qlen > 0x7fffffffLL || x.qlen < -0x80000000LL || x.qn_len > 0xFFFF)
                atomicMin(status, ((unsigned long long)table[s] << 8) | (unsigned)(-GCI_E_INVALID));
            else {
                const bool block = x.te - x.ts > -1;                  // (merge_span's "longest block" starts from length -1)
                r.name_hash = x.qhash; r.contig = x.t; r.start = block ? (int32_t)x.ts : 0; r.end = block ? (int32_t)x.te : 0;
                r.qlen = (int32_t)x.qlen; r.rec_idx = s; r.mapq = 0;
                r.flags = (uint8_t)(GCI_REC_PASS | ((hq[s] >> 32) ? GCI_REC_HQ : 0)); r.name_len = (uint16_t)x.qn_len;
                name_off = x.qn_off;
                emit = true;
            }
        } else if (m > a0) {
            bool have = false;
            double best_rank = 0;
            int32_t best_t = -1;
            int64_t best_s = 0, best_e = 0, best_qlen = 0;
            for (uint32_t a = a0; a < m; a++) {
                const int32_t t = hits[order[a]].t;
                bool dup = false;
                for (uint32_t c = a0; c < a; c++) dup = dup || hits[order[c]].t == t;
                if (dup) continue;
                uint32_t n_aln = 0;
                int64_t qlen = 0;
                double total = 0.0;
                for (uint32_t c = a; c < m; c++) {
                    const PafHitD& x = hits[order[c]];
                    if (x.t != t) continue;
                    if (n_aln == 0) qlen = x.qlen;                       // qlen of the first block
                    pa[a0 + n_aln] = x.qs; pb[a0 + n_aln] = x.qe;
                    total = total + x.identity;                          // file order, as sum() does
                    n_aln++;
                }
                int64_t covered, s0, e0;
                merge_span(pa + a0, pb + a0, n_aln, covered, s0, e0);
                if (qlen == 0) {                                         // aligned / qlen: ZeroDivisionError
                    atomicMin(status, ((unsigned long long)order[a] << 8) | (unsigned)(-GCI_E_ZERO_DIV));
                    have = false;
                    break;
                }
                const double rank = total / (double)n_aln * ((double)covered / (double)qlen);
                if (!have || rank > best_rank || (rank == best_rank && trank[t] > trank[best_t])) {
                    uint32_t k = 0;
                    for (uint32_t c = a; c < m; c++) {
                        const PafHitD& x = hits[order[c]];
                        if (x.t == t) { pa[a0 + k] = x.ts; pb[a0 + k] = x.te; k++; }
                    }
                    merge_span(pa + a0, pb + a0, k, covered, s0, e0);
                    have = true; best_rank = rank; best_t = t; best_s = s0; best_e = e0; best_qlen = qlen;
                }
            }
            if (have) {
                const PafHitD& q0 = hits[table[s]];
                if (best_s > 0x7fffffffLL || best_s < -0x80000000LL || best_e > 0x7fffffffLL || best_e < -0x80000000LL ||
                    best_qlen > 0x7fffffffLL || best_qlen < -0x80000000LL || q0.qn_len > 0xFFFF) {
                    atomicMin(status, ((unsigned long long)table[s] << 8) | (unsigned)(-GCI_E_INVALID));
                } else {
                    r.name_hash = q0.qhash; r.contig = best_t; r.start = (int32_t)best_s; r.end = (int32_t)best_e;
                    r.qlen = (int32_t)best_qlen; r.rec_idx = s; r.mapq = 0;
                    r.flags = (uint8_t)(GCI_REC_PASS | ((hq[s] >> 32) ? GCI_REC_HQ : 0)); r.name_len = (uint16_t)q0.qn_len;
                    name_off = q0.qn_off;
                    emit = true;
                }
            }
        }
    }
    // ONE returning atomic per workgroup on the output counter: a returning atomic on one address costs ~6 - 12 ns whoever asks, and
    // one per wave -- nearly every wave emits something -- was the kernel: 131 k of them in 0.77 ms for a 438 MB file, 8.4 M of
    // them in the 96 ms of a 20 GB one
    __shared__ uint32_t s_cnt[BLOCK / 64];
    __shared__ uint32_t s_base;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long bal = __ballot(emit);
    if (lane == 0) s_cnt[wave] = (uint32_t)__builtin_popcountll(bal);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tot = 0;
        for (int k = 0; k < BLOCK / 64; k++) tot += s_cnt[k];
        s_base = tot ? atomicAdd(n_out, tot) : 0u;
    }
    __syncthreads();
    if (emit) {
        uint32_t w = s_base + (uint32_t)__builtin_popcountll(bal & ((1ull << lane) - 1ull));
        for (int k = 0; k < wave; k++) w += s_cnt[k];
        r.rec_idx = w;
        out[w] = r;
        out_name_off[w] = name_off;
    }
}

}  // namespace

struct gci_paf_dev {
    gci_ctx* ctx = nullptr;
    std::vector<void*> recs, name_off;       // per file (device)
    std::vector<uint32_t> count;
};

// Stage A's result for the sharded run: per file the hits of this rank's byte range, dense and in line order (device).
struct gci_paf_hits {
    gci_ctx* ctx = nullptr;
    std::vector<void*> hits;
    std::vector<uint32_t> count;
};

static void paf_dev_release(gci_paf_dev* h)
{
    for (void* p : h->recs) if (p) (void)gci_dfree(p);
    for (void* p : h->name_off) if (p) (void)gci_dfree(p);
    delete h;
}

static void paf_hits_release(gci_paf_hits* h)
{
    for (void* p : h->hits) if (p) (void)gci_dfree(p);
    delete h;
}

namespace {

// Device scratch of one call.  The buffers belong to the context and are handed out in the order they are asked for -- the same
// order in every call with as many files -- so a second call of the same size allocates nothing: a step of configs[3] makes two
// such calls, and twenty hipMalloc + hipFree each (every hipFree a device-wide synchronisation) were part of every one.  A call
// whose scratch is beyond PAF_POOL_KEEP (a PAF of tens of GB) gives it back when it ends, as before.
#define PAF_POOL_KEEP (16ull << 30)
struct PafScratch {
    gci_ctx* ctx;
    size_t next = 0;
    explicit PafScratch(gci_ctx* c) : ctx(c) {}
    ~PafScratch()
    {
        size_t held = 0;
        for (const DevBuf& b : ctx->paf_pool) held += b.cap;
        // (GCI_PAF_POOL_KEEP_GB: a harness that repeats a pass over tens of GB keeps the scratch instead -- giving ~70 GB back to the
        // driver and asking for it again made single passes of bench.py --workload paf take 2 - 3 s instead of 0.22 s)
        static const size_t keep = [] { const char* e = getenv("GCI_PAF_POOL_KEEP_GB"); return e ? (size_t)atoll(e) << 30 : (size_t)PAF_POOL_KEEP; }();
        if (held <= keep) return;
        (void)hipStreamSynchronize(ctx->stream);
        for (DevBuf& b : ctx->paf_pool) { if (b.p) (void)gci_dfree(b.p); b.p = nullptr; b.cap = 0; }
    }
    void* alloc(size_t bytes)
    {
        if (next >= ctx->paf_pool.size()) ctx->paf_pool.emplace_back();
        DevBuf& b = ctx->paf_pool[next++];
        return gci_ensure(ctx, b, bytes ? bytes : 16) == GCI_OK ? b.p : nullptr;
    }
    // p leaves the pool: the caller owns it from here on (hipFree)
    void keep(void* p) { for (DevBuf& b : ctx->paf_pool) if (b.p == p) { b.p = nullptr; b.cap = 0; return; } }
};
}  // namespace

// The pooled scratch of the PAF filter given back to the driver (the pipeline calls this when the PAF stage of filter() is over: the
// BAM ingestion, the join and the depth build behind it allocate through the host's allocator, and up to 16 GB held here outside of
// it would be theirs to miss).  The next PAF call allocates again.
extern "C" int gci_paf_pool_release(gci_ctx* ctx)
{
    if (!ctx) return GCI_E_INVALID;
    bool any = false;
    for (const DevBuf& b : ctx->paf_pool) any = any || b.p;
    if (!any) return GCI_OK;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    for (DevBuf& b : ctx->paf_pool) { if (b.p) (void)gci_dfree(b.p); b.p = nullptr; b.cap = 0; }
    return GCI_OK;
}

namespace {

#define PAF_ALLOC(var, type, count)                                         \
    type* var = (type*)S.alloc(sizeof(type) * (size_t)(count));             \
    if (!var) return GCI_E_NOMEM

// the selected contigs on the device: hash table, names, rank in sorted(name) order
int paf_targets(gci_ctx* ctx, PafScratch& S, const char* const* targets, int n_targets, PafTargets& T)
{
    hipStream_t st = ctx->stream;
    uint32_t tslots = 16;
    while (tslots < 2u * (uint32_t)n_targets + 2u) tslots <<= 1;
    std::vector<int32_t> h_slot(tslots, -1), h_rank(n_targets > 0 ? n_targets : 1, 0);
    std::vector<uint64_t> h_hash(n_targets > 0 ? n_targets : 1, 0), h_off(n_targets + 1, 0);
    std::string names;
    for (int t = 0; t < n_targets; t++) {
        const std::string nm(targets[t]);
        h_off[t] = names.size();
        names += nm;
        h_hash[t] = gci_name_hash((const uint8_t*)nm.data(), (uint32_t)nm.size());
        for (uint32_t s = (uint32_t)(h_hash[t] ^ (h_hash[t] >> 29)) & (tslots - 1);; s = (s + 1) & (tslots - 1)) {
            if (h_slot[s] < 0) { h_slot[s] = t; break; }
            if (nm == targets[h_slot[s]]) break;                         // a name listed twice: the first index, as the host map does
        }
    }
    h_off[n_targets] = names.size();
    {
        std::vector<int32_t> idx(n_targets);
        for (int t = 0; t < n_targets; t++) idx[t] = t;
        std::sort(idx.begin(), idx.end(), [&](int32_t a, int32_t b) { return std::string(targets[a]) < std::string(targets[b]); });
        for (int k = 0; k < n_targets; k++) h_rank[idx[k]] = k;
        for (int k = 1; k < n_targets; k++)                              // equal names share a rank
            if (std::string(targets[idx[k]]) == std::string(targets[idx[k - 1]])) h_rank[idx[k]] = h_rank[idx[k - 1]];
    }
    PAF_ALLOC(d_slot, int32_t, tslots);
    PAF_ALLOC(d_thash, uint64_t, h_hash.size());
    PAF_ALLOC(d_toff, uint64_t, h_off.size());
    PAF_ALLOC(d_tnames, uint8_t, names.size() + 16);
    PAF_ALLOC(d_trank, int32_t, h_rank.size());
    HIPCHK(hipMemcpyAsync(d_slot, h_slot.data(), tslots * 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_thash, h_hash.data(), h_hash.size() * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_toff, h_off.data(), h_off.size() * 8, hipMemcpyHostToDevice, st));
    if (!names.empty()) HIPCHK(hipMemcpyAsync(d_tnames, names.data(), names.size(), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_trank, h_rank.data(), h_rank.size() * 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st));                        // the host vectors above may go out of scope safely
    T.slot = d_slot; T.hash = d_thash; T.off = d_toff; T.names = d_tnames; T.rank = d_trank; T.mask = tslots - 1;
    return GCI_OK;
}

// Stage A -- per file: lines -> hits, dense and in line order.  A line the reference raises on ends the stage: the files in
// front of it are complete (n_ok of them), pending_status / pending_line say what it was.
struct PafStageA {
    std::vector<PafHitD*> file_hits;
    std::vector<uint32_t> hits_upto;
    int n_ok = 0, pending_status = GCI_OK;
    uint64_t pending_line = 0;
};

int paf_stage_a(gci_ctx* ctx, PafScratch& S, const uint8_t* d_text, const uint64_t* h_file_end, int n_files, const PafTargets& T,
                int map_qual, int mq_cutoff, double iden_percent, PafStageA& A)
{
    hipStream_t st = ctx->stream;
    PAF_ALLOC(d_status, unsigned long long, 1);
    A.file_hits.assign(n_files, nullptr);
    A.hits_upto.assign(n_files + 1, 0);
    A.n_ok = n_files;
    for (int f = 0; f < n_files; f++) {
        const uint64_t lo = f ? h_file_end[f - 1] : 0, hi = h_file_end[f];
        if (hi < lo) return GCI_E_INVALID;
        A.hits_upto[f + 1] = A.hits_upto[f];
        if (hi == lo) continue;
        const uint64_t n_tiles64 = (hi - lo + 15 + TILE - 1) / TILE;          // (tiles start at the 16-byte boundary at or below `lo`)
        if (n_tiles64 > 0x7fffffffULL) return GCI_E_INVALID;
        const uint32_t n_tiles = (uint32_t)n_tiles64;
        PAF_ALLOC(d_tile, uint32_t, n_tiles + 1);
        PAF_ALLOC(d_blk, uint32_t, n_tiles / TILE + 2);
        hipLaunchKernelGGL(k_paf_lines<false>, dim3(n_tiles), dim3(BLOCK), 0, st, d_text, lo, hi, d_tile, (const uint32_t*)nullptr,
                           (uint64_t*)nullptr);
        LAUNCHCHK("k_paf_lines<count>");
        int r = device_exclusive_scan<uint32_t, uint32_t>(ctx, d_tile, d_tile, d_blk, (int64_t)n_tiles, true);
        if (r) return r;
        uint32_t n_lines = 0;
        HIPCHK(hipMemcpyAsync(&n_lines, d_tile + n_tiles, 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        if (n_lines == 0) continue;
        PAF_ALLOC(d_starts, uint64_t, n_lines);
        hipLaunchKernelGGL(k_paf_lines<true>, dim3(n_tiles), dim3(BLOCK), 0, st, d_text, lo, hi, (uint32_t*)nullptr, (const uint32_t*)d_tile,
                           d_starts);
        LAUNCHCHK("k_paf_lines<write>");
        PAF_ALLOC(d_hit, PafHitD, n_lines);
        PAF_ALLOC(d_flag, uint32_t, n_lines + 1);
        PAF_ALLOC(d_blk2, uint32_t, n_lines / TILE + 2);
        HIPCHK(hipMemsetAsync(d_status, 0xFF, 8, st));
        hipLaunchKernelGGL(k_paf_tokenise, dim3((n_lines + BLOCK - 1) / BLOCK), dim3(BLOCK), PAF_LDS_BYTES + 16, st, d_text, hi, (const uint64_t*)d_starts,
                           n_lines, (uint64_t)0, T, map_qual, mq_cutoff, iden_percent, d_hit, d_flag, d_status);
        LAUNCHCHK("k_paf_tokenise");
        unsigned long long h_status = 0;
        HIPCHK(hipMemcpyAsync(&h_status, d_status, 8, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        if (h_status != ~0ull) {
            // The reference reads and scores file after file: an error in the scoring of an EARLIER file comes first.
            // Remember this one; the caller scores the files before it.
            A.pending_status = -(int)(h_status & 0xFF);
            A.pending_line = h_status >> 8;
            A.n_ok = f;
            break;
        }
        PAF_ALLOC(d_pos, uint32_t, n_lines + 1);
        r = device_exclusive_scan<uint32_t, uint32_t>(ctx, d_flag, d_pos, d_blk2, (int64_t)n_lines, true);
        if (r) return r;
        uint32_t n_hits = 0;
        HIPCHK(hipMemcpyAsync(&n_hits, d_pos + n_lines, 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        if ((uint64_t)A.hits_upto[f] + n_hits > 0x7fffffffULL) return GCI_E_INVALID;
        A.hits_upto[f + 1] = A.hits_upto[f] + n_hits;
        if (n_hits == n_lines) A.file_hits[f] = d_hit;                    // every line passed: nothing to squeeze out
        else if (n_hits) {
            PAF_ALLOC(d_dense, PafHitD, n_hits);
            hipLaunchKernelGGL(k_paf_compact, dim3((n_lines + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, st, (const PafHitD*)d_hit,
                               (const uint32_t*)d_flag, (const uint32_t*)d_pos, n_lines, d_dense);
            LAUNCHCHK("k_paf_compact");
            A.file_hits[f] = d_dense;
        }
    }
    for (int f = A.n_ok; f < n_files; f++) A.hits_upto[f + 1] = A.hits_upto[A.n_ok];
    return GCI_OK;
}

// Stage B -- the hits of files 0 .. n_ok - 1 in one array (file after file, every file in line order; qn_off relative to
// d_names): queries numbered, their hits side by side, and per file the queries scored over the hits so far.
int paf_stage_b(gci_ctx* ctx, PafScratch& S, const uint8_t* d_names, PafHitD* d_hits, const std::vector<uint32_t>& hits_upto, int n_ok,
                const int32_t* d_trank, gci_paf_dev* H)
{
    hipStream_t st = ctx->stream;
    const uint32_t total = hits_upto[n_ok];
    if (total == 0) return GCI_OK;
    if (total > (1u << 29)) return GCI_E_INVALID;
    PAF_ALLOC(d_status, unsigned long long, 1);
    PAF_ALLOC(d_n, uint32_t, 4);
    uint32_t n_slots = 1024;
    while (n_slots < 2ull * total) n_slots <<= 1;
    PAF_ALLOC(d_table, uint32_t, n_slots);
    PAF_ALLOC(d_count, unsigned long long, n_slots + 1);
    PAF_ALLOC(d_start, uint32_t, n_slots + 1);
    PAF_ALLOC(d_slot_of, uint32_t, total);
    PAF_ALLOC(d_place_of, uint32_t, total);
    PAF_ALLOC(d_order, uint32_t, total);
    PAF_ALLOC(d_blk3, uint32_t, n_slots / TILE + 2);
    PAF_ALLOC(d_pa, int64_t, total);
    PAF_ALLOC(d_pb, int64_t, total);
    HIPCHK(hipMemsetAsync(d_table, 0xFF, 4ull * n_slots, st));
    HIPCHK(hipMemsetAsync(d_count, 0, 8ull * (n_slots + 1), st));
    hipLaunchKernelGGL(k_paf_insert, dim3((total + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, st, d_names, (const PafHitD*)d_hits, 0u, total, d_table,
                       n_slots - 1, d_count, d_slot_of, d_place_of);
    LAUNCHCHK("k_paf_insert");
    // (the scan reads the low half of every counter: the number of hits)
    int rc = device_exclusive_scan<unsigned long long, uint32_t>(ctx, d_count, d_start, d_blk3, (int64_t)n_slots, true);
    if (rc) return rc;
    hipLaunchKernelGGL(k_paf_scatter, dim3((total + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, st, (const uint32_t*)d_slot_of,
                       (const uint32_t*)d_place_of, total, (const uint32_t*)d_start, d_order);
    LAUNCHCHK("k_paf_scatter");
    HIPCHK(hipMemsetAsync(d_status, 0xFF, 8, st));
    bool lists_sorted = false;
    for (int f = 0; f < n_ok; f++) {
        const uint32_t limit = hits_upto[f + 1];
        if (limit == 0) continue;
        void *p_recs = nullptr, *p_off = nullptr;
        if (gci_dmalloc(ctx->device, &p_recs, sizeof(gci_rec) * (size_t)limit) != hipSuccess) return GCI_E_NOMEM;
        H->recs[f] = p_recs;
        if (gci_dmalloc(ctx->device, &p_off, 8ull * limit) != hipSuccess) return GCI_E_NOMEM;
        H->name_off[f] = p_off;
        HIPCHK(hipMemsetAsync(d_n, 0, 4, st));
        hipLaunchKernelGGL(k_paf_score, dim3((n_slots + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, st, (const PafHitD*)d_hits, (const uint32_t*)d_table,
                           n_slots, (const uint32_t*)d_start, d_order, limit, (const unsigned long long*)d_count, d_trank, d_pa, d_pb,
                           lists_sorted ? 0 : 1, (gci_rec*)p_recs, (uint64_t*)p_off, d_n, d_status);
        LAUNCHCHK("k_paf_score");
        lists_sorted = true;
        uint32_t n_q = 0;
        unsigned long long h_status = 0;
        HIPCHK(hipMemcpyAsync(&n_q, d_n, 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(&h_status, d_status, 8, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        if (h_status != ~0ull) return -(int)(h_status & 0xFF);
        H->count[f] = n_q;
    }
    return GCI_OK;
}

}  // namespace

extern "C" int gci_paf_filter_device(gci_ctx* ctx, const uint8_t* d_text, const uint64_t* h_file_end, int n_files,
                                     const char* const* targets, int n_targets, int map_qual, int mq_cutoff, double iden_percent,
                                     gci_paf_dev** out, uint64_t* err_line)
{
    if (!ctx || !out || n_files < 0 || (n_files && (!d_text || !h_file_end)) || (n_targets && !targets)) return GCI_E_INVALID;
    *out = nullptr;
    if (err_line) *err_line = 0;
    hipStream_t st = ctx->stream;
    PafScratch S(ctx);
    PafTargets T;
    int rc = paf_targets(ctx, S, targets, n_targets, T);
    if (rc) return rc;
    PafStageA A;
    rc = paf_stage_a(ctx, S, d_text, h_file_end, n_files, T, map_qual, mq_cutoff, iden_percent, A);
    if (rc) return rc;
    auto pending = [&]() { if (err_line) *err_line = A.pending_line; return A.pending_status; };
    gci_paf_dev* H = new (std::nothrow) gci_paf_dev();
    if (!H) return GCI_E_NOMEM;
    H->ctx = ctx;
    H->recs.assign(n_files, nullptr); H->name_off.assign(n_files, nullptr); H->count.assign(n_files, 0);
    const uint32_t total = A.hits_upto[A.n_ok];
    if (total) {
        // one array of all hits, files in command-line order (the hits of the only file with any are that array)
        int with_hits = 0, only = -1;
        for (int f = 0; f < A.n_ok; f++) if (A.file_hits[f]) { with_hits++; only = f; }
        PafHitD* d_hits = with_hits == 1 ? A.file_hits[only] : (PafHitD*)S.alloc(sizeof(PafHitD) * (size_t)total);
        if (!d_hits) { paf_dev_release(H); return GCI_E_NOMEM; }
        for (int f = 0; f < A.n_ok && with_hits != 1; f++)
            if (A.file_hits[f]) {
                hipError_t e = hipMemcpyAsync(d_hits + A.hits_upto[f], A.file_hits[f],
                                              sizeof(PafHitD) * (size_t)(A.hits_upto[f + 1] - A.hits_upto[f]), hipMemcpyDeviceToDevice, st);
                if (e != hipSuccess) { paf_dev_release(H); return gci_fail(ctx, e, "hipMemcpyAsync(hits)"); }
            }
        rc = paf_stage_b(ctx, S, d_text, d_hits, A.hits_upto, A.n_ok, T.rank, H);
        if (rc) { paf_dev_release(H); return rc; }                       // (a scoring error of an earlier file comes before a pending line)
    }
    if (A.pending_status != GCI_OK) { paf_dev_release(H); return pending(); }
    *out = H;
    return GCI_OK;
}

// ---- the same in two halves, for runs that shard a PAF file by byte range (gci_amd/shard.py: ShardedPaf) -------------------
// Stage A over this rank's range of every file (the bytes of the ranges back to back in d_text): the lines that pass, as
// 80-byte hits whose qn_off points into d_text.  A line the reference would raise on is reported as by
// gci_paf_filter_device, but *err_line counts from the start of the RANGE: the sharded caller does not interpret it -- all
// ranks fall back to the whole files, where the reference's exception comes out exactly.
extern "C" int gci_paf_hits_device(gci_ctx* ctx, const uint8_t* d_text, const uint64_t* h_file_end, int n_files,
                                   const char* const* targets, int n_targets, int map_qual, int mq_cutoff, double iden_percent,
                                   gci_paf_hits** out, uint64_t* err_line)
{
    if (!ctx || !out || n_files < 0 || (n_files && (!d_text || !h_file_end)) || (n_targets && !targets)) return GCI_E_INVALID;
    *out = nullptr;
    if (err_line) *err_line = 0;
    PafScratch S(ctx);
    PafTargets T;
    int rc = paf_targets(ctx, S, targets, n_targets, T);
    if (rc) return rc;
    PafStageA A;
    rc = paf_stage_a(ctx, S, d_text, h_file_end, n_files, T, map_qual, mq_cutoff, iden_percent, A);
    if (rc) return rc;
    if (A.pending_status != GCI_OK) { if (err_line) *err_line = A.pending_line; return A.pending_status; }
    gci_paf_hits* H = new (std::nothrow) gci_paf_hits();
    if (!H) return GCI_E_NOMEM;
    H->ctx = ctx;
    H->hits.assign(n_files, nullptr); H->count.assign(n_files, 0);
    for (int f = 0; f < n_files; f++) {
        H->count[f] = A.hits_upto[f + 1] - A.hits_upto[f];
        if (A.file_hits[f]) { H->hits[f] = A.file_hits[f]; S.keep(A.file_hits[f]); }
    }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    *out = H;
    return GCI_OK;
}

extern "C" uint64_t gci_paf_hits_count(const gci_paf_hits* h, int file)
{
    return h && file >= 0 && (size_t)file < h->count.size() ? h->count[file] : 0;
}

// copies file `file`'s hits (gci_paf_hits_count() entries of GCI_PAF_HIT_BYTES) into the caller's device buffer
extern "C" int gci_paf_hits_export(const gci_paf_hits* h, int file, uint8_t* d_hits)
{
    if (!h || !h->ctx || file < 0 || (size_t)file >= h->count.size()) return GCI_E_INVALID;
    gci_ctx* ctx = h->ctx;
    const size_t n = h->count[file];
    if (n == 0) return GCI_OK;
    if (!d_hits) return GCI_E_INVALID;
    HIPCHK(hipMemcpyAsync(d_hits, h->hits[file], n * sizeof(PafHitD), hipMemcpyDeviceToDevice, ctx->stream));
    return GCI_OK;
}

extern "C" int gci_paf_hits_free(gci_paf_hits* h)
{
    if (!h) return GCI_OK;
    if (h->ctx) (void)hipStreamSynchronize(h->ctx->stream);
    paf_hits_release(h);
    return GCI_OK;
}

// Stage B over hits gathered from every rank: d_hits = the hits of the queries this rank owns, file after file
// (h_hits_upto[f] = first hit of file f, h_hits_upto[n_files] = their number), every file in line order, qn_off relative to
// d_names.  d_hits is modified (the query slots).  -> the handle gci_paf_filter_device returns (name offsets relative to d_names).
extern "C" int gci_paf_score_device(gci_ctx* ctx, const uint8_t* d_names, uint8_t* d_hits, const uint32_t* h_hits_upto, int n_files,
                                    const char* const* targets, int n_targets, gci_paf_dev** out)
{
    if (!ctx || !out || n_files < 0 || !h_hits_upto || (n_targets && !targets)) return GCI_E_INVALID;
    *out = nullptr;
    std::vector<uint32_t> upto(h_hits_upto, h_hits_upto + n_files + 1);
    for (int f = 0; f < n_files; f++) if (upto[f + 1] < upto[f]) return GCI_E_INVALID;
    if (upto[n_files] && (!d_hits || !d_names)) return GCI_E_INVALID;
    PafScratch S(ctx);
    PafTargets T;
    int rc = paf_targets(ctx, S, targets, n_targets, T);
    if (rc) return rc;
    gci_paf_dev* H = new (std::nothrow) gci_paf_dev();
    if (!H) return GCI_E_NOMEM;
    H->ctx = ctx;
    H->recs.assign(n_files, nullptr); H->name_off.assign(n_files, nullptr); H->count.assign(n_files, 0);
    rc = paf_stage_b(ctx, S, d_names, (PafHitD*)d_hits, upto, n_files, T.rank, H);
    if (rc) { paf_dev_release(H); return rc; }
    *out = H;
    return GCI_OK;
}

extern "C" uint64_t gci_paf_dev_count(const gci_paf_dev* h, int file)
{
    return h && file >= 0 && (size_t)file < h->count.size() ? h->count[file] : 0;
}

// copies file `file`'s records and name offsets into the caller's device buffers (gci_paf_dev_count() entries each)
namespace {
// (a device-to-device hipMemcpyAsync of 7 GB goes at 180 GB/s here -- the copy engines --, a kernel at the memory's rate)
__global__ __launch_bounds__(BLOCK) void k_copy8(const uint2* __restrict__ src, uint2* __restrict__ dst, uint64_t n)
{
    for (uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (uint64_t)gridDim.x * BLOCK) dst[i] = src[i];
}
__global__ __launch_bounds__(BLOCK) void k_copy16(const uint4* __restrict__ src, uint4* __restrict__ dst, uint64_t n)
{
    for (uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (uint64_t)gridDim.x * BLOCK) dst[i] = src[i];
}
}  // namespace

extern "C" int gci_paf_dev_export(const gci_paf_dev* h, int file, gci_rec* d_recs, uint64_t* d_name_off)
{
    if (!h || !h->ctx || file < 0 || (size_t)file >= h->count.size()) return GCI_E_INVALID;
    gci_ctx* ctx = h->ctx;
    const size_t n = h->count[file];
    if (n == 0) return GCI_OK;
    if (!d_recs || !d_name_off) return GCI_E_INVALID;
    static_assert(sizeof(gci_rec) == 32, "two 16-byte pieces per record");
    if (((uintptr_t)d_recs & 15u) || ((uintptr_t)d_name_off & 7u)) {
        HIPCHK(hipMemcpyAsync(d_recs, h->recs[file], n * sizeof(gci_rec), hipMemcpyDeviceToDevice, ctx->stream));
        HIPCHK(hipMemcpyAsync(d_name_off, h->name_off[file], n * 8, hipMemcpyDeviceToDevice, ctx->stream));
        return GCI_OK;
    }
    const uint32_t g16 = (uint32_t)std::min<uint64_t>((2 * n + BLOCK - 1) / BLOCK, 65536), g8 = (uint32_t)std::min<uint64_t>((n + BLOCK - 1) / BLOCK, 65536);
    hipLaunchKernelGGL(k_copy16, dim3(g16), dim3(BLOCK), 0, ctx->stream, (const uint4*)h->recs[file], (uint4*)d_recs, (uint64_t)(2 * n));
    LAUNCHCHK("k_copy16");
    hipLaunchKernelGGL(k_copy8, dim3(g8), dim3(BLOCK), 0, ctx->stream, (const uint2*)h->name_off[file], (uint2*)d_name_off, (uint64_t)n);
    LAUNCHCHK("k_copy8");
    return GCI_OK;
}

extern "C" int gci_paf_dev_free(gci_paf_dev* h)
{
    if (!h) return GCI_OK;
    if (h->ctx) (void)hipStreamSynchronize(h->ctx->stream);
    paf_dev_release(h);
    return GCI_OK;
}
