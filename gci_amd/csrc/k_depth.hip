// k_depth.hip -- K4 / K5: per-base depth from the surviving intervals
// (/root/reference/GCI.py:302-306: depths[target][start+fl : end-fl+1] += 1).
//
// Difference array + prefix sum, with the difference array kept OUT of HBM:
//
//   k_evt_count    per interval: Python-slice-normalised [a, b); counts one +1 event in tile(a), one -1
//                  event in tile(b), and adds +1 / -1 to a coarse per-tile table (one int per 4096 bases).
//   k_scan2_*      exclusive scans of both per-tile tables: bucket offsets and every tile's carry-in.
//   k_evt_scatter  writes each event (12-bit position in its tile + sign) into its tile's bucket.
//   k_tile_build   one workgroup per tile: zero a 16 KiB difference array in LDS, LDS-atomicAdd the tile's
//                  events (about 2 * coverage * 4096 / read length of them: ~20 at 40x HiFi), wave/workgroup
//                  prefix sum seeded with the carry-in, and
//                    pass 1 (by-products, no HBM write of depth): per-tile sum, decimal-text byte count,
//                            optionally the issue-scan run boundaries;
//                    pass 2: the depth track (16-byte coalesced stores) and, optionally, its decimal text.
//
// HBM traffic of the depth build is therefore its OUTPUT only: 4 B/base (+ text bytes), instead of
// memset 4 + scan read 4 + write 4 (+ 4 to count text + 4 to render it + 4 to scan issues + 4 to sum).
// Recomputing a tile in pass 2 costs LDS atomics and shuffles, not bandwidth.  Tiles are independent
// (the carry-in comes from the coarse table), so there is no look-back chain across the 8 XCDs, whose
// L2s are not coherent with each other.
//
// The -1 of an interval that reaches the contig end is kept in the coarse table (last tile) so every
// contig sums to zero and ONE unsegmented scan over all tiles serves all contigs; its event is dropped
// (or lands in the tail padding, which keeps the padding at zero).
#include "gci_ctx.hpp"

// ---- per interval ---------------------------------------------------------------------------------

struct IvlSpan { int64_t tile_a, tile_b; uint32_t pos_a, pos_b; bool valid, has_b; int64_t tile_bc; };

__device__ __forceinline__ IvlSpan span_of(const gci_ivl v, int flank, const int64_t* __restrict__ len,
                                           const int64_t* __restrict__ tile_first, int32_t n_contigs)
{
    IvlSpan s;
    s.valid = false; s.has_b = false; s.tile_a = s.tile_b = s.tile_bc = 0; s.pos_a = s.pos_b = 0;
    if (v.contig < 0 || v.contig >= n_contigs) return s;
    const int64_t L = len[v.contig];
    const int64_t a = gci_slice_bound((int64_t)v.start + flank, L);
    const int64_t b = gci_slice_bound((int64_t)v.end - flank + 1, L);
    if (a >= b) return s;
    const int64_t t0 = tile_first[v.contig];
    s.valid = true;
    s.tile_a = t0 + a / TILE; s.pos_a = (uint32_t)(a % TILE);
    s.has_b = b < (L + TILE - 1) / TILE * TILE;          // b == L lands in tail padding when there is any
    s.tile_b = t0 + b / TILE; s.pos_b = (uint32_t)(b % TILE);
    s.tile_bc = t0 + (b < L ? b : L - 1) / TILE;         // coarse -1: last tile of the contig when b == L
    return s;
}

// One 64-bit atomic per interval end: the low word of tile_cd counts the events of a tile, the high word carries
// the coarse difference (+1 in the tile of the start, -1 in the tile of the stop).
__global__ __launch_bounds__(BLOCK) void k_evt_count(const gci_ivl* __restrict__ ivl, const uint32_t* __restrict__ d_n,
                                                     uint32_t max_n, int flank, const int64_t* __restrict__ len,
                                                     const int64_t* __restrict__ tile_first, int32_t n_contigs,
                                                     unsigned long long* __restrict__ tile_cd)
{
    const uint32_t n = d_n ? min(*d_n, max_n) : max_n;
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const IvlSpan s = span_of(ivl[i], flank, len, tile_first, n_contigs);
    if (!s.valid) return;
    const unsigned long long minus1 = 0xFFFFFFFFull << 32;         // -1 in the high word (the low word never carries)
    atomicAdd(tile_cd + s.tile_a, 1ull | (1ull << 32));
    if (s.has_b && s.tile_b == s.tile_bc) atomicAdd(tile_cd + s.tile_b, 1ull | minus1);
    else {
        if (s.has_b) atomicAdd(tile_cd + s.tile_b, 1ull);
        atomicAdd(tile_cd + s.tile_bc, minus1);
    }
}

// the counts are decremented back to zero while handing out bucket slots: no memset next time
__global__ __launch_bounds__(BLOCK) void k_evt_scatter(const gci_ivl* __restrict__ ivl, const uint32_t* __restrict__ d_n,
                                                       uint32_t max_n, int flank, const int64_t* __restrict__ len,
                                                       const int64_t* __restrict__ tile_first, int32_t n_contigs,
                                                       uint32_t* __restrict__ tile_cd_words, const uint32_t* __restrict__ evt_off,
                                                       uint16_t* __restrict__ events)
{
    const uint32_t n = d_n ? min(*d_n, max_n) : max_n;
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const IvlSpan s = span_of(ivl[i], flank, len, tile_first, n_contigs);
    if (!s.valid) return;
    events[evt_off[s.tile_a] + atomicSub(tile_cd_words + 2 * s.tile_a, 1u) - 1u] = (uint16_t)(s.pos_a << 1);
    if (s.has_b) events[evt_off[s.tile_b] + atomicSub(tile_cd_words + 2 * s.tile_b, 1u) - 1u] = (uint16_t)((s.pos_b << 1) | 1u);
}

// Both per-tile scans in one launch: blockIdx.y == 0 coarse difference (high words) -> carry, == 1 counts (low
// words) -> offsets.  The y == 0 blocks also zero the high words they consumed and the small outputs of the build.
__global__ __launch_bounds__(BLOCK) void k_scan2_local(uint32_t* __restrict__ cd_words, int32_t* __restrict__ carry,
                                                       int32_t* __restrict__ blk_a, uint32_t* __restrict__ off,
                                                       uint32_t* __restrict__ blk_b, int64_t n, uint32_t* __restrict__ n_keys,
                                                       long long* __restrict__ sums, int32_t n_contigs)
{
    if (blockIdx.y == 0) {
        scan_local_body<int32_t, int32_t>((const int32_t*)cd_words + 1, carry, blk_a, n, blockIdx.x, 2);
        const int64_t base = (int64_t)blockIdx.x * TILE + (int64_t)threadIdx.x * 16;
#pragma unroll
        for (int i = 0; i < 16; i++) if (base + i < n) cd_words[2 * (base + i) + 1] = 0u;      // own elements only
        if (blockIdx.x == 0) {
            if (threadIdx.x == 0 && n_keys) *n_keys = 0;
            if (sums) for (int32_t c = threadIdx.x; c < n_contigs; c += BLOCK) sums[c] = 0;
        }
    } else scan_local_body<uint32_t, uint32_t>(cd_words, off, blk_b, n, blockIdx.x, 2);
}

__global__ __launch_bounds__(BLOCK) void k_scan2_add(int32_t* __restrict__ carry, const int32_t* __restrict__ blk_a,
                                                     uint32_t* __restrict__ off, const uint32_t* __restrict__ blk_b,
                                                     int64_t n, int32_t n_blocks)
{
    if (blockIdx.y == 0) { if ((int32_t)blockIdx.x < n_blocks) scan_add_body<int32_t>(carry, blk_a, n, n_blocks, blockIdx.x); }
    else scan_add_body<uint32_t>(off, blk_b, n, n_blocks, blockIdx.x);     // block n_blocks writes off[n] = total
}

// ---- per tile -------------------------------------------------------------------------------------


struct IssueArgs {
    unsigned long long* keys;      // nullptr: no fused issue scan
    uint32_t* n_keys;
    uint32_t cap;
    int flank;
    int32_t lo, hi;                // depth in [lo, hi]  (gci_int_range of the caller's lo < d <= hi)
};

// Body of one tile.  PASS 1: by-products only (sum, text bytes, issue boundaries).  PASS 2: depth (+ text).
// FULL: every element of the tile lies inside its contig (all tiles but the last of each contig).
template <int PASS, bool FULL>
__device__ __forceinline__ void tile_body(
    int32_t* lds, const int4 (&v)[4], int32_t carry_in, int64_t tile, int32_t c, int64_t elem0, int64_t L,
    long long* __restrict__ tile_sum, uint32_t* __restrict__ tile_bytes, const IssueArgs& iss,
    int32_t* __restrict__ depth, const uint64_t* __restrict__ tile_text_off, uint8_t* __restrict__ text,
    uint64_t text_cap, uint32_t (*twtot)[BLOCK / 64], const uint32_t* lut, int t, int lane, int wave)
{
    const int64_t valid = L - elem0;                              // elements of this tile inside the contig
    int4* l4 = reinterpret_cast<int4*>(lds);
    if (PASS == 1) {
        long long s = 0;
        uint32_t bytes = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int64_t i = (int64_t)(j * BLOCK + t) * 4;
            const int32_t d[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const bool in = FULL || i + k < valid;
                s += in ? d[k] : 0;
                bytes += in ? ndigits_fast((uint32_t)d[k]) + 1u : 0u;
            }
        }
        s = wave_sum<long long>(s);
        bytes = wave_sum<uint32_t>(bytes);
        __shared__ long long ssum[BLOCK / 64];
        if (lane == 0) { ssum[wave] = s; twtot[0][wave] = bytes; }
        if (iss.keys) {
            // depth of the whole tile to LDS so each thread can see its predecessor element
#pragma unroll
            for (int j = 0; j < 4; j++) l4[j * BLOCK + t] = v[j];
        }
        __syncthreads();
        if (t == 0) {
            tile_sum[tile] = ssum[0] + ssum[1] + ssum[2] + ssum[3];
            tile_bytes[tile] = twtot[0][0] + twtot[0][1] + twtot[0][2] + twtot[0][3];
        }
        if (iss.keys) {
            // window of this contig: depth_list[flank : L - flank] with Python slice normalisation (GCI.py:374)
            const int64_t wa = gci_slice_bound(iss.flank, L);
            int64_t wb = gci_slice_bound(L - iss.flank, L);
            if (wb < wa) wb = wa;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int64_t i = (int64_t)(j * BLOCK + t) * 4;
                const int64_t e = elem0 + i;                       // contig coordinate of this thread's first element
                const int32_t d[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
                const int32_t dprev = i > 0 ? lds[i - 1] : carry_in;
                bool gp = (e - 1 >= wa) && (e - 1 < wb) && (dprev >= iss.lo) && (dprev <= iss.hi);
                bool g[4];
                bool any = gp;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    g[k] = (e + k >= wa) && (e + k < wb) && (d[k] >= iss.lo) && (d[k] <= iss.hi);
                    any |= g[k];
                }
                if (!any) continue;                                 // nearly always: low-depth runs are rare
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int64_t p = e + k;
                    if (g[k] != gp && p >= wa && (g[k] || p < wb)) {   // run boundary at p: start if g, else end
                        const uint32_t slot = atomicAdd(iss.n_keys, 1u);
                        if (slot < iss.cap) iss.keys[slot] = issue_key((uint32_t)c, p - wa, !g[k]);
                    }
                    if (g[k] && p == wb - 1) {                      // run reaches the end of the window
                        const uint32_t slot = atomicAdd(iss.n_keys, 1u);
                        if (slot < iss.cap) iss.keys[slot] = issue_key((uint32_t)c, wb - wa, true);
                    }
                    gp = g[k];
                }
            }
        }
    } else {
        int4* g4 = reinterpret_cast<int4*>(depth + (size_t)tile * TILE);
#pragma unroll
        for (int j = 0; j < 4; j++) g4[j * BLOCK + t] = v[j];
        if (text)
            text_tile<FULL>(v, valid, reinterpret_cast<uint8_t*>(lds), twtot, lut, text, tile_text_off[tile], text_cap, t,
                            lane, wave);
    }
}

// One tile by one workgroup: difference array in LDS, scan, tile_body.
template <int PASS>
__device__ __forceinline__ void tile_dense(
    int64_t tile, const uint16_t* __restrict__ events, const uint32_t* __restrict__ evt_off,
    const int32_t* __restrict__ tile_carry, const int64_t* __restrict__ tile_first, const int64_t* __restrict__ len,
    int32_t n_contigs, long long* __restrict__ tile_sum, uint32_t* __restrict__ tile_bytes, const IssueArgs& iss,
    int32_t* __restrict__ depth, const uint64_t* __restrict__ tile_text_off, uint8_t* __restrict__ text, uint64_t text_cap,
    const uint32_t* __restrict__ g_lut)
{
    __shared__ __attribute__((aligned(16))) int32_t lds[TILE];    // difference array; later depth (pass 1) / text staging
    __shared__ int32_t wtot[4][BLOCK / 64];
    __shared__ uint32_t twtot[4][BLOCK / 64];
    __shared__ uint32_t lut[PASS == 2 ? TEXT_LUT : 1];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    int4* l4 = reinterpret_cast<int4*>(lds);
    // every global load this tile depends on is issued before the LDS work so that their latencies overlap:
    // bucket bounds, the first event of this thread (buckets hold ~20 events: one per thread at most), carry-in
    const uint32_t e0 = evt_off[tile], e1 = evt_off[tile + 1];
    const int32_t carry_in = tile_carry[tile];
    const int32_t c = contig_of_tile(tile_first, n_contigs, tile);
    const int64_t elem0 = (tile - tile_first[c]) * TILE;          // index of the tile's first element in its contig
    const int64_t L = len[c];
#pragma unroll
    for (int j = 0; j < 4; j++) l4[j * BLOCK + t] = make_int4(0, 0, 0, 0);
    if (PASS == 2 && text) text_lut_load(lut, g_lut, t);
    const uint32_t ev0 = e0 + t < e1 ? (uint32_t)events[e0 + t] : 0xFFFFFFFFu;
    __syncthreads();
    if (ev0 != 0xFFFFFFFFu) atomicAdd(&lds[ev0 >> 1], (ev0 & 1u) ? -1 : 1);
    for (uint32_t e = e0 + BLOCK + t; e < e1; e += BLOCK) {
        const uint32_t ev = events[e];
        atomicAdd(&lds[ev >> 1], (ev & 1u) ? -1 : 1);
    }
    __syncthreads();
    int4 v[4];
#pragma unroll
    for (int j = 0; j < 4; j++) v[j] = l4[j * BLOCK + t];
    int32_t tot[4], inc[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        v[j].y += v[j].x; v[j].z += v[j].y; v[j].w += v[j].z;
        tot[j] = v[j].w;
        inc[j] = wave_inclusive_i32(tot[j]);
    }
    if (lane == 63) {
#pragma unroll
        for (int j = 0; j < 4; j++) wtot[j][wave] = inc[j];
    }
    __syncthreads();                       // also: every thread has read its slice of the LDS difference array
    int32_t carry = carry_in;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        int32_t pre = 0, all = 0;
#pragma unroll
        for (int w = 0; w < BLOCK / 64; w++) { const int32_t x = wtot[j][w]; if (w < wave) pre += x; all += x; }
        const int32_t ex = carry + pre + inc[j] - tot[j];
        v[j].x += ex; v[j].y += ex; v[j].z += ex; v[j].w += ex;
        carry += all;
    }
    // v[j] now holds the depth of elements (j * 256 + t) * 4 .. + 3 of this tile
    if (L - elem0 >= TILE)
        tile_body<PASS, true>(lds, v, carry_in, tile, c, elem0, L, tile_sum, tile_bytes, iss, depth, tile_text_off, text,
                              text_cap, twtot, lut, t, lane, wave);
    else
        tile_body<PASS, false>(lds, v, carry_in, tile, c, elem0, L, tile_sum, tile_bytes, iss, depth, tile_text_off, text,
                               text_cap, twtot, lut, t, lane, wave);
}

// PASS 2 (depth + text): one workgroup per tile.
__global__ __launch_bounds__(BLOCK) void k_tile_build(
    const uint16_t* __restrict__ events, const uint32_t* __restrict__ evt_off, const int32_t* __restrict__ tile_carry,
    const int64_t* __restrict__ tile_first, const int64_t* __restrict__ len, int32_t n_contigs,
    int32_t* __restrict__ depth, const uint64_t* __restrict__ tile_text_off, uint8_t* __restrict__ text, uint64_t text_cap,
    const uint32_t* __restrict__ g_lut)
{
    IssueArgs none;
    none.keys = nullptr; none.n_keys = nullptr; none.cap = 0; none.flank = 0; none.lo = 0; none.hi = 0;
    tile_dense<2>(blockIdx.x, events, evt_off, tile_carry, tile_first, len, n_contigs, nullptr, nullptr, none, depth,
                  tile_text_off, text, text_cap, g_lut);
}

// PASS 1 (by-products: depth sum, text bytes, issue-run boundaries of every tile).  The depth inside a tile is
// piecewise constant with one piece per event, and a tile of long-read data holds ~20 events: so a tile with at most
// SPARSE_MAX events is done by ONE WAVE straight from its event list -- rank the events by position, scan the
// signs, and every lane owns one constant-depth segment: sum += depth * length, bytes += (digits + 1) * length, run
// boundaries only where a segment meets its neighbour or the window edge.  Tiles with more events (short reads,
// pile-ups) fall back to the dense path, one after the other, by the whole workgroup.
#define SPARSE_MAX 63

__device__ __forceinline__ void tile_sparse1(
    int64_t tile, uint32_t e0, uint32_t n_ev, const uint16_t* __restrict__ events, int32_t carry_in, int32_t c,
    int64_t elem0, int64_t L, long long* __restrict__ tile_sum, uint32_t* __restrict__ tile_bytes, const IssueArgs& iss,
    int lane)
{
    const int32_t valid = (int32_t)min((int64_t)TILE, L - elem0);
    // lane 0: a null event at position 0 (the segment that continues the previous tile); lanes 1 .. n_ev: the events
    const bool has = lane >= 1 && (uint32_t)lane <= n_ev;
    const uint32_t ev = has ? (uint32_t)events[e0 + lane - 1] : 0u;
    const uint32_t key = lane == 0 ? 0u : has ? ((ev >> 1) << 7) | (uint32_t)lane : 0xFFFFFFFFu;   // position, then lane: distinct
    uint32_t rank = 0;
    for (uint32_t j = 0; j <= n_ev; j++) rank += (uint32_t)__builtin_amdgcn_readlane((int)key, (int)j) < key ? 1u : 0u;
    // forward permute: the lane with rank r hands its event to lane r (idle lanes keep rank > n_ev among themselves)
    const int32_t delta = has ? ((ev & 1u) ? -1 : 1) : 0;
    const uint32_t packed = ((has ? ev >> 1 : lane == 0 ? 0u : (uint32_t)TILE) << 2) | (uint32_t)(delta + 1);
    const uint32_t dst = lane <= (int)n_ev ? rank : (uint32_t)lane;
    const uint32_t got = (uint32_t)__builtin_amdgcn_ds_permute((int)(dst << 2), (int)packed);
    const int32_t my_delta = (int32_t)(got & 3u) - 1;
    const int32_t p = min((int32_t)(got >> 2), valid);                       // segment start, clipped to the contig
    const int32_t d = carry_in + wave_inclusive_i32(my_delta);              // depth of the segment
    int32_t p_next = __shfl_down(p, 1, 64);
    if (lane == 63) p_next = valid;
    p_next = min(p_next, valid);                                             // (idle lanes sit at TILE >= valid)
    const int32_t seg = p_next - p;                                         // >= 0: positions are sorted
    long long s = (long long)seg * d;
    uint32_t bytes = (uint32_t)seg * (ndigits_fast((uint32_t)d) + 1u);
    s = wave_sum<long long>(s);
    bytes = wave_sum<uint32_t>(bytes);
    if (lane == 0) { tile_sum[tile] = s; tile_bytes[tile] = bytes; }
    if (!iss.keys) return;
    // ---- issue-run boundaries (same rules as the dense path: tile_body) -----------------------------------------
    const int64_t wa = gci_slice_bound(iss.flank, L);
    int64_t wb = gci_slice_bound(L - iss.flank, L);
    if (wb < wa) wb = wa;
    const bool low = d >= iss.lo && d <= iss.hi;
    const bool nonempty = seg > 0;
    const unsigned long long m_ne = __ballot(nonempty), m_low = __ballot(nonempty && low);
    if (m_ne == 0ull) return;
    const unsigned long long below = lane ? m_ne & ((1ull << lane) - 1ull) : 0ull;
    const unsigned long long above = lane < 63 ? m_ne >> (lane + 1) : 0ull;
    // the element before this segment: previous non-empty segment, or the last element of the previous tile
    const bool first = below == 0ull;
    const bool prev_in_window = elem0 - 1 >= wa && elem0 - 1 < wb;
    const bool low_prev = first ? (prev_in_window && carry_in >= iss.lo && carry_in <= iss.hi)
                                : ((m_low >> (63 - __builtin_clzll(below))) & 1ull) != 0ull;
    const bool has_next = above != 0ull;
    const bool low_next = has_next && ((m_low >> (lane + 1 + __builtin_ctzll(above))) & 1ull) != 0ull;
    if (!nonempty) return;
    const int64_t A = elem0 + p, B = elem0 + p_next;
    const int64_t a = max(A, wa), b = min(B, wb);
    auto put = [&](int64_t rel, bool is_end) {
        const uint32_t slot = atomicAdd(iss.n_keys, 1u);
        if (slot < iss.cap) iss.keys[slot] = issue_key((uint32_t)c, rel, is_end);
    };
    if (low && a < b) {
        if (A <= wa || !low_prev) put(a - wa, false);                        // a run starts at a
        if (b == wb) put(wb - wa, true);                                     // ... and reaches the end of the window
        else if (has_next && !low_next) put(b - wa, true);                   // ... or ends where the next segment begins
    } else if (first && low_prev && A < wb) {
        put(A - wa, true);                                                   // the previous tile's run ends at this tile's first element
    }
}

__global__ __launch_bounds__(BLOCK) void k_tile_pass1(
    const uint16_t* __restrict__ events, const uint32_t* __restrict__ evt_off, const int32_t* __restrict__ tile_carry,
    const int64_t* __restrict__ tile_first, const int64_t* __restrict__ len, int32_t n_contigs, int64_t n_tiles,
    long long* __restrict__ tile_sum, uint32_t* __restrict__ tile_bytes, IssueArgs iss, const uint32_t* __restrict__ g_lut)
{
    __shared__ int dense[BLOCK / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t tile = (int64_t)blockIdx.x * (BLOCK / 64) + wave;
    bool is_dense = false;
    if (tile < n_tiles) {
        const uint32_t e0 = evt_off[tile], e1 = evt_off[tile + 1];
        is_dense = e1 - e0 > SPARSE_MAX;
        if (!is_dense) {
            const int32_t c = contig_of_tile(tile_first, n_contigs, tile);
            tile_sparse1(tile, e0, e1 - e0, events, tile_carry[tile], c, (tile - tile_first[c]) * TILE, len[c], tile_sum,
                         tile_bytes, iss, lane);
        }
    }
    if (lane == 0) dense[wave] = is_dense ? 1 : 0;
    __syncthreads();
    for (int w = 0; w < BLOCK / 64; w++) {
        if (!dense[w]) continue;                                             // uniform over the workgroup
        tile_dense<1>((int64_t)blockIdx.x * (BLOCK / 64) + w, events, evt_off, tile_carry, tile_first, len, n_contigs,
                      tile_sum, tile_bytes, iss, nullptr, nullptr, nullptr, 0, g_lut);
        __syncthreads();
    }
}

// ---- host -----------------------------------------------------------------------------------------

static int launch_tile_build(gci_ctx* ctx, int pass, IssueArgs iss, int32_t* d_depth, uint8_t* d_text, uint64_t text_cap)
{
    const dim3 grid((uint32_t)ctx->n_tiles), block(BLOCK);
    const uint16_t* ev = (const uint16_t*)ctx->events.p;
    const uint32_t* eo = (const uint32_t*)ctx->evt_off.p;
    const int32_t* tc = (const int32_t*)ctx->tile_carry.p;
    const int64_t* tf = (const int64_t*)ctx->d_tile_first.p;
    const int64_t* ln = (const int64_t*)ctx->d_len.p;
    if (pass == 1) {
        ProfScope _ps(ctx, GCI_PROF_TILE_PASS1);
        const int64_t per = BLOCK / 64;
        hipLaunchKernelGGL(k_tile_pass1, dim3((uint32_t)((ctx->n_tiles + per - 1) / per)), block, 0, ctx->stream, ev, eo, tc,
                           tf, ln, ctx->n_contigs, ctx->n_tiles, (long long*)ctx->tile_sum.p, (uint32_t*)ctx->tile_u32.p, iss,
                           (const uint32_t*)ctx->text_lut.p);
    } else {
        ProfScope _ps(ctx, GCI_PROF_DEPTH_SCAN);
        hipLaunchKernelGGL(k_tile_build, grid, block, 0, ctx->stream, ev, eo, tc, tf, ln, ctx->n_contigs, d_depth,
                           (const uint64_t*)ctx->tile_u64.p, d_text, text_cap, (const uint32_t*)ctx->text_lut.p);
    }
    LAUNCHCHK("k_tile_build");
    return GCI_OK;
}

extern "C" int gci_depth_build_begin(gci_ctx* ctx, const gci_ivl* d_ivl, const uint32_t* d_n, uint32_t max_n,
                                     const gci_build_opts* o)
{
    if (!ctx || !o || (max_n && !d_ivl)) return GCI_E_INVALID;
    if (!ctx->n_contigs) return GCI_E_NO_LAYOUT;
    if (o->want_text && !o->d_contig_text_off) return GCI_E_INVALID;
    if (o->d_n_keys && o->key_cap && !o->d_keys) return GCI_E_INVALID;
    ctx->build_pending = false;
    const int64_t nt = ctx->n_tiles;
    if (nt == 0) {
        if (o->d_contig_text_off) HIPCHK(hipMemsetAsync(o->d_contig_text_off, 0, (size_t)(ctx->n_contigs + 1) * 8, ctx->stream));
        if (o->d_sums) HIPCHK(hipMemsetAsync(o->d_sums, 0, (size_t)ctx->n_contigs * 8, ctx->stream));
        if (o->d_n_keys) HIPCHK(hipMemsetAsync(o->d_n_keys, 0, 4, ctx->stream));
        ctx->build_pending = true; ctx->build_text = false;
        return GCI_OK;
    }
    GCI_TRY(gci_ensure(ctx, ctx->events, (size_t)max_n * 2 * sizeof(uint16_t) + 16));
    const int64_t* ln = (const int64_t*)ctx->d_len.p;
    const int64_t* tf = (const int64_t*)ctx->d_tile_first.p;
    unsigned long long* cd = (unsigned long long*)ctx->tile_cd.p;
    uint32_t* off = (uint32_t*)ctx->evt_off.p;
    const int32_t nb = (int32_t)((nt + TILE - 1) / TILE);
    if (ctx->cd_dirty) HIPCHK(hipMemsetAsync(cd, 0, (size_t)(nt + 1) * 8, ctx->stream));
    ctx->cd_dirty = true;
    if (max_n) {
        ProfScope _ps(ctx, GCI_PROF_DEPTH_DIFF);
        hipLaunchKernelGGL(k_evt_count, dim3((max_n + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, ctx->stream, d_ivl, d_n, max_n,
                           o->flank, ln, tf, ctx->n_contigs, cd);
        LAUNCHCHK("k_evt_count");
    }
    {
        ProfScope _ps(ctx, GCI_PROF_SCAN_TILES);
        hipLaunchKernelGGL(k_scan2_local, dim3(nb, 2), dim3(BLOCK), 0, ctx->stream, (uint32_t*)cd, (int32_t*)ctx->tile_carry.p,
                           (int32_t*)ctx->blk_a.p, off, (uint32_t*)ctx->blk_b.p, nt, o->d_n_keys, (long long*)o->d_sums,
                           ctx->n_contigs);
        LAUNCHCHK("k_scan2_local");
        hipLaunchKernelGGL(k_scan2_add, dim3(nb + 1, 2), dim3(BLOCK), 0, ctx->stream, (int32_t*)ctx->tile_carry.p,
                           (const int32_t*)ctx->blk_a.p, off, (const uint32_t*)ctx->blk_b.p, nt, nb);
        LAUNCHCHK("k_scan2_add");
    }
    if (max_n) {
        ProfScope _ps(ctx, GCI_PROF_DEPTH_DIFF);
        hipLaunchKernelGGL(k_evt_scatter, dim3((max_n + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, ctx->stream, d_ivl, d_n, max_n,
                           o->flank, ln, tf, ctx->n_contigs, (uint32_t*)cd, (const uint32_t*)off, (uint16_t*)ctx->events.p);
        LAUNCHCHK("k_evt_scatter");
    }
    ctx->cd_dirty = false;
    const bool by_products = o->want_text || o->d_sums || o->d_n_keys;
    if (by_products) {
        IssueArgs iss;
        memset(&iss, 0, sizeof iss);
        if (o->d_n_keys) {
            iss.keys = (unsigned long long*)o->d_keys; iss.n_keys = o->d_n_keys; iss.cap = o->key_cap;
            const IntRange rg = gci_int_range(o->lo, o->hi);
            iss.flank = o->issue_flank; iss.lo = rg.lo; iss.hi = rg.hi;
            if (!iss.keys) { static unsigned long long dummy; iss.keys = &dummy; iss.cap = 0; }   // count only
        }
        GCI_TRY(launch_tile_build(ctx, 1, iss, nullptr, nullptr, 0));
        if (o->d_sums) {
            ProfScope _ps(ctx, GCI_PROF_DEPTH_SUM);
            hipLaunchKernelGGL(k_reduce_tiles, dim3(ctx->n_contigs, REDUCE_SPLIT), dim3(BLOCK), 0, ctx->stream,
                               (const long long*)ctx->tile_sum.p, tf, (unsigned long long*)o->d_sums);
            LAUNCHCHK("k_reduce_tiles");
        }
        if (o->want_text) {
            ProfScope _ps(ctx, GCI_PROF_TEXT_COUNT);
            GCI_TRY((device_exclusive_scan<uint32_t, unsigned long long>(ctx, (const uint32_t*)ctx->tile_u32.p,
                                                                         (unsigned long long*)ctx->tile_u64.p,
                                                                         (unsigned long long*)ctx->blk_u64.p, nt, true)));
            hipLaunchKernelGGL(k_contig_text_off, dim3((ctx->n_contigs + 1 + 63) / 64), dim3(64), 0, ctx->stream,
                               (const uint64_t*)ctx->tile_u64.p, tf, ctx->n_contigs, nt, o->d_contig_text_off);
            LAUNCHCHK("k_contig_text_off");
        }
    }
    ctx->build_pending = true;
    ctx->build_text = o->want_text != 0;
    return GCI_OK;
}

extern "C" int gci_depth_build_finish(gci_ctx* ctx, int32_t* d_depth, uint8_t* d_text, uint64_t text_cap)
{
    if (!ctx || !d_depth) return GCI_E_INVALID;
    if (!ctx->n_contigs) return GCI_E_NO_LAYOUT;
    if (!ctx->build_pending) return GCI_E_INVALID;          // gci_depth_build_begin() first
    if (d_text && !ctx->build_text) return GCI_E_INVALID;   // begin() was not asked for text offsets
    if (ctx->n_tiles == 0) return GCI_OK;
    IssueArgs none;
    memset(&none, 0, sizeof none);
    return launch_tile_build(ctx, 2, none, d_depth, d_text, text_cap);
}

extern "C" int gci_depth_build(gci_ctx* ctx, const gci_ivl* d_ivl, const uint32_t* d_n, uint32_t max_n, int flank,
                               int32_t* d_depth)
{
    if (!d_depth) return GCI_E_INVALID;
    gci_build_opts o;
    memset(&o, 0, sizeof o);
    o.flank = flank;
    GCI_TRY(gci_depth_build_begin(ctx, d_ivl, d_n, max_n, &o));
    return gci_depth_build_finish(ctx, d_depth, nullptr, 0);
}
