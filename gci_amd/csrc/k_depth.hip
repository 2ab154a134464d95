// k_depth.hip -- K4 / K5: per-base depth from the surviving intervals
// (/root/reference/GCI.py:302-306: depths[target][start+fl : end-fl+1] += 1).
//
// Difference array + prefix sum, with the difference array kept OUT of HBM:
//
//   k_evt_count    per interval: Python-slice-normalised [a, b); counts one +1 event in tile(a), one -1
//                  event in tile(b), and adds +1 / -1 to a coarse per-tile table (one int per 4096 bases).
//   k_scan2_*      exclusive scans of both per-tile tables: bucket offsets and every tile's carry-in.
//   k_evt_scatter  writes each event (12-bit position in its tile + sign) into its tile's bucket.
//   k_evp_*        the same bucketing for large inputs by radix partition over tile ranges (no device-scope atomics).
//   k_tile_pass1   by-products without any HBM write of depth: per-tile sum, decimal-text byte count, optionally the
//                  issue-scan run boundaries.  A tile of long-read data holds ~20 events, so it is never materialised:
//                  the events are ranked by position and every lane owns one constant-depth segment (two tiles per wave).
//   k_tile_build   the depth track (16-byte stores) and, optionally, its decimal text, again from the segments; four
//                  waves per tile, workgroups mapped to tiles so that every XCD owns a contiguous part of the genome.
//   k_tile_dense   tiles with many events (short reads, pile-ups) or four-digit depths: a workgroup zeroes a 16 KiB
//                  difference array in LDS, LDS-atomicAdds the tile's events, prefix-sums it seeded with the carry-in.
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
#include <stdlib.h>

#ifndef TB_TRACK_NT
#define TB_TRACK_NT 1                // bulk stores of the track non-temporal (A/B: tools/gpu_boxab.sh)
#endif
#ifndef TB_TEXT_NT
#define TB_TEXT_NT 0
#endif
typedef int i32x4 __attribute__((ext_vector_type(4)));          // native vector type: __builtin_nontemporal_store takes it

// ---- per interval ---------------------------------------------------------------------------------

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
    count_span(s, tile_cd);
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

// ---- the same bucketing for large inputs, without device-scope atomics ----------------------------------------------
// At genome scale (10^7 events over 10^6 tiles) every atomic above misses every cache and is executed by the memory side:
// 22 G/s, 0.6 ms per build plus the counting.  Instead the events are radix-partitioned by tile RANGE (2^sh consecutive
// tiles, at most EVP_BUCKETS ranges): a histogram and a scatter pass over the intervals with workgroup-local LDS
// counters (the scan of the [range][workgroup] matrix gives every workgroup its slots), then one workgroup per range
// counts its tiles' events and coarse differences in LDS (k_evp_tiles<0>, which writes the range's part of tile_cd in the
// format the scans expect) and, after the scans, places the events in their tiles' buckets (k_evp_tiles<1>).
// An item: tile within its range << 14 | position in the tile << 2 | kind (0: +1 event, 1: -1 event, 2: the coarse -1 of an
// interval that reaches the end of its contig -- no event, see span_of).
#define EVP_CHUNK 8192               // intervals per workgroup of the two partition passes
#define EVP_BUCKETS 1024

struct EvpItems { uint32_t b0, w0, b1, w1; };

__device__ __forceinline__ EvpItems evp_items(const IvlSpan& s, int sh)
{
    const uint32_t mask = (1u << sh) - 1u;
    EvpItems e;
    e.b0 = (uint32_t)(s.tile_a >> sh);
    e.w0 = (((uint32_t)s.tile_a & mask) << 14) | (s.pos_a << 2);
    const int64_t tb = s.has_b ? s.tile_b : s.tile_bc;              // (has_b implies tile_b == tile_bc)
    e.b1 = (uint32_t)(tb >> sh);
    e.w1 = (((uint32_t)tb & mask) << 14) | (s.has_b ? (s.pos_b << 2) | 1u : 2u);
    return e;
}

template <bool SCATTER>
__global__ __launch_bounds__(BLOCK) void k_evp_part(const gci_ivl* __restrict__ ivl, const uint32_t* __restrict__ d_n, uint32_t max_n,
                                                    int flank, const int64_t* __restrict__ len, const int64_t* __restrict__ tile_first,
                                                    int32_t n_contigs, int sh, uint32_t nb, uint32_t n_wg, uint32_t* __restrict__ hist,
                                                    uint32_t* __restrict__ items)
{
    __shared__ uint32_t h[EVP_BUCKETS];
    // SCATTER: hist has been scanned: entry [range][workgroup] = the first slot of this workgroup's items of that range
    for (uint32_t i = threadIdx.x; i < nb; i += BLOCK) h[i] = SCATTER ? hist[(size_t)i * n_wg + blockIdx.x] : 0u;
    __syncthreads();
    const uint32_t n = d_n ? min(*d_n, max_n) : max_n;
    const uint32_t lo = blockIdx.x * EVP_CHUNK, hi = min(n, lo + EVP_CHUNK);
    for (uint32_t i = lo + threadIdx.x; i < hi; i += BLOCK) {
        const IvlSpan s = span_of(ivl[i], flank, len, tile_first, n_contigs);
        if (!s.valid) continue;
        const EvpItems e = evp_items(s, sh);
        if (SCATTER) {
            items[atomicAdd(&h[e.b0], 1u)] = e.w0;
            items[atomicAdd(&h[e.b1], 1u)] = e.w1;
        } else {
            atomicAdd(&h[e.b0], 1u);
            atomicAdd(&h[e.b1], 1u);
        }
    }
    if (SCATTER) return;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < nb; i += BLOCK) hist[(size_t)i * n_wg + blockIdx.x] = h[i];
}

// One workgroup per tile range.  PLACE == 0: tile_cd of the range's tiles (low word: events, high word: coarse difference).
// PLACE == 1: the events into their buckets (evt_off from the scans), and the low words back to zero (k_tile_build does
// the high words): the table is clean for the next build.
template <int PLACE>
__global__ __launch_bounds__(BLOCK) void k_evp_tiles(const uint32_t* __restrict__ items, const uint32_t* __restrict__ hist, uint32_t n_wg,
                                                     int sh, int64_t n_tiles, unsigned long long* __restrict__ tile_cd,
                                                     const uint32_t* __restrict__ evt_off, uint16_t* __restrict__ events, uint32_t stage_cap)
{
    extern __shared__ unsigned long long evp_lds[];                 // 2^sh entries (+ PLACE: stage_cap events)
    const uint32_t per = 1u << sh;
    const int64_t t0 = (int64_t)blockIdx.x << sh;
    const uint32_t nt = (uint32_t)min((int64_t)per, n_tiles - t0);
    uint32_t* cur = reinterpret_cast<uint32_t*>(evp_lds);
    // PLACE: the range's events form one contiguous stretch of the bucket array; when it fits (stage_cap events, the usual
    // case) it is put together in LDS behind the cursors and leaves as one coalesced copy -- 2-byte stores scattered over
    // the stretch cost 19 bytes of HBM writes per byte.
    uint16_t* stage = reinterpret_cast<uint16_t*>(evp_lds + per);
    const uint32_t e_lo = PLACE ? evt_off[t0] : 0u, e_hi = PLACE ? evt_off[t0 + nt] : 0u;
    const bool staged = PLACE && e_hi - e_lo <= stage_cap;
    for (uint32_t i = threadIdx.x; i < nt; i += BLOCK) {
        if (PLACE) cur[i] = evt_off[t0 + i] - (staged ? e_lo : 0u);
        else evp_lds[i] = 0ull;
    }
    __syncthreads();
    const uint32_t a = hist[(size_t)blockIdx.x * n_wg], b = hist[(size_t)(blockIdx.x + 1) * n_wg];
    for (uint32_t i = a + threadIdx.x; i < b; i += BLOCK) {
        const uint32_t w = items[i], tl = w >> 14, kind = w & 3u;
        if (PLACE) {
            if (kind < 2u) {
                const uint16_t v = (uint16_t)((((w >> 2) & 0xFFFu) << 1) | kind);
                const uint32_t at = atomicAdd(&cur[tl], 1u);
                if (staged) stage[at] = v; else events[at] = v;
            }
        } else {
            atomicAdd(&evp_lds[tl], kind == 0u ? (1ull | (1ull << 32)) : kind == 1u ? (1ull | (0xFFFFFFFFull << 32)) : (0xFFFFFFFFull << 32));
        }
    }
    __syncthreads();
    if (staged) {
        // dwords where the stretch allows (it starts at an even or odd event index), single events at its ends
        uint16_t* out = events + e_lo;
        const uint32_t n_e = e_hi - e_lo, head = (e_lo & 1u) && n_e ? 1u : 0u;
        if (head && threadIdx.x == 0) out[0] = stage[0];
        const uint32_t pairs = (n_e - head) >> 1;
        uint32_t* out32 = reinterpret_cast<uint32_t*>(out + head);
        for (uint32_t i = threadIdx.x; i < pairs; i += BLOCK)
            out32[i] = (uint32_t)stage[head + 2 * i] | ((uint32_t)stage[head + 2 * i + 1] << 16);
        if (((n_e - head) & 1u) && threadIdx.x == 0) out[n_e - 1] = stage[n_e - 1];
    }
    __syncthreads();
    if (PLACE) {
        uint32_t* words = reinterpret_cast<uint32_t*>(tile_cd);
        for (uint32_t i = threadIdx.x; i < nt; i += BLOCK) words[2 * (t0 + i)] = 0u;
    } else {
        // (the low word of an entry never carries into the high one: both halves are sums of their own)
        for (uint32_t i = threadIdx.x; i < nt; i += BLOCK) tile_cd[t0 + i] = evp_lds[i];
        if (t0 + nt == n_tiles && threadIdx.x == 0) tile_cd[n_tiles] = 0ull;
    }
}

// Both per-tile scans in one launch: blockIdx.y == 0 coarse difference (high words) -> carry, == 1 counts (low
// words) -> offsets.  The y == 0 blocks also zero the small outputs of the build (the high words are zeroed by
// k_tile_build, tile by tile).
__global__ __launch_bounds__(BLOCK) void k_scan2_local(uint32_t* __restrict__ cd_words, int32_t* __restrict__ carry,
                                                       int32_t* __restrict__ blk_a, uint32_t* __restrict__ off,
                                                       uint32_t* __restrict__ blk_b, int64_t n, uint32_t* __restrict__ n_keys,
                                                       long long* __restrict__ sums, int32_t n_contigs)
{
    if (blockIdx.y == 0) {
        scan_local_body<int32_t, int32_t>((const int32_t*)cd_words + 1, carry, blk_a, n, blockIdx.x, 2);
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

// ---- the same table work in ONE launch ------------------------------------------------------------------
// A kernel on this stream costs ~4.5 us however little it does, and a build of one chromosome (15 000 tiles, four
// workgroups' worth of table) spent a quarter of its time in two-launch scans.  Up to FEW_BLOCKS workgroups every
// workgroup simply re-reduces the entries in front of its own instead of waiting for a second launch.
#define FEW_BLOCKS 8

// sum of in[0 .. n_before) (entry i at in[i * stride]) by the whole workgroup
template <typename TIn, typename TOut>
__device__ __forceinline__ TOut block_sum_before(const TIn* __restrict__ in, int64_t n_before, int64_t stride)
{
    __shared__ TOut part[BLOCK / 64];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    TOut s = 0;
    if (stride == 1 && sizeof(TIn) == 4) {                       // n_before is a multiple of TILE: whole 16-byte loads
        const uint4* in4 = reinterpret_cast<const uint4*>(in);
        const int64_t n4 = n_before >> 2;
        for (int64_t i = t; i < n4; i += 8 * BLOCK) {
            uint4 v[8];
#pragma unroll
            for (int k = 0; k < 8; k++) v[k] = i + k * BLOCK < n4 ? in4[i + k * BLOCK] : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
            for (int k = 0; k < 8; k++) s += (TOut)(TIn)v[k].x + (TOut)(TIn)v[k].y + (TOut)(TIn)v[k].z + (TOut)(TIn)v[k].w;
        }
    } else
    for (int64_t i = t; i < n_before; i += 8 * BLOCK) {            // eight independent loads in flight per thread
        TIn v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = i + k * BLOCK < n_before ? in[(i + k * BLOCK) * stride] : (TIn)0;
#pragma unroll
        for (int k = 0; k < 8; k++) s += (TOut)v[k];
    }
    s = wave_sum<TOut>(s);
    if (lane == 0) part[wave] = s;
    __syncthreads();
    TOut all = 0;
#pragma unroll
    for (int w = 0; w < BLOCK / 64; w++) all += part[w];
    __syncthreads();
    return all;
}

// exclusive scan of this workgroup's TILE entries on top of `before`; returns before + the workgroup's total
template <typename TIn, typename TOut>
__device__ __forceinline__ TOut scan_block_from(const TIn* __restrict__ in, TOut* __restrict__ out, int64_t n, uint32_t blk,
                                                int64_t stride, TOut before)
{
    __shared__ TOut wtot[BLOCK / 64];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int64_t base = (int64_t)blk * TILE + (int64_t)t * 16;
    TOut v[16], run = 0;
    if (stride == 1 && sizeof(TIn) == 4 && base + 16 <= n) {       // the table starts 16-byte aligned, base is a multiple of 16
        const uint4* in4 = reinterpret_cast<const uint4*>(in + base);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint4 q = in4[k];
            v[4 * k] = (TOut)(TIn)q.x; v[4 * k + 1] = (TOut)(TIn)q.y; v[4 * k + 2] = (TOut)(TIn)q.z; v[4 * k + 3] = (TOut)(TIn)q.w;
        }
    } else {
#pragma unroll
        for (int i = 0; i < 16; i++) v[i] = base + i < n ? (TOut)in[(base + i) * stride] : (TOut)0;
    }
#pragma unroll
    for (int i = 0; i < 16; i++) run += v[i];
    const TOut inc = wave_inclusive<TOut>(run, lane);
    if (lane == 63) wtot[wave] = inc;
    __syncthreads();
    TOut pre = before + inc - run, all = before;
#pragma unroll
    for (int w = 0; w < BLOCK / 64; w++) { const TOut x = wtot[w]; if (w < wave) pre += x; all += x; }
#pragma unroll
    for (int i = 0; i < 16; i++) { if (base + i < n) out[base + i] = pre; pre += v[i]; }
    return all;
}

// Both scans of the (count, difference) table by the same workgroups: 16-byte loads bring two tiles' pairs at a time.
// (The difference words are zeroed by k_tile_build: other workgroups still read them here.)
__global__ __launch_bounds__(BLOCK) void k_scan2_few(const uint32_t* __restrict__ cd_words, int32_t* __restrict__ carry,
                                                     uint32_t* __restrict__ off, int64_t n, uint32_t* __restrict__ n_keys,
                                                     long long* __restrict__ sums, int32_t n_contigs)
{
    __shared__ uint32_t part_c[BLOCK / 64];
    __shared__ int32_t part_d[BLOCK / 64];
    __shared__ uint32_t wtot_c[BLOCK / 64];
    __shared__ int32_t wtot_d[BLOCK / 64];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const uint2* cd = reinterpret_cast<const uint2*>(cd_words);            // .x = count, .y = difference
    // everything in front of this workgroup, eight independent loads in flight per thread
    const int64_t n_before = (int64_t)blockIdx.x * TILE;
    uint32_t bc = 0; int32_t bd = 0;
    {
        const uint4* cdq = reinterpret_cast<const uint4*>(cd_words);        // two tiles per 16-byte load; n_before is even
        const int64_t n_pairs = n_before >> 1;
        for (int64_t i = t; i < n_pairs; i += 8 * BLOCK) {
            uint4 v[8];
#pragma unroll
            for (int k = 0; k < 8; k++) v[k] = i + k * BLOCK < n_pairs ? cdq[i + k * BLOCK] : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
            for (int k = 0; k < 8; k++) { bc += v[k].x + v[k].z; bd += (int32_t)v[k].y + (int32_t)v[k].w; }
        }
    }
    bc = wave_sum<uint32_t>(bc); bd = wave_sum<int32_t>(bd);
    if (lane == 0) { part_c[wave] = bc; part_d[wave] = bd; }
    // this workgroup's own TILE entries: 16 per thread
    const int64_t base = n_before + (int64_t)t * 16;
    uint32_t c[16], run_c = 0; int32_t d[16], run_d = 0;
    const uint4* cd4 = reinterpret_cast<const uint4*>(cd_words);           // two tiles per load; tile_cd has n + 1 entries
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const int64_t i = base + 2 * k;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (i + 1 < n) v = cd4[i >> 1];
        else if (i < n) { const uint2 u = cd[i]; v.x = u.x; v.y = u.y; }
        c[2 * k] = v.x; d[2 * k] = (int32_t)v.y; c[2 * k + 1] = v.z; d[2 * k + 1] = (int32_t)v.w;
    }
#pragma unroll
    for (int i = 0; i < 16; i++) { run_c += c[i]; run_d += d[i]; }
    const uint32_t inc_c = (uint32_t)wave_inclusive_i32((int32_t)run_c);
    const int32_t inc_d = wave_inclusive_i32(run_d);
    if (lane == 63) { wtot_c[wave] = inc_c; wtot_d[wave] = inc_d; }
    __syncthreads();
    uint32_t pre_c = inc_c - run_c, all_c = 0; int32_t pre_d = inc_d - run_d;
#pragma unroll
    for (int w = 0; w < BLOCK / 64; w++) {
        pre_c += part_c[w]; pre_d += part_d[w]; all_c += part_c[w] + wtot_c[w];
        if (w < wave) { pre_c += wtot_c[w]; pre_d += wtot_d[w]; }
    }
#pragma unroll
    for (int i = 0; i < 16; i++) {
        if (base + i < n) { off[base + i] = pre_c; carry[base + i] = pre_d; }
        pre_c += c[i]; pre_d += d[i];
    }
    if (blockIdx.x == gridDim.x - 1 && t == 0) off[n] = all_c;
    if (blockIdx.x == 0) {
        if (t == 0 && n_keys) *n_keys = 0;
        if (sums) for (int32_t k = t; k < n_contigs; k += BLOCK) sums[k] = 0;
    }
}

// after pass 1, one launch: workgroups [0, nb): text offset of every tile (+ total) and of every contig;
// workgroups [nb, nb + n_contigs * REDUCE_SPLIT): depth sum of every contig (sums zeroed by k_scan2_*)
__global__ __launch_bounds__(BLOCK) void k_after_pass1_few(const uint32_t* __restrict__ tile_bytes, unsigned long long* __restrict__ tile_off,
                                                           int64_t n, uint32_t nb, const int64_t* __restrict__ tile_first,
                                                           int32_t n_contigs, uint64_t* __restrict__ contig_off,
                                                           const long long* __restrict__ tile_sum, unsigned long long* __restrict__ sums)
{
    const int t = threadIdx.x;
    if (blockIdx.x < nb) {
        if (!contig_off) return;
        const uint32_t blk = blockIdx.x;
        const unsigned long long before = block_sum_before<uint32_t, unsigned long long>(tile_bytes, (int64_t)blk * TILE, 1);
        const unsigned long long all = scan_block_from<uint32_t, unsigned long long>(tile_bytes, tile_off, n, blk, 1, before);
        if (blk == nb - 1 && t == 0) { tile_off[n] = all; contig_off[n_contigs] = all; }
        __syncthreads();                                   // this workgroup's offsets are visible to it
        const int64_t lo = (int64_t)blk * TILE, hi = min(n, lo + TILE);
        for (int32_t c = t; c < n_contigs; c += BLOCK) {
            const int64_t ft = tile_first[c];
            if (ft >= lo && ft < hi) contig_off[c] = tile_off[ft];
            else if (ft >= n && blk == nb - 1) contig_off[c] = all;          // contigs of length zero at the end
        }
    } else {
        if (!sums) return;
        __shared__ long long part[BLOCK / 64];
        const uint32_t w = blockIdx.x - nb, c = w / REDUCE_SPLIT, y = w % REDUCE_SPLIT;
        const int lane = t & 63, wave = t >> 6;
        const int64_t a = tile_first[c], b = tile_first[c + 1];
        long long s = 0;
        for (int64_t i = a + (int64_t)y * BLOCK + t; i < b; i += (int64_t)REDUCE_SPLIT * BLOCK) s += tile_sum[i];
        s = wave_sum<long long>(s);
        if (lane == 0) part[wave] = s;
        __syncthreads();
        if (t == 0) {
            const long long v = part[0] + part[1] + part[2] + part[3];
            if (v) atomicAdd(sums + c, (unsigned long long)v);
        }
    }
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

// PASS 1 (by-products: depth sum, text bytes, issue-run boundaries of every tile).  The depth inside a tile is
// piecewise constant with one piece per event, and a tile of long-read data holds ~20 events: so a tile with at most
// SPARSE_MAX events is done by ONE WAVE straight from its event list -- rank the events by position, scan the
// signs, and every lane owns one constant-depth segment: sum += depth * length, bytes += (digits + 1) * length, run
// boundaries only where a segment meets its neighbour or the window edge.  Tiles with more events (short reads,
// pile-ups) are left to the dense path (k_tile_dense: one workgroup per tile).
#define SPARSE_MAX 62     // + the null event of lane 0 + one lane for the padding behind a contig end

// The constant-depth segments of a tile with at most SPARSE_MAX events, one per lane and sorted by position:
// lane r owns elements [p, p_next) of the tile (clipped to `valid`, possibly empty) at depth d.
struct Seg { int32_t p, p_next, d, len; };

__device__ __forceinline__ Seg sparse_segments(uint32_t e0, uint32_t n_ev, const uint16_t* __restrict__ events,
                                               int32_t carry_in, int32_t valid, int lane)
{
    // lane 0: a null event at position 0 (the segment that continues the previous tile); lanes 1 .. n_ev: the events
    const bool has = lane >= 1 && (uint32_t)lane <= n_ev;
    const uint32_t ev = has ? (uint32_t)events[e0 + lane - 1] : 0u;
    const uint32_t key = lane == 0 ? 0u : has ? ((ev >> 1) << 7) | (uint32_t)lane : 0xFFFFFFFFu;   // position, then lane: distinct
    uint32_t rank = 0;
    for (uint32_t j = 0; j <= n_ev; j++) rank += (uint32_t)__builtin_amdgcn_readlane((int)key, (int)j) < key ? 1u : 0u;
    // forward permute: the lane with rank r hands its event to lane r (idle lanes keep their own place behind them)
    const int32_t delta = has ? ((ev & 1u) ? -1 : 1) : 0;
    const uint32_t packed = ((has ? ev >> 1 : lane == 0 ? 0u : (uint32_t)TILE) << 2) | (uint32_t)(delta + 1);
    const uint32_t dst = lane <= (int)n_ev ? rank : (uint32_t)lane;
    const uint32_t got = (uint32_t)__builtin_amdgcn_ds_permute((int)(dst << 2), (int)packed);
    Seg sg;
    sg.p = min((int32_t)(got >> 2), valid);                                  // segment start, clipped to the contig
    sg.d = carry_in + wave_inclusive_i32((int32_t)(got & 3u) - 1);          // depth of the segment
    int32_t p_next = __shfl_down(sg.p, 1, 64);
    if (lane == 63) p_next = valid;
    sg.p_next = min(p_next, valid);                                          // (idle lanes sit at TILE >= valid)
    sg.len = sg.p_next - sg.p;                                               // >= 0: positions are sorted
    return sg;
}

__device__ __forceinline__ void tile_sparse1(
    int64_t tile, uint32_t e0, uint32_t n_ev, const uint16_t* __restrict__ events, int32_t carry_in, int32_t c,
    int64_t elem0, int64_t L, long long* __restrict__ tile_sum, uint32_t* __restrict__ tile_bytes, const IssueArgs& iss,
    int lane)
{
    const int32_t valid = (int32_t)min((int64_t)TILE, L - elem0);
    const Seg sg = sparse_segments(e0, n_ev, events, carry_in, valid, lane);
    const int32_t p = sg.p, p_next = sg.p_next, d = sg.d, seg = sg.len;
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

template <int PASS>
__device__ __forceinline__ void tile_dense(
    int64_t tile, const uint16_t* __restrict__ events, const uint32_t* __restrict__ evt_off,
    const int32_t* __restrict__ tile_carry, const int64_t* __restrict__ tile_first, const int64_t* __restrict__ len,
    int32_t n_contigs, long long* __restrict__ tile_sum, uint32_t* __restrict__ tile_bytes, const IssueArgs& iss,
    int32_t* __restrict__ depth, const uint64_t* __restrict__ tile_text_off, uint8_t* __restrict__ text, uint64_t text_cap,
    const uint32_t* __restrict__ g_lut);

// The same by-products for TWO tiles per wave, 32 lanes each: a tile of 40x long-read data holds ~20 events, so a whole wave
// per tile leaves two thirds of its lanes idle, and the pass is bound by instruction issue.  Everything that is per tile is
// per lane here (uniform over a half); ranks come through ds_bpermute instead of v_readlane, scans stop at 32 lanes (the DPP
// sequence without its last step), the 64-bit sum is two 32-bit ones (sum of segment lengths <= 4096: length * low / high
// half of the depth cannot overflow).  For tiles with at most HALF_MAX events.
#define HALF_MAX 30       // + the null event of the half's first lane + one lane for the padding behind a contig end

__device__ __forceinline__ int32_t half_inclusive_i32(int32_t v)
{
    v = dpp_add<0x111, 0xF>(v);
    v = dpp_add<0x112, 0xF>(v);
    v = dpp_add<0x114, 0xF>(v);
    v = dpp_add<0x118, 0xF>(v);
    v = dpp_add<0x142, 0xA>(v);
    return v;
}

__device__ __forceinline__ void tile_sparse1_half(
    int64_t tile, uint32_t e0, uint32_t n_ev, uint32_t n_ev_max, const uint16_t* __restrict__ events, int32_t carry_in, int32_t c,
    int64_t elem0, int64_t L, long long* __restrict__ tile_sum, uint32_t* __restrict__ tile_bytes, const IssueArgs& iss, int lane)
{
    const int hl = lane & 31, hb = lane & 32;
    const int32_t valid = (int32_t)min((int64_t)TILE, L - elem0);
    // first lane of the half: a null event at position 0; lanes 1 .. n_ev: the events
    const bool has = hl >= 1 && (uint32_t)hl <= n_ev;
    const uint32_t ev = has ? (uint32_t)events[e0 + hl - 1] : 0u;
    const uint32_t key = hl == 0 ? 0u : has ? ((ev >> 1) << 7) | (uint32_t)hl : 0xFFFFFFFFu;
    uint32_t rank = 0;
    for (uint32_t j = 0; j <= n_ev_max; j++)
        rank += (uint32_t)__builtin_amdgcn_ds_bpermute((int)((hb + j) << 2), (int)key) < key ? 1u : 0u;
    const int32_t delta = has ? ((ev & 1u) ? -1 : 1) : 0;
    const uint32_t packed = ((has ? ev >> 1 : hl == 0 ? 0u : (uint32_t)TILE) << 2) | (uint32_t)(delta + 1);
    const uint32_t dst = (uint32_t)hl <= n_ev ? (uint32_t)hb + rank : (uint32_t)lane;
    const uint32_t got = (uint32_t)__builtin_amdgcn_ds_permute((int)(dst << 2), (int)packed);
    const int32_t p = min((int32_t)(got >> 2), valid);
    const int32_t d = carry_in + half_inclusive_i32((int32_t)(got & 3u) - 1);
    int32_t p_next = __shfl_down(p, 1, 64);
    if (hl == 31) p_next = valid;
    p_next = min(p_next, valid);
    const int32_t seg = p_next - p;
    const int32_t s_lo = half_inclusive_i32(seg * (int32_t)((uint32_t)d & 0xFFFFu));
    const int32_t s_hi = half_inclusive_i32(seg * (d >> 16));
    const int32_t bytes = half_inclusive_i32(seg * (int32_t)(ndigits_fast((uint32_t)d) + 1u));
    if (hl == 31) { tile_sum[tile] = (long long)s_hi * 65536ll + (long long)s_lo; tile_bytes[tile] = (uint32_t)bytes; }
    if (!iss.keys) return;
    // ---- issue-run boundaries (the rules of tile_sparse1, masks cut to the half) ----------------------------------------
    const int64_t wa = gci_slice_bound(iss.flank, L);
    int64_t wb = gci_slice_bound(L - iss.flank, L);
    if (wb < wa) wb = wa;
    const bool low = d >= iss.lo && d <= iss.hi;
    const bool nonempty = seg > 0;
    const uint32_t m_ne = (uint32_t)(__ballot(nonempty) >> hb), m_low = (uint32_t)(__ballot(nonempty && low) >> hb);
    const uint32_t below = hl ? m_ne & ((1u << hl) - 1u) : 0u;
    const uint32_t above = hl < 31 ? m_ne >> (hl + 1) : 0u;
    const bool first = below == 0u;
    const bool prev_in_window = elem0 - 1 >= wa && elem0 - 1 < wb;
    const bool low_prev = first ? (prev_in_window && carry_in >= iss.lo && carry_in <= iss.hi)
                                : ((m_low >> (31 - __builtin_clz(below))) & 1u) != 0u;
    const bool has_next = above != 0u;
    const bool low_next = has_next && ((m_low >> (hl + 1 + __builtin_ctz(above))) & 1u) != 0u;
    if (!nonempty) return;
    const int64_t A = elem0 + p, B = elem0 + p_next;
    const int64_t a = max(A, wa), b = min(B, wb);
    auto put = [&](int64_t rel, bool is_end) {
        const uint32_t slot = atomicAdd(iss.n_keys, 1u);
        if (slot < iss.cap) iss.keys[slot] = issue_key((uint32_t)c, rel, is_end);
    };
    if (low && a < b) {
        if (A <= wa || !low_prev) put(a - wa, false);
        if (b == wb) put(wb - wa, true);
        else if (has_next && !low_next) put(b - wa, true);
    } else if (first && low_prev && A < wb) {
        put(A - wa, true);
    }
}

// One workgroup, eight tiles: every wave does two tiles from their event lists (together when both are small, else one after
// the other); a tile with too many events is then done densely by the whole workgroup (no second launch: a dependent launch
// costs ~4.6 us even when it finds nothing).
#define PASS1_TILES (2 * (BLOCK / 64))
__global__ __launch_bounds__(BLOCK) void k_tile_pass1(
    const uint16_t* __restrict__ events, const uint32_t* __restrict__ evt_off, const int32_t* __restrict__ tile_carry,
    const int64_t* __restrict__ tile_first, const int64_t* __restrict__ len, int32_t n_contigs, int64_t n_tiles,
    long long* __restrict__ tile_sum, uint32_t* __restrict__ tile_bytes, IssueArgs iss, int32_t sparse_max,
    const uint32_t* __restrict__ g_lut)
{
    const int lane = threadIdx.x & 63;
    // the wave index is uniform: say so, and everything per tile of the one-tile path (bounds, carry, offsets, loop counts) lives in SGPRs
    const int64_t t0 = ((int64_t)blockIdx.x * (BLOCK / 64) + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6))) * 2;
    if (t0 < n_tiles) {
        const bool pair = t0 + 1 < n_tiles;
        const int64_t mine = min(t0 + (lane >> 5), n_tiles - 1);            // the tile of this lane's half
        const uint32_t e0 = evt_off[mine], n_ev = evt_off[mine + 1] - e0;
        if (pair && __all((int64_t)n_ev <= (int64_t)min(sparse_max, HALF_MAX))) {
            const int32_t c = contig_of_tile(tile_first, n_contigs, mine);
            const uint32_t n_max = max((uint32_t)__builtin_amdgcn_readlane((int)n_ev, 0), (uint32_t)__builtin_amdgcn_readlane((int)n_ev, 32));
            tile_sparse1_half(mine, e0, n_ev, n_max, events, tile_carry[mine], c, (mine - tile_first[c]) * TILE, len[c], tile_sum, tile_bytes,
                              iss, lane);
        } else {
            for (int k = 0; k < (pair ? 2 : 1); k++) {
                const int64_t tile = t0 + k;
                const uint32_t f0 = evt_off[tile], f1 = evt_off[tile + 1];
                if ((int64_t)(f1 - f0) <= sparse_max) {
                    const int32_t c = contig_of_tile(tile_first, n_contigs, tile);
                    tile_sparse1(tile, f0, f1 - f0, events, tile_carry[tile], c, (tile - tile_first[c]) * TILE, len[c], tile_sum, tile_bytes,
                                 iss, lane);
                }
            }
        }
    }
    // (written out rather than looped: a loop around tile_dense doubles its register count)
#define DENSE_ONE(k)                                                                                                        \
    {                                                                                                                       \
        const int64_t td = (int64_t)blockIdx.x * PASS1_TILES + (k);                                                         \
        if (td < n_tiles && (int64_t)(evt_off[td + 1] - evt_off[td]) > sparse_max) {                                        \
            __syncthreads();                                                                                                \
            tile_dense<1>(td, events, evt_off, tile_carry, tile_first, len, n_contigs, tile_sum, tile_bytes, iss, nullptr,  \
                          nullptr, nullptr, 0, g_lut);                                                                      \
        }                                                                                                                   \
    }
    DENSE_ONE(0) DENSE_ONE(1) DENSE_ONE(2) DENSE_ONE(3) DENSE_ONE(4) DENSE_ONE(5) DENSE_ONE(6) DENSE_ONE(7)
#undef DENSE_ONE
}

// ---- PASS 2 (depth + text) ------------------------------------------------------------------------
// Same idea as pass 1: a tile with few events is a handful of constant-depth segments, so one wave writes it segment
// by segment -- 16-byte splat stores for the depth, and for the text the decimal pattern of the segment ("37\n")
// repeated through a 64-bit register, 16 bytes per lane and store, phase taken from the distance to the segment
// start.  The 16-byte groups that hold a boundary between two segments are composed in registers, all boundaries of
// the tile at once (lane r owns the boundary at the start of segment r), so that nothing but whole aligned groups
// reaches memory, except at the two ends of the tile's text.  No LDS, no workgroup barrier.  A tile with more than
// SPARSE_MAX events, or with a depth of four or more digits when text is wanted, is done densely by a whole
// workgroup (k_tile_dense).

#ifdef GCI_TILE_TRACE           // tools/exp_tile_trace.py: shader-clock stamps per tile
#define TT(i) do { if ((threadIdx.x & 63) == 0) g_tt[(i)] = clock64(); } while (0)
#define TT_PARAM , unsigned long long* g_tt
#define TT_ARG , g_tt
#else
#define TT(i) do {} while (0)
#define TT_PARAM
#define TT_ARG
#endif

__device__ __forceinline__ uint32_t mod_w(uint32_t x, uint32_t w)        // x mod w for x < 2^16, w in {2, 3, 4}
{
    const uint32_t m3 = x - 3u * ((x * 43691u) >> 17);
    return w == 3u ? m3 : x & (w - 1u);
}

// the 16 bytes of a w-byte pattern (held repeated in X, 8 valid bytes) whose first byte has phase ph < w
__device__ __forceinline__ uint4 pattern16(unsigned long long X, uint32_t w, uint32_t ph)
{
    // 4 = 1 (mod 3) and 0 (mod 2 or 4): the phase moves by one per dword only for w == 3
    const bool three = w == 3u;
    uint32_t p1 = three ? ph + 1u : ph;  p1 = three && p1 >= 3u ? p1 - 3u : p1;
    uint32_t p2 = three ? p1 + 1u : ph;  p2 = three && p2 >= 3u ? p2 - 3u : p2;
    uint4 c;
    c.x = (uint32_t)(X >> (8u * ph)); c.y = (uint32_t)(X >> (8u * p1)); c.z = (uint32_t)(X >> (8u * p2)); c.w = c.x;
    return c;
}

// dword k of: bytes [0, cut) from a, bytes [cut, 16) from b
__device__ __forceinline__ uint32_t cut_dword(uint32_t a, uint32_t b, int32_t cut, int k)
{
    const int32_t n = cut - 4 * k;                                       // bytes of this dword taken from a
    if (n <= 0) return b;
    if (n >= 4) return a;
    const uint32_t m = (1u << (8 * n)) - 1u;
    return (a & m) | (b & ~m);
}

// returns false when the tile needs the dense path (nothing has been written then)
//
// What bounds this kernel besides the HBM write rate is VALU issue (a wave64 instruction takes two issue cycles of its
// SIMD-32 and a CU works through ~60 tiles): everything that is uniform over a segment is kept in scalar registers
// (v_readlane of the owning lane), so a group of 16 text bytes costs a handful of vector instructions.
__device__ __forceinline__ bool tile_sparse2(
    int64_t tile, uint32_t e0, uint32_t n_ev, const uint16_t* __restrict__ events, int32_t carry_in, int32_t valid,
    int32_t* __restrict__ depth, uint64_t T0, uint8_t* __restrict__ text, uint64_t text_cap, int lane, uint32_t wi,
    uint32_t nw, int2* __restrict__ run_list TT_PARAM)
{
    // wi / nw: this wave is one of nw that share the tile: every one derives the segments, wave 0 writes the groups
    // that hold boundaries, and the whole groups of segment r go to wave r mod nw
    TT(1);
    Seg sg = sparse_segments(e0, n_ev, events, carry_in, valid, lane);
    TT(2);
    // want_runs: the segments -- {depth, length}, in order, possibly empty, neighbours possibly of equal depth -- are what the
    // .depth.gz encoder (k_deflate.hip) wants to know about this tile: one coalesced store of 8 bytes per lane instead of a
    // second read of the tile's 16 KB
    if (run_list && wi == 0 && (uint32_t)lane <= n_ev) run_list[(size_t)tile * GCI_RUN_MAX + lane] = make_int2(sg.d, sg.len);
    // padding behind the contig end: the first idle lane owns [valid, TILE) at depth 0 and has no text
    const bool pad_lane = (uint32_t)lane == n_ev + 1u && valid < TILE;
    if (pad_lane) { sg.p = valid; sg.p_next = TILE; sg.len = TILE - valid; sg.d = 0; }
    const uint32_t n_seg = n_ev + 2u;                                    // lanes 0 .. n_ev + 1 can own something
    const uint32_t d = (uint32_t)sg.d;
    const int32_t len_t = pad_lane ? 0 : sg.len;
    if (text && __ballot(len_t > 0 && d >= 1000u) != 0ull) return false;
    const unsigned long long lanes_below = lane ? (1ull << lane) - 1ull : 0ull;

    // ---- depth ---------------------------------------------------------------------------------------------------
    if (depth) {
        int32_t* dt = depth + (size_t)tile * TILE;
        int4* dt4 = reinterpret_cast<int4*>(dt);
        const bool ne = sg.len > 0;
        const unsigned long long m_ne = __ballot(ne);
        const unsigned long long below = m_ne & lanes_below;
        const bool has_prev = below != 0ull;
        const int q = has_prev ? 63 - __builtin_clzll(below) : 0;           // previous non-empty segment
        const int32_t p_q = __shfl(sg.p, q, 64), d_q = __shfl(sg.d, q, 64);
        const int32_t al = sg.p & 3, gs = sg.p - al;                        // the group of four that holds the boundary
        const bool hb = ne && al != 0;
        const bool prev_covers = has_prev && p_q <= gs;                     // ... from its first element
        const bool simple = hb && prev_covers && sg.p_next >= gs + 4;
        if (simple && wi == 0) dt4[gs >> 2] = make_int4(d_q, al > 1 ? d_q : sg.d, al > 2 ? d_q : sg.d, sg.d);
        // several boundaries inside one group: every segment writes its own elements, and the previous segment's when
        // that one began before the group (its own boundary logic then never sees this group).  One boundary at a
        // time, four lanes.
        for (unsigned long long m = wi == 0 ? __ballot(hb && !simple) : 0ull; m; m &= m - 1ull) {
            const int r = __builtin_ctzll(m);
            const int32_t pr = __builtin_amdgcn_readlane(sg.p, r), pn = __builtin_amdgcn_readlane(sg.p_next, r);
            const int32_t e = (pr & ~3) + lane;
            const bool own = lane < 4 && e >= pr && e < pn;
            const bool prv = lane < 4 && __builtin_amdgcn_readlane((int)prev_covers, r) && e < pr;
            if (own || prv) dt[e] = own ? __builtin_amdgcn_readlane(sg.d, r) : __builtin_amdgcn_readlane(d_q, r);
        }
        // whole groups of every segment: one store instruction per segment and 1024 bytes
        const int32_t g0v = (sg.p + 3) >> 2, g1v = sg.p_next >> 2;
        for (uint32_t r = wi; r < n_seg; r += nw) {
            const int32_t g0 = __builtin_amdgcn_readlane(g0v, (int)r), g1 = __builtin_amdgcn_readlane(g1v, (int)r);
            if (g0 >= g1) continue;
            const int32_t dr = __builtin_amdgcn_readlane(sg.d, (int)r);
            // Non-temporal: the 247 MB of the track stream past the caches, and the 185 MB of text -- written with plain
            // stores below -- stay in the 256 MB Infinity Cache.  Same kernel time, the kernels around it 8 - 15 us faster
            // per step; with the text non-temporal as well (or instead) the kernel itself slows down by 10 - 25 us.
            const i32x4 v4 = {dr, dr, dr, dr};
#pragma clang loop vectorize(disable) unroll(disable)
            for (int32_t g = g0 + lane; g < g1; g += 64) {
#if TB_TRACK_NT
                __builtin_nontemporal_store(v4, reinterpret_cast<i32x4*>(dt4) + g);
#else
                reinterpret_cast<i32x4*>(dt4)[g] = v4;
#endif
            }
        }
    }
    TT(3);
    if (!text) return true;

    // ---- text ----------------------------------------------------------------------------------------------------
    // "row": the 16-byte aligned byte string TB that holds the tile's text; the text starts at row byte A0
    const uint32_t A0 = (uint32_t)(((uint64_t)(uintptr_t)text + T0) & 15ull);
    const int64_t row0 = (int64_t)T0 - (int64_t)A0;                         // offset of TB in the caller's buffer (>= -15:
    uint8_t* TB = text + row0;                                              //  a buffer need not start 16-byte aligned)
    const int64_t room = (int64_t)text_cap - row0;
    const uint32_t capu = room <= 0 ? 0u : room > 0x7FFFFFFFll ? 0x7FFFFFFFu : (uint32_t)room;   // row bytes [0, capu) may be written
    // decimal pattern of this lane's segment: "d\n", w = digits + 1 bytes, repeated to 8 bytes
    const uint32_t q10 = d / 10u, r0 = d - 10u * q10, q100 = q10 / 10u, r1 = q10 - 10u * q100;
    const uint32_t w = d >= 100u ? 4u : d >= 10u ? 3u : 2u;
    const uint32_t pd = d >= 100u ? (('0' + q100) | (('0' + r1) << 8) | (('0' + r0) << 16) | ((uint32_t)'\n' << 24))
                      : d >= 10u ? (('0' + r1) | (('0' + r0) << 8) | ((uint32_t)'\n' << 16))
                                 : (('0' + r0) | ((uint32_t)'\n' << 8));
    const unsigned long long X = w == 4u ? (unsigned long long)pd | ((unsigned long long)pd << 32)
                               : w == 3u ? (unsigned long long)pd * 0x0001000001000001ull
                                         : (unsigned long long)pd * 0x0001000100010001ull;
    const uint32_t Xlo = (uint32_t)X, Xhi = (uint32_t)(X >> 32);
    const uint32_t bytes = (uint32_t)len_t * w;
    const uint32_t u2 = A0 + (uint32_t)wave_inclusive_i32((int32_t)bytes), u1 = u2 - bytes;   // the segment is row bytes [u1, u2)
    const bool net = len_t > 0;
    const unsigned long long m_net = __ballot(net);
    if (m_net == 0ull) return true;
    const unsigned long long below = m_net & lanes_below;
    const bool has_prev = below != 0ull;
    const int q = has_prev ? 63 - __builtin_clzll(below) : 0;
    const uint32_t u1_q = (uint32_t)__shfl((int)u1, q, 64), w_q = (uint32_t)__shfl((int)w, q, 64);
    const unsigned long long X_q = ((unsigned long long)(uint32_t)__shfl((int)Xhi, q, 64) << 32) | (uint32_t)__shfl((int)Xlo, q, 64);
    const uint32_t al = u1 & 15u, cs = u1 - al;                              // the 16-byte group that holds the boundary
    const bool hb = net && al != 0u;
    const bool prev_covers = has_prev && u1_q <= cs;
    const bool to_group_end = u2 >= cs + 16u;
    const bool simple = hb && prev_covers && to_group_end && cs + 16u <= capu;
    if (simple && wi == 0) {
        const uint4 a = pattern16(X_q, w_q, mod_w(cs - u1_q, w_q));
        const uint4 b = pattern16(X, w, mod_w(cs + 48u - u1, w));           // (cs - u1) mod w: 48 = 0 (mod 2, 3, 4)
        uint4 c;
        c.x = cut_dword(a.x, b.x, (int32_t)al, 0); c.y = cut_dword(a.y, b.y, (int32_t)al, 1);
        c.z = cut_dword(a.z, b.z, (int32_t)al, 2); c.w = cut_dword(a.w, b.w, (int32_t)al, 3);
        *reinterpret_cast<uint4*>(TB + cs) = c;
    }
    // The two ends of the tile's text, one byte-store instruction for both: lanes 0-15 the head (first segment, when
    // the text starts inside a group: the bytes before belong to the previous tile), lanes 16-31 the tail (last
    // segment, when the text ends inside a group; when that segment begins inside the group its boundary logic owns it).
    const int first = __builtin_ctzll(m_net), last = 63 - __builtin_clzll(m_net);
    const bool head_own = hb && !has_prev && to_group_end;
    if (wi == 0) {
        const uint32_t fu1 = (uint32_t)__builtin_amdgcn_readlane((int)u1, first), fw = (uint32_t)__builtin_amdgcn_readlane((int)w, first);
        const unsigned long long fX = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)Xhi, first) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)Xlo, first);
        const bool f_head = __builtin_amdgcn_readlane((int)head_own, first) != 0;
        const uint32_t lu1 = (uint32_t)__builtin_amdgcn_readlane((int)u1, last), lu2 = (uint32_t)__builtin_amdgcn_readlane((int)u2, last);
        const uint32_t lw = (uint32_t)__builtin_amdgcn_readlane((int)w, last);
        const unsigned long long lX = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)Xhi, last) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)Xlo, last);
        const uint32_t lce = lu2 & ~15u;
        const bool l_tail = (lu2 & 15u) != 0u && lu1 <= lce;
        const bool is_head = lane < 16;
        const uint32_t y = is_head ? fu1 + (uint32_t)lane : lce + (uint32_t)(lane - 16);
        const bool on = lane < 32 && (is_head ? (f_head && y < ((fu1 + 15u) & ~15u)) : (l_tail && y < lu2));
        if (on && y < capu) TB[y] = is_head ? (uint8_t)(fX >> (8u * mod_w(y - fu1, fw))) : (uint8_t)(lX >> (8u * mod_w(y - lu1, lw)));
    }
    // anything else (several boundaries inside one group, a write limit inside the group): byte by byte, every segment
    // its own bytes of the group, and the previous segment's when that one began before the group.  One boundary at
    // a time, sixteen lanes.
    for (unsigned long long m = wi == 0 ? __ballot(hb && !simple && !head_own) : 0ull; m; m &= m - 1ull) {
        const int r = __builtin_ctzll(m);
        const uint32_t ru1 = (uint32_t)__builtin_amdgcn_readlane((int)u1, r), ru2 = (uint32_t)__builtin_amdgcn_readlane((int)u2, r);
        const uint32_t rw = (uint32_t)__builtin_amdgcn_readlane((int)w, r);
        const unsigned long long rX = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)Xhi, r) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)Xlo, r);
        const uint32_t qu1 = (uint32_t)__builtin_amdgcn_readlane((int)u1_q, r), qw = (uint32_t)__builtin_amdgcn_readlane((int)w_q, r);
        const unsigned long long qX = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(X_q >> 32), r) << 32) |
                                      (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)X_q, r);
        const bool q_covers = __builtin_amdgcn_readlane((int)prev_covers, r) != 0;
        const uint32_t y = (ru1 & ~15u) + (uint32_t)lane;
        const bool own = lane < 16 && y >= ru1 && y < ru2, prv = lane < 16 && q_covers && y < ru1;
        if ((own || prv) && y < capu) TB[y] = own ? (uint8_t)(rX >> (8u * mod_w(y - ru1, rw))) : (uint8_t)(qX >> (8u * mod_w(y - qu1, qw)));
    }
    TT(4);
    // whole groups of every segment.  The pattern of a segment is wave-uniform, and so is its phase at a group start
    // when w divides 16 (w = 2, 4): the group is four copies of one scalar dword.  For w = 3 the phase moves by one
    // per group (16 = 1 mod 3) and the group is a rotation of three scalar dwords.
    {
        const uint32_t c0v = (u1 + 15u) >> 4, c1v = min(u2, capu) >> 4;
        const uint32_t lane3 = (uint32_t)lane - 3u * (((uint32_t)lane * 43691u) >> 17);        // lane mod 3
        for (uint32_t r = wi; r < n_seg; r += nw) {
            const uint32_t c0 = (uint32_t)__builtin_amdgcn_readlane((int)c0v, (int)r), c1 = (uint32_t)__builtin_amdgcn_readlane((int)c1v, (int)r);
            if (c0 >= c1 || !__builtin_amdgcn_readlane((int)net, (int)r)) continue;
            const uint32_t wr = (uint32_t)__builtin_amdgcn_readlane((int)w, (int)r);
            const uint32_t x0 = 16u * c0 - (uint32_t)__builtin_amdgcn_readlane((int)u1, (int)r);   // < 16
            const unsigned long long Xr = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)Xhi, (int)r) << 32) |
                                          (uint32_t)__builtin_amdgcn_readlane((int)Xlo, (int)r);
            if (wr != 3u) {
                const uint32_t D = (uint32_t)(Xr >> (8u * (x0 & (wr - 1u))));
                const uint4 v = make_uint4(D, D, D, D);
#pragma clang loop vectorize(disable) unroll(disable)
                for (uint32_t c = c0 + lane; c < c1; c += 64) {
#if TB_TEXT_NT
                    const i32x4 vv = {(int)v.x, (int)v.y, (int)v.z, (int)v.w};
                    __builtin_nontemporal_store(vv, reinterpret_cast<i32x4*>(TB + 16u * c));
#else
                    *reinterpret_cast<uint4*>(TB + 16u * c) = v;
#endif
                }
            } else {
                const uint32_t D0 = (uint32_t)Xr, D1 = (uint32_t)(Xr >> 8), D2 = (uint32_t)(Xr >> 16);
                uint32_t ph = x0 - 3u * ((x0 * 43691u) >> 17) + lane3;                          // scalar part + lane part
                ph = ph >= 3u ? ph - 3u : ph;
#pragma clang loop vectorize(disable) unroll(disable)
                for (uint32_t c = c0 + lane; c < c1; c += 64) {
                    uint4 v;
                    v.x = ph == 0u ? D0 : ph == 1u ? D1 : D2;
                    v.y = ph == 0u ? D1 : ph == 1u ? D2 : D0;
                    v.z = ph == 0u ? D2 : ph == 1u ? D0 : D1;
                    v.w = v.x;
#if TB_TEXT_NT
                    { const i32x4 vv = {(int)v.x, (int)v.y, (int)v.z, (int)v.w}; __builtin_nontemporal_store(vv, reinterpret_cast<i32x4*>(TB + 16u * c)); }
#else
                    *reinterpret_cast<uint4*>(TB + 16u * c) = v;
#endif
                    ph = ph == 2u ? 0u : ph + 1u;                                               // 64 groups on: 64 = 1 (mod 3)
                }
            }
        }
    }
    return true;
}

// Sparse tiles, SHARE waves each; the others are flagged for k_tile_dense.
#ifndef SHARE
#define SHARE 4
#endif
// Every load a tile needs is issued before its first store: on gfx9 stores count in vmcnt, so a load issued behind
// them (s_waitcnt vmcnt(0)) would wait for every store of the wave to be acknowledged by a write-saturated memory.
__global__ __launch_bounds__(BLOCK) void k_tile_build(
    const uint16_t* __restrict__ events, const uint32_t* __restrict__ evt_off, const int32_t* __restrict__ tile_carry,
    const int32_t* __restrict__ tile_valid, int64_t n_tiles, int32_t* __restrict__ depth,
    const uint64_t* __restrict__ tile_text_off, uint8_t* __restrict__ text, uint64_t text_cap,
    uint8_t* __restrict__ dense_flag, uint32_t* __restrict__ dense_list, int32_t sparse_max, uint32_t* __restrict__ cd_words,
    uint32_t* __restrict__ run_n, int2* __restrict__ run_list
#ifdef GCI_TILE_TRACE
    , unsigned long long* __restrict__ trace
#endif
    )
{
    const int lane = threadIdx.x & 63;
    // the wave index is uniform: say so, and everything per tile (bounds, carry, offsets, loop counts) lives in SGPRs
    const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // Workgroups go to the 8 XCDs round robin: give every XCD a contiguous eighth of the tiles, so that each of them streams
    // its own stretch of the track and the text instead of all eight interleaving 16 KiB pieces of one front
    // (3.33 instead of 3.55 - 4.6 ms at genome scale, and the same from run to run).
    const uint32_t per8 = gridDim.x / 8;
    const uint32_t bx = blockIdx.x < per8 * 8 ? (blockIdx.x % 8) * per8 + blockIdx.x / 8 : blockIdx.x;
    const int64_t gw = (int64_t)bx * (BLOCK / 64) + wv;                       // SHARE consecutive waves share a tile
    const int64_t tile = gw / SHARE;
    const uint32_t wi = (uint32_t)(gw % SHARE);
    if (tile >= n_tiles) return;
    if (gw == 0 && lane == 0) dense_list[0] = 0u;                             // k_dense_list (next on the stream) counts from here
#ifdef GCI_TILE_TRACE
    unsigned long long* g_tt = trace + tile * 8;
    TT(0);
#endif
    const uint32_t e0 = evt_off[tile], e1 = evt_off[tile + 1];
    const int32_t carry_in = tile_carry[tile], valid = tile_valid[tile];
    const uint64_t T0 = text ? tile_text_off[tile] : 0ull;
    bool done = false;
    if ((int64_t)(e1 - e0) <= sparse_max)
        done = tile_sparse2(tile, e0, e1 - e0, events, carry_in, valid, depth, T0, text, text_cap, lane, wi, SHARE, run_list TT_ARG);
    if (lane == 0 && wi == 0) {
        dense_flag[tile] = done ? 0 : 1;                                     // k_tile_dense takes the rest
        if (run_n) run_n[tile] = done ? (e1 - e0) + 1u : GCI_RUNS_WALK;      // (a dense tile: the encoder reads the track)
        cd_words[2 * tile + 1] = 0u;                                         // the coarse difference has been consumed: table clean again
    }
    TT(7);
}

// The dense tiles: those k_tile_build flagged (more than sparse_max events, or a shape its sparse path declined).
// k_dense_list: a workgroup looks at the flags of BLOCK consecutive tiles, leaves when (as usual) none is set, and appends
// the flagged ones to a list (order irrelevant: tiles are independent).
__global__ __launch_bounds__(BLOCK) void k_dense_list(const uint8_t* __restrict__ dense_flag, int64_t n_tiles, uint32_t* __restrict__ list)
{
    const int64_t mine = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (mine < n_tiles && dense_flag[mine]) list[1 + atomicAdd(&list[0], 1u)] = (uint32_t)mine;
}

// One workgroup per DENSE_SPAN entries of that list; the grid is sized for "every tile is dense", and a workgroup beyond
// the list's end costs one scalar load.
#define DENSE_SPAN 8
__global__ __launch_bounds__(BLOCK) void k_tile_dense(
    const uint32_t* __restrict__ list,
    const uint16_t* __restrict__ events, const uint32_t* __restrict__ evt_off, const int32_t* __restrict__ tile_carry,
    const int64_t* __restrict__ tile_first, const int64_t* __restrict__ len, int32_t n_contigs,
    int32_t* __restrict__ depth, const uint64_t* __restrict__ tile_text_off, uint8_t* __restrict__ text, uint64_t text_cap,
    const uint32_t* __restrict__ g_lut)
{
    const uint32_t n = list[0];
    IssueArgs none;
    memset(&none, 0, sizeof none);
    // (written out rather than looped: a loop around tile_dense doubles its register count)
#define DENSE_ONE(k)                                                                                                       \
    {                                                                                                                      \
        const uint32_t at = blockIdx.x * DENSE_SPAN + (k);                                                                 \
        if (at >= n) return;                                                                                               \
        tile_dense<2>((int64_t)list[1 + at], events, evt_off, tile_carry, tile_first, len, n_contigs, nullptr, nullptr, none, depth, \
                      tile_text_off, text, text_cap, g_lut);                                                               \
        __syncthreads();                                                                                                   \
    }
    DENSE_ONE(0) DENSE_ONE(1) DENSE_ONE(2) DENSE_ONE(3) DENSE_ONE(4) DENSE_ONE(5) DENSE_ONE(6) DENSE_ONE(7)
#undef DENSE_ONE
}

// ---- host -----------------------------------------------------------------------------------------

#ifdef GCI_TILE_TRACE
#define TILE_TRACE_ARG , (unsigned long long*)strtoull(getenv("GCI_TILE_TRACE_PTR") ? getenv("GCI_TILE_TRACE_PTR") : "0", nullptr, 0)
#else
#define TILE_TRACE_ARG
#endif

static int launch_tile_build(gci_ctx* ctx, int pass, IssueArgs iss, int32_t* d_depth, uint8_t* d_text, uint64_t text_cap)
{
    const dim3 block(BLOCK);
    const uint16_t* ev = (const uint16_t*)ctx->events.p;
    const uint32_t* eo = (const uint32_t*)ctx->evt_off.p;
    const int32_t* tc = (const int32_t*)ctx->tile_carry.p;
    const int64_t* tf = (const int64_t*)ctx->d_tile_first.p;
    const int64_t* ln = (const int64_t*)ctx->d_len.p;
    const int32_t* tv = (const int32_t*)ctx->d_tile_valid.p;
    const int64_t per = BLOCK / 64;
    const dim3 grid((uint32_t)((ctx->n_tiles + PASS1_TILES - 1) / PASS1_TILES));
    const dim3 grid2((uint32_t)((ctx->n_tiles * SHARE + per - 1) / per));
    const dim3 list_grid((uint32_t)((ctx->n_tiles + BLOCK - 1) / BLOCK));
    const dim3 dense_grid((uint32_t)((ctx->n_tiles + DENSE_SPAN - 1) / DENSE_SPAN));
    uint8_t* flag = (uint8_t*)ctx->dense_flag.p;          // written by k_tile_build for every tile, read by k_dense_list
    uint32_t* list = (uint32_t*)ctx->dense_list.p;        // its counter is zeroed by k_tile_build
    if (pass == 1) {
        ProfScope _ps(ctx, GCI_PROF_TILE_PASS1);
        hipLaunchKernelGGL(k_tile_pass1, grid, block, 0, ctx->stream, ev, eo, tc, tf, ln, ctx->n_contigs, ctx->n_tiles,
                           (long long*)ctx->tile_sum.p, (uint32_t*)ctx->tile_u32.p, iss, ctx->sparse_max,
                           (const uint32_t*)ctx->text_lut.p);
        LAUNCHCHK("k_tile_pass1");
        return GCI_OK;
    } else {
        {
            ProfScope _ps(ctx, GCI_PROF_DEPTH_SCAN);
            hipLaunchKernelGGL(k_tile_build, grid2, block, 0, ctx->stream, ev, eo, tc, tv, ctx->n_tiles, d_depth,
                               (const uint64_t*)ctx->tile_u64.p, d_text, text_cap, flag, list, ctx->sparse_max,
                               (uint32_t*)ctx->tile_cd.p, ctx->build_runs_wanted ? (uint32_t*)ctx->build_nruns.p : nullptr,
                               ctx->build_runs_wanted ? (int2*)ctx->build_runs.p : nullptr TILE_TRACE_ARG);
            LAUNCHCHK("k_tile_build");
        }
        ProfScope _ps(ctx, GCI_PROF_TILE_DENSE);
        hipLaunchKernelGGL(k_dense_list, list_grid, block, 0, ctx->stream, (const uint8_t*)flag, ctx->n_tiles, list);
        hipLaunchKernelGGL(k_tile_dense, dense_grid, block, 0, ctx->stream, (const uint32_t*)list, ev, eo, tc, tf, ln,
                           ctx->n_contigs, d_depth, (const uint64_t*)ctx->tile_u64.p, d_text, text_cap, (const uint32_t*)ctx->text_lut.p);
    }
    LAUNCHCHK("k_tile_dense");
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
    ctx->build_runs_track = nullptr; ctx->build_runs_armed = false;                        // whatever lists an earlier build kept describe another track now
    ctx->build_runs_wanted = false;
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
    // tile_cd: clean (0), counted by gci_name_join_count for exactly this build (1), or in use / left over (2)
    const bool counted = o->counted != 0;
    if (counted && (ctx->cd_state != 1 || ctx->counted_flank != o->flank)) return GCI_E_INVALID;
    // large inputs: the events go through the radix partition (k_evp_*), which also fills tile_cd -- every entry, so that
    // its state does not matter; a join that left the counting to the build says so in count_deferred
    const bool radix = max_n && (counted ? ctx->count_deferred : gci_evp_wanted(ctx, max_n));
    int evp_sh = 10;
    uint32_t evp_nb = 0, evp_wg = 0;
    if (radix) {
        while (((nt + (int64_t(1) << evp_sh) - 1) >> evp_sh) > EVP_BUCKETS) evp_sh++;
        evp_nb = (uint32_t)((nt + (int64_t(1) << evp_sh) - 1) >> evp_sh);
        evp_wg = (max_n + EVP_CHUNK - 1) / EVP_CHUNK;
        const size_t n_hist = (size_t)evp_nb * evp_wg;
        GCI_TRY(gci_ensure(ctx, ctx->evp_items, (size_t)max_n * 2 * sizeof(uint32_t) + 16));
        GCI_TRY(gci_ensure(ctx, ctx->evp_hist, (n_hist + 1) * sizeof(uint32_t)));
        GCI_TRY(gci_ensure(ctx, ctx->evp_blk, ((n_hist + TILE - 1) / TILE + 2) * sizeof(uint32_t)));
        ProfScope _ps(ctx, GCI_PROF_DEPTH_DIFF);
        uint32_t* hist = (uint32_t*)ctx->evp_hist.p;
        uint32_t* items = (uint32_t*)ctx->evp_items.p;
        hipLaunchKernelGGL(k_evp_part<false>, dim3(evp_wg), dim3(BLOCK), 0, ctx->stream, d_ivl, d_n, max_n, o->flank, ln, tf,
                           ctx->n_contigs, evp_sh, evp_nb, evp_wg, hist, items);
        LAUNCHCHK("k_evp_part<hist>");
        GCI_TRY((device_exclusive_scan<uint32_t, uint32_t>(ctx, hist, hist, (uint32_t*)ctx->evp_blk.p, (int64_t)n_hist, true)));
        hipLaunchKernelGGL(k_evp_part<true>, dim3(evp_wg), dim3(BLOCK), 0, ctx->stream, d_ivl, d_n, max_n, o->flank, ln, tf,
                           ctx->n_contigs, evp_sh, evp_nb, evp_wg, hist, items);
        LAUNCHCHK("k_evp_part<scatter>");
        hipLaunchKernelGGL(k_evp_tiles<0>, dim3(evp_nb), dim3(BLOCK), sizeof(unsigned long long) << evp_sh, ctx->stream,
                           (const uint32_t*)items, (const uint32_t*)hist, evp_wg, evp_sh, nt, cd, (const uint32_t*)off, (uint16_t*)nullptr, 0u);
        LAUNCHCHK("k_evp_tiles<count>");
    } else if (!counted && ctx->cd_state != 0) {
        HIPCHK(hipMemsetAsync(cd, 0, (size_t)(nt + 1) * 8, ctx->stream));
    }
    ctx->cd_state = 2;
    ctx->count_deferred = false;
    if (max_n && !counted && !radix) {
        ProfScope _ps(ctx, GCI_PROF_DEPTH_DIFF);
        hipLaunchKernelGGL(k_evt_count, dim3((max_n + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, ctx->stream, d_ivl, d_n, max_n,
                           o->flank, ln, tf, ctx->n_contigs, cd);
        LAUNCHCHK("k_evt_count");
    }
    {
        ProfScope _ps(ctx, GCI_PROF_SCAN_TILES);
        if (nb <= FEW_BLOCKS) {
            hipLaunchKernelGGL(k_scan2_few, dim3(nb), dim3(BLOCK), 0, ctx->stream, (const uint32_t*)cd, (int32_t*)ctx->tile_carry.p,
                               off, nt, o->d_n_keys, (long long*)o->d_sums, ctx->n_contigs);
            LAUNCHCHK("k_scan2_few");
        } else {
            hipLaunchKernelGGL(k_scan2_local, dim3(nb, 2), dim3(BLOCK), 0, ctx->stream, (uint32_t*)cd, (int32_t*)ctx->tile_carry.p,
                               (int32_t*)ctx->blk_a.p, off, (uint32_t*)ctx->blk_b.p, nt, o->d_n_keys, (long long*)o->d_sums,
                               ctx->n_contigs);
            LAUNCHCHK("k_scan2_local");
            hipLaunchKernelGGL(k_scan2_add, dim3(nb + 1, 2), dim3(BLOCK), 0, ctx->stream, (int32_t*)ctx->tile_carry.p,
                               (const int32_t*)ctx->blk_a.p, off, (const uint32_t*)ctx->blk_b.p, nt, nb);
            LAUNCHCHK("k_scan2_add");
        }
    }
    if (radix) {
        ProfScope _ps(ctx, GCI_PROF_DEPTH_DIFF);
        // (a workgroup may have 64 KiB of LDS: what the cursors leave is the staging area of the range's events)
        const size_t cursors = sizeof(unsigned long long) << evp_sh;
        const uint32_t stage_cap = (uint32_t)((65536 - cursors) / sizeof(uint16_t));
        hipLaunchKernelGGL(k_evp_tiles<1>, dim3(evp_nb), dim3(BLOCK), cursors + stage_cap * sizeof(uint16_t), ctx->stream,
                           (const uint32_t*)ctx->evp_items.p, (const uint32_t*)ctx->evp_hist.p, evp_wg, evp_sh, nt, cd,
                           (const uint32_t*)off, (uint16_t*)ctx->events.p, stage_cap);
        LAUNCHCHK("k_evp_tiles<place>");
    } else if (max_n) {
        ProfScope _ps(ctx, GCI_PROF_DEPTH_DIFF);
        hipLaunchKernelGGL(k_evt_scatter, dim3((max_n + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, ctx->stream, d_ivl, d_n, max_n,
                           o->flank, ln, tf, ctx->n_contigs, (uint32_t*)cd, (const uint32_t*)off, (uint16_t*)ctx->events.p);
        LAUNCHCHK("k_evt_scatter");
    }
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
        if (nb <= FEW_BLOCKS && (o->want_text || o->d_sums)) {
            ProfScope _ps(ctx, GCI_PROF_TEXT_COUNT);
            const uint32_t reduce_blocks = o->d_sums ? (uint32_t)ctx->n_contigs * REDUCE_SPLIT : 0u;
            hipLaunchKernelGGL(k_after_pass1_few, dim3((uint32_t)nb + reduce_blocks), dim3(BLOCK), 0, ctx->stream,
                               (const uint32_t*)ctx->tile_u32.p, (unsigned long long*)ctx->tile_u64.p, nt, (uint32_t)nb, tf,
                               ctx->n_contigs, o->want_text ? o->d_contig_text_off : (uint64_t*)nullptr,
                               (const long long*)ctx->tile_sum.p, (unsigned long long*)o->d_sums);
            LAUNCHCHK("k_after_pass1_few");
        } else {
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
    }
    if (o->want_runs) {
        GCI_TRY(gci_ensure(ctx, ctx->build_nruns, (size_t)nt * sizeof(uint32_t)));
        GCI_TRY(gci_ensure(ctx, ctx->build_runs, (size_t)nt * GCI_RUN_MAX * sizeof(int2)));
        ctx->build_runs_wanted = true;
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
    GCI_TRY(launch_tile_build(ctx, 2, none, d_depth, d_text, text_cap));
    if (ctx->build_runs_wanted) ctx->build_runs_track = d_depth;
    ctx->cd_state = 0;                                      // k_evt_scatter returned the counts, k_tile_build the differences
    return GCI_OK;
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
