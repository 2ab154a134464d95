// gci_ctx.hpp -- internals shared by the translation units of libgci_hip.so: the context, the
// error / launch macros, the per-kernel HIP-event scopes and small device helpers.
//
// Everything on this path is integer, streaming or scatter work: no MFMA anywhere.  The bound is
// HBM bandwidth (depth build, issue scan, text), atomic rate (join) or instruction issue (record
// decode).  Wave = 64 lanes, 256-thread workgroups, 16-byte vector accesses, 4096-element
// (16 KiB) tiles that never straddle contigs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>

#include "../../include/gci_hip.h"
#include "gci_common.h"

#define TILE GCI_TILE
#define TEXT_LUT 1000                    // depths below this come out of a 4-byte-per-entry table
#define BLOCK 256
static_assert(TILE == BLOCK * 16, "a tile is 16 elements per thread");

constexpr int GCI_RUN_MAX = 64;                  // run lists of a tile (k_tile_build / k_depth_runs -> k_depth_deflate): entries per tile
constexpr uint32_t GCI_RUNS_WALK = 0xFFFFFFFFu;  // ... "not listed: walk the track"

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
};

struct gci_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string err;
    // layout
    int32_t n_contigs = 0;
    std::vector<int64_t> len, off;          // host copies
    std::vector<int64_t> tile_first;        // n_contigs + 1
    int64_t total = 0, n_tiles = 0;
    DevBuf d_len, d_off, d_tile_first;      // int64 each
    // depth-build scratch (per tile unless noted)
    DevBuf tile_cd;                         // uint64 per tile: low word = events in the tile, high word = coarse
                                            // difference (int32); both return to zero by the end of a build
    DevBuf tile_carry;                      // int32: exclusive scan of the coarse difference = depth entering a tile
    DevBuf evt_off;                         // uint32: bucket offsets (n_tiles + 1)
    DevBuf d_tile_valid;                    // int32 per tile: elements of the tile inside its contig (TILE but for the last)
    DevBuf dense_flag;                      // uint8 per tile: pass 2 left it to the dense kernel
    DevBuf dense_list;                      // uint32: [0] = number of flagged tiles, [1 ..] their indices (k_dense_list)
    int32_t sparse_max = 62;                // tiles with more events take the dense path (GCI_FORCE_DENSE=1: all of them)
    bool join_dirty = false;                // the join tables are not in their clean state (a join was cut short)
    int join_mode = 0;                      // gci_join_mode: 0 by size, 1 classic table, 2 radix-partitioned
    uint32_t k1_parity = 0;                 // which of the two K1 counter sets the next gci_bam_filter uses
    int cd_state = 0;                       // tile_cd: 0 clean, 1 counted by gci_name_join_count, 2 in use / left over
    int counted_flank = 0;                  // the flank gci_name_join_count counted with
    bool count_deferred = false;            // cd_state == 1 without counts: the build buckets the events itself (gci_evp_wanted)
    DevBuf evp_items, evp_hist, evp_blk;    // large inputs: the events partitioned by tile range (k_evp_*)
    DevBuf events;                          // uint16 per event: local position << 1 | is_minus
    DevBuf blk_a, blk_b;                    // block totals of the two scans
    DevBuf tile_sum;                        // int64: sum of depth per tile
    DevBuf tile_u32, tile_u64, blk_u64;     // text bytes per tile, byte offsets (n_tiles + 1), block totals
    uint32_t build_max_n = 0;               // capacity the pending begin() was issued with
    int build_flank = 0;
    bool build_pending = false, build_text = false;
    bool build_runs_wanted = false;         // gci_build_opts.want_runs of the pending build
    DevBuf build_nruns, build_runs;         // ... per tile of the layout its constant-depth runs, as k_tile_build saw them
    const void* build_runs_track = nullptr; // ... and the track they describe (nullptr: none; cleared by whatever writes a track)
    bool build_runs_armed = false;          // ... gci_depth_deflate_from_build() has said that track is untouched: the next size / write pair may use them
    // join scratch
    DevBuf join_table, join_last, join_hq;
    DevBuf join_bucket;                     // partitioned join: per bucket its survivor count, first slot and output offset
    DevBuf part_a, part_b, part_hist, part_blk;   // partitioned join: entry ping-pong, histograms + segment table, scan totals
    DevBuf route_tab;                       // gci_route_*: per (part, chunk) counts and their scan
    DevBuf deflate_nruns, deflate_runs;     // gci_depth_deflate_*: per tile its constant-depth runs (k_depth_runs)
    DevBuf deflate_tab;                     // ... the CRC tables of the size pass (k_deflate.hip: host_crc_tab)
    bool deflate_tab_ready = false;
    bool deflate_from_build = false;        // ... the last size call took the lists of the build (build_runs) instead
    uint32_t deflate_members = 0;           // ... of the members the last size call measured,
    const void* deflate_key_depth = nullptr;    // ... over this track
    const void* deflate_key_elem = nullptr;     // ... and this member table (the write call reuses the lists only for the same three)
    DevBuf conflict_table;                  // gci_hash_conflicts' own open-addressing tables (two, used alternately)
    uint64_t conflict_slots = 0;            // slots per table
    uint32_t conflict_parity = 0;           // which one the next call inserts into (the other one is clean by then)
    DevBuf text_lut;                        // uint32[TEXT_LUT]: decimal characters of 0..999
    DevBuf long_items;                      // K1: queue of long-CIGAR records + its counter
    DevBuf pg_cost, pg_scan, pg_first;      // record pages: per-record cost / blob bytes, their scans, first record of a page
    uint32_t pg_n_rec = 0, pg_page_bytes = 0, pg_n_pages = 0;
    uint64_t pg_blob_off = 0;
    uint64_t pg_blob_bytes = 0;               // total size of the blob the size call measured (without its 16 guard bytes)
    std::vector<DevBuf> paf_pool;           // K2's scratch, in the order a call asks for it (k_paf.hip: PafScratch)
    DevBuf crc_tabs;                        // k_bgzf_crc: the 32 look-up tables (k_crc_tables), made on first use
    bool crc_tabs_ready = false;            // ... set once k_crc_tables has been launched without an error
    uint32_t inflate_last_n = 0;            // members of the last gci_bgzf_inflate_device call (0: the wave decoder was not used)
    DevBuf inflate_sym, inflate_nsym, inflate_wstatus, inflate_lists, inflate_prof, inflate_next;
    DevBuf inflate_sym2, inflate_lists2;    // ... of the batches that run on the second stream
    int inflate_streams = 1;                // gci_bgzf_inflate_streams(): 2 = every other batch of members on inflate_stream2
    hipStream_t inflate_stream2 = nullptr;  // k_inflate_wave.hip: every other batch of members on a stream of its own (made on first use)
    hipEvent_t inflate_ev_in = nullptr, inflate_ev_out = nullptr;   // k_inflate_wave.hip: a batch's symbol streams, per member its symbols and how it fared
    DevBuf tail_sums;                       // gci_two_type_tail: per-tile sums of the three tracks
    DevBuf tail_gaps;                       // gci_two_type_tail: the N runs as absolute sorted [begin, end) element ranges
    std::vector<int64_t> tail_gaps_host;    // ... what was uploaded last
    // issue-scan windows
    DevBuf win, win_tile_first;
    int win_flank = INT32_MIN;              // flank the cached per-contig windows were built for
    uint32_t win_n = 0;
    int64_t win_tiles = 0;
    void* h_pinned = nullptr;               // staging for small uploads
    size_t h_pinned_cap = 0;
    // optional per-kernel HIP-event timing (gci_profile_*)
    int prof_mask = 0;
    struct ProfEv { int id; hipEvent_t a, b; };
    std::vector<ProfEv> prof_live, prof_free;
    double prof_ms[GCI_PROF_COUNT] = {0};
    uint64_t prof_n[GCI_PROF_COUNT] = {0};
};

// Scoped HIP-event pair around one launch (or a few), recorded on the ctx stream when that
// kernel id is enabled.  Events are pooled; elapsed times are folded in by gci_profile_read().
struct ProfScope {
    gci_ctx* c; gci_ctx::ProfEv ev; bool on;
    ProfScope(gci_ctx* ctx, int kid) : c(ctx), on(false) {
        if (!(ctx->prof_mask & (1 << kid))) return;
        if (!ctx->prof_free.empty()) { ev = ctx->prof_free.back(); ctx->prof_free.pop_back(); }
        else if (hipEventCreate(&ev.a) != hipSuccess || hipEventCreate(&ev.b) != hipSuccess) return;
        ev.id = kid;
        on = hipEventRecord(ev.a, ctx->stream) == hipSuccess;
    }
    ~ProfScope() {
        if (!on) return;
        (void)hipEventRecord(ev.b, c->stream);
        c->prof_live.push_back(ev);
    }
};

int gci_fail(gci_ctx* c, hipError_t e, const char* what);
int gci_ensure(gci_ctx* ctx, DevBuf& b, size_t bytes);
int gci_upload_small(gci_ctx* ctx, void* d_dst, const void* h_src, size_t bytes);

// every device allocation of the library comes from the arena of k_hbm.hip (slabs of GBs, cut up here: no driver call per buffer)
hipError_t gci_dmalloc(int device, void** out, size_t bytes);
hipError_t gci_dfree(void* p);

#define HIPCHK(call)                                      \
    do {                                                  \
        hipError_t _e = (call);                           \
        if (_e != hipSuccess) return gci_fail(ctx, _e, #call); \
    } while (0)
#define LAUNCHCHK(name)                                   \
    do {                                                  \
        hipError_t _e = hipGetLastError();                \
        if (_e != hipSuccess) return gci_fail(ctx, _e, name); \
    } while (0)
#define GCI_TRY(expr)                                     \
    do {                                                  \
        int _r = (expr);                                  \
        if (_r) return _r;                                \
    } while (0)

// ---------------------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------------------

// contig of a tile: largest c with tile_first[c] <= tile   (tile_first has n + 1 entries)
__device__ __forceinline__ int32_t contig_of_tile(const int64_t* __restrict__ tile_first, int32_t n, int64_t tile)
{
    int32_t lo = 0, hi = n;          // invariant: tile_first[lo] <= tile < tile_first[hi]
    while (hi - lo > 1) {
        int32_t mid = (lo + hi) >> 1;
        if (tile_first[mid] <= tile) lo = mid; else hi = mid;
    }
    return lo;
}

__device__ __forceinline__ uint32_t ld_u32(const uint8_t* p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }
__device__ __forceinline__ int32_t ld_i32(const uint8_t* p) { int32_t v; __builtin_memcpy(&v, p, 4); return v; }
__device__ __forceinline__ uint16_t ld_u16(const uint8_t* p) { uint16_t v; __builtin_memcpy(&v, p, 2); return v; }

// Large inputs bucket their events by radix partition instead of one device-scope atomic per event (k_depth.hip, k_evp_*);
// gci_name_join_count then leaves the counting to the build.  GCI_EVENTS=atomic|radix overrides the size rule.
#define EVP_MAX_TILES (int64_t(1) << 23)
static inline bool gci_evp_wanted(const gci_ctx* ctx, uint64_t n_items)
{
    if (ctx->n_tiles == 0 || ctx->n_tiles > EVP_MAX_TILES) return false;
    const char* e = getenv("GCI_EVENTS");
    if (e && !strcmp(e, "atomic")) return false;
    if (e && !strcmp(e, "radix")) return true;
    return n_items >= (1u << 20);
}

// ---- interval -> tile events (shared by k_evt_count / k_evt_scatter and the counting join) -------------------------
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
__device__ __forceinline__ void count_span(const IvlSpan& s, unsigned long long* __restrict__ tile_cd)
{
    const unsigned long long minus1 = 0xFFFFFFFFull << 32;         // -1 in the high word (the low word never carries)
    atomicAdd(tile_cd + s.tile_a, 1ull | (1ull << 32));
    if (s.has_b && s.tile_b == s.tile_bc) atomicAdd(tile_cd + s.tile_b, 1ull | minus1);
    else {
        if (s.has_b) atomicAdd(tile_cd + s.tile_b, 1ull);
        atomicAdd(tile_cd + s.tile_bc, minus1);
    }
}

// ---- wave-level scans and sums -------------------------------------------------------------------
// 32-bit values go through DPP (one v_add_*_dpp per step: row_shr 1/2/4/8 inside each 16-lane row, then
// row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3); 64-bit values through ds_bpermute.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int32_t dpp_add(int32_t v)
{
    return v + __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xF, true);
}

__device__ __forceinline__ int32_t wave_inclusive_i32(int32_t v)
{
    v = dpp_add<0x111, 0xF>(v);
    v = dpp_add<0x112, 0xF>(v);
    v = dpp_add<0x114, 0xF>(v);
    v = dpp_add<0x118, 0xF>(v);
    v = dpp_add<0x142, 0xA>(v);
    v = dpp_add<0x143, 0xC>(v);
    return v;
}

template <typename T>
__device__ __forceinline__ T wave_inclusive(T v, int lane)
{
    if constexpr (sizeof(T) == 4) {
        return (T)wave_inclusive_i32((int32_t)v);
    } else {
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { T n = __shfl_up(v, d, 64); if (lane >= d) v += n; }
        return v;
    }
}

template <typename T>
__device__ __forceinline__ T wave_sum(T v)
{
    if constexpr (sizeof(T) == 4) {
        return (T)__builtin_amdgcn_readlane(wave_inclusive_i32((int32_t)v), 63);
    } else {
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
        return v;
    }
}

__device__ __forceinline__ uint32_t ndigits(uint32_t v)
{
    return 1u + (v >= 10u) + (v >= 100u) + (v >= 1000u) + (v >= 10000u) + (v >= 100000u) + (v >= 1000000u) +
           (v >= 10000000u) + (v >= 100000000u) + (v >= 1000000000u);
}

// depths are small almost everywhere: five compares, the long tail behind a rarely taken branch
__device__ __forceinline__ uint32_t ndigits_fast(uint32_t v)
{
    uint32_t n = 1u + (v >= 10u) + (v >= 100u) + (v >= 1000u) + (v >= 10000u);
    if (__builtin_expect(v >= 100000u, 0)) n = ndigits(v);
    return n;
}

// `lo < d <= hi` for an integer d as integer bounds: d in [ilo, ihi]  (empty range: ilo > ihi)
struct IntRange { int32_t lo, hi; };
static inline IntRange gci_int_range(double lo, double hi)
{
    IntRange r;
    if (!(lo == lo) || !(hi == hi)) { r.lo = 1; r.hi = 0; return r; }           // NaN: every comparison is false
    double fl = __builtin_floor(lo) + 1.0, fh = __builtin_floor(hi);
    if (fl < -2147483648.0) fl = -2147483648.0;
    if (fh > 2147483647.0) fh = 2147483647.0;
    if (fl > fh) { r.lo = 1; r.hi = 0; return r; }
    r.lo = (int32_t)fl; r.hi = (int32_t)fh;
    return r;
}

// issue-scan boundary key: (window << 33) | (rel << 1) | is_end
__device__ __forceinline__ unsigned long long issue_key(uint32_t window, int64_t rel, bool is_end)
{
    return ((unsigned long long)window << 33) | ((unsigned long long)rel << 1) | (is_end ? 1ull : 0ull);
}

// ---------------------------------------------------------------------------------------------
// exclusive scan over per-tile tables (n up to millions): local scan per 4096 entries + add
// ---------------------------------------------------------------------------------------------

template <typename TIn, typename TOut>
__device__ __forceinline__ void scan_local_body(const TIn* __restrict__ in, TOut* __restrict__ out,
                                                TOut* __restrict__ blk_tot, int64_t n, uint32_t blk, int64_t stride = 1)
{
    __shared__ TOut wtot[BLOCK / 64];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int64_t base = (int64_t)blk * TILE + (int64_t)t * 16;
    TOut v[16];
    TOut run = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) { v[i] = base + i < n ? (TOut)in[(base + i) * stride] : (TOut)0; run += v[i]; }
    const TOut inc = wave_inclusive<TOut>(run, lane);
    if (lane == 63) wtot[wave] = inc;
    __syncthreads();
    TOut pre = inc - run;
    TOut all = 0;
#pragma unroll
    for (int w = 0; w < BLOCK / 64; w++) { if (w < wave) pre += wtot[w]; all += wtot[w]; }
#pragma unroll
    for (int i = 0; i < 16; i++) { if (base + i < n) out[base + i] = pre; pre += v[i]; }
    if (t == 0) blk_tot[blk] = all;
}

// out[i] += sum of blk_tot[0 .. block(i) - 1]; the extra block n_blocks writes the grand total to out[n]
template <typename TOut>
__device__ __forceinline__ void scan_add_body(TOut* __restrict__ out, const TOut* __restrict__ blk_tot, int64_t n,
                                              int32_t n_blocks, uint32_t me)
{
    __shared__ TOut part[BLOCK / 64];
    __shared__ TOut s_pre;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    TOut s = 0;
    for (int32_t b = t; b < (int32_t)me; b += BLOCK) s += blk_tot[b];
    s = wave_sum<TOut>(s);
    if (lane == 0) part[wave] = s;
    __syncthreads();
    if (t == 0) { TOut a = 0; for (int w = 0; w < BLOCK / 64; w++) a += part[w]; s_pre = a; }
    __syncthreads();
    const TOut pre = s_pre;
    if ((int32_t)me == n_blocks) { if (t == 0) out[n] = pre; return; }
    if (pre == 0) return;
    const int64_t base = (int64_t)me * TILE + (int64_t)t * 16;
#pragma unroll
    for (int i = 0; i < 16; i++) if (base + i < n) out[base + i] += pre;
}

template <typename TIn, typename TOut>
__global__ __launch_bounds__(BLOCK) void k_scan_local(const TIn* __restrict__ in, TOut* __restrict__ out,
                                                      TOut* __restrict__ blk_tot, int64_t n)
{
    scan_local_body<TIn, TOut>(in, out, blk_tot, n, blockIdx.x);
}

template <typename TOut>
__global__ __launch_bounds__(BLOCK) void k_scan_add(TOut* __restrict__ out, const TOut* __restrict__ blk_tot, int64_t n,
                                                    int32_t n_blocks)
{
    scan_add_body<TOut>(out, blk_tot, n, n_blocks, blockIdx.x);
}

// out must have n + 1 entries when write_total is set
template <typename TIn, typename TOut>
static inline int device_exclusive_scan(gci_ctx* ctx, const TIn* in, TOut* out, TOut* blk_tot, int64_t n, bool write_total)
{
    const int32_t nb = (int32_t)((n + TILE - 1) / TILE);
    if (nb == 0) {
        if (write_total) HIPCHK(hipMemsetAsync(out, 0, sizeof(TOut), ctx->stream));
        return GCI_OK;
    }
    hipLaunchKernelGGL((k_scan_local<TIn, TOut>), dim3(nb), dim3(BLOCK), 0, ctx->stream, in, out, blk_tot, n);
    LAUNCHCHK("k_scan_local");
    if (nb > 1 || write_total) {
        hipLaunchKernelGGL((k_scan_add<TOut>), dim3(nb + (write_total ? 1 : 0)), dim3(BLOCK), 0, ctx->stream, out,
                           blk_tot, n, nb);
        LAUNCHCHK("k_scan_add");
    }
    return GCI_OK;
}

// ---------------------------------------------------------------------------------------------
// decimal text of depth values through LDS staging (K10)
// ---------------------------------------------------------------------------------------------

#define TEXT_STAGE 16384                 // LDS bytes of staging: 16 for the alignment shift + text

// lut[x] = the characters of "x\n" packed little endian (x < 1000: at most 3 digits + newline).
// Built once per context in global memory (gci_text_lut_host); every workgroup copies it to LDS.
static inline void gci_text_lut_host(uint32_t* lut)
{
    for (uint32_t x = 0; x < TEXT_LUT; x++) {
        const uint32_t d2 = x / 100u, d1 = (x / 10u) % 10u, d0 = x % 10u;
        if (x >= 100u) lut[x] = ('0' + d2) | (('0' + d1) << 8) | (('0' + d0) << 16) | ((uint32_t)'\n' << 24);
        else if (x >= 10u) lut[x] = ('0' + d1) | (('0' + d0) << 8) | ((uint32_t)'\n' << 16);
        else lut[x] = ('0' + d0) | ((uint32_t)'\n' << 8);
    }
}

__device__ __forceinline__ void text_lut_load(uint32_t* lut, const uint32_t* __restrict__ g_lut, int t)
{
    for (uint32_t x = t; x < TEXT_LUT; x += BLOCK) lut[x] = g_lut[x];
}

__device__ __forceinline__ void lds_put_u32(uint8_t* p, uint32_t v) { __builtin_memcpy(p, &v, 4); }   // any alignment

// write the decimal lines of 4 consecutive elements into the staging buffer at byte offset o.
// Common case -- all four below 1000 with the same width W (digits + newline): the 4 * W bytes are assembled
// from the table words in registers and leave as W unaligned dword stores (gfx950 LDS takes any alignment).
__device__ __forceinline__ void text_put4(const uint32_t (&v)[4], const uint32_t (&nd)[4], uint32_t o, uint8_t* stage,
                                          const uint32_t* lut)
{
    const uint32_t W = nd[0];
    const bool uniform = W >= 2u && nd[1] == W && nd[2] == W && nd[3] == W &&
                         v[0] < TEXT_LUT && v[1] < TEXT_LUT && v[2] < TEXT_LUT && v[3] < TEXT_LUT;
    if (uniform) {
        const uint32_t w0 = lut[v[0]], w1 = lut[v[1]], w2 = lut[v[2]], w3 = lut[v[3]];
        if (W == 3u) {
            lds_put_u32(stage + o, w0 | (w1 << 24));
            lds_put_u32(stage + o + 4, (w1 >> 8) | (w2 << 16));
            lds_put_u32(stage + o + 8, (w2 >> 16) | (w3 << 8));
        } else if (W == 2u) {
            lds_put_u32(stage + o, w0 | (w1 << 16));
            lds_put_u32(stage + o + 4, w2 | (w3 << 16));
        } else {
            lds_put_u32(stage + o, w0); lds_put_u32(stage + o + 4, w1);
            lds_put_u32(stage + o + 8, w2); lds_put_u32(stage + o + 12, w3);
        }
        return;
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (!nd[k]) continue;
        if (v[k] < TEXT_LUT) {
            const uint32_t w = lut[v[k]];
            stage[o] = (uint8_t)w;
            stage[o + 1] = (uint8_t)(w >> 8);
            if (nd[k] > 2) stage[o + 2] = (uint8_t)(w >> 16);
            if (nd[k] > 3) stage[o + 3] = (uint8_t)(w >> 24);
        } else {
            uint32_t x = v[k];
            const uint32_t e = o + nd[k] - 1;
            stage[e] = '\n';
            for (uint32_t d = 1; d < nd[k]; d++) { stage[e - d] = (uint8_t)('0' + x % 10u); x /= 10u; }
        }
        o += nd[k];
    }
}

// staging -> global.  The staging buffer mirrors the 16-byte alignment of the destination (the text starts
// at stage[shift], shift = dst & 15), so whole chunks move as ds_read_b128 + global_store_dwordx4; the first
// and last partial chunk are written byte-wise and only where this tile owns the bytes.
__device__ __forceinline__ void text_copy_out(uint8_t* __restrict__ g, const uint8_t* stage, uint32_t shift,
                                              uint32_t total, int t)
{
    uint8_t* ga = g - shift;
    const uint32_t end = shift + total;
    const uint32_t nchunks = (end + 15u) >> 4;
    for (uint32_t c = t; c < nchunks; c += BLOCK) {
        const uint32_t lo = c << 4, hi = lo + 16u;
        if (lo >= shift && hi <= end) {
            *reinterpret_cast<int4*>(ga + lo) = *reinterpret_cast<const int4*>(stage + lo);
        } else {
            const uint32_t a = lo > shift ? lo : shift, b = hi < end ? hi : end;
            for (uint32_t i = a; i < b; i++) ga[i] = stage[i];
        }
    }
}

// Decimal text of one tile held in registers (thread t, group j: elements (j * 256 + t) * 4 .. + 3) to
// out[dst ...].  One staging round
// when the tile's text fits the LDS buffer (it does unless depths reach 5+ digits), else one per group.
// Needs: stage[TEXT_STAGE] (16-byte aligned), wtot[4][BLOCK / 64], lut[TEXT_LUT] built and synced.
template <bool FULL>
__device__ __forceinline__ void text_tile(const int4 (&v)[4], int64_t valid, uint8_t* stage,
                                          uint32_t (*wtot)[BLOCK / 64], const uint32_t* lut, uint8_t* __restrict__ out,
                                          uint64_t dst, uint64_t cap, int t, int lane, int wave)
{
    uint32_t u[4][4], nd[4][4], mine[4], inc[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        u[j][0] = (uint32_t)v[j].x; u[j][1] = (uint32_t)v[j].y; u[j][2] = (uint32_t)v[j].z; u[j][3] = (uint32_t)v[j].w;
        const int64_t i = (int64_t)(j * BLOCK + t) * 4;
        mine[j] = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            nd[j][k] = (FULL || i + k < valid) ? ndigits_fast(u[j][k]) + 1u : 0u;
            mine[j] += nd[j][k];
        }
        inc[j] = (uint32_t)wave_inclusive_i32((int32_t)mine[j]);
    }
    if (lane == 63) {
#pragma unroll
        for (int j = 0; j < 4; j++) wtot[j][wave] = inc[j];
    }
    __syncthreads();
    uint32_t off[4], gtot[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        uint32_t pre = 0, all = 0;
#pragma unroll
        for (int w = 0; w < BLOCK / 64; w++) { const uint32_t x = wtot[j][w]; if (w < wave) pre += x; all += x; }
        off[j] = pre + inc[j] - mine[j];
        gtot[j] = all;
    }
    const uint32_t tile_bytes = gtot[0] + gtot[1] + gtot[2] + gtot[3];
    if (dst + tile_bytes > cap) return;                       // caller's buffer too small: write nothing
    if (tile_bytes + 16u <= TEXT_STAGE) {
        const uint32_t shift = (uint32_t)((uintptr_t)(out + dst) & 15u);
        uint32_t base = shift;
#pragma unroll
        for (int j = 0; j < 4; j++) { text_put4(u[j], nd[j], base + off[j], stage, lut); base += gtot[j]; }
        __syncthreads();
        text_copy_out(out + dst, stage, shift, tile_bytes, t);
    } else {
        for (int j = 0; j < 4; j++) {
            const uint32_t shift = (uint32_t)((uintptr_t)(out + dst) & 15u);
            text_put4(u[j], nd[j], shift + off[j], stage, lut);
            __syncthreads();
            text_copy_out(out + dst, stage, shift, gtot[j], t);
            dst += gtot[j];
            __syncthreads();
        }
    }
}

// ---------------------------------------------------------------------------------------------
// small kernels used by more than one translation unit
// ---------------------------------------------------------------------------------------------

// per-contig reduction of the per-tile sums: REDUCE_SPLIT workgroups per contig, each adds its share with one
// 64-bit atomic (sums must be zeroed first)
#define REDUCE_SPLIT 32
__attribute__((unused)) static __global__ __launch_bounds__(BLOCK) void k_reduce_tiles(const long long* __restrict__ tile_sum,
                                                        const int64_t* __restrict__ tile_first,
                                                        unsigned long long* __restrict__ sums)
{
    __shared__ long long part[BLOCK / 64];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int64_t a = tile_first[blockIdx.x], b = tile_first[blockIdx.x + 1];
    long long s = 0;
    for (int64_t i = a + (int64_t)blockIdx.y * BLOCK + t; i < b; i += (int64_t)REDUCE_SPLIT * BLOCK) s += tile_sum[i];
    s = wave_sum<long long>(s);
    if (lane == 0) part[wave] = s;
    __syncthreads();
    if (t == 0) {
        const long long v = part[0] + part[1] + part[2] + part[3];
        if (v) atomicAdd(sums + blockIdx.x, (unsigned long long)v);
    }
}

__attribute__((unused)) static __global__ void k_contig_text_off(const uint64_t* __restrict__ tile_off, const int64_t* __restrict__ tile_first,
                                  int32_t n_contigs, int64_t n_tiles, uint64_t* __restrict__ contig_off)
{
    const int32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c > n_contigs) return;
    contig_off[c] = c == n_contigs ? tile_off[n_tiles] : tile_off[tile_first[c]];
}

