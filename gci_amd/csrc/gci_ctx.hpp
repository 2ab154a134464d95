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
#include <string.h>
#include <string>
#include <vector>

#include "../../include/gci_hip.h"
#include "gci_common.h"

#define TILE GCI_TILE
#define BLOCK 256
static_assert(TILE == BLOCK * 16, "a tile is 16 elements per thread");

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
    DevBuf tile_diff, tile_carry;           // int32: coarse difference table, its exclusive scan
    DevBuf evt_cnt, evt_off;                // uint32: events per tile, bucket offsets (n_tiles + 1)
    DevBuf events;                          // uint16 per event: local position << 1 | is_minus
    DevBuf blk_a, blk_b;                    // block totals of the two scans
    DevBuf tile_sum;                        // int64: sum of depth per tile
    DevBuf tile_u32, tile_u64, blk_u64;     // text bytes per tile, byte offsets (n_tiles + 1), block totals
    uint32_t build_max_n = 0;               // capacity the pending begin() was issued with
    int build_flank = 0;
    bool build_pending = false, build_text = false;
    // join scratch
    DevBuf join_table, join_last, join_hq;
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

// inclusive scan across the 64 lanes of a wave
template <typename T>
__device__ __forceinline__ T wave_inclusive(T v, int lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { T n = __shfl_up(v, d, 64); if (lane >= d) v += n; }
    return v;
}

template <typename T>
__device__ __forceinline__ T wave_sum(T v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

__device__ __forceinline__ uint32_t ndigits(uint32_t v)
{
    return 1u + (v >= 10u) + (v >= 100u) + (v >= 1000u) + (v >= 10000u) + (v >= 100000u) + (v >= 1000000u) +
           (v >= 10000000u) + (v >= 100000000u) + (v >= 1000000000u);
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
                                                TOut* __restrict__ blk_tot, int64_t n, uint32_t blk)
{
    __shared__ TOut wtot[BLOCK / 64];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int64_t base = (int64_t)blk * TILE + (int64_t)t * 16;
    TOut v[16];
    TOut run = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) { v[i] = base + i < n ? (TOut)in[base + i] : (TOut)0; run += v[i]; }
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

#define TEXT_SUB 1024                    // elements per text staging round
#define TEXT_STAGE (TEXT_SUB * 11)       // worst case: 10 digits + '\n'

// staging -> global: byte head up to a 4-byte boundary, dword body, byte tail
__device__ __forceinline__ void copy_out(uint8_t* __restrict__ g, const uint8_t* stage, uint32_t total, int t)
{
    const uint32_t head = min((uint32_t)((4 - ((uintptr_t)g & 3)) & 3), total);
    const uint32_t nw = (total - head) >> 2;
    if ((uint32_t)t < head) g[t] = stage[t];
    uint32_t* gw = reinterpret_cast<uint32_t*>(g + head);
    for (uint32_t wi = t; wi < nw; wi += BLOCK) {
        const uint8_t* s = stage + head + 4 * wi;
        gw[wi] = (uint32_t)s[0] | ((uint32_t)s[1] << 8) | ((uint32_t)s[2] << 16) | ((uint32_t)s[3] << 24);
    }
    const uint32_t done = head + 4 * nw;
    if ((uint32_t)t < total - done) g[done + t] = stage[done + t];
}

// Render the decimal lines of up to TEXT_SUB elements (4 per thread) through LDS staging.
// Returns the bytes this round produced.  Ends with a __syncthreads().
__device__ __forceinline__ uint32_t text_round(const uint32_t (&v)[4], int64_t i0, int64_t valid, uint8_t* stage,
                                               uint32_t* wtot, uint8_t* __restrict__ out, uint64_t dst, uint64_t cap,
                                               int t, int lane, int wave)
{
    uint32_t nd[4], mine = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) { nd[k] = i0 + k < valid ? ndigits(v[k]) + 1 : 0; mine += nd[k]; }
    const uint32_t inc = wave_inclusive<uint32_t>(mine, lane);
    if (lane == 63) wtot[wave] = inc;
    __syncthreads();
    uint32_t o = inc - mine, total = 0;
#pragma unroll
    for (int w = 0; w < BLOCK / 64; w++) { if (w < wave) o += wtot[w]; total += wtot[w]; }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (nd[k]) {
            uint32_t x = v[k];
            const uint32_t e = o + nd[k] - 1;
            stage[e] = '\n';
            for (uint32_t d = 1; d < nd[k]; d++) { stage[e - d] = (uint8_t)('0' + x % 10u); x /= 10u; }
            o += nd[k];
        }
    }
    __syncthreads();
    if (dst + total <= cap) copy_out(out + dst, stage, total, t);
    __syncthreads();
    return total;
}

// ---------------------------------------------------------------------------------------------
// small kernels used by more than one translation unit
// ---------------------------------------------------------------------------------------------

// per-contig reduction of the per-tile sums (one workgroup per contig)
__attribute__((unused)) static __global__ __launch_bounds__(BLOCK) void k_reduce_tiles(const long long* __restrict__ tile_sum,
                                                        const int64_t* __restrict__ tile_first,
                                                        long long* __restrict__ sums)
{
    __shared__ long long part[BLOCK / 64];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int64_t a = tile_first[blockIdx.x], b = tile_first[blockIdx.x + 1];
    long long s = 0;
    for (int64_t i = a + t; i < b; i += BLOCK) s += tile_sum[i];
    s = wave_sum<long long>(s);
    if (lane == 0) part[wave] = s;
    __syncthreads();
    if (t == 0) sums[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

__attribute__((unused)) static __global__ void k_contig_text_off(const uint64_t* __restrict__ tile_off, const int64_t* __restrict__ tile_first,
                                  int32_t n_contigs, int64_t n_tiles, uint64_t* __restrict__ contig_off)
{
    const int32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c > n_contigs) return;
    contig_off[c] = c == n_contigs ? tile_off[n_tiles] : tile_off[tile_first[c]];
}

