// k_track.hip -- kernels over an existing depth track: K6 gap mask, K7 two-type max, per-contig sums,
// K8 issue scan, K10 decimal text.  These serve the seams that run on a track AFTER it was built
// (masked, merged or uploaded tracks); a fresh build gets the same results fused into
// k_tile_build (k_depth.hip) without re-reading the track.
#include "gci_ctx.hpp"
#include <stdlib.h>
#include <algorithm>
#include <utility>

// ============================================================================================
// K6: gap mask (GCI.py:324-328), K7: two-type max (GCI.py:350)
// ============================================================================================

__global__ __launch_bounds__(BLOCK) void k_gap_mask(int32_t* __restrict__ depth, const gci_ivl* __restrict__ gaps,
                                                    const int64_t* __restrict__ len, const int64_t* __restrict__ off,
                                                    int32_t n_contigs)
{
    const gci_ivl g = gaps[blockIdx.y];
    if (g.contig < 0 || g.contig >= n_contigs) return;
    const int64_t L = len[g.contig];
    const int64_t a = gci_slice_bound(g.start, L), b = gci_slice_bound(g.end, L);
    int32_t* d = depth + off[g.contig];
    for (int64_t p = a + (int64_t)blockIdx.x * BLOCK + threadIdx.x; p < b; p += (int64_t)gridDim.x * BLOCK) d[p] = 0;
}

extern "C" int gci_gap_mask(gci_ctx* ctx, int32_t* d_depth, const gci_ivl* d_gaps, uint32_t n_gaps)
{
    if (!ctx || !d_depth || (n_gaps && !d_gaps)) return GCI_E_INVALID;
    if (!ctx->n_contigs) return GCI_E_NO_LAYOUT;
    ctx->build_runs_track = nullptr; ctx->build_runs_armed = false;                       // (a build's run lists, gci_build_opts.want_runs, describe no track any more)
    ProfScope _ps(ctx, GCI_PROF_GAP_MASK);
    for (uint32_t done = 0; done < n_gaps; done += 65535) {
        const uint32_t n = n_gaps - done < 65535 ? n_gaps - done : 65535;
        hipLaunchKernelGGL(k_gap_mask, dim3(32, n), dim3(BLOCK), 0, ctx->stream, d_depth, d_gaps + done,
                           (const int64_t*)ctx->d_len.p, (const int64_t*)ctx->d_off.p, ctx->n_contigs);
        LAUNCHCHK("k_gap_mask");
    }
    return GCI_OK;
}

__global__ __launch_bounds__(BLOCK) void k_max2(const int4* __restrict__ a, const int4* __restrict__ b,
                                                int4* __restrict__ o, int64_t n4)
{
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n4; i += (int64_t)gridDim.x * BLOCK) {
        const int4 x = a[i], y = b[i];
        int4 r;
        r.x = max(x.x, y.x); r.y = max(x.y, y.y); r.z = max(x.z, y.z); r.w = max(x.w, y.w);
        o[i] = r;
    }
}

extern "C" int gci_max2(gci_ctx* ctx, const int32_t* a, const int32_t* b, int32_t* o)
{
    if (!ctx || !a || !b || !o) return GCI_E_INVALID;
    if (!ctx->n_contigs) return GCI_E_NO_LAYOUT;
    ctx->build_runs_track = nullptr; ctx->build_runs_armed = false;
    const int64_t n4 = ctx->total / 4;
    if (n4 == 0) return GCI_OK;
    const int64_t want = (n4 + BLOCK * 4 - 1) / (BLOCK * 4);
    const uint32_t grid = (uint32_t)(want < 1 ? 1 : want > 16384 ? 16384 : want);
    ProfScope _ps(ctx, GCI_PROF_MAX2);
    hipLaunchKernelGGL(k_max2, dim3(grid), dim3(BLOCK), 0, ctx->stream, (const int4*)a, (const int4*)b, (int4*)o, n4);
    LAUNCHCHK("k_max2");
    return GCI_OK;
}

// ============================================================================================
// R15: per-contig sums -- per-tile partials, then one workgroup per contig (no same-address atomics:
// one returning atomic per tile on a single word costs ~12 ns each and serialises)
// ============================================================================================

__global__ __launch_bounds__(BLOCK) void k_tile_sum(const int32_t* __restrict__ depth, long long* __restrict__ tile_sum)
{
    __shared__ long long part[BLOCK / 64];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int4* base = reinterpret_cast<const int4*>(depth + (size_t)blockIdx.x * TILE);
    long long s = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) { const int4 v = base[j * BLOCK + t]; s += (long long)v.x + v.y + v.z + v.w; }
    s = wave_sum<long long>(s);
    if (lane == 0) part[wave] = s;
    __syncthreads();
    if (t == 0) tile_sum[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

extern "C" int gci_depth_sum(gci_ctx* ctx, const int32_t* d_depth, int64_t* d_sums)
{
    if (!ctx || !d_depth || !d_sums) return GCI_E_INVALID;
    if (!ctx->n_contigs) return GCI_E_NO_LAYOUT;
    if (ctx->n_tiles == 0) { HIPCHK(hipMemsetAsync(d_sums, 0, (size_t)ctx->n_contigs * 8, ctx->stream)); return GCI_OK; }
    ProfScope _ps(ctx, GCI_PROF_DEPTH_SUM);
    HIPCHK(hipMemsetAsync(d_sums, 0, (size_t)ctx->n_contigs * 8, ctx->stream));
    hipLaunchKernelGGL(k_tile_sum, dim3((uint32_t)ctx->n_tiles), dim3(BLOCK), 0, ctx->stream, d_depth,
                       (long long*)ctx->tile_sum.p);
    LAUNCHCHK("k_tile_sum");
    hipLaunchKernelGGL(k_reduce_tiles, dim3(ctx->n_contigs, REDUCE_SPLIT), dim3(BLOCK), 0, ctx->stream,
                       (const long long*)ctx->tile_sum.p, (const int64_t*)ctx->d_tile_first.p, (unsigned long long*)d_sums);
    LAUNCHCHK("k_reduce_tiles");
    return GCI_OK;
}

// ============================================================================================
// K8: issue scan (collapse_depth_range, GCI.py:369-390) as run-boundary detection
// ============================================================================================
//
// g[p] = (lo < depth[p] <= hi) and p inside the window.  A boundary sits at p when g[p] != g[p-1]:
// a run START if g[p], else the (exclusive) END of the run before it; a run that reaches the window
// end is closed at the window end by the thread holding its last element.  Only the predecessor is
// needed, so each thread reads one extra element.  Low-depth runs are rare (CHM13: 11, MH63: 2328):
// boundaries are appended with one atomic each and the kernel is a pure 4 B/base read stream.

__global__ __launch_bounds__(BLOCK) void k_issue_scan(const int32_t* __restrict__ depth,
                                                      const gci_window* __restrict__ win,
                                                      const int64_t* __restrict__ win_tile_first, int32_t n_win,
                                                      int32_t lo, int32_t hi, unsigned long long* __restrict__ keys,
                                                      uint32_t cap, uint32_t* __restrict__ n_keys)
{
    const int t = threadIdx.x, lane = t & 63;
    const int32_t w = contig_of_tile(win_tile_first, n_win, blockIdx.x);
    const gci_window W = win[w];
    const int64_t p0 = (W.begin / TILE + ((int64_t)blockIdx.x - win_tile_first[w])) * TILE;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int64_t p = p0 + (int64_t)(j * BLOCK + t) * 4;
        const int4 v = *reinterpret_cast<const int4*>(depth + p);
        const int32_t d[4] = {v.x, v.y, v.z, v.w};
        bool g[4];
#pragma unroll
        for (int k = 0; k < 4; k++) g[k] = (p + k >= W.begin) && (p + k < W.end) && (d[k] >= lo) && (d[k] <= hi);
        int gp = __shfl_up((int)g[3], 1, 64);
        if (lane == 0) {
            gp = 0;
            if (p - 1 >= W.begin && p - 1 < W.end) { const int32_t x = depth[p - 1]; gp = (x >= lo) && (x <= hi); }
        }
        if (!(g[0] | g[1] | g[2] | g[3] | (bool)gp)) continue;
        bool prev = (bool)gp;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int64_t q = p + k;
            if (g[k] != prev && q >= W.begin && (g[k] || q < W.end)) {
                const uint32_t s = atomicAdd(n_keys, 1u);
                if (s < cap) keys[s] = issue_key((uint32_t)w, q - W.begin, !g[k]);
            }
            if (g[k] && q == W.end - 1) {
                const uint32_t s = atomicAdd(n_keys, 1u);
                if (s < cap) keys[s] = issue_key((uint32_t)w, W.end - W.begin, true);
            }
            prev = g[k];
        }
    }
}

static int issue_scan_launch(gci_ctx* ctx, const int32_t* d_depth, uint32_t n_win, int64_t n_tiles, double lo, double hi,
                             uint64_t* d_keys, uint32_t cap, uint32_t* d_n_keys)
{
    HIPCHK(hipMemsetAsync(d_n_keys, 0, 4, ctx->stream));
    if (n_tiles == 0) return GCI_OK;
    ProfScope _ps(ctx, GCI_PROF_ISSUE_SCAN);
    const IntRange rg = gci_int_range(lo, hi);
    hipLaunchKernelGGL(k_issue_scan, dim3((uint32_t)n_tiles), dim3(BLOCK), 0, ctx->stream, d_depth,
                       (const gci_window*)ctx->win.p, (const int64_t*)ctx->win_tile_first.p, (int32_t)n_win, rg.lo, rg.hi,
                       (unsigned long long*)d_keys, cap, d_n_keys);
    LAUNCHCHK("k_issue_scan");
    return GCI_OK;
}

static int set_windows(gci_ctx* ctx, const gci_window* h_win, uint32_t n_win)
{
    std::vector<gci_window> ws;
    std::vector<int64_t> first;
    ws.reserve(n_win + 1);
    first.reserve(n_win + 2);
    int64_t tiles = 0;
    for (uint32_t i = 0; i < n_win; i++) {
        gci_window w = h_win[i];
        if (w.begin < 0) w.begin = 0;
        if (w.end > ctx->total) w.end = ctx->total;
        if (w.end < w.begin) w.end = w.begin;
        first.push_back(tiles);
        if (w.end > w.begin) tiles += (w.end + TILE - 1) / TILE - w.begin / TILE;
        ws.push_back(w);
    }
    first.push_back(tiles);
    GCI_TRY(gci_ensure(ctx, ctx->win, (size_t)(n_win + 1) * sizeof(gci_window)));
    GCI_TRY(gci_ensure(ctx, ctx->win_tile_first, (size_t)(n_win + 2) * 8));
    if (n_win) GCI_TRY(gci_upload_small(ctx, ctx->win.p, ws.data(), n_win * sizeof(gci_window)));
    GCI_TRY(gci_upload_small(ctx, ctx->win_tile_first.p, first.data(), first.size() * 8));
    ctx->win_n = n_win;
    ctx->win_tiles = tiles;
    return GCI_OK;
}

extern "C" int gci_issue_scan_windows(gci_ctx* ctx, const int32_t* d_depth, const gci_window* h_windows,
                                      uint32_t n_windows, double lo, double hi, uint64_t* d_keys, uint32_t cap,
                                      uint32_t* d_n_keys)
{
    if (!ctx || !d_depth || !d_n_keys || (cap && !d_keys) || (n_windows && !h_windows)) return GCI_E_INVALID;
    if (!ctx->n_contigs) return GCI_E_NO_LAYOUT;
    if (n_windows >= (1u << 31)) return GCI_E_INVALID;
    GCI_TRY(set_windows(ctx, h_windows, n_windows));
    ctx->win_flank = INT32_MIN;
    return issue_scan_launch(ctx, d_depth, n_windows, ctx->win_tiles, lo, hi, d_keys, cap, d_n_keys);
}

extern "C" int gci_issue_scan(gci_ctx* ctx, const int32_t* d_depth, double lo, double hi, int flank, uint64_t* d_keys,
                              uint32_t cap, uint32_t* d_n_keys)
{
    if (!ctx || !d_depth || !d_n_keys || (cap && !d_keys)) return GCI_E_INVALID;
    if (!ctx->n_contigs) return GCI_E_NO_LAYOUT;
    if (ctx->win_flank != flank) {
        // depth_list[flank_len : chr_len - flank_len] with Python slice normalisation (GCI.py:374)
        std::vector<gci_window> ws(ctx->n_contigs);
        for (int32_t c = 0; c < ctx->n_contigs; c++) {
            const int64_t L = ctx->len[c];
            int64_t a = gci_slice_bound(flank, L), b = gci_slice_bound(L - flank, L);
            if (b < a) b = a;
            ws[c].begin = ctx->off[c] + a;
            ws[c].end = ctx->off[c] + b;
        }
        GCI_TRY(set_windows(ctx, ws.data(), (uint32_t)ctx->n_contigs));
        ctx->win_flank = flank;
    }
    return issue_scan_launch(ctx, d_depth, ctx->win_n, ctx->win_tiles, lo, hi, d_keys, cap, d_n_keys);
}

// ============================================================================================
// The tail of a two-read-type run in ONE pass (GCI.py:1014-1024): gap masks of both tracks (GCI.py:324-328), their per-base
// maximum (GCI.py:350) and the issue-run boundaries of all three tracks (GCI.py:369-390).
// ============================================================================================
//
// Done seam by seam -- gci_gap_mask x 2, gci_max2, gci_issue_scan x 3 -- the two tracks are read twice and the merged one once
// more: 24 bytes per base.  Here a workgroup takes one 4096-base tile of both tracks, zeroes the bases inside N runs in
// registers (and writes a tile back only if a run touched it), writes the maximum and tests every base of the three tracks
// against `lo < d <= hi` inside the contig's window [flank, L - flank): 12 bytes per base.  The N runs arrive as absolute,
// sorted, disjoint [begin, end) element ranges of the track (made on the host from the reference's contig coordinates with
// Python's slice rules); a tile finds its first run by binary search, and nearly every tile has none.
struct TailArgs {
    int32_t* a; int32_t* b; int32_t* out;
    const int64_t* gaps; uint32_t n_gaps;                    // [begin, end) pairs, sorted, disjoint
    const int64_t* len; const int64_t* off; const int64_t* tile_first; int32_t n_contigs;
    int32_t lo, hi, flank;
    unsigned long long* keys; uint32_t cap; uint32_t* n_keys;     // three key arrays of `cap` entries, three counters
    long long* tile_sums; int64_t n_tiles;                        // nullable: 3 x n_tiles sums of depth (a, b, the maximum)
};

__device__ __forceinline__ bool tail_in_gap(const int64_t* __restrict__ gaps, uint32_t g0, uint32_t g1, int64_t p)
{
    for (uint32_t g = g0; g < g1; g++) if (p >= gaps[2 * g] && p < gaps[2 * g + 1]) return true;
    return false;
}

__global__ __launch_bounds__(BLOCK) void k_two_type_tail(TailArgs A)
{
    const int t = threadIdx.x, lane = t & 63;
    const int32_t c = contig_of_tile(A.tile_first, A.n_contigs, blockIdx.x);
    const int64_t L = A.len[c], o = A.off[c];
    int64_t wa = gci_slice_bound(A.flank, L), wb = gci_slice_bound(L - A.flank, L);
    if (wb < wa) wb = wa;
    const int64_t Wb = o + wa, We = o + wb;                  // the window of this contig, in track elements
    const int64_t p0 = (int64_t)blockIdx.x * TILE;
    // the N runs that touch this tile: [g0, g1)
    uint32_t g0 = 0, g1 = 0;
    if (A.n_gaps) {
        uint32_t lo_i = 0, hi_i = A.n_gaps;                  // first run with end > p0
        while (lo_i < hi_i) { const uint32_t mid = (lo_i + hi_i) >> 1; if (A.gaps[2 * mid + 1] > p0) hi_i = mid; else lo_i = mid + 1; }
        g0 = lo_i;
        g1 = g0;
        while (g1 < A.n_gaps && A.gaps[2 * g1] < p0 + TILE) g1++;
    }
    const bool gapped = g1 > g0;
    long long sum[3] = {0, 0, 0};                            // (the padding behind a contig's last base is kept at zero)
    // (not unrolled: four groups in flight are 110 VGPRs and four waves per SIMD with the sums, 19.0 ms for the 6.1 Gb diploid; one
    // group is 58 VGPRs and eight waves, 14.4 ms -- the kernel streams, it wants the waves more than the groups)
#pragma unroll 1
    for (int j = 0; j < 4; j++) {
        const int64_t p = p0 + (int64_t)(j * BLOCK + t) * 4;
        int4 va = *reinterpret_cast<const int4*>(A.a + p), vb = *reinterpret_cast<const int4*>(A.b + p);
        int32_t da[4] = {va.x, va.y, va.z, va.w}, db[4] = {vb.x, vb.y, vb.z, vb.w};
        if (gapped) {
            bool touched = false;
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (tail_in_gap(A.gaps, g0, g1, p + k)) { da[k] = 0; db[k] = 0; touched = true; }
            if (touched) {
                *reinterpret_cast<int4*>(A.a + p) = make_int4(da[0], da[1], da[2], da[3]);
                *reinterpret_cast<int4*>(A.b + p) = make_int4(db[0], db[1], db[2], db[3]);
            }
        }
        int32_t dm[4];
#pragma unroll
        for (int k = 0; k < 4; k++) dm[k] = max(da[k], db[k]);
        *reinterpret_cast<int4*>(A.out + p) = make_int4(dm[0], dm[1], dm[2], dm[3]);
#pragma unroll
        for (int k = 0; k < 4; k++) { sum[0] += da[k]; sum[1] += db[k]; sum[2] += dm[k]; }
        // run boundaries of the three tracks (k_issue_scan's rule, the window being the contig's)
        uint32_t g[3][4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const bool in = (p + k >= Wb) && (p + k < We);
            g[0][k] = in && da[k] >= A.lo && da[k] <= A.hi;
            g[1][k] = in && db[k] >= A.lo && db[k] <= A.hi;
            g[2][k] = in && dm[k] >= A.lo && dm[k] <= A.hi;
        }
        int gp[3];
#pragma unroll
        for (int x = 0; x < 3; x++) gp[x] = __shfl_up((int)g[x][3], 1, 64);
        if (lane == 0) {
            gp[0] = gp[1] = gp[2] = 0;
            if (p - 1 >= Wb && p - 1 < We) {                 // (p - 1 lies in the same contig: the window starts inside it)
                int32_t xa = A.a[p - 1], xb = A.b[p - 1];
                if (A.n_gaps) {                               // (the base before may lie in a run of the tile in front)
                    uint32_t lo_i = 0, hi_i = A.n_gaps;       // first run with end > p - 1
                    while (lo_i < hi_i) { const uint32_t mid = (lo_i + hi_i) >> 1; if (A.gaps[2 * mid + 1] > p - 1) hi_i = mid; else lo_i = mid + 1; }
                    if (lo_i < A.n_gaps && A.gaps[2 * lo_i] <= p - 1) xa = xb = 0;
                }
                const int32_t xm = max(xa, xb);
                gp[0] = xa >= A.lo && xa <= A.hi; gp[1] = xb >= A.lo && xb <= A.hi; gp[2] = xm >= A.lo && xm <= A.hi;
            }
        }
#pragma unroll
        for (int x = 0; x < 3; x++) {
            if (!(g[x][0] | g[x][1] | g[x][2] | g[x][3] | (uint32_t)gp[x])) continue;
            unsigned long long* keys = A.keys + (size_t)x * A.cap;
            bool prev = (bool)gp[x];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int64_t q = p + k;
                const bool gk = (bool)g[x][k];
                if (gk != prev && q >= Wb && (gk || q < We)) {
                    const uint32_t s = atomicAdd(A.n_keys + x, 1u);
                    if (s < A.cap) keys[s] = issue_key((uint32_t)c, q - Wb, !gk);
                }
                if (gk && q == We - 1) {
                    const uint32_t s = atomicAdd(A.n_keys + x, 1u);
                    if (s < A.cap) keys[s] = issue_key((uint32_t)c, We - Wb, true);
                }
                prev = gk;
            }
        }
    }
    if (A.tile_sums) {
        __shared__ long long part[3][BLOCK / 64];
        const int wave = t >> 6;
#pragma unroll
        for (int x = 0; x < 3; x++) { const long long v = wave_sum<long long>(sum[x]); if (lane == 0) part[x][wave] = v; }
        __syncthreads();
        if (t < 3) A.tile_sums[(int64_t)t * A.n_tiles + blockIdx.x] = part[t][0] + part[t][1] + part[t][2] + part[t][3];
    }
}

// h_gaps: the N runs in the reference's coordinates (contig = index in the layout, [start, end) with Python's slice rules, as
// gci_gap_mask takes them on the device); d_a / d_b are masked IN PLACE, d_out receives their maximum; d_keys: 3 x cap keys
// (track a, track b, the maximum; the keys gci_issue_scan(track, lo, hi, flank) gives), d_n_keys: 3 counters.
extern "C" int gci_two_type_tail(gci_ctx* ctx, int32_t* d_a, int32_t* d_b, int32_t* d_out, const gci_ivl* h_gaps, uint32_t n_gaps,
                                 double lo, double hi, int flank, uint64_t* d_keys, uint32_t cap, uint32_t* d_n_keys, int64_t* d_sums)
{
    if (!ctx || !d_a || !d_b || !d_out || !d_n_keys || (cap && !d_keys) || (n_gaps && !h_gaps)) return GCI_E_INVALID;
    if (!ctx->n_contigs) return GCI_E_NO_LAYOUT;
    ctx->build_runs_track = nullptr; ctx->build_runs_armed = false;
    HIPCHK(hipMemsetAsync(d_n_keys, 0, 12, ctx->stream));
    if (d_sums) HIPCHK(hipMemsetAsync(d_sums, 0, (size_t)3 * ctx->n_contigs * 8, ctx->stream));
    if (ctx->n_tiles == 0) return GCI_OK;
    if (d_sums) GCI_TRY(gci_ensure(ctx, ctx->tail_sums, (size_t)3 * ctx->n_tiles * 8));
    // absolute, sorted, merged element ranges of the N runs (uploaded again only when they change)
    std::vector<std::pair<int64_t, int64_t>> g;
    g.reserve(n_gaps);
    for (uint32_t i = 0; i < n_gaps; i++) {
        const gci_ivl v = h_gaps[i];
        if (v.contig < 0 || v.contig >= ctx->n_contigs) continue;
        const int64_t L = ctx->len[v.contig];
        const int64_t a = gci_slice_bound(v.start, L), b = gci_slice_bound(v.end, L);
        if (b > a) g.emplace_back(ctx->off[v.contig] + a, ctx->off[v.contig] + b);
    }
    std::sort(g.begin(), g.end());
    std::vector<int64_t> flat;
    for (const auto& r : g) {
        if (!flat.empty() && r.first <= flat.back()) { if (r.second > flat.back()) flat.back() = r.second; }
        else { flat.push_back(r.first); flat.push_back(r.second); }
    }
    if (flat != ctx->tail_gaps_host) {
        GCI_TRY(gci_ensure(ctx, ctx->tail_gaps, flat.size() * 8 + 16));
        if (!flat.empty()) GCI_TRY(gci_upload_small(ctx, ctx->tail_gaps.p, flat.data(), flat.size() * 8));
        ctx->tail_gaps_host = flat;
    }
    const IntRange rg = gci_int_range(lo, hi);
    TailArgs A;
    A.a = d_a; A.b = d_b; A.out = d_out;
    A.gaps = (const int64_t*)ctx->tail_gaps.p; A.n_gaps = (uint32_t)(flat.size() / 2);
    A.len = (const int64_t*)ctx->d_len.p; A.off = (const int64_t*)ctx->d_off.p; A.tile_first = (const int64_t*)ctx->d_tile_first.p;
    A.n_contigs = ctx->n_contigs; A.lo = rg.lo; A.hi = rg.hi; A.flank = flank;
    A.keys = (unsigned long long*)d_keys; A.cap = cap; A.n_keys = d_n_keys;
    A.tile_sums = d_sums ? (long long*)ctx->tail_sums.p : nullptr; A.n_tiles = ctx->n_tiles;
    ProfScope _ps(ctx, GCI_PROF_MAX2);
    hipLaunchKernelGGL(k_two_type_tail, dim3((uint32_t)ctx->n_tiles), dim3(BLOCK), 0, ctx->stream, A);
    LAUNCHCHK("k_two_type_tail");
    if (d_sums)
        for (int x = 0; x < 3; x++) {
            hipLaunchKernelGGL(k_reduce_tiles, dim3(ctx->n_contigs, REDUCE_SPLIT), dim3(BLOCK), 0, ctx->stream,
                               (const long long*)ctx->tail_sums.p + (size_t)x * ctx->n_tiles, (const int64_t*)ctx->d_tile_first.p,
                               (unsigned long long*)d_sums + (size_t)x * ctx->n_contigs);
            LAUNCHCHK("k_reduce_tiles");
        }
    return GCI_OK;
}

// ============================================================================================
// K10: depth -> decimal text (write_depth body, GCI.py:115-117) for an existing track
// ============================================================================================

// bytes of text per tile: sum over the contig's valid elements of (digits + 1)
__global__ __launch_bounds__(BLOCK) void k_text_count(const int32_t* __restrict__ depth,
                                                      const int64_t* __restrict__ tile_first,
                                                      const int64_t* __restrict__ len, int32_t n_contigs,
                                                      uint32_t* __restrict__ tile_bytes)
{
    __shared__ uint32_t part[BLOCK / 64];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int32_t c = contig_of_tile(tile_first, n_contigs, blockIdx.x);
    const int64_t valid = len[c] - ((int64_t)blockIdx.x - tile_first[c]) * TILE;
    const int4* base = reinterpret_cast<const int4*>(depth + (size_t)blockIdx.x * TILE);
    uint32_t s = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int64_t i = (int64_t)(j * BLOCK + t) * 4;
        const int4 v = base[j * BLOCK + t];
        if (i + 0 < valid) s += ndigits((uint32_t)v.x) + 1;
        if (i + 1 < valid) s += ndigits((uint32_t)v.y) + 1;
        if (i + 2 < valid) s += ndigits((uint32_t)v.z) + 1;
        if (i + 3 < valid) s += ndigits((uint32_t)v.w) + 1;
    }
    s = wave_sum<uint32_t>(s);
    if (lane == 0) part[wave] = s;
    __syncthreads();
    if (t == 0) tile_bytes[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

__global__ __launch_bounds__(BLOCK) void k_text_write(const int32_t* __restrict__ depth,
                                                      const int64_t* __restrict__ tile_first,
                                                      const int64_t* __restrict__ len, int32_t n_contigs,
                                                      const uint64_t* __restrict__ tile_off, uint8_t* __restrict__ out,
                                                      uint64_t cap, const uint32_t* __restrict__ g_lut)
{
    __shared__ __attribute__((aligned(16))) uint8_t stage[TEXT_STAGE];
    __shared__ uint32_t wtot[4][BLOCK / 64];
    __shared__ uint32_t lut[TEXT_LUT];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    text_lut_load(lut, g_lut, t);
    const int32_t c = contig_of_tile(tile_first, n_contigs, blockIdx.x);
    const int64_t valid = len[c] - ((int64_t)blockIdx.x - tile_first[c]) * TILE;
    const int4* base = reinterpret_cast<const int4*>(depth + (size_t)blockIdx.x * TILE);
    int4 v[4];
#pragma unroll
    for (int j = 0; j < 4; j++) v[j] = base[j * BLOCK + t];
    __syncthreads();
    if (valid >= TILE) text_tile<true>(v, valid, stage, wtot, lut, out, tile_off[blockIdx.x], cap, t, lane, wave);
    else text_tile<false>(v, valid, stage, wtot, lut, out, tile_off[blockIdx.x], cap, t, lane, wave);
}

extern "C" int gci_depth_text_size(gci_ctx* ctx, const int32_t* d_depth, uint64_t* d_contig_off)
{
    if (!ctx || !d_depth || !d_contig_off) return GCI_E_INVALID;
    if (!ctx->n_contigs) return GCI_E_NO_LAYOUT;
    const int64_t nt = ctx->n_tiles;
    if (nt == 0) { HIPCHK(hipMemsetAsync(d_contig_off, 0, (size_t)(ctx->n_contigs + 1) * 8, ctx->stream)); return GCI_OK; }
    ProfScope _ps(ctx, GCI_PROF_TEXT_COUNT);
    hipLaunchKernelGGL(k_text_count, dim3((uint32_t)nt), dim3(BLOCK), 0, ctx->stream, d_depth,
                       (const int64_t*)ctx->d_tile_first.p, (const int64_t*)ctx->d_len.p, ctx->n_contigs,
                       (uint32_t*)ctx->tile_u32.p);
    LAUNCHCHK("k_text_count");
    GCI_TRY((device_exclusive_scan<uint32_t, unsigned long long>(ctx, (const uint32_t*)ctx->tile_u32.p,
                                                                 (unsigned long long*)ctx->tile_u64.p,
                                                                 (unsigned long long*)ctx->blk_u64.p, nt, true)));
    hipLaunchKernelGGL(k_contig_text_off, dim3((ctx->n_contigs + 1 + 63) / 64), dim3(64), 0, ctx->stream,
                       (const uint64_t*)ctx->tile_u64.p, (const int64_t*)ctx->d_tile_first.p, ctx->n_contigs, nt,
                       d_contig_off);
    LAUNCHCHK("k_contig_text_off");
    return GCI_OK;
}

extern "C" int gci_depth_text_write(gci_ctx* ctx, const int32_t* d_depth, uint8_t* d_out, uint64_t cap)
{
    if (!ctx || !d_depth || !d_out) return GCI_E_INVALID;
    if (!ctx->n_contigs) return GCI_E_NO_LAYOUT;
    if (ctx->n_tiles == 0) return GCI_OK;
    ProfScope _ps(ctx, GCI_PROF_TEXT_WRITE);
    hipLaunchKernelGGL(k_text_write, dim3((uint32_t)ctx->n_tiles), dim3(BLOCK), 0, ctx->stream, d_depth,
                       (const int64_t*)ctx->d_tile_first.p, (const int64_t*)ctx->d_len.p, ctx->n_contigs,
                       (const uint64_t*)ctx->tile_u64.p, d_out, cap, (const uint32_t*)ctx->text_lut.p);
    LAUNCHCHK("k_text_write");
    return GCI_OK;
}

// ---- N3: window sums for the -p numeric front-end (sliding_window_average_depth, GCI.py:660-705) ----------------
// The reference walks a contig base by base, restarting its window at every zero-depth base.  Which bases emit a
// value follows from the zero runs alone (gci_issue_scan_windows with depth == 0) and the window size; what is left
// for the device is the sum of each window: one wave per [begin, end) range of the track, coalesced 4-byte loads.
__global__ __launch_bounds__(BLOCK) void k_range_sums(const int32_t* __restrict__ depth, const int64_t* __restrict__ ranges,
                                                      uint64_t n, long long* __restrict__ sums)
{
    const int lane = threadIdx.x & 63;
    const uint64_t r = (uint64_t)blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
    if (r >= n) return;
    const int64_t b = ranges[2 * r], e = ranges[2 * r + 1];
    long long s = 0;
    for (int64_t i = b + lane; i < e; i += 64) s += depth[i];
    s = wave_sum<long long>(s);
    if (lane == 0) sums[r] = s;
}

extern "C" int gci_range_sums(gci_ctx* ctx, const int32_t* d_depth, const int64_t* d_ranges, uint64_t n_ranges, int64_t* d_sums)
{
    if (!ctx || !d_depth || (n_ranges && (!d_ranges || !d_sums))) return GCI_E_INVALID;
    if (!ctx->n_contigs) return GCI_E_NO_LAYOUT;
    if (!n_ranges) return GCI_OK;
    const uint64_t blocks = (n_ranges + BLOCK / 64 - 1) / (BLOCK / 64);
    if (blocks > 0x7FFFFFFFull) return GCI_E_INVALID;
    hipLaunchKernelGGL(k_range_sums, dim3((uint32_t)blocks), dim3(BLOCK), 0, ctx->stream, d_depth, d_ranges, n_ranges,
                       (long long*)d_sums);
    LAUNCHCHK("k_range_sums");
    return GCI_OK;
}

// ---- N4 (first half): N runs of the assembly (get_Ns_ref, GCI.py:27-35) ---------------------------------------------
// The FASTA bytes go to the device as they are in the file.  Sequence coordinates do not count line ends, carriage
// returns and blanks (Bio.SeqIO's reader), and runs do not cross records.  One pass over the bytes: every workgroup
// counts the bytes of its 4096-byte tile that do count (the host's prefix sum over tiles turns a byte offset into a
// sequence coordinate) and emits the byte offsets where a run of N / n begins or ends; "the base before this one" is
// found by stepping back over dropped bytes (a line end or two), never beyond the start of the record's body.
__device__ __forceinline__ bool fasta_dropped(uint8_t c) { return c == '\n' || c == '\r' || c == ' '; }
__device__ __forceinline__ bool fasta_is_n(uint8_t c) { return c == 'N' || c == 'n'; }

__global__ __launch_bounds__(BLOCK) void k_fasta_n_scan(const uint8_t* __restrict__ text, uint64_t n_bytes,
                                                        const int64_t* __restrict__ body, uint32_t n_records,
                                                        uint32_t* __restrict__ tile_kept, unsigned long long* __restrict__ keys,
                                                        uint32_t cap, uint32_t* __restrict__ n_keys)
{
    __shared__ uint32_t part[BLOCK / 64];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const uint64_t tile0 = (uint64_t)blockIdx.x * TILE;
    // the last record whose body begins at or before this tile's end (bodies are sorted and disjoint)
    uint32_t lo = 0, hi = n_records;
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if ((uint64_t)body[2 * mid] < tile0 + TILE) lo = mid + 1; else hi = mid; }
    int64_t r = (int64_t)lo - 1;                                   // candidate record for the bytes of this tile
    uint32_t kept = 0;
    for (uint32_t k = 0; k < TILE / BLOCK; k++) {
        const uint64_t i = tile0 + (uint64_t)k * BLOCK + t;
        if (i >= n_bytes) break;
        // the record whose body holds byte i: step back from the candidate (a tile rarely meets more than one)
        int64_t rr = r;
        while (rr >= 0 && (uint64_t)body[2 * rr] > i) rr--;
        if (rr < 0 || i >= (uint64_t)body[2 * rr + 1]) continue;   // title line or in front of the first record
        const uint8_t c = text[i];
        if (fasta_dropped(c)) continue;
        kept++;
        const uint64_t b0 = (uint64_t)body[2 * rr];
        bool prev_n = false;
        for (uint64_t j = i; j > b0;) { const uint8_t p = text[--j]; if (!fasta_dropped(p)) { prev_n = fasta_is_n(p); break; } }
        const bool cur_n = fasta_is_n(c);
        if (cur_n != prev_n) {
            const uint32_t slot = atomicAdd(n_keys, 1u);
            if (slot < cap) keys[slot] = ((unsigned long long)i << 1) | (cur_n ? 0ull : 1ull);    // low bit: the run ended before i
        }
    }
    kept = wave_sum<uint32_t>(kept);
    if (lane == 0) part[wave] = kept;
    __syncthreads();
    if (t == 0) tile_kept[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

extern "C" int gci_fasta_n_scan(gci_ctx* ctx, const uint8_t* d_text, uint64_t n_bytes, const int64_t* d_body, uint32_t n_records,
                                uint32_t* d_tile_kept, uint64_t* d_keys, uint32_t cap, uint32_t* d_n_keys)
{
    if (!ctx || !d_n_keys || (n_bytes && (!d_text || !d_tile_kept)) || (n_records && !d_body) || (cap && !d_keys)) return GCI_E_INVALID;
    HIPCHK(hipMemsetAsync(d_n_keys, 0, 4, ctx->stream));
    if (!n_bytes) return GCI_OK;
    const uint64_t tiles = (n_bytes + TILE - 1) / TILE;
    if (tiles > 0x7FFFFFFFull) return GCI_E_INVALID;
    hipLaunchKernelGGL(k_fasta_n_scan, dim3((uint32_t)tiles), dim3(BLOCK), 0, ctx->stream, d_text, n_bytes, d_body, n_records,
                       d_tile_kept, (unsigned long long*)d_keys, cap, d_n_keys);
    LAUNCHCHK("k_fasta_n_scan");
    return GCI_OK;
}
