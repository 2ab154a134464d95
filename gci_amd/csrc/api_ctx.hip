// api_ctx.hip -- context, memory helpers, per-kernel event timing and the genome layout of
// libgci_hip.so (include/gci_hip.h).
#include "gci_ctx.hpp"
#include <stdlib.h>
#include <vector>

int gci_fail(gci_ctx* c, hipError_t e, const char* what)
{
    if (c) c->err = std::string(what) + ": " + hipGetErrorString(e);
    return GCI_E_HIP;
}

int gci_ensure(gci_ctx* ctx, DevBuf& b, size_t bytes)
{
    if (bytes <= b.cap) return GCI_OK;
    if (b.p) { HIPCHK(hipStreamSynchronize(ctx->stream)); HIPCHK(gci_dfree(b.p)); b.p = nullptr; b.cap = 0; }
    size_t want = bytes + bytes / 4 + 256;
    HIPCHK(gci_dmalloc(ctx->device, &b.p, want));
    b.cap = want;
    return GCI_OK;
}

int gci_upload_small(gci_ctx* ctx, void* d_dst, const void* h_src, size_t bytes)
{
    // staged through pinned memory so the async copy does not race with the caller's buffer
    HIPCHK(hipStreamSynchronize(ctx->stream));          // previous use of the staging buffer is done
    if (bytes > ctx->h_pinned_cap) {
        if (ctx->h_pinned) HIPCHK(hipHostFree(ctx->h_pinned));
        ctx->h_pinned = nullptr;
        ctx->h_pinned_cap = 0;
        size_t want = bytes * 2 + 4096;
        HIPCHK(hipHostMalloc(&ctx->h_pinned, want, hipHostMallocDefault));
        ctx->h_pinned_cap = want;
    }
    memcpy(ctx->h_pinned, h_src, bytes);
    HIPCHK(hipMemcpyAsync(d_dst, ctx->h_pinned, bytes, hipMemcpyHostToDevice, ctx->stream));
    return GCI_OK;
}

extern "C" int gci_abi_version(void) { return GCI_ABI_VERSION; }

extern "C" int gci_ctx_create(int device, void* stream, int own_stream, gci_ctx** out)
{
    if (!out) return GCI_E_INVALID;
    gci_ctx* ctx = new (std::nothrow) gci_ctx();
    if (!ctx) return GCI_E_NOMEM;
    ctx->device = device;
    { const char* fd = getenv("GCI_FORCE_DENSE"); if (fd && fd[0] == '1') ctx->sparse_max = -1; }   // testing / A-B timing
    { const char* m = getenv("GCI_JOIN"); ctx->join_mode = m && m[0] == 'c' ? 1 : m && m[0] == 'p' ? 2 : 0; }
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) { delete ctx; return GCI_E_HIP; }
    if (!own_stream) { ctx->stream = (hipStream_t)stream; }    // NULL = the device's default stream
    else {
        e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
        if (e != hipSuccess) { delete ctx; return GCI_E_HIP; }
        ctx->own_stream = true;
    }
    {
        uint32_t lut[TEXT_LUT];
        gci_text_lut_host(lut);
        if (gci_dmalloc(device, &ctx->text_lut.p, sizeof lut) != hipSuccess ||
            hipMemcpy(ctx->text_lut.p, lut, sizeof lut, hipMemcpyHostToDevice) != hipSuccess) {
            gci_ctx_destroy(ctx);
            return GCI_E_HIP;
        }
        ctx->text_lut.cap = sizeof lut;
    }
    *out = ctx;
    return GCI_OK;
}

extern "C" int gci_ctx_destroy(gci_ctx* ctx)
{
    if (!ctx) return GCI_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    DevBuf* bufs[] = {&ctx->d_len, &ctx->d_off, &ctx->d_tile_first, &ctx->tile_cd, &ctx->tile_carry, &ctx->dense_flag, &ctx->dense_list, &ctx->evp_items, &ctx->evp_hist, &ctx->evp_blk, &ctx->d_tile_valid,
                      &ctx->evt_off, &ctx->events, &ctx->blk_a, &ctx->blk_b, &ctx->tile_sum, &ctx->tile_u32,
                      &ctx->tile_u64, &ctx->blk_u64, &ctx->join_table, &ctx->join_last, &ctx->join_hq, &ctx->part_a, &ctx->part_b, &ctx->part_hist, &ctx->part_blk, &ctx->conflict_table, &ctx->win,
                      &ctx->win_tile_first, &ctx->text_lut, &ctx->long_items, &ctx->pg_cost, &ctx->pg_scan, &ctx->pg_first, &ctx->route_tab, &ctx->deflate_nruns, &ctx->deflate_runs, &ctx->deflate_tab, &ctx->build_nruns, &ctx->build_runs, &ctx->join_bucket, &ctx->tail_gaps, &ctx->tail_sums, &ctx->crc_tabs, &ctx->inflate_sym, &ctx->inflate_nsym, &ctx->inflate_wstatus, &ctx->inflate_lists, &ctx->inflate_prof, &ctx->inflate_next, &ctx->inflate_sym2, &ctx->inflate_lists2};
    for (DevBuf* b : bufs) if (b->p) (void)gci_dfree(b->p);
    for (DevBuf& b : ctx->paf_pool) if (b.p) (void)gci_dfree(b.p);
    if (ctx->h_pinned) (void)hipHostFree(ctx->h_pinned);
    for (auto* v : {&ctx->prof_live, &ctx->prof_free}) for (auto& e : *v) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    if (ctx->inflate_stream2) { (void)hipStreamSynchronize(ctx->inflate_stream2); (void)hipStreamDestroy(ctx->inflate_stream2); }
    if (ctx->inflate_ev_in) (void)hipEventDestroy(ctx->inflate_ev_in);
    if (ctx->inflate_ev_out) (void)hipEventDestroy(ctx->inflate_ev_out);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return GCI_OK;
}

extern "C" int gci_sync(gci_ctx* ctx)
{
    if (!ctx) return GCI_E_INVALID;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return GCI_OK;
}

extern "C" const char* gci_strerror(int s)
{
    switch (s) {
    case GCI_OK: return "ok";
    case GCI_E_INVALID: return "invalid argument";
    case GCI_E_HIP: return "HIP runtime error";
    case GCI_E_NO_NM: return "record without NM tag (reference raises KeyError, GCI.py:163)";
    case GCI_E_ZERO_DIV: return "zero denominator (reference raises ZeroDivisionError, GCI.py:165/292)";
    case GCI_E_BAD_NM_TYPE: return "NM tag is not an integer";
    case GCI_E_NO_END: return "record has no CIGAR: reference_end is None";
    case GCI_E_MALFORMED: return "malformed BAM record";
    case GCI_E_CAPACITY: return "output capacity exceeded";
    case GCI_E_NOMEM: return "out of memory";
    case GCI_E_NO_LAYOUT: return "gci_layout_set() not called";
    default: return "unknown status";
    }
}

extern "C" const char* gci_last_error(gci_ctx* ctx) { return ctx ? ctx->err.c_str() : ""; }

static const char* const PROF_NAMES[GCI_PROF_COUNT] = {
    "k_bam_filter", "k_join_insert", "k_join_fold", "k_evt_count+scatter", "k_scan2", "k_tile_build", "k_gap_mask",
    "k_max2", "k_issue_scan", "k_text_count", "k_text_write", "k_depth_sum", "memset", "k_tile_pass1", "k_tile_dense",
    "k_part1+k_part2 (radix partition of the join)", "k_join_part", "k_pg_measure+scans+k_pg_first (record pages: sizes)", "k_pg_write (record pages)"};

extern "C" int gci_profile_enable(gci_ctx* ctx, int mask)
{
    if (!ctx) return GCI_E_INVALID;
    ctx->prof_mask = mask;
    return GCI_OK;
}

extern "C" int gci_profile_read(gci_ctx* ctx, int kernel_id, double* total_ms, uint64_t* launches, int reset)
{
    if (!ctx || kernel_id < 0 || kernel_id >= GCI_PROF_COUNT) return GCI_E_INVALID;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    for (auto& e : ctx->prof_live) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) { ctx->prof_ms[e.id] += ms; ctx->prof_n[e.id]++; }
        ctx->prof_free.push_back(e);
    }
    ctx->prof_live.clear();
    if (total_ms) *total_ms = ctx->prof_ms[kernel_id];
    if (launches) *launches = ctx->prof_n[kernel_id];
    if (reset) { ctx->prof_ms[kernel_id] = 0; ctx->prof_n[kernel_id] = 0; }
    return GCI_OK;
}

extern "C" const char* gci_profile_name(int kernel_id)
{
    return kernel_id >= 0 && kernel_id < GCI_PROF_COUNT ? PROF_NAMES[kernel_id] : "";
}

extern "C" int gci_malloc(gci_ctx* ctx, size_t bytes, void** d_out)
{
    if (!ctx || !d_out) return GCI_E_INVALID;
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(gci_dmalloc(ctx->device, d_out, bytes ? bytes : 16));
    return GCI_OK;
}
extern "C" int gci_free(gci_ctx* ctx, void* p)
{
    if (!ctx) return GCI_E_INVALID;
    if (p) { HIPCHK(hipStreamSynchronize(ctx->stream)); HIPCHK(gci_dfree(p)); }
    return GCI_OK;
}
extern "C" int gci_memcpy_h2d(gci_ctx* ctx, void* d, const void* h, size_t n)
{
    if (!ctx) return GCI_E_INVALID;
    if (n) { HIPCHK(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, ctx->stream)); HIPCHK(hipStreamSynchronize(ctx->stream)); }
    return GCI_OK;
}
extern "C" int gci_memcpy_d2h(gci_ctx* ctx, void* h, const void* d, size_t n)
{
    if (!ctx) return GCI_E_INVALID;
    if (n) { HIPCHK(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, ctx->stream)); HIPCHK(hipStreamSynchronize(ctx->stream)); }
    return GCI_OK;
}
extern "C" int gci_memset(gci_ctx* ctx, void* d, int byte, size_t n)
{
    if (!ctx) return GCI_E_INVALID;
    if (n) HIPCHK(hipMemsetAsync(d, byte, n, ctx->stream));
    return GCI_OK;
}

// ============================================================================================
// layout
// ============================================================================================

extern "C" int gci_layout_set(gci_ctx* ctx, int32_t n, const int64_t* h_len)
{
    if (!ctx || n <= 0 || !h_len) return GCI_E_INVALID;
    ctx->n_contigs = n;
    ctx->len.assign(h_len, h_len + n);
    ctx->off.resize(n);
    ctx->tile_first.resize(n + 1);
    int64_t tiles = 0;
    for (int32_t c = 0; c < n; c++) {
        if (h_len[c] < 0 || h_len[c] > 0x7fffffffLL) return GCI_E_INVALID;
        ctx->tile_first[c] = tiles;
        ctx->off[c] = tiles * TILE;
        tiles += (h_len[c] + TILE - 1) / TILE;
    }
    ctx->tile_first[n] = tiles;
    ctx->n_tiles = tiles;
    ctx->total = tiles * TILE;
    ctx->win_flank = INT32_MIN;
    int r;
    if ((r = gci_ensure(ctx, ctx->d_len, n * 8))) return r;
    if ((r = gci_ensure(ctx, ctx->d_off, n * 8))) return r;
    if ((r = gci_ensure(ctx, ctx->d_tile_first, (n + 1) * 8))) return r;
    if ((r = gci_upload_small(ctx, ctx->d_len.p, ctx->len.data(), n * 8))) return r;
    if ((r = gci_upload_small(ctx, ctx->d_off.p, ctx->off.data(), n * 8))) return r;
    if ((r = gci_upload_small(ctx, ctx->d_tile_first.p, ctx->tile_first.data(), (n + 1) * 8))) return r;
    const size_t nb = (size_t)(tiles / TILE + 2);
    GCI_TRY(gci_ensure(ctx, ctx->tile_cd, (size_t)(tiles + 1) * 8));
    GCI_TRY(gci_ensure(ctx, ctx->tile_carry, (size_t)(tiles + 1) * 4));
    GCI_TRY(gci_ensure(ctx, ctx->evt_off, (size_t)(tiles + 2) * 4));
    GCI_TRY(gci_ensure(ctx, ctx->dense_flag, (size_t)tiles + 16));
    GCI_TRY(gci_ensure(ctx, ctx->dense_list, ((size_t)tiles + 1) * sizeof(uint32_t)));
    GCI_TRY(gci_ensure(ctx, ctx->d_tile_valid, (size_t)(tiles + 1) * 4));
    {
        std::vector<int32_t> tv((size_t)tiles + 1, TILE);
        for (int32_t c = 0; c < n; c++)
            if (h_len[c] > 0) tv[(size_t)ctx->tile_first[c + 1] - 1] = (int32_t)(h_len[c] - ((h_len[c] - 1) / TILE) * TILE);
        GCI_TRY(gci_upload_small(ctx, ctx->d_tile_valid.p, tv.data(), tv.size() * 4));
    }
    GCI_TRY(gci_ensure(ctx, ctx->blk_a, nb * 4));
    GCI_TRY(gci_ensure(ctx, ctx->blk_b, nb * 4));
    GCI_TRY(gci_ensure(ctx, ctx->tile_sum, (size_t)(tiles + 1) * 8));
    GCI_TRY(gci_ensure(ctx, ctx->tile_u32, (size_t)(tiles + 1) * 4));
    GCI_TRY(gci_ensure(ctx, ctx->tile_u64, (size_t)(tiles + 2) * 8));
    GCI_TRY(gci_ensure(ctx, ctx->blk_u64, nb * 8));
    // the per-tile (count, difference) table is self-cleaning (k_tile_build zeroes the differences, k_evt_scatter
    // returns the counts to zero): zero it once here
    HIPCHK(hipMemsetAsync(ctx->tile_cd.p, 0, (size_t)(tiles + 1) * 8, ctx->stream));
    ctx->cd_state = 0;
    ctx->build_pending = false;
    ctx->build_runs_track = nullptr; ctx->build_runs_armed = false;
    return GCI_OK;
}

extern "C" int64_t gci_layout_total(gci_ctx* ctx) { return ctx ? ctx->total : 0; }

extern "C" int gci_layout_offsets(gci_ctx* ctx, int64_t* h)
{
    if (!ctx || !h) return GCI_E_INVALID;
    if (!ctx->n_contigs) return GCI_E_NO_LAYOUT;
    memcpy(h, ctx->off.data(), ctx->n_contigs * 8);
    return GCI_OK;
}

