// gci_hip.hip -- libgci_hip.so: gfx950 (MI355X, CDNA4) kernels + the C ABI of include/gci_hip.h.
//
// Everything on this path is integer, streaming or scatter work: no MFMA anywhere.  The
// bound is HBM bandwidth (depth scan, issue scan, text), atomic rate (depth diff, join) or
// latency (record decode).  Wave = 64 lanes, 256-thread workgroups, 16-byte vector accesses,
// 4096-element (16 KiB) tiles that never straddle contigs.
//
// Kernel  replaces (reference /root/reference/GCI.py)          algorithmic bytes
//   K1 k_bam_filter     read_sam 146-169                        36+name+4*n_cigar+NM in, 32 out / record
//   K3 k_join_*         cross-file join 272-301                 48 / probe, 16 / interval
//   K4 k_depth_diff     slice += 1, 302-306                     12 read + 4 atomics / interval
//   K5 k_depth_scan     (implicit in the slice add)             4 read + 4 write / base
//   K6 k_gap_mask       merge_gaps_depths 324-328               4 write / masked base
//   K7 k_max2           merge_two_type_depth 350                8 read + 4 write / base
//   K8 k_issue_scan     collapse_depth_range 369-390            4 read / base
//   K10 k_text_*        write_depth body 115-117                4 read (x2) + text write / base
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include <vector>

#include "../../include/gci_hip.h"
#include "gci_common.h"

#define TILE GCI_TILE
#define BLOCK 256
static_assert(TILE == BLOCK * 16, "a tile is 16 elements per thread");

// ============================================================================================
// context
// ============================================================================================

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
    // scratch
    DevBuf tile_i32, blk_i32;               // coarse diff / carries + block totals
    DevBuf tile_u32, tile_u64, blk_u64;     // text byte counts, offsets
    DevBuf join_table, join_last, join_hq;
    DevBuf win, win_tile_first;             // issue-scan windows
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

// Scoped HIP-event pair around one launch, recorded on the ctx stream when that kernel id is
// enabled.  Events are pooled; elapsed times are folded in by gci_profile_read().
struct ProfScope {
    gci_ctx* c; int id; gci_ctx::ProfEv ev; bool on;
    ProfScope(gci_ctx* ctx, int kid) : c(ctx), id(kid), on(false) {
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

static int fail(gci_ctx* c, hipError_t e, const char* what)
{
    if (c) c->err = std::string(what) + ": " + hipGetErrorString(e);
    return GCI_E_HIP;
}
#define HIPCHK(call)                                  \
    do {                                              \
        hipError_t _e = (call);                       \
        if (_e != hipSuccess) return fail(ctx, _e, #call); \
    } while (0)
#define LAUNCHCHK(name)                               \
    do {                                              \
        hipError_t _e = hipGetLastError();            \
        if (_e != hipSuccess) return fail(ctx, _e, name); \
    } while (0)

static int ensure(gci_ctx* ctx, DevBuf& b, size_t bytes)
{
    if (bytes <= b.cap) return GCI_OK;
    if (b.p) { HIPCHK(hipStreamSynchronize(ctx->stream)); HIPCHK(hipFree(b.p)); b.p = nullptr; b.cap = 0; }
    size_t want = bytes + bytes / 4 + 256;
    HIPCHK(hipMalloc(&b.p, want));
    b.cap = want;
    return GCI_OK;
}

static int upload_small(gci_ctx* ctx, void* d_dst, const void* h_src, size_t bytes)
{
    // staged through pinned memory so the async copy does not race with the caller's buffer
    if (bytes > ctx->h_pinned_cap) {
        HIPCHK(hipStreamSynchronize(ctx->stream));
        if (ctx->h_pinned) HIPCHK(hipHostFree(ctx->h_pinned));
        ctx->h_pinned = nullptr;
        ctx->h_pinned_cap = 0;
        size_t want = bytes * 2 + 4096;
        HIPCHK(hipHostMalloc(&ctx->h_pinned, want, hipHostMallocDefault));
        ctx->h_pinned_cap = want;
    } else {
        HIPCHK(hipStreamSynchronize(ctx->stream));   // previous use of the staging buffer is done
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
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) { delete ctx; return GCI_E_HIP; }
    if (!own_stream) { ctx->stream = (hipStream_t)stream; }    // NULL = the device's default stream
    else {
        e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
        if (e != hipSuccess) { delete ctx; return GCI_E_HIP; }
        ctx->own_stream = true;
    }
    *out = ctx;
    return GCI_OK;
}

extern "C" int gci_ctx_destroy(gci_ctx* ctx)
{
    if (!ctx) return GCI_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    DevBuf* bufs[] = {&ctx->d_len, &ctx->d_off, &ctx->d_tile_first, &ctx->tile_i32, &ctx->blk_i32, &ctx->tile_u32,
                      &ctx->tile_u64, &ctx->blk_u64, &ctx->join_table, &ctx->join_last, &ctx->join_hq, &ctx->win,
                      &ctx->win_tile_first};
    for (DevBuf* b : bufs) if (b->p) (void)hipFree(b->p);
    if (ctx->h_pinned) (void)hipHostFree(ctx->h_pinned);
    for (auto* v : {&ctx->prof_live, &ctx->prof_free}) for (auto& e : *v) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
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
    "k_bam_filter", "k_join_insert", "k_join_fold", "k_depth_diff", "k_scan_tiles", "k_depth_scan", "k_gap_mask",
    "k_max2", "k_issue_scan", "k_text_count", "k_text_write", "k_depth_sum", "memset"};

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
    HIPCHK(hipMalloc(d_out, bytes ? bytes : 16));
    return GCI_OK;
}
extern "C" int gci_free(gci_ctx* ctx, void* p)
{
    if (!ctx) return GCI_E_INVALID;
    if (p) { HIPCHK(hipStreamSynchronize(ctx->stream)); HIPCHK(hipFree(p)); }
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
    if ((r = ensure(ctx, ctx->d_len, n * 8))) return r;
    if ((r = ensure(ctx, ctx->d_off, n * 8))) return r;
    if ((r = ensure(ctx, ctx->d_tile_first, (n + 1) * 8))) return r;
    if ((r = upload_small(ctx, ctx->d_len.p, ctx->len.data(), n * 8))) return r;
    if ((r = upload_small(ctx, ctx->d_off.p, ctx->off.data(), n * 8))) return r;
    if ((r = upload_small(ctx, ctx->d_tile_first.p, ctx->tile_first.data(), (n + 1) * 8))) return r;
    if ((r = ensure(ctx, ctx->tile_i32, (size_t)(tiles + 1) * 4))) return r;
    if ((r = ensure(ctx, ctx->blk_i32, (size_t)(tiles / TILE + 2) * 4))) return r;
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

// ============================================================================================
// small device utilities
// ============================================================================================

__device__ __forceinline__ uint32_t ld_u32(const uint8_t* p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }
__device__ __forceinline__ int32_t ld_i32(const uint8_t* p) { int32_t v; __builtin_memcpy(&v, p, 4); return v; }
__device__ __forceinline__ uint16_t ld_u16(const uint8_t* p) { uint16_t v; __builtin_memcpy(&v, p, 2); return v; }

template <int G>
__device__ __forceinline__ int64_t group_sum_i64(int64_t v)
{
#pragma unroll
    for (int m = G / 2; m >= 1; m >>= 1) v += __shfl_xor(v, m, G);
    return v;
}
template <int G>
__device__ __forceinline__ uint32_t group_min_u32(uint32_t v)
{
#pragma unroll
    for (int m = G / 2; m >= 1; m >>= 1) { uint32_t o = __shfl_xor(v, m, G); v = o < v ? o : v; }
    return v;
}

// inclusive scan across the 64 lanes of a wave
template <typename T>
__device__ __forceinline__ T wave_inclusive(T v, int lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { T n = __shfl_up(v, d, 64); if (lane >= d) v += n; }
    return v;
}

// ============================================================================================
// K1: BAM record filter (read_sam, GCI.py:146-169)
// ============================================================================================
//
// G lanes cooperate on one record (G = 16: four records per wave): the CIGAR words and the
// name bytes are read group-strided (contiguous 4*G bytes per step), op totals are reduced with
// xor-shuffles, the aux walk (NM, CG) is done redundantly by every lane of the group (same
// addresses: one transaction), the two IEEE f64 divisions decide, lane 0 writes the 32-byte
// compact record.  SEQ and QUAL are skipped by pointer arithmetic and never touched.

__device__ __forceinline__ void report(unsigned long long* status, uint32_t rec, int code)
{
    atomicMin(status, ((unsigned long long)rec << 8) | (unsigned long long)(uint8_t)(-code));
}

// size of an aux value of type t at p; -1 if malformed / past end
__device__ __forceinline__ int64_t aux_value_size(const uint8_t* p, const uint8_t* end, uint8_t t)
{
    switch (t) {
    case 'A': case 'c': case 'C': return 1;
    case 's': case 'S': return 2;
    case 'i': case 'I': case 'f': return 4;
    case 'Z': case 'H': {
        const uint8_t* q = p;
        while (q < end && *q) q++;
        return q < end ? (q - p) + 1 : -1;
    }
    case 'B': {
        if (p + 5 > end) return -1;
        uint8_t sub = p[0];
        int64_t n = ld_u32(p + 1);
        int64_t es = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2
                   : (sub == 'i' || sub == 'I' || sub == 'f') ? 4 : -1;
        return es < 0 ? -1 : 5 + n * es;
    }
    default: return -1;
    }
}

template <int G>
__global__ __launch_bounds__(BLOCK) void k_bam_filter(
    const uint8_t* __restrict__ bam, uint64_t n_bytes, const uint64_t* __restrict__ rec_off, uint32_t n_rec,
    const int32_t* __restrict__ ref_sel, int32_t n_ref, int map_qual, int mq_cutoff, double clip_percent,
    double iden_percent, uint32_t rec_idx_base, gci_rec* __restrict__ out, unsigned long long* __restrict__ status)
{
    const int gl = threadIdx.x % G;
    const uint32_t rec = (uint32_t)(((uint64_t)blockIdx.x * BLOCK + threadIdx.x) / G);
    if (rec >= n_rec) return;

    gci_rec r;
    r.name_hash = 0; r.contig = -1; r.start = 0; r.end = 0; r.qlen = 0; r.rec_idx = rec + rec_idx_base; r.mapq = 0; r.flags = 0;
    r.name_len = 0;
    const uint64_t off = rec_off[rec];
    bool ok = off + 36 <= n_bytes;
    int32_t block_size = 0;
    if (ok) { block_size = ld_i32(bam + off); ok = block_size >= 32 && off + 4 + (uint64_t)block_size <= n_bytes; }
    if (!ok) {
        if (gl == 0) { report(status, rec, GCI_E_MALFORMED); out[rec] = r; }
        return;
    }
    const uint8_t* p = bam + off;
    const int32_t ref_id = ld_i32(p + 4);
    const int32_t pos = ld_i32(p + 8);
    const uint32_t l_read_name = p[12];
    const int mapq = p[13];
    const uint32_t n_cigar = ld_u16(p + 16);
    const uint32_t flag = ld_u16(p + 18);
    const int32_t l_seq = ld_i32(p + 20);
    const uint8_t* name = p + 36;
    const uint8_t* rec_end = p + 4 + block_size;
    const uint8_t* cig = name + l_read_name;
    const uint8_t* aux = cig + 4 * (uint64_t)n_cigar + (((uint64_t)(uint32_t)l_seq + 1) >> 1) + (uint64_t)(uint32_t)l_seq;
    if (l_seq < 0 || aux > rec_end) {
        if (gl == 0) { report(status, rec, GCI_E_MALFORMED); out[rec] = r; }
        return;
    }
    r.mapq = (uint8_t)mapq;

    // fetch(contig=target) only ever yields records of selected contigs (GCI.py:151, 260);
    // then GCI.py:152-156: mapped, not secondary, not supplementary, MAPQ >= -mq.
    const bool sel = ref_id >= 0 && ref_id < n_ref && ref_sel[ref_id] >= 0;
    if (!sel || (flag & (0x4u | 0x100u | 0x800u)) || mapq < map_qual) {
        if (gl == 0) out[rec] = r;
        return;
    }

    // ---- query_name: bytes up to the first NUL; hash of its 8-byte words ----------------------
    uint32_t nul = l_read_name;
    for (uint32_t i = gl; i < l_read_name; i += G) if (name[i] == 0) { nul = i; break; }
    const uint32_t name_len = group_min_u32<G>(nul);
    uint64_t acc = 0;
    for (uint32_t k = gl; k * 8 < name_len; k += G) {
        uint64_t w = 0;
        const uint32_t b0 = k * 8;
#pragma unroll
        for (int b = 0; b < 8; b++) if (b0 + b < name_len) w |= (uint64_t)name[b0 + b] << (8 * b);
        acc += gci_hash_word(w, k);
    }
    acc = (uint64_t)group_sum_i64<G>((int64_t)acc);
    r.name_hash = gci_hash_finish(acc, name_len);
    r.name_len = (uint16_t)name_len;

    // ---- aux walk: first NM, first CG (bam_aux_get semantics) ----------------------------------
    const uint8_t* nm_p = nullptr;
    const uint8_t* cg_p = nullptr;
    {
        const uint8_t* q = aux;
        while (q + 3 <= rec_end) {
            const uint8_t t0 = q[0], t1 = q[1], ty = q[2];
            const int64_t sz = aux_value_size(q + 3, rec_end, ty);
            if (sz < 0 || q + 3 + sz > rec_end) break;
            if (t0 == 'N' && t1 == 'M' && !nm_p) nm_p = q + 2;
            if (t0 == 'C' && t1 == 'G' && !cg_p) cg_p = q + 2;
            q += 3 + sz;
        }
    }

    // ---- htslib moves a >65535-op CIGAR back from CG:B,I when op0 == <l_seq>S --------------------
    const uint8_t* ops = cig;
    uint64_t n_ops = n_cigar;
    if (n_cigar > 0 && pos >= 0) {
        const uint32_t op0 = ld_u32(cig);
        if ((op0 & 0xF) == 4 && (op0 >> 4) == (uint32_t)l_seq && cg_p && cg_p[0] == 'B' &&
            (cg_p[1] == 'I' || cg_p[1] == 'i')) {
            const uint32_t cg_len = ld_u32(cg_p + 2);
            if (cg_len >= n_cigar && cg_len < (1u << 29)) { ops = cg_p + 6; n_ops = cg_len; }
        }
    }

    // ---- get_cigar_stats()[0] (GCI.py:157-162): base totals per op ------------------------------
    int64_t sM = 0, sI = 0, sD = 0, sN = 0, sS = 0, sE = 0, sX = 0;
    for (uint64_t k = gl; k < n_ops; k += G) {
        const uint32_t v = ld_u32(ops + 4 * k);
        const int64_t len = v >> 4;
        const uint32_t op = v & 0xF;
        sM += op == 0 ? len : 0;
        sI += op == 1 ? len : 0;
        sD += op == 2 ? len : 0;
        sN += op == 3 ? len : 0;
        sS += op == 4 ? len : 0;
        sE += op == 7 ? len : 0;
        sX += op == 8 ? len : 0;
    }
    sM = group_sum_i64<G>(sM); sI = group_sum_i64<G>(sI); sD = group_sum_i64<G>(sD); sN = group_sum_i64<G>(sN);
    sS = group_sum_i64<G>(sS); sE = group_sum_i64<G>(sE); sX = group_sum_i64<G>(sX);

    if (gl != 0) return;        // the rest is scalar per record

    // ---- get_tag('NM') (GCI.py:163) ---------------------------------------------------------------
    if (!nm_p) { report(status, rec, GCI_E_NO_NM); out[rec] = r; return; }
    int64_t NM;
    switch (nm_p[0]) {
    case 'c': NM = (int8_t)nm_p[1]; break;
    case 'C': NM = nm_p[1]; break;
    case 's': NM = (int16_t)ld_u16(nm_p + 1); break;
    case 'S': NM = ld_u16(nm_p + 1); break;
    case 'i': NM = ld_i32(nm_p + 1); break;
    case 'I': NM = ld_u32(nm_p + 1); break;
    default: report(status, rec, GCI_E_BAD_NM_TYPE); out[rec] = r; return;
    }
    const int64_t mm = NM - (sI + sD);                                             // GCI.py:164
    const int64_t den1 = sM + sE + sX + sI + sS, den2 = sM + sE + sX + sI + sD;
    if (den1 == 0) { report(status, rec, GCI_E_ZERO_DIV); out[rec] = r; return; }
    // Python's `and` short-circuits: the identity division only runs when the clip test passed
    if (!((double)sS / (double)den1 <= clip_percent)) { out[rec] = r; return; }   // GCI.py:165
    if (den2 == 0) { report(status, rec, GCI_E_ZERO_DIV); out[rec] = r; return; }
    if (!((double)(sM + sE + sX - mm) / (double)den2 >= iden_percent)) { out[rec] = r; return; }
    if (n_cigar == 0) { report(status, rec, GCI_E_NO_END); out[rec] = r; return; }
    const int64_t rlen = sM + sD + sN + sE + sX;
    r.contig = ref_sel[ref_id];
    r.start = pos;
    r.end = (int32_t)((int64_t)pos + (rlen > 0 ? rlen : 1));                        // bam_endpos
    r.qlen = l_seq;                                                                // query_length
    r.flags = GCI_REC_PASS | (mapq >= mq_cutoff ? GCI_REC_HQ : 0);                 // GCI.py:166-168
    out[rec] = r;
}

extern "C" int gci_bam_filter(gci_ctx* ctx, const uint8_t* d_bam, uint64_t n_bytes, const uint64_t* d_rec_off,
                              uint32_t n_rec, const int32_t* d_ref_sel, int32_t n_ref, int map_qual, int mq_cutoff,
                              double clip_percent, double iden_percent, uint32_t rec_idx_base, gci_rec* d_out,
                              uint64_t* d_status)
{
    if (!ctx || !d_out || !d_status || (n_rec && (!d_bam || !d_rec_off || !d_ref_sel))) return GCI_E_INVALID;
    HIPCHK(hipMemsetAsync(d_status, 0xFF, 8, ctx->stream));
    if (n_rec == 0) return GCI_OK;
    constexpr int G = 16;
    const uint64_t threads = (uint64_t)n_rec * G;
    const uint32_t grid = (uint32_t)((threads + BLOCK - 1) / BLOCK);
    { ProfScope _ps(ctx, GCI_PROF_BAM_FILTER);
    hipLaunchKernelGGL(k_bam_filter<G>, dim3(grid), dim3(BLOCK), 0, ctx->stream, d_bam, n_bytes, d_rec_off, n_rec,
                       d_ref_sel, n_ref, map_qual, mq_cutoff, clip_percent, iden_percent, rec_idx_base, d_out,
                       (unsigned long long*)d_status);
    }
    LAUNCHCHK("k_bam_filter");
    return GCI_OK;
}

extern "C" int gci_decode_status(uint64_t w, uint32_t* rec_idx)
{
    if (w == ~0ull) return GCI_OK;
    if (rec_idx) *rec_idx = (uint32_t)(w >> 8);
    return -(int)(w & 0xFF);
}

extern "C" uint64_t gci_name_hash(const uint8_t* name, uint32_t len)
{
    uint64_t acc = 0;
    for (uint32_t k = 0; k * 8 < len; k++) {
        uint64_t w = 0;
        for (int b = 0; b < 8; b++) if (k * 8 + b < len) w |= (uint64_t)name[k * 8 + b] << (8 * b);
        acc += gci_hash_word(w, k);
    }
    return gci_hash_finish(acc, len);
}

// ============================================================================================
// generic exclusive scan over per-tile tables (n up to millions): local scan + add
// ============================================================================================

template <typename TIn, typename TOut>
__global__ __launch_bounds__(BLOCK) void k_scan_local(const TIn* __restrict__ in, TOut* __restrict__ out,
                                                      TOut* __restrict__ blk_tot, int64_t n)
{
    __shared__ TOut wtot[BLOCK / 64];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int64_t base = (int64_t)blockIdx.x * TILE + (int64_t)t * 16;
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
    if (t == 0) blk_tot[blockIdx.x] = all;
}

// out[i] += sum of blk_tot[0 .. block(i) - 1]; optionally writes the grand total to out[n]
template <typename TOut>
__global__ __launch_bounds__(BLOCK) void k_scan_add(TOut* __restrict__ out, const TOut* __restrict__ blk_tot,
                                                    int64_t n, int32_t n_blocks, bool write_total)
{
    __shared__ TOut part[BLOCK / 64];
    __shared__ TOut s_pre;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int32_t me = blockIdx.x;
    const int32_t upto = write_total && me == n_blocks ? n_blocks : me;   // extra block computes the total
    TOut s = 0;
    for (int32_t b = t; b < upto; b += BLOCK) s += blk_tot[b];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
    if (lane == 0) part[wave] = s;
    __syncthreads();
    if (t == 0) { TOut a = 0; for (int w = 0; w < BLOCK / 64; w++) a += part[w]; s_pre = a; }
    __syncthreads();
    const TOut pre = s_pre;
    if (me == n_blocks) { if (t == 0) out[n] = pre; return; }
    if (pre == 0) return;
    const int64_t base = (int64_t)me * TILE + (int64_t)t * 16;
#pragma unroll
    for (int i = 0; i < 16; i++) if (base + i < n) out[base + i] += pre;
}

template <typename TIn, typename TOut>
static int device_exclusive_scan(gci_ctx* ctx, const TIn* in, TOut* out, TOut* blk_tot, int64_t n, bool write_total)
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
                           blk_tot, n, nb, write_total);
        LAUNCHCHK("k_scan_add");
    }
    return GCI_OK;
}

// ============================================================================================
// K4 + K5: depth build (GCI.py:302-306) = difference array + per-tile carries + tile scan
// ============================================================================================
//
// K4 writes +1 / -1 into the (zeroed) track itself and, per interval, the same +1 / -1 into a
// coarse per-tile table.  An exclusive scan of the coarse table (one int per 4096 bases) gives
// every tile its carry-in, so the big pass K5 is ONE read + ONE write of the track with no
// inter-workgroup dependency (no look-back chain across XCDs whose L2s are not coherent).
// The -1 of an interval reaching the contig end is kept in the coarse table (last tile) so each
// contig sums to zero and one unsegmented scan over all tiles serves every contig.

__global__ __launch_bounds__(BLOCK) void k_depth_diff(const gci_ivl* __restrict__ ivl, const uint32_t* __restrict__ d_n,
                                                      uint32_t max_n, int flank, const int64_t* __restrict__ len,
                                                      const int64_t* __restrict__ off,
                                                      const int64_t* __restrict__ tile_first, int32_t n_contigs,
                                                      int32_t* __restrict__ depth, int32_t* __restrict__ tile_diff)
{
    const uint32_t n = d_n ? min(*d_n, max_n) : max_n;
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const gci_ivl v = ivl[i];
    if (v.contig < 0 || v.contig >= n_contigs) return;
    const int64_t L = len[v.contig];
    const int64_t a = gci_slice_bound((int64_t)v.start + flank, L);
    const int64_t b = gci_slice_bound((int64_t)v.end - flank + 1, L);
    if (a >= b) return;
    int32_t* d = depth + off[v.contig];
    int32_t* td = tile_diff + tile_first[v.contig];
    atomicAdd(d + a, 1);
    atomicAdd(td + a / TILE, 1);
    // b == L lands in the contig's tail padding when there is any, which keeps the padding at zero
    if (b < (L + TILE - 1) / TILE * TILE) atomicAdd(d + b, -1);
    atomicAdd(td + (b < L ? b : L - 1) / TILE, -1);
}

__global__ __launch_bounds__(BLOCK) void k_depth_scan(int32_t* __restrict__ depth, const int32_t* __restrict__ tile_carry)
{
    __shared__ int32_t wtot[4][BLOCK / 64];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    int4* base = reinterpret_cast<int4*>(depth + (size_t)blockIdx.x * TILE);
    int4 v[4];
#pragma unroll
    for (int j = 0; j < 4; j++) v[j] = base[j * BLOCK + t];
    int32_t tot[4], inc[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        v[j].y += v[j].x; v[j].z += v[j].y; v[j].w += v[j].z;
        tot[j] = v[j].w;
        inc[j] = tot[j];
    }
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
#pragma unroll
        for (int j = 0; j < 4; j++) { int32_t n = __shfl_up(inc[j], d, 64); if (lane >= d) inc[j] += n; }
    }
    if (lane == 63) {
#pragma unroll
        for (int j = 0; j < 4; j++) wtot[j][wave] = inc[j];
    }
    __syncthreads();
    int32_t carry = tile_carry[blockIdx.x];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        int32_t pre = 0, all = 0;
#pragma unroll
        for (int w = 0; w < BLOCK / 64; w++) { const int32_t x = wtot[j][w]; if (w < wave) pre += x; all += x; }
        const int32_t ex = carry + pre + inc[j] - tot[j];
        v[j].x += ex; v[j].y += ex; v[j].z += ex; v[j].w += ex;
        base[j * BLOCK + t] = v[j];
        carry += all;
    }
}

extern "C" int gci_depth_build(gci_ctx* ctx, const gci_ivl* d_ivl, const uint32_t* d_n, uint32_t max_n, int flank,
                               int32_t* d_depth)
{
    if (!ctx || !d_depth || (max_n && !d_ivl)) return GCI_E_INVALID;
    if (!ctx->n_contigs) return GCI_E_NO_LAYOUT;
    if (ctx->n_tiles == 0) return GCI_OK;
    int32_t* td = (int32_t*)ctx->tile_i32.p;
    {
        ProfScope _ps(ctx, GCI_PROF_MEMSET);
        HIPCHK(hipMemsetAsync(d_depth, 0, (size_t)ctx->total * 4, ctx->stream));
        HIPCHK(hipMemsetAsync(td, 0, (size_t)ctx->n_tiles * 4, ctx->stream));
    }
    if (max_n) {
        { ProfScope _ps(ctx, GCI_PROF_DEPTH_DIFF);
    hipLaunchKernelGGL(k_depth_diff, dim3((max_n + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, ctx->stream, d_ivl, d_n,
                           max_n, flank, (const int64_t*)ctx->d_len.p, (const int64_t*)ctx->d_off.p,
                           (const int64_t*)ctx->d_tile_first.p, ctx->n_contigs, d_depth, td);
        }
    LAUNCHCHK("k_depth_diff");
    }
    int r;
    { ProfScope _ps(ctx, GCI_PROF_SCAN_TILES);
      r = device_exclusive_scan<int32_t, int32_t>(ctx, td, td, (int32_t*)ctx->blk_i32.p, ctx->n_tiles, false); }
    if (r) return r;
    { ProfScope _ps(ctx, GCI_PROF_DEPTH_SCAN);
    hipLaunchKernelGGL(k_depth_scan, dim3((uint32_t)ctx->n_tiles), dim3(BLOCK), 0, ctx->stream, d_depth, td);
    }
    LAUNCHCHK("k_depth_scan");
    return GCI_OK;
}

// ============================================================================================
// K6: gap mask (GCI.py:324-328), K7: two-type max (GCI.py:350), R15: per-contig sums
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
    const int64_t n4 = ctx->total / 4;
    if (n4 == 0) return GCI_OK;
    const int64_t want = (n4 + BLOCK * 4 - 1) / (BLOCK * 4);
    const uint32_t grid = (uint32_t)(want < 1 ? 1 : want > 16384 ? 16384 : want);
    { ProfScope _ps(ctx, GCI_PROF_MAX2);
    hipLaunchKernelGGL(k_max2, dim3(grid), dim3(BLOCK), 0, ctx->stream, (const int4*)a, (const int4*)b, (int4*)o, n4);
    }
    LAUNCHCHK("k_max2");
    return GCI_OK;
}

__global__ __launch_bounds__(BLOCK) void k_depth_sum(const int32_t* __restrict__ depth,
                                                     const int64_t* __restrict__ tile_first, int32_t n_contigs,
                                                     unsigned long long* __restrict__ sums)
{
    __shared__ long long part[BLOCK / 64];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int4* base = reinterpret_cast<const int4*>(depth + (size_t)blockIdx.x * TILE);
    long long s = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) { const int4 v = base[j * BLOCK + t]; s += (long long)v.x + v.y + v.z + v.w; }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
    if (lane == 0) part[wave] = s;
    __syncthreads();
    if (t == 0) {
        long long a = 0;
        for (int w = 0; w < BLOCK / 64; w++) a += part[w];
        if (a) atomicAdd(sums + contig_of_tile(tile_first, n_contigs, blockIdx.x), (unsigned long long)a);
    }
}

extern "C" int gci_depth_sum(gci_ctx* ctx, const int32_t* d_depth, int64_t* d_sums)
{
    if (!ctx || !d_depth || !d_sums) return GCI_E_INVALID;
    if (!ctx->n_contigs) return GCI_E_NO_LAYOUT;
    HIPCHK(hipMemsetAsync(d_sums, 0, (size_t)ctx->n_contigs * 8, ctx->stream));
    if (ctx->n_tiles == 0) return GCI_OK;
    { ProfScope _ps(ctx, GCI_PROF_DEPTH_SUM);
    hipLaunchKernelGGL(k_depth_sum, dim3((uint32_t)ctx->n_tiles), dim3(BLOCK), 0, ctx->stream, d_depth,
                       (const int64_t*)ctx->d_tile_first.p, ctx->n_contigs, (unsigned long long*)d_sums);
    }
    LAUNCHCHK("k_depth_sum");
    return GCI_OK;
}

// ============================================================================================
// K8: issue scan (collapse_depth_range, GCI.py:369-390) as run-boundary detection
// ============================================================================================
//
// g[p] = (lo < depth[p] <= hi) and p inside the window.  A run starts where g[p] && !g[p-1] and
// ends (exclusive) after g[p] && !g[p+1].  Low-depth runs are rare (CHM13: 11, MH63: 2328), so
// boundaries are appended with one atomic each; the kernel is a pure 4 B/base read stream.

__global__ __launch_bounds__(BLOCK) void k_issue_scan(const int32_t* __restrict__ depth,
                                                      const gci_window* __restrict__ win,
                                                      const int64_t* __restrict__ win_tile_first, int32_t n_win,
                                                      double lo, double hi, unsigned long long* __restrict__ keys,
                                                      uint32_t cap, uint32_t* __restrict__ n_keys)
{
    const int t = threadIdx.x, lane = t & 63;
    const int32_t w = contig_of_tile(win_tile_first, n_win, blockIdx.x);
    const gci_window W = win[w];
    const int64_t tile0 = W.begin / TILE + ((int64_t)blockIdx.x - win_tile_first[w]);
    const int64_t p0 = tile0 * TILE;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int64_t p = p0 + (int64_t)(j * BLOCK + t) * 4;
        const int4 v = *reinterpret_cast<const int4*>(depth + p);
        const int32_t d[4] = {v.x, v.y, v.z, v.w};
        bool g[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const double x = (double)d[k];
            g[k] = (p + k >= W.begin) && (p + k < W.end) && (lo < x) && (x <= hi);
        }
        int gp = __shfl_up((int)g[3], 1, 64);
        int gn = __shfl_down((int)g[0], 1, 64);
        if (lane == 0) {
            gp = 0;
            if (p - 1 >= W.begin && p - 1 < W.end) { const double x = (double)depth[p - 1]; gp = (lo < x) && (x <= hi); }
        }
        if (lane == 63) {
            gn = 0;
            if (p + 4 >= W.begin && p + 4 < W.end) { const double x = (double)depth[p + 4]; gn = (lo < x) && (x <= hi); }
        }
        const bool any = g[0] | g[1] | g[2] | g[3];
        if (any) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const bool prev = k == 0 ? (bool)gp : g[k - 1];
                const bool next = k == 3 ? (bool)gn : g[k + 1];
                if (g[k] && !prev) {
                    const uint32_t s = atomicAdd(n_keys, 1u);
                    if (s < cap) keys[s] = ((unsigned long long)w << 33) | ((unsigned long long)(p + k - W.begin) << 1);
                }
                if (g[k] && !next) {
                    const uint32_t s = atomicAdd(n_keys, 1u);
                    if (s < cap) keys[s] = ((unsigned long long)w << 33) | ((unsigned long long)(p + k + 1 - W.begin) << 1) | 1ull;
                }
            }
        }
    }
}

static int issue_scan_launch(gci_ctx* ctx, const int32_t* d_depth, uint32_t n_win, int64_t n_tiles, double lo, double hi,
                             uint64_t* d_keys, uint32_t cap, uint32_t* d_n_keys)
{
    HIPCHK(hipMemsetAsync(d_n_keys, 0, 4, ctx->stream));
    if (n_tiles == 0) return GCI_OK;
    { ProfScope _ps(ctx, GCI_PROF_ISSUE_SCAN);
    hipLaunchKernelGGL(k_issue_scan, dim3((uint32_t)n_tiles), dim3(BLOCK), 0, ctx->stream, d_depth,
                       (const gci_window*)ctx->win.p, (const int64_t*)ctx->win_tile_first.p, (int32_t)n_win, lo, hi,
                       (unsigned long long*)d_keys, cap, d_n_keys);
    }
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
    int r;
    if ((r = ensure(ctx, ctx->win, (size_t)(n_win + 1) * sizeof(gci_window)))) return r;
    if ((r = ensure(ctx, ctx->win_tile_first, (size_t)(n_win + 2) * 8))) return r;
    if (n_win) if ((r = upload_small(ctx, ctx->win.p, ws.data(), n_win * sizeof(gci_window)))) return r;
    if ((r = upload_small(ctx, ctx->win_tile_first.p, first.data(), first.size() * 8))) return r;
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
    int r = set_windows(ctx, h_windows, n_windows);
    if (r) return r;
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
        int r = set_windows(ctx, ws.data(), (uint32_t)ctx->n_contigs);
        if (r) return r;
        ctx->win_flank = flank;
    }
    return issue_scan_launch(ctx, d_depth, ctx->win_n, ctx->win_tiles, lo, hi, d_keys, cap, d_n_keys);
}

// ============================================================================================
// K10: depth -> decimal text (write_depth body, GCI.py:115-117)
// ============================================================================================

__device__ __forceinline__ uint32_t ndigits(uint32_t v)
{
    return 1u + (v >= 10u) + (v >= 100u) + (v >= 1000u) + (v >= 10000u) + (v >= 100000u) + (v >= 1000000u) +
           (v >= 10000000u) + (v >= 100000000u) + (v >= 1000000000u);
}

// bytes of text per tile: sum over the contig's valid elements of (digits + 1)
__global__ __launch_bounds__(BLOCK) void k_text_count(const int32_t* __restrict__ depth,
                                                      const int64_t* __restrict__ tile_first,
                                                      const int64_t* __restrict__ len, int32_t n_contigs,
                                                      uint32_t* __restrict__ tile_bytes)
{
    __shared__ uint32_t part[BLOCK / 64];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int32_t c = contig_of_tile(tile_first, n_contigs, blockIdx.x);
    const int64_t e0 = ((int64_t)blockIdx.x - tile_first[c]) * TILE;     // element index of the tile in its contig
    const int64_t valid = len[c] - e0;                                   // elements of this tile inside the contig
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
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
    if (lane == 0) part[wave] = s;
    __syncthreads();
    if (t == 0) tile_bytes[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

__global__ void k_text_contig_off(const uint64_t* __restrict__ tile_off, const int64_t* __restrict__ tile_first,
                                  int32_t n_contigs, int64_t n_tiles, uint64_t* __restrict__ contig_off)
{
    const int32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c > n_contigs) return;
    contig_off[c] = c == n_contigs ? tile_off[n_tiles] : tile_off[tile_first[c]];
}

#define TEXT_SUB 1024                    // elements per staging round
#define TEXT_STAGE (TEXT_SUB * 11)       // worst case: 10 digits + '\n'

__global__ __launch_bounds__(BLOCK) void k_text_write(const int32_t* __restrict__ depth,
                                                      const int64_t* __restrict__ tile_first,
                                                      const int64_t* __restrict__ len, int32_t n_contigs,
                                                      const uint64_t* __restrict__ tile_off, uint8_t* __restrict__ out,
                                                      uint64_t cap)
{
    __shared__ uint8_t stage[TEXT_STAGE];
    __shared__ uint32_t wtot[BLOCK / 64];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int32_t c = contig_of_tile(tile_first, n_contigs, blockIdx.x);
    const int64_t valid = len[c] - ((int64_t)blockIdx.x - tile_first[c]) * TILE;
    const int4* base = reinterpret_cast<const int4*>(depth + (size_t)blockIdx.x * TILE);
    uint64_t dst = tile_off[blockIdx.x];
    for (int j = 0; j < 4; j++) {
        const int64_t i = (int64_t)(j * BLOCK + t) * 4;
        const int4 q = base[j * BLOCK + t];
        const uint32_t v[4] = {(uint32_t)q.x, (uint32_t)q.y, (uint32_t)q.z, (uint32_t)q.w};
        uint32_t nd[4], mine = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) { nd[k] = i + k < valid ? ndigits(v[k]) + 1 : 0; mine += nd[k]; }
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
        // staging -> global: byte head to a 4-byte boundary, dword body, byte tail
        if (dst + total <= cap) {
            uint8_t* g = out + dst;
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
        dst += total;
        __syncthreads();
    }
}

extern "C" int gci_depth_text_size(gci_ctx* ctx, const int32_t* d_depth, uint64_t* d_contig_off)
{
    if (!ctx || !d_depth || !d_contig_off) return GCI_E_INVALID;
    if (!ctx->n_contigs) return GCI_E_NO_LAYOUT;
    int r;
    const int64_t nt = ctx->n_tiles;
    if (nt == 0) { HIPCHK(hipMemsetAsync(d_contig_off, 0, (size_t)(ctx->n_contigs + 1) * 8, ctx->stream)); return GCI_OK; }
    if ((r = ensure(ctx, ctx->tile_u32, (size_t)nt * 4))) return r;
    if ((r = ensure(ctx, ctx->tile_u64, (size_t)(nt + 1) * 8))) return r;
    if ((r = ensure(ctx, ctx->blk_u64, (size_t)(nt / TILE + 2) * 8))) return r;
    { ProfScope _ps(ctx, GCI_PROF_TEXT_COUNT);
    hipLaunchKernelGGL(k_text_count, dim3((uint32_t)nt), dim3(BLOCK), 0, ctx->stream, d_depth,
                       (const int64_t*)ctx->d_tile_first.p, (const int64_t*)ctx->d_len.p, ctx->n_contigs,
                       (uint32_t*)ctx->tile_u32.p);
    }
    LAUNCHCHK("k_text_count");
    r = device_exclusive_scan<uint32_t, unsigned long long>(ctx, (const uint32_t*)ctx->tile_u32.p,
                                                            (unsigned long long*)ctx->tile_u64.p,
                                                            (unsigned long long*)ctx->blk_u64.p, nt, true);
    if (r) return r;
    hipLaunchKernelGGL(k_text_contig_off, dim3((ctx->n_contigs + 1 + 63) / 64), dim3(64), 0, ctx->stream,
                       (const uint64_t*)ctx->tile_u64.p, (const int64_t*)ctx->d_tile_first.p, ctx->n_contigs, nt,
                       d_contig_off);
    LAUNCHCHK("k_text_contig_off");
    return GCI_OK;
}

extern "C" int gci_depth_text_write(gci_ctx* ctx, const int32_t* d_depth, uint8_t* d_out, uint64_t cap)
{
    if (!ctx || !d_depth || !d_out) return GCI_E_INVALID;
    if (!ctx->n_contigs) return GCI_E_NO_LAYOUT;
    if (ctx->n_tiles == 0) return GCI_OK;
    if (!ctx->tile_u64.p) return GCI_E_INVALID;    // gci_depth_text_size() first
    { ProfScope _ps(ctx, GCI_PROF_TEXT_WRITE);
    hipLaunchKernelGGL(k_text_write, dim3((uint32_t)ctx->n_tiles), dim3(BLOCK), 0, ctx->stream, d_depth,
                       (const int64_t*)ctx->d_tile_first.p, (const int64_t*)ctx->d_len.p, ctx->n_contigs,
                       (const uint64_t*)ctx->tile_u64.p, d_out, cap);
    }
    LAUNCHCHK("k_text_write");
    return GCI_OK;
}

// ============================================================================================
// K3: cross-file join by read name (GCI.py:272-301) + dict "last record wins" (166, 269)
// ============================================================================================
//
// Open-addressing table keyed by the 64-bit name hash, one slot per DISTINCT name.  A slot holds
// the id (file << 32 | index) of the record that claimed it; keys are compared through the
// immutable record arrays, and a hash match is confirmed on the full name bytes, so a 64-bit
// collision can never merge two reads.  Per (slot, file) an atomicMax keeps the record that the
// reference's dict would keep: the last one in (contig order, file order).  The fold over files
// is then independent per name: one thread per slot.

struct JoinFiles {
    gci_join_file f[GCI_MAX_JOIN_FILES];
    int n;
};

#define SLOT_EMPTY 0xFFFFFFFFFFFFFFFFull

__device__ __forceinline__ const uint8_t* name_ptr(const gci_join_file& f, const gci_rec& r)
{
    return f.d_name_base + f.d_name_off[r.rec_idx] + f.name_delta;
}

__device__ __forceinline__ bool same_name(const JoinFiles& F, int fa, const gci_rec& a, unsigned long long owner)
{
    const int fb = (int)(owner >> 32);
    const gci_rec& b = F.f[fb].d_recs[(uint32_t)owner];
    if (a.name_hash != b.name_hash || a.name_len != b.name_len) return false;
    const uint8_t* pa = name_ptr(F.f[fa], a);
    const uint8_t* pb = name_ptr(F.f[fb], b);
    for (uint32_t i = 0; i < a.name_len; i++) if (pa[i] != pb[i]) return false;
    return true;
}

__global__ __launch_bounds__(BLOCK) void k_join_insert(JoinFiles F, int file, unsigned long long* __restrict__ table,
                                                       uint64_t mask, unsigned long long* __restrict__ last,
                                                       uint32_t* __restrict__ hq)
{
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= F.f[file].n_recs) return;
    const gci_rec r = F.f[file].d_recs[i];
    if (!(r.flags & GCI_REC_PASS)) return;
    const unsigned long long me = ((unsigned long long)file << 32) | i;
    uint64_t slot = r.name_hash & mask;
    for (;;) {
        unsigned long long cur = table[slot];
        if (cur == SLOT_EMPTY) {
            cur = atomicCAS(table + slot, SLOT_EMPTY, me);
            if (cur == SLOT_EMPTY) break;               // claimed
        }
        if (same_name(F, file, r, cur)) break;
        slot = (slot + 1) & mask;
    }
    // order of dict insertion in the reference: contig by contig (header order), file order inside
    const unsigned long long ord = (((unsigned long long)(uint32_t)r.contig << 32) | i) + 1ull;
    atomicMax(last + slot * F.n + file, ord);
    if (r.flags & GCI_REC_HQ) atomicOr(hq + slot, 1u);
}

__global__ __launch_bounds__(BLOCK) void k_join_fold(JoinFiles F, const unsigned long long* __restrict__ table,
                                                     uint64_t n_slots, const unsigned long long* __restrict__ last,
                                                     const uint32_t* __restrict__ hq, double ovlp_percent,
                                                     const int32_t* __restrict__ contig_map, gci_ivl* __restrict__ out,
                                                     uint32_t cap, uint32_t* __restrict__ n_out,
                                                     unsigned long long* __restrict__ status)
{
    const uint64_t slot = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (slot >= n_slots || table[slot] == SLOT_EMPTY) return;
    const bool high = hq[slot] != 0;
    bool comm = true;
    for (int f = 0; f < F.n; f++) comm = comm && last[slot * F.n + f] != 0;
    // file1 = entries of files[0] whose name is in high_qual | comm   (GCI.py:279-280)
    bool have = false;
    int32_t contig = -1, s = 0, e = 0;
    {
        const unsigned long long v = last[slot * F.n];
        if (v && (F.n == 1 || high || comm)) {
            const gci_rec& r = F.f[0].d_recs[(uint32_t)(v - 1)];
            have = true; contig = r.contig; s = r.start; e = r.end;
        }
    }
    for (int f = 1; f < F.n; f++) {                                          // GCI.py:281-299
        const unsigned long long v = last[slot * F.n + f];
        if (!v) continue;
        const gci_rec& r = F.f[f].d_recs[(uint32_t)(v - 1)];
        if (have) {
            if (r.contig == contig) {
                const int32_t ms = max(r.start, s), me = min(r.end, e);
                const int64_t ovlp = (int64_t)me - (int64_t)ms;
                if (r.qlen == 0) { atomicMin(status, ((unsigned long long)r.rec_idx << 8) | (unsigned)(-GCI_E_ZERO_DIV)); return; }
                if ((double)ovlp / (double)r.qlen < ovlp_percent) have = false;
                else { s = ms; e = me; }
            } else have = false;
        } else if (high) {
            have = true; contig = r.contig; s = r.start; e = r.end;
        }
    }
    if (!have) return;
    if (contig_map) { contig = contig_map[contig]; if (contig < 0) return; }
    const uint32_t k = atomicAdd(n_out, 1u);
    if (k < cap) { gci_ivl o; o.contig = contig; o.start = s; o.end = e; o.pad = 0; out[k] = o; }
}

extern "C" int gci_name_join(gci_ctx* ctx, const gci_join_file* h_files, int n_files, double ovlp_percent,
                             const int32_t* d_contig_map, gci_ivl* d_out, uint32_t cap, uint32_t* d_n_out,
                             uint64_t* d_status)
{
    if (!ctx || !h_files || n_files < 1 || n_files > GCI_MAX_JOIN_FILES || !d_n_out || !d_status || (cap && !d_out))
        return GCI_E_INVALID;
    JoinFiles F;
    memset(&F, 0, sizeof F);
    F.n = n_files;
    uint64_t total = 0;
    for (int f = 0; f < n_files; f++) { F.f[f] = h_files[f]; total += h_files[f].n_recs; }
    uint64_t slots = 1024;
    while (slots < 2 * total) slots <<= 1;
    int r;
    if ((r = ensure(ctx, ctx->join_table, slots * 8))) return r;
    if ((r = ensure(ctx, ctx->join_last, slots * 8 * n_files))) return r;
    if ((r = ensure(ctx, ctx->join_hq, slots * 4))) return r;
    HIPCHK(hipMemsetAsync(ctx->join_table.p, 0xFF, slots * 8, ctx->stream));
    HIPCHK(hipMemsetAsync(ctx->join_last.p, 0, slots * 8 * n_files, ctx->stream));
    HIPCHK(hipMemsetAsync(ctx->join_hq.p, 0, slots * 4, ctx->stream));
    HIPCHK(hipMemsetAsync(d_n_out, 0, 4, ctx->stream));
    HIPCHK(hipMemsetAsync(d_status, 0xFF, 8, ctx->stream));
    for (int f = 0; f < n_files; f++) {
        if (!F.f[f].n_recs) continue;
        { ProfScope _ps(ctx, GCI_PROF_JOIN_INSERT);
    hipLaunchKernelGGL(k_join_insert, dim3((F.f[f].n_recs + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, ctx->stream, F, f,
                           (unsigned long long*)ctx->join_table.p, slots - 1, (unsigned long long*)ctx->join_last.p,
                           (uint32_t*)ctx->join_hq.p);
        }
    LAUNCHCHK("k_join_insert");
    }
    { ProfScope _ps(ctx, GCI_PROF_JOIN_FOLD);
    hipLaunchKernelGGL(k_join_fold, dim3((uint32_t)((slots + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, ctx->stream, F,
                       (const unsigned long long*)ctx->join_table.p, slots, (const unsigned long long*)ctx->join_last.p,
                       (const uint32_t*)ctx->join_hq.p, ovlp_percent, d_contig_map, d_out, cap, d_n_out,
                       (unsigned long long*)d_status);
    }
    LAUNCHCHK("k_join_fold");
    return GCI_OK;
}

// ---- names blob for the multi-GPU exchange ------------------------------------------------------

__global__ __launch_bounds__(BLOCK) void k_name_len(const gci_rec* __restrict__ recs, uint32_t n, uint32_t* __restrict__ out)
{
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i < n) out[i] = recs[i].name_len;
}

__global__ __launch_bounds__(BLOCK) void k_pack_names(gci_join_file f, const unsigned long long* __restrict__ off,
                                                      uint8_t* __restrict__ out, uint64_t cap)
{
    // 16 lanes per record copy its name bytes
    const uint32_t i = (blockIdx.x * BLOCK + threadIdx.x) / 16, gl = threadIdx.x % 16;
    if (i >= f.n_recs) return;
    const gci_rec r = f.d_recs[i];
    const uint64_t o = off[i];
    if (o + r.name_len > cap) return;
    const uint8_t* src = f.d_name_base + f.d_name_off[r.rec_idx] + f.name_delta;
    for (uint32_t b = gl; b < r.name_len; b += 16) out[o + b] = src[b];
}

extern "C" int gci_pack_names(gci_ctx* ctx, const gci_join_file* h_file, uint8_t* d_out_names, uint64_t cap,
                              uint64_t* d_out_off)
{
    if (!ctx || !h_file || !d_out_off || (cap && !d_out_names)) return GCI_E_INVALID;
    const uint32_t n = h_file->n_recs;
    int r;
    if ((r = ensure(ctx, ctx->tile_u32, (size_t)(n + 1) * 4))) return r;
    if ((r = ensure(ctx, ctx->blk_u64, (size_t)(n / TILE + 2) * 8))) return r;
    if (n) {
        hipLaunchKernelGGL(k_name_len, dim3((n + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, ctx->stream, h_file->d_recs, n,
                           (uint32_t*)ctx->tile_u32.p);
        LAUNCHCHK("k_name_len");
    }
    r = device_exclusive_scan<uint32_t, unsigned long long>(ctx, (const uint32_t*)ctx->tile_u32.p,
                                                            (unsigned long long*)d_out_off,
                                                            (unsigned long long*)ctx->blk_u64.p, n, true);
    if (r) return r;
    if (n) {
        hipLaunchKernelGGL(k_pack_names, dim3((uint32_t)(((uint64_t)n * 16 + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0,
                           ctx->stream, *h_file, (const unsigned long long*)d_out_off, d_out_names, cap);
        LAUNCHCHK("k_pack_names");
    }
    return GCI_OK;
}
