// k_shard.hip -- multi-GPU: the exchange steps of the name-hash-sharded join (replaces the fan-out of
// /root/reference/GCI.py:257-270 and keeps the join of GCI.py:272-301 exact across GPUs).
//
// Contigs are sharded over the ranks, so a rank's record filter sees the records of ITS contigs -- but the join is by read
// name: a read aligned to contigs of two ranks must be dropped (GCI.py:296-297), a name repeated across contigs keeps its
// last record (GCI.py:269).  The join treats every name independently, so it shards by name: every passing record goes
// to rank (name hash >> 33) % N (one all-to-all of 32-byte records and one of their names per input file), each rank joins
// the names it owns with the same kernels as a single GPU, and the surviving 16-byte intervals go to the rank that owns
// their contig (one more all-to-all).  Per step and rank that is (32 + name slot) * R * (N - 1) / N^2 bytes out for R records in
// the job, against (32 + name) * R * (N - 1) / N of the replicated join of round 2.
//
// Buckets have a fixed capacity (the collectives keep one shape from step to step, nothing is sized on the host per step):
//   records:   n_parts buckets of (cap + 1) gci_rec slots; slot 0 is a header (name_hash = number of records routed here,
//              flags = 0), slots 1 .. hold the records IN FILE ORDER (the routing is stable: among records of one file with
//              the same name on the same contig the last one must stay the last one);
//              names: n_parts * cap slots of name_slot bytes (a multiple of 16 the caller picks: at least the longest
//              name), zero padded;
//   intervals: n_parts buckets of (cap + 1) gci_ivl slots, slot 0 = {contig = -1, start = count}.
// A count beyond the capacity (or a name longer than a slot) is reported as GCI_E_CAPACITY: the caller grows the buckets.
#include "gci_ctx.hpp"

#define ROUTE_CHUNK 4096
#define ROUTE_MAX_PARTS 64

struct RouteSrc {
    const gci_rec* recs; const uint8_t* name_base; const uint64_t* name_off; uint32_t name_delta;   // records
    const gci_ivl* ivl; const uint32_t* d_n; const int32_t* owner; int32_t n_owner;                  // intervals
    const gci_paf_hit* hits;                                                                        // PAF hits (names: name_base + qn_off)
    uint32_t n;                                                                                     // items (intervals: at most)
};
static_assert(sizeof(gci_paf_hit) == GCI_PAF_HIT_BYTES, "a PAF hit is 80 bytes");
#define ROUTE_RECS 0
#define ROUTE_IVL 1
#define ROUTE_HITS 2

template <int KIND>
__device__ __forceinline__ int route_dest(const RouteSrc& S, uint32_t i, uint32_t n, uint32_t n_parts)
{
    constexpr bool IVL = KIND == ROUTE_IVL;
    if (i >= n) return -1;
    if (KIND == ROUTE_HITS) return (int)((S.hits[i].qhash >> 33) % n_parts);
    if (IVL) {
        const int32_t c = S.ivl[i].contig;
        if (c < 0 || c >= S.n_owner) return -1;
        const int32_t o = S.owner[c];
        return o >= 0 && (uint32_t)o < n_parts ? o : -1;
    }
    const gci_rec r = S.recs[i];
    if (!(r.flags & GCI_REC_PASS)) return -1;
    return (int)((r.name_hash >> 33) % n_parts);
}

template <int KIND>
__global__ __launch_bounds__(BLOCK) void k_route_count(const RouteSrc S, uint32_t n_parts, uint32_t n_chunks, uint32_t* __restrict__ counts)
{
    constexpr bool IVL = KIND == ROUTE_IVL;
    __shared__ uint32_t h[ROUTE_MAX_PARTS];
    const uint32_t t = threadIdx.x, chunk = blockIdx.x;
    if (t < ROUTE_MAX_PARTS) h[t] = 0;
    __syncthreads();
    const uint32_t n = IVL ? min(*S.d_n, S.n) : S.n;
    for (uint32_t k = 0; k < ROUTE_CHUNK / BLOCK; k++) {
        const int d = route_dest<KIND>(S, chunk * ROUTE_CHUNK + k * BLOCK + t, n, n_parts);
        if (d >= 0) atomicAdd(&h[d], 1u);
    }
    __syncthreads();
    if (t < n_parts) counts[(size_t)t * n_chunks + chunk] = h[t];
}

// dword d of the len bytes at p (any alignment), zero past len; naturally aligned loads only
__device__ __forceinline__ uint32_t route_name_dword(const uint8_t* p, uint32_t len, uint32_t d)
{
    if (4u * d >= len) return 0u;
    const uint8_t* q = p + 4u * d;
    const uint32_t sh = (uint32_t)((uintptr_t)q & 3u);
    const uint32_t* a = reinterpret_cast<const uint32_t*>(q - sh);
    const uint32_t left = len - 4u * d;
    const uint32_t lo = a[0];
    const uint32_t hi = (sh && sh + (left < 4u ? left : 4u) > 4u) ? a[1] : 0u;      // only if the bytes reach into the next dword
    uint32_t w = __builtin_amdgcn_alignbyte(hi, lo, sh);
    if (left < 4u) w &= (1u << (8u * left)) - 1u;
    return w;
}

template <int KIND>
__global__ __launch_bounds__(BLOCK) void k_route_scatter(const RouteSrc S, uint32_t n_parts, uint32_t n_chunks, uint32_t cap,
                                                         const uint32_t* __restrict__ off, uint8_t* __restrict__ out,
                                                         uint8_t* __restrict__ out_names, uint32_t name_slot, unsigned long long* __restrict__ status)
{
    constexpr bool IVL = KIND == ROUTE_IVL;
    __shared__ uint32_t run[ROUTE_MAX_PARTS], wcnt[BLOCK / 64][ROUTE_MAX_PARTS];
    const uint32_t t = threadIdx.x, chunk = blockIdx.x, lane = t & 63, wave = t >> 6;
    if (t < n_parts) run[t] = off[(size_t)t * n_chunks + chunk] - off[(size_t)t * n_chunks];     // position inside the bucket
    if (chunk == 0 && t < n_parts) {                              // headers: how many items were routed to each part
        const uint32_t total = off[(size_t)(t + 1) * n_chunks] - off[(size_t)t * n_chunks];
        if (KIND == ROUTE_HITS) {
            gci_paf_hit h; memset(&h, 0, sizeof h); h.qhash = total; h.t = -1;
            reinterpret_cast<gci_paf_hit*>(out)[(size_t)t * (cap + 1)] = h;
        } else if (IVL) {
            gci_ivl h; h.contig = -1; h.start = (int32_t)total; h.end = 0; h.pad = 0;
            reinterpret_cast<gci_ivl*>(out)[(size_t)t * (cap + 1)] = h;
        } else {
            gci_rec h; memset(&h, 0, sizeof h); h.name_hash = total; h.contig = -1;
            reinterpret_cast<gci_rec*>(out)[(size_t)t * (cap + 1)] = h;
        }
        if (total > cap) atomicMin(status, (unsigned long long)(unsigned)(-GCI_E_CAPACITY));
    }
    const uint32_t n = IVL ? min(*S.d_n, S.n) : S.n;
    for (uint32_t k = 0; k < ROUTE_CHUNK / BLOCK; k++) {
        for (uint32_t i = t; i < (BLOCK / 64) * ROUTE_MAX_PARTS; i += BLOCK) (&wcnt[0][0])[i] = 0;
        __syncthreads();
        const uint32_t i = chunk * ROUTE_CHUNK + k * BLOCK + t;
        const int d = route_dest<KIND>(S, i, n, n_parts);
        // rank among the lanes of this wave that go to the same part, in lane order
        uint32_t lane_rank = 0;
        for (unsigned long long todo = __ballot(d >= 0); todo;) {
            const int d0 = __builtin_amdgcn_readlane(d, __builtin_ctzll(todo));
            const unsigned long long m = __ballot(d == d0);
            if (d == d0) lane_rank = (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull));
            if (lane == (uint32_t)__builtin_ctzll(m)) wcnt[wave][d0] = (uint32_t)__builtin_popcountll(m);
            todo &= ~m;
        }
        __syncthreads();
        uint32_t pos = 0;
        if (d >= 0) {
            pos = run[d] + lane_rank;
            for (uint32_t w = 0; w < wave; w++) pos += wcnt[w][d];
        }
        __syncthreads();
        if (t < n_parts) { uint32_t a = 0; for (uint32_t w = 0; w < BLOCK / 64; w++) a += wcnt[w][t]; run[t] += a; }
        if (d >= 0 && pos < cap) {
            const size_t slot = (size_t)d * (cap + 1) + 1 + pos;
            if (KIND == ROUTE_HITS) {
                const gci_paf_hit h = S.hits[i];
                reinterpret_cast<gci_paf_hit*>(out)[slot] = h;
                uint32_t* nd = reinterpret_cast<uint32_t*>(out_names + ((size_t)d * cap + pos) * name_slot);
                if (h.qn_len > name_slot) atomicMin(status, (unsigned long long)(unsigned)(-GCI_E_CAPACITY));
                const uint8_t* src = S.name_base + h.qn_off;
                const uint32_t len = h.qn_len < name_slot ? h.qn_len : name_slot;
                for (uint32_t w = 0; w < name_slot / 4; w++) nd[w] = route_name_dword(src, len, w);
            } else if (IVL) reinterpret_cast<gci_ivl*>(out)[slot] = S.ivl[i];
            else {
                gci_rec r = S.recs[i];
                r.flags |= GCI_REC_NAME16;                        // a routed name slot: 16-byte aligned, zero padded
                reinterpret_cast<gci_rec*>(out)[slot] = r;
                uint32_t* nd = reinterpret_cast<uint32_t*>(out_names + ((size_t)d * cap + pos) * name_slot);
                if (r.name_len > name_slot) atomicMin(status, (unsigned long long)(unsigned)(-GCI_E_CAPACITY));
                const uint8_t* src = S.name_base + S.name_off[i] + S.name_delta;
                const uint32_t len = r.name_len < name_slot ? r.name_len : name_slot;
                for (uint32_t w = 0; w < name_slot / 4; w++) nd[w] = route_name_dword(src, len, w);
            }
        }
        __syncthreads();
    }
}

static int route_impl(gci_ctx* ctx, const RouteSrc& S, int kind, uint32_t n_parts, uint32_t cap, uint8_t* d_out, uint8_t* d_out_names,
                      uint32_t name_slot, uint64_t* d_status)
{
    if (n_parts == 0 || n_parts > ROUTE_MAX_PARTS || !d_out || !d_status) return GCI_E_INVALID;
    const uint32_t n_chunks = S.n ? (S.n + ROUTE_CHUNK - 1) / ROUTE_CHUNK : 1;
    const size_t n_tab = (size_t)n_parts * n_chunks;
    GCI_TRY(gci_ensure(ctx, ctx->route_tab, (n_tab + 2) * 4));
    GCI_TRY(gci_ensure(ctx, ctx->part_blk, (n_tab / TILE + 2) * 4));
    uint32_t* tab = (uint32_t*)ctx->route_tab.p;
    HIPCHK(hipMemsetAsync(d_status, 0xFF, 8, ctx->stream));
    if (kind == ROUTE_IVL) hipLaunchKernelGGL((k_route_count<ROUTE_IVL>), dim3(n_chunks), dim3(BLOCK), 0, ctx->stream, S, n_parts, n_chunks, tab);
    else if (kind == ROUTE_HITS) hipLaunchKernelGGL((k_route_count<ROUTE_HITS>), dim3(n_chunks), dim3(BLOCK), 0, ctx->stream, S, n_parts, n_chunks, tab);
    else hipLaunchKernelGGL((k_route_count<ROUTE_RECS>), dim3(n_chunks), dim3(BLOCK), 0, ctx->stream, S, n_parts, n_chunks, tab);
    LAUNCHCHK("k_route_count");
    int r = device_exclusive_scan<uint32_t, uint32_t>(ctx, tab, tab, (uint32_t*)ctx->part_blk.p, (int64_t)n_tab, true);
    if (r) return r;
    if (kind == ROUTE_IVL) hipLaunchKernelGGL((k_route_scatter<ROUTE_IVL>), dim3(n_chunks), dim3(BLOCK), 0, ctx->stream, S, n_parts, n_chunks, cap,
                                              (const uint32_t*)tab, d_out, d_out_names, name_slot, (unsigned long long*)d_status);
    else if (kind == ROUTE_HITS) hipLaunchKernelGGL((k_route_scatter<ROUTE_HITS>), dim3(n_chunks), dim3(BLOCK), 0, ctx->stream, S, n_parts, n_chunks,
                                                    cap, (const uint32_t*)tab, d_out, d_out_names, name_slot, (unsigned long long*)d_status);
    else hipLaunchKernelGGL((k_route_scatter<ROUTE_RECS>), dim3(n_chunks), dim3(BLOCK), 0, ctx->stream, S, n_parts, n_chunks, cap,
                            (const uint32_t*)tab, d_out, d_out_names, name_slot, (unsigned long long*)d_status);
    LAUNCHCHK("k_route_scatter");
    return GCI_OK;
}

extern "C" int gci_route_records(gci_ctx* ctx, const gci_join_file* h_file, uint32_t n_parts, uint32_t cap, gci_rec* d_out_recs,
                                 uint8_t* d_out_names, uint32_t name_slot, uint64_t* d_status)
{
    if (name_slot < 16 || name_slot > 65536 || (name_slot & 15u)) return GCI_E_INVALID;
    if (!ctx || !h_file || !d_out_names || (h_file->n_recs && (!h_file->d_recs || !h_file->d_name_base || !h_file->d_name_off)))
        return GCI_E_INVALID;
    RouteSrc S;
    memset(&S, 0, sizeof S);
    S.recs = h_file->d_recs; S.name_base = h_file->d_name_base; S.name_off = h_file->d_name_off; S.name_delta = h_file->name_delta;
    S.n = h_file->n_recs;
    return route_impl(ctx, S, ROUTE_RECS, n_parts, cap, (uint8_t*)d_out_recs, d_out_names, name_slot, d_status);
}

extern "C" int gci_route_intervals(gci_ctx* ctx, const gci_ivl* d_ivl, const uint32_t* d_n, uint32_t max_n, const int32_t* d_owner,
                                   int32_t n_contigs, uint32_t n_parts, uint32_t cap, gci_ivl* d_out, uint64_t* d_status)
{
    if (!ctx || !d_n || !d_owner || (max_n && !d_ivl)) return GCI_E_INVALID;
    RouteSrc S;
    memset(&S, 0, sizeof S);
    S.ivl = d_ivl; S.d_n = d_n; S.owner = d_owner; S.n_owner = n_contigs; S.n = max_n;
    return route_impl(ctx, S, ROUTE_IVL, n_parts, cap, (uint8_t*)d_out, nullptr, 16, d_status);
}

// PAF hits of a byte range -> the ranks that own their query names (stable: a query's hits keep their line order).
extern "C" int gci_route_hits(gci_ctx* ctx, const uint8_t* d_hits, uint32_t n, const uint8_t* d_name_base, uint32_t n_parts, uint32_t cap,
                              uint8_t* d_out_hits, uint8_t* d_out_names, uint32_t name_slot, uint64_t* d_status)
{
    if (name_slot < 16 || name_slot > 65536 || (name_slot & 15u)) return GCI_E_INVALID;
    if (!ctx || !d_out_names || (n && (!d_hits || !d_name_base))) return GCI_E_INVALID;
    RouteSrc S;
    memset(&S, 0, sizeof S);
    S.hits = reinterpret_cast<const gci_paf_hit*>(d_hits); S.name_base = d_name_base; S.n = n;
    return route_impl(ctx, S, ROUTE_HITS, n_parts, cap, d_out_hits, d_out_names, name_slot, d_status);
}

// ---- after the all-to-all: what arrived beyond a bucket's count is not data ---------------------------------------------

__global__ __launch_bounds__(BLOCK) void k_seal_records(gci_rec* __restrict__ recs, uint32_t n_parts, uint32_t cap,
                                                        unsigned long long* __restrict__ status)
{
    const uint32_t d = blockIdx.y;
    gci_rec* b = recs + (size_t)d * (cap + 1);
    const unsigned long long cnt = b[0].name_hash;
    if (cnt > cap && blockIdx.x == 0 && threadIdx.x == 0) atomicMin(status, (unsigned long long)(unsigned)(-GCI_E_CAPACITY));
    for (uint32_t k = blockIdx.x * BLOCK + threadIdx.x; k <= cap; k += gridDim.x * BLOCK)
        if (k == 0 || k > cnt) b[k].flags = 0;
}

__global__ __launch_bounds__(BLOCK) void k_seal_intervals(gci_ivl* __restrict__ ivl, uint32_t n_parts, uint32_t cap,
                                                          const int32_t* __restrict__ cmap, int32_t n_map,
                                                          unsigned long long* __restrict__ status)
{
    const uint32_t d = blockIdx.y;
    gci_ivl* b = ivl + (size_t)d * (cap + 1);
    const uint32_t cnt = (uint32_t)b[0].start;
    if (cnt > cap && blockIdx.x == 0 && threadIdx.x == 0) atomicMin(status, (unsigned long long)(unsigned)(-GCI_E_CAPACITY));
    __syncthreads();
    for (uint32_t k = blockIdx.x * BLOCK + threadIdx.x + 1; k <= cap; k += gridDim.x * BLOCK) {
        const int32_t c = b[k].contig;
        b[k].contig = (k <= cnt && c >= 0 && c < n_map) ? cmap[c] : -1;
    }
}

extern "C" int gci_route_seal_records(gci_ctx* ctx, gci_rec* d_recs, uint32_t n_parts, uint32_t cap, uint64_t* d_status)
{
    if (!ctx || !d_recs || !d_status || n_parts == 0 || n_parts > ROUTE_MAX_PARTS) return GCI_E_INVALID;
    HIPCHK(hipMemsetAsync(d_status, 0xFF, 8, ctx->stream));
    const uint32_t gx = (cap + 1 + BLOCK - 1) / BLOCK;
    hipLaunchKernelGGL(k_seal_records, dim3(gx > 256 ? 256 : gx, n_parts), dim3(BLOCK), 0, ctx->stream, d_recs, n_parts, cap,
                       (unsigned long long*)d_status);
    LAUNCHCHK("k_seal_records");
    return GCI_OK;
}

// d_cmap[global contig] = index of the contig among this rank's own (the track layout), -1 for the others
extern "C" int gci_route_seal_intervals(gci_ctx* ctx, gci_ivl* d_ivl, uint32_t n_parts, uint32_t cap, const int32_t* d_cmap,
                                        int32_t n_contigs, uint64_t* d_status)
{
    if (!ctx || !d_ivl || !d_cmap || !d_status || n_parts == 0 || n_parts > ROUTE_MAX_PARTS) return GCI_E_INVALID;
    HIPCHK(hipMemsetAsync(d_status, 0xFF, 8, ctx->stream));
    const uint32_t gx = (cap + 1 + BLOCK - 1) / BLOCK;
    hipLaunchKernelGGL(k_seal_intervals, dim3(gx > 256 ? 256 : gx, n_parts), dim3(BLOCK), 0, ctx->stream, d_ivl, n_parts, cap, d_cmap,
                       n_contigs, (unsigned long long*)d_status);
    LAUNCHCHK("k_seal_intervals");
    return GCI_OK;
}
