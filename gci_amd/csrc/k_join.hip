// k_join.hip -- K3: cross-file join by read name (/root/reference/GCI.py:272-301) and the dict
// "last record wins" semantics (GCI.py:166, 269); names blob for the multi-GPU exchange.
#include "gci_ctx.hpp"
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

// name bytes of the record at POSITION idx of file f's record array
__device__ __forceinline__ const uint8_t* name_ptr(const gci_join_file& f, uint32_t idx)
{
    return f.d_name_base + f.d_name_off[idx] + f.name_delta;
}

__device__ __forceinline__ bool same_name(const JoinFiles& F, int fa, uint32_t ia, const gci_rec& a,
                                          unsigned long long owner)
{
    const int fb = (int)(owner >> 32);
    const uint32_t ib = (uint32_t)owner;
    const gci_rec& b = F.f[fb].d_recs[ib];
    if (a.name_hash != b.name_hash || a.name_len != b.name_len) return false;
    const uint8_t* pa = name_ptr(F.f[fa], ia);
    const uint8_t* pb = name_ptr(F.f[fb], ib);
    for (uint32_t i = 0; i < a.name_len; i++) if (pa[i] != pb[i]) return false;
    return true;
}

__global__ __launch_bounds__(BLOCK) void k_join_insert(JoinFiles F, int file, unsigned long long* __restrict__ table,
                                                       uint64_t mask, unsigned long long* __restrict__ last,
                                                       uint32_t* __restrict__ hq, uint32_t* __restrict__ reset_n_out,
                                                       unsigned long long* __restrict__ reset_status)
{
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i == 0 && reset_n_out) { *reset_n_out = 0; *reset_status = SLOT_EMPTY; }    // first launch of a join: the fold appends later
    if (i >= F.f[file].n_recs) return;
    const gci_rec r = F.f[file].d_recs[i];
    if (!(r.flags & GCI_REC_PASS)) return;
    const unsigned long long me = ((unsigned long long)file << 32) | i;
    uint64_t slot = r.name_hash & mask;
    if (F.n == 1) {
        // One file: only the dict's "last record of a name wins" (GCI.py:166, 269).  The slot word itself is the
        // order key (contig, index) + 1 of the record that holds the name: one CAS for a new name, no second table.
        const unsigned long long ord1 = (((unsigned long long)(uint32_t)r.contig << 32) | i) + 1ull;
        for (;;) {
            unsigned long long cur = table[slot];
            if (cur == SLOT_EMPTY) {
                cur = atomicCAS(table + slot, SLOT_EMPTY, ord1);
                if (cur == SLOT_EMPTY) return;
            }
            if (same_name(F, 0, i, r, (unsigned long long)(uint32_t)(cur - 1ull))) { atomicMax(table + slot, ord1); return; }
            slot = (slot + 1) & mask;
        }
    }
    for (;;) {
        unsigned long long cur = table[slot];
        if (cur == SLOT_EMPTY) {
            cur = atomicCAS(table + slot, SLOT_EMPTY, me);
            if (cur == SLOT_EMPTY) break;               // claimed
        }
        if (same_name(F, file, i, r, cur)) break;
        slot = (slot + 1) & mask;
    }
    // order of dict insertion in the reference: contig by contig (header order), file order inside
    const unsigned long long ord = (((unsigned long long)(uint32_t)r.contig << 32) | i) + 1ull;
    atomicMax(last + slot * F.n + file, ord);
    if (F.n > 1 && (r.flags & GCI_REC_HQ)) atomicOr(hq + slot, 1u);    // (a single file keeps every passing name: GCI.py:270)
}

// Fold of one name over the files (GCI.py:279-299); returns true when an interval survives.
__device__ __forceinline__ bool fold_slot(const JoinFiles& F, uint64_t slot, const unsigned long long* last,
                                          bool high, double ovlp_percent, const int32_t* __restrict__ contig_map,
                                          unsigned long long* __restrict__ status, gci_ivl& o)
{
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
                if (r.qlen == 0) {                                           // ZeroDivisionError at GCI.py:292
                    atomicMin(status, ((unsigned long long)r.rec_idx << 8) | (unsigned)(-GCI_E_ZERO_DIV));
                    return false;
                }
                if ((double)ovlp / (double)r.qlen < ovlp_percent) have = false;
                else { s = ms; e = me; }
            } else have = false;
        } else if (high) {
            have = true; contig = r.contig; s = r.start; e = r.end;
        }
    }
    if (!have) return false;
    if (contig_map) { contig = contig_map[contig]; if (contig < 0) return false; }
    o.contig = contig; o.start = s; o.end = e; o.pad = 0;
    return true;
}

// One thread per slot, FOLD_PER_THREAD consecutive slots each; survivors are appended with ONE
// returning atomic per workgroup (a same-address atomic costs ~12 ns on this chip, so per-wave
// appends would serialise for ~100 us at 10^5 intervals).
#define FOLD_PER_THREAD 4
#define COUNT_LDS 256
struct CountArgs { unsigned long long* tile_cd; const int64_t* len; const int64_t* tile_first; int32_t n_contigs; int flank; };
__global__ __launch_bounds__(BLOCK) void k_join_fold(JoinFiles F, unsigned long long* __restrict__ table,
                                                     uint64_t n_slots, unsigned long long* last,
                                                     uint32_t* __restrict__ hq, double ovlp_percent,
                                                     const int32_t* __restrict__ contig_map, gci_ivl* __restrict__ out,
                                                     uint32_t cap, uint32_t* __restrict__ n_out,
                                                     unsigned long long* __restrict__ status, const CountArgs cnt)
{
    __shared__ uint32_t wtot[BLOCK / 64];
    __shared__ uint32_t s_base;
    __shared__ int64_t s_len[COUNT_LDS], s_tf[COUNT_LDS];        // contig tables for the counting pass, loaded up front
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const bool tables_in_lds = cnt.tile_cd && cnt.n_contigs <= COUNT_LDS;
    if (tables_in_lds) for (int c = t; c < cnt.n_contigs; c += BLOCK) { s_len[c] = cnt.len[c]; s_tf[c] = cnt.tile_first[c]; }
    const uint64_t slot0 = ((uint64_t)blockIdx.x * BLOCK + t) * FOLD_PER_THREAD;
    gci_ivl keep[FOLD_PER_THREAD];                  // indexed with compile-time constants only: stays in registers
    bool ok[FOLD_PER_THREAD];
    uint32_t mine = 0;
#pragma unroll
    for (int k = 0; k < FOLD_PER_THREAD; k++) {
        const uint64_t slot = slot0 + k;
        const unsigned long long owner = slot < n_slots ? table[slot] : SLOT_EMPTY;
        const bool used = owner != SLOT_EMPTY;
        if (F.n == 1) {                             // the slot word is the order key of the name's last record
            ok[k] = false;
            if (used) {
                const gci_rec& r = F.f[0].d_recs[(uint32_t)(owner - 1ull)];
                int32_t contig = r.contig;
                if (contig_map) contig = contig_map[contig];
                ok[k] = contig >= 0;
                keep[k].contig = contig; keep[k].start = r.start; keep[k].end = r.end; keep[k].pad = 0;
                table[slot] = SLOT_EMPTY;
            }
        } else {
            ok[k] = used && fold_slot(F, slot, last, hq[slot] != 0, ovlp_percent, contig_map, status, keep[k]);
            if (used) {                             // leave the tables as they were found: no clearing launch next time
                table[slot] = SLOT_EMPTY;
                hq[slot] = 0u;
                for (int f = 0; f < F.n; f++) last[slot * F.n + f] = 0ull;
            }
        }
        mine += ok[k] ? 1u : 0u;
    }
    const uint32_t inc = wave_inclusive<uint32_t>(mine, lane);
    if (lane == 63) wtot[wave] = inc;
    __syncthreads();
    uint32_t pre = inc - mine, all = 0;
#pragma unroll
    for (int w = 0; w < BLOCK / 64; w++) { if (w < wave) pre += wtot[w]; all += wtot[w]; }
    if (t == 0) s_base = all ? atomicAdd(n_out, all) : 0u;
    __syncthreads();
    uint32_t w = s_base + pre;
#pragma unroll
    for (int k = 0; k < FOLD_PER_THREAD; k++) {
        if (ok[k]) {
            if (w < cap) out[w] = keep[k];
            w++;
            if (cnt.tile_cd) {                      // gci_name_join_count: the first pass of the depth build, here
                const IvlSpan sp = tables_in_lds ? span_of(keep[k], cnt.flank, s_len, s_tf, cnt.n_contigs)
                                                 : span_of(keep[k], cnt.flank, cnt.len, cnt.tile_first, cnt.n_contigs);
                if (sp.valid) count_span(sp, cnt.tile_cd);
            }
        }
    }
}

static int name_join_impl(gci_ctx* ctx, const gci_join_file* h_files, int n_files, double ovlp_percent,
                          const int32_t* d_contig_map, gci_ivl* d_out, uint32_t cap, uint32_t* d_n_out,
                          uint64_t* d_status, const CountArgs& cnt)
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
    // The three tables are kept clean between calls (k_join_fold empties every slot it reads), whatever the slot count
    // and the number of files of the next call: a memset only when a buffer is new or a
    // join was cut short.
    const size_t cap_t = ctx->join_table.cap, cap_l = ctx->join_last.cap, cap_h = ctx->join_hq.cap;
    GCI_TRY(gci_ensure(ctx, ctx->join_table, slots * 8));
    GCI_TRY(gci_ensure(ctx, ctx->join_last, slots * 8 * n_files));
    GCI_TRY(gci_ensure(ctx, ctx->join_hq, slots * 4));
    if (ctx->join_table.cap != cap_t || ctx->join_dirty) HIPCHK(hipMemsetAsync(ctx->join_table.p, 0xFF, ctx->join_table.cap, ctx->stream));
    if (ctx->join_last.cap != cap_l || ctx->join_dirty) HIPCHK(hipMemsetAsync(ctx->join_last.p, 0, ctx->join_last.cap, ctx->stream));
    if (ctx->join_hq.cap != cap_h || ctx->join_dirty) HIPCHK(hipMemsetAsync(ctx->join_hq.p, 0, ctx->join_hq.cap, ctx->stream));
    ctx->join_dirty = true;
    bool reset_done = false;
    for (int f = 0; f < n_files; f++) {
        if (!F.f[f].n_recs) continue;
        ProfScope _ps(ctx, GCI_PROF_JOIN_INSERT);
        hipLaunchKernelGGL(k_join_insert, dim3((F.f[f].n_recs + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, ctx->stream, F, f,
                           (unsigned long long*)ctx->join_table.p, slots - 1, (unsigned long long*)ctx->join_last.p,
                           (uint32_t*)ctx->join_hq.p, reset_done ? (uint32_t*)nullptr : d_n_out,
                           reset_done ? (unsigned long long*)nullptr : (unsigned long long*)d_status);
        LAUNCHCHK("k_join_insert");
        reset_done = true;
    }
    if (!reset_done) {
        HIPCHK(hipMemsetAsync(d_n_out, 0, 4, ctx->stream));
        HIPCHK(hipMemsetAsync(d_status, 0xFF, 8, ctx->stream));
    }
    {
        ProfScope _ps(ctx, GCI_PROF_JOIN_FOLD);
        const uint64_t per_block = (uint64_t)BLOCK * FOLD_PER_THREAD;
        hipLaunchKernelGGL(k_join_fold, dim3((uint32_t)((slots + per_block - 1) / per_block)), dim3(BLOCK), 0, ctx->stream,
                           F, (unsigned long long*)ctx->join_table.p, slots,
                           (unsigned long long*)ctx->join_last.p, (uint32_t*)ctx->join_hq.p, ovlp_percent,
                           d_contig_map, d_out, cap, d_n_out, (unsigned long long*)d_status, cnt);
        LAUNCHCHK("k_join_fold");
    }
    ctx->join_dirty = false;
    return GCI_OK;
}

extern "C" int gci_name_join(gci_ctx* ctx, const gci_join_file* h_files, int n_files, double ovlp_percent,
                             const int32_t* d_contig_map, gci_ivl* d_out, uint32_t cap, uint32_t* d_n_out,
                             uint64_t* d_status)
{
    CountArgs none;
    memset(&none, 0, sizeof none);
    return name_join_impl(ctx, h_files, n_files, ovlp_percent, d_contig_map, d_out, cap, d_n_out, d_status, none);
}

// The join, and in the same kernel that emits an interval the first pass of the depth build over it (the per-tile event
// counts and coarse differences that gci_depth_build_begin would otherwise get from a launch of its own).
extern "C" int gci_name_join_count(gci_ctx* ctx, const gci_join_file* h_files, int n_files, double ovlp_percent,
                                   const int32_t* d_contig_map, gci_ivl* d_out, uint32_t cap, uint32_t* d_n_out,
                                   uint64_t* d_status, int flank)
{
    if (!ctx) return GCI_E_INVALID;
    if (!ctx->n_contigs) return GCI_E_NO_LAYOUT;
    if (ctx->n_tiles && ctx->cd_state != 0)
        HIPCHK(hipMemsetAsync(ctx->tile_cd.p, 0, (size_t)(ctx->n_tiles + 1) * 8, ctx->stream));
    ctx->cd_state = 2;
    CountArgs cnt;
    cnt.tile_cd = ctx->n_tiles ? (unsigned long long*)ctx->tile_cd.p : nullptr;
    cnt.len = (const int64_t*)ctx->d_len.p; cnt.tile_first = (const int64_t*)ctx->d_tile_first.p;
    cnt.n_contigs = ctx->n_contigs; cnt.flank = flank;
    const int st = name_join_impl(ctx, h_files, n_files, ovlp_percent, d_contig_map, d_out, cap, d_n_out, d_status, cnt);
    if (st == GCI_OK) { ctx->cd_state = 1; ctx->counted_flank = flank; }
    return st;
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
    const uint8_t* src = f.d_name_base + f.d_name_off[i] + f.name_delta;
    for (uint32_t b = gl; b < r.name_len; b += 16) out[o + b] = src[b];
}

extern "C" int gci_pack_names(gci_ctx* ctx, const gci_join_file* h_file, uint8_t* d_out_names, uint64_t cap,
                              uint64_t* d_out_off)
{
    if (!ctx || !h_file || !d_out_off || (cap && !d_out_names)) return GCI_E_INVALID;
    const uint32_t n = h_file->n_recs;
    int r;
    GCI_TRY(gci_ensure(ctx, ctx->tile_u32, (size_t)(n + 1) * 4));
    GCI_TRY(gci_ensure(ctx, ctx->blk_u64, (size_t)(n / TILE + 2) * 8));
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


// ---- cross-rank name check for the contig-sharded run ---------------------------------------------------------------
//
// A rank that holds the records of its own contigs can run the join locally iff no query name also occurs on
// another rank.  Equal names have equal hashes, so it is enough (and exact) to look for 64-bit hashes that
// arrive from two different ranks: every rank sends each passing record's hash to rank (hash >> 33) % n_parts
// (ONE all-to-all of fixed-size buckets, 8 bytes per record, constant per rank), and the receiver counts hashes
// seen from more than one source.  Zero conflicts on every rank => local joins are exact; otherwise the caller
// falls back to the replicated join over gathered records + names.
// Bucket layout (uint64 words): [0] = number of hashes the sender had for this bucket (may exceed the capacity:
// overflow), [1 .. part_cap] = hashes.

// One global atomic per (workgroup, destination): a same-address returning atomic costs ~12 ns on this chip, so the
// 2137 waves of a chr19 record set appending wave by wave took 29 us; 1024 records per workgroup rank themselves in
// LDS first.  next_out (may be NULL): the bucket array of the NEXT call, whose count words this call zeroes.
#define BUCKET_BLOCK 1024
__global__ __launch_bounds__(BUCKET_BLOCK) void k_hash_bucket(const gci_rec* __restrict__ recs, uint32_t n, uint32_t n_parts,
                                                              uint32_t part_cap, unsigned long long* __restrict__ out,
                                                              unsigned long long* __restrict__ next_out)
{
    __shared__ uint32_t s_cnt[256];
    __shared__ unsigned long long s_base[256];
    const uint32_t t = threadIdx.x;
    const uint32_t i = blockIdx.x * BUCKET_BLOCK + t;
    const size_t stride = (size_t)part_cap + 1;
    if (t < n_parts) s_cnt[t] = 0;
    if (next_out && blockIdx.x == 0 && t < n_parts) next_out[(size_t)t * stride] = 0ull;
    __syncthreads();
    unsigned long long h = 0;
    bool live = false;
    if (i < n) { const gci_rec r = recs[i]; h = r.name_hash; live = (r.flags & GCI_REC_PASS) != 0; }
    const uint32_t d = (uint32_t)((h >> 33) % n_parts);
    uint32_t mine = 0;
    if (live) mine = atomicAdd(&s_cnt[d], 1u);                    // rank among this workgroup's records for bucket d
    __syncthreads();
    if (t < n_parts) s_base[t] = s_cnt[t] ? atomicAdd(out + (size_t)t * stride, (unsigned long long)s_cnt[t]) : 0ull;
    __syncthreads();
    if (live) {
        const unsigned long long slot = s_base[d] + mine;
        if (slot < part_cap) out[(size_t)d * stride + 1 + slot] = h;
    }
}

__global__ void k_hash_bucket_clear(unsigned long long* out, uint32_t n_parts, uint32_t part_cap)
{
    const uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d < n_parts) out[(size_t)d * ((size_t)part_cap + 1)] = 0ull;
}

// Two tables used alternately: a call inserts into one and wipes the other for the next call (a strip per thread,
// streaming stores), so no fill is launched per call.
__global__ __launch_bounds__(BLOCK) void k_hash_conflicts(const unsigned long long* __restrict__ buckets, uint32_t n_parts,
                                                          uint32_t part_cap, unsigned long long* __restrict__ table,
                                                          uint64_t mask, uint32_t* __restrict__ n_conflicts,
                                                          unsigned long long* __restrict__ next_table)
{
    const uint32_t src = blockIdx.y;
    const size_t n_threads = (size_t)gridDim.x * gridDim.y * BLOCK;
    for (size_t k = ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * BLOCK + threadIdx.x; k <= mask; k += n_threads)
        __builtin_nontemporal_store(SLOT_EMPTY, next_table + k);
    const unsigned long long* b = buckets + (size_t)src * ((size_t)part_cap + 1);
    const unsigned long long cnt = b[0];
    if (cnt > part_cap) { if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(n_conflicts, 1u); }   // overflow: force the fallback
    const uint32_t m = (uint32_t)(cnt < part_cap ? cnt : part_cap);
    for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < m; i += gridDim.x * BLOCK) {
        const unsigned long long h = b[1 + i];
        const unsigned long long word = (h << 8) | src;               // 56 hash bits + source rank (n_parts <= 255)
        uint64_t slot = (h ^ (h >> 29)) & mask;
        for (;;) {
            unsigned long long cur = table[slot];
            if (cur == SLOT_EMPTY) {
                cur = atomicCAS(table + slot, SLOT_EMPTY, word);
                if (cur == SLOT_EMPTY) break;
            }
            if ((cur >> 8) == (word >> 8)) { if ((cur & 0xFF) != src) atomicAdd(n_conflicts, 1u); break; }
            slot = (slot + 1) & mask;
        }
    }
}

extern "C" int gci_hash_bucket(gci_ctx* ctx, const gci_rec* d_recs, uint32_t n, uint32_t n_parts, uint32_t part_cap,
                               uint64_t* d_out, uint64_t* d_next_out)
{
    if (!ctx || !d_out || n_parts == 0 || n_parts > 255 || (n && !d_recs)) return GCI_E_INVALID;
    if (!d_next_out) {                                     // no ping-pong partner: clear the count words here
        hipLaunchKernelGGL(k_hash_bucket_clear, dim3(1), dim3(256), 0, ctx->stream, (unsigned long long*)d_out, n_parts, part_cap);
        LAUNCHCHK("k_hash_bucket_clear");
    }
    if (n || d_next_out) {
        hipLaunchKernelGGL(k_hash_bucket, dim3(n ? (n + BUCKET_BLOCK - 1) / BUCKET_BLOCK : 1), dim3(BUCKET_BLOCK), 0, ctx->stream,
                           d_recs, n, n_parts, part_cap, (unsigned long long*)d_out, (unsigned long long*)d_next_out);
        LAUNCHCHK("k_hash_bucket");
    }
    return GCI_OK;
}

// *d_n_conflicts is ADDED to (the caller zeroes it when it wants a fresh count)
extern "C" int gci_hash_conflicts(gci_ctx* ctx, const uint64_t* d_buckets, uint32_t n_parts, uint32_t part_cap,
                                  uint32_t* d_n_conflicts)
{
    if (!ctx || !d_buckets || !d_n_conflicts || n_parts == 0 || n_parts > 255) return GCI_E_INVALID;
    uint64_t slots = 1024;
    while (slots < 2ull * n_parts * part_cap) slots <<= 1;
    // its own tables (the join's stay in their clean state), two of them: [0, slots) and [slots, 2 slots)
    const size_t cap_before = ctx->conflict_table.cap;
    GCI_TRY(gci_ensure(ctx, ctx->conflict_table, 2 * slots * 8));
    if (ctx->conflict_table.cap != cap_before || ctx->conflict_slots != slots) {
        HIPCHK(hipMemsetAsync(ctx->conflict_table.p, 0xFF, 2 * slots * 8, ctx->stream));
        ctx->conflict_slots = slots;
        ctx->conflict_parity = 0;
    }
    unsigned long long* t0 = (unsigned long long*)ctx->conflict_table.p + (size_t)ctx->conflict_parity * slots;
    unsigned long long* t1 = (unsigned long long*)ctx->conflict_table.p + (size_t)(ctx->conflict_parity ^ 1u) * slots;
    ctx->conflict_parity ^= 1u;
    const uint32_t gx = (part_cap + BLOCK - 1) / BLOCK;
    hipLaunchKernelGGL(k_hash_conflicts, dim3(gx ? (gx > 1024 ? 1024 : gx) : 1, n_parts), dim3(BLOCK), 0, ctx->stream,
                       (const unsigned long long*)d_buckets, n_parts, part_cap, t0, slots - 1, d_n_conflicts, t1);
    LAUNCHCHK("k_hash_conflicts");
    return GCI_OK;
}
