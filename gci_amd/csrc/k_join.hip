// k_join.hip -- K3: cross-file join by read name (/root/reference/GCI.py:272-301) and the dict
// "last record wins" semantics (GCI.py:166, 269); names blob for the multi-GPU exchange.
#include "gci_ctx.hpp"
#include <stdlib.h>
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


// =====================================================================================================================
// Partitioned join (genome scale).  With 10^7 names the classic table above is ~1 GB of open addressing: every record
// costs several random device-scope atomics that miss every cache (measured: 5 G/s, a third of the BASELINE configs[2]
// step).  Here the passing records are first radix-partitioned by bits of the name hash -- two passes of
// histogram / scan / scatter, sequential traffic only -- into buckets small enough that the whole dict logic of one
// bucket (claim a slot per distinct name, last-record-wins per file, high-quality bit, then the fold over files) runs
// in LDS.
//
// Round 3: the entry carries everything the fold needs (contig, start, end, qlen: the record's interval travels with
// its hash through both scatters) and where its name bytes are, so the join kernel reads its bucket SEQUENTIALLY and
// never goes back to the record arrays (round 2 fetched a 128-byte line per (name, file) winner: 8x the algorithmic
// bytes, profiles/r02t_pmc_summary.txt).  What is left of the random traffic is what exactness needs -- the name bytes
// of an entry against those of the slot's claimant -- and it is no longer inside the probe loop: the insert trusts the
// 48 hash bits, the name comparisons of a bucket are issued afterwards, all independent of each other, and a bucket in
// which two DIFFERENT names share their hash bits (one in ~10 genome-scale joins has such a pair) is redone by its
// workgroup with the names compared inside the probe loop.  Same results as the classic path (tests run both);
// chosen by size (gci_join_mode / GCI_JOIN=classic|partition override).
//
//   entry (32 B): key = hash48 | name_len << 48 | (name offset & 15) << 60;  idx = position in its file;
//                 meta = contig | file << 26 | hq << 30 | name16 << 31;  start, end, qlen;  name offset >> 4
//   level 1: B1 = 2^b1 buckets by hash bits [47, 48 - b1); chunks of 8192 records of ONE file
//   level 2: every level-1 bucket again by the next b2 bits; chunks never straddle level-1 buckets
//   join   : one workgroup per final bucket, table in LDS: slot = {tag40 | claimant23 | hq} + one order key per file

struct __attribute__((aligned(16))) PartEntry {
    unsigned long long key;
    uint32_t idx;
    uint32_t meta;
    int32_t start, end, qlen;
    uint32_t name_off16;
};
static_assert(sizeof(PartEntry) == 32, "partition entry is 32 bytes");
#define PART_CHUNK 8192
#define PART_CONTIG_BITS 26
#define PART_KEY_MASK 0xFFFFFFFFFFFFull            // the hash bits of a key
#define PART_CMP_MASK 0x0FFFFFFFFFFFFFFFull         // hash bits + name length
#define PART_MAX_NAME 4095u
#define PART_WIN (1ull << 63)                       // order-key word of a (slot, file) once its winner is known: WIN | entry index
#define PART_NONE 0xFFFFFFFFu
struct PartFiles { uint32_t chunk_first[GCI_MAX_JOIN_FILES + 1]; };

// A record takes part in the partition iff this holds (histogram and scatter must agree on it).
__device__ __forceinline__ bool part_takes(const gci_rec& r)
{
    return (r.flags & GCI_REC_PASS) && !((uint32_t)r.contig >> PART_CONTIG_BITS) && r.name_len <= PART_MAX_NAME;
}

// A passing record the entry format cannot hold (contig index beyond 26 bits, a name of 4 KiB and more, name bytes
// beyond 64 GiB from their base): the caller takes the classic table.  Reported with record index 0, so that it wins
// the min over whatever else this (abandoned) join reports.
__device__ __forceinline__ void part_refuse(unsigned long long* status)
{
    atomicMin(status, (unsigned long long)(unsigned)(-GCI_E_CAPACITY));
}

__device__ __forceinline__ int part_file_of(const PartFiles& P, int n, uint32_t chunk)
{
    int f = 0;
    while (f + 1 < n && chunk >= P.chunk_first[f + 1]) f++;
    return f;
}

__device__ __forceinline__ const uint8_t* entry_name(const JoinFiles& F, const PartEntry& e)
{
    return F.f[(e.meta >> PART_CONTIG_BITS) & 15u].d_name_base + (((uint64_t)e.name_off16 << 4) | (uint64_t)(e.key >> 60));
}

__device__ __forceinline__ void store_entry(PartEntry* __restrict__ dst, const PartEntry& e)
{
    int4* d = reinterpret_cast<int4*>(dst);
    d[0] = make_int4((int)(uint32_t)e.key, (int)(uint32_t)(e.key >> 32), (int)e.idx, (int)e.meta);
    d[1] = make_int4(e.start, e.end, e.qlen, (int)e.name_off16);
}

__device__ __forceinline__ PartEntry load_entry(const PartEntry* __restrict__ src)
{
    const int4* s = reinterpret_cast<const int4*>(src);
    const int4 a = s[0], b = s[1];
    PartEntry e;
    e.key = (unsigned long long)(uint32_t)a.x | ((unsigned long long)(uint32_t)a.y << 32);
    e.idx = (uint32_t)a.z; e.meta = (uint32_t)a.w;
    e.start = b.x; e.end = b.y; e.qlen = b.z; e.name_off16 = (uint32_t)b.w;
    return e;
}

__global__ __launch_bounds__(BLOCK) void k_part1_hist(JoinFiles F, PartFiles P, int shift, uint32_t n_bins, uint32_t n_chunks,
                                                      uint32_t* __restrict__ hist, uint32_t* __restrict__ reset_n_out,
                                                      unsigned long long* __restrict__ status)
{
    __shared__ uint32_t h[256];
    const uint32_t t = threadIdx.x, chunk = blockIdx.x;
    if (chunk == 0 && t == 0) *reset_n_out = 0;                   // (the status word was reset by a memset before this launch)
    h[t] = 0;
    __syncthreads();
    const int f = part_file_of(P, F.n, chunk);
    const uint32_t i0 = (chunk - P.chunk_first[f]) * PART_CHUNK, n = F.f[f].n_recs;
    const gci_rec* __restrict__ recs = F.f[f].d_recs;
#pragma unroll 4
    for (int k = 0; k < PART_CHUNK / BLOCK; k++) {
        const uint32_t i = i0 + k * BLOCK + t;
        if (i < n) {
            const gci_rec r = recs[i];
            if (part_takes(r)) atomicAdd(&h[(uint32_t)((r.name_hash & PART_KEY_MASK) >> shift) & (n_bins - 1)], 1u);
            else if (r.flags & GCI_REC_PASS) part_refuse(status);
        }
    }
    __syncthreads();
    if (t < n_bins) hist[(size_t)t * n_chunks + chunk] = h[t];
}

// ---- scatter through LDS ----------------------------------------------------------------------------------------------
// A lane that stores its entry where its bin's cursor points writes 32 bytes into a line nobody else of its wave touches:
// the runs of a (chunk, bin) fill over the whole life of the workgroup, a CU's worth of half-written lines does not fit
// the L2 and the memory side sees partial writes (round 2: 1.6x the bytes; with 32-byte entries 0.98 ms for both levels,
// profiles/r03a_*).  So a workgroup ranks SC_TILE entries by bin in LDS first and copies the sorted tile out: consecutive
// lanes write consecutive 16-byte pieces of their bin's run.
#define SC_TILE 1024
#define SC_PER (SC_TILE / BLOCK)
struct ScatterLds {
    int4 stage[SC_TILE * 2];
    uint32_t tcnt[256], toff[256], gofs[256], cnt[256], base[256];
    uint32_t wtot[BLOCK / 64], total;
    uint8_t sbin[SC_TILE];
};

// One tile: every thread brings SC_PER entries with their bins (d < 0: no entry).  All threads of the workgroup call it.
__device__ __forceinline__ void scatter_tile(ScatterLds& L, const PartEntry (&e)[SC_PER], const int (&d)[SC_PER], uint32_t n_bins,
                                             PartEntry* __restrict__ out, int t)
{
    const int lane = t & 63, wave = t >> 6;
    uint32_t r[SC_PER];
#pragma unroll
    for (int k = 0; k < SC_PER; k++) r[k] = d[k] >= 0 ? atomicAdd(&L.tcnt[d[k]], 1u) : 0u;
    __syncthreads();
    const uint32_t c = (uint32_t)t < n_bins ? L.tcnt[t] : 0u;
    const uint32_t inc = wave_inclusive<uint32_t>(c, lane);
    if (lane == 63) L.wtot[wave] = inc;
    __syncthreads();
    uint32_t pre = inc - c, all = 0;
#pragma unroll
    for (int w = 0; w < BLOCK / 64; w++) { if (w < wave) pre += L.wtot[w]; all += L.wtot[w]; }
    L.toff[t] = pre;
    L.gofs[t] = L.base[t] + L.cnt[t] - pre;                       // global index of the tile's staged position 0 of this bin, - 0
    if (t == 0) L.total = all;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < SC_PER; k++) {
        if (d[k] >= 0) {
            const uint32_t p = L.toff[d[k]] + r[k];
            L.stage[2 * p] = make_int4((int)(uint32_t)e[k].key, (int)(uint32_t)(e[k].key >> 32), (int)e[k].idx, (int)e[k].meta);
            L.stage[2 * p + 1] = make_int4(e[k].start, e[k].end, e[k].qlen, (int)e[k].name_off16);
            L.sbin[p] = (uint8_t)d[k];
        }
    }
    __syncthreads();
    const uint32_t total = L.total;
    for (uint32_t q = t; q < 2 * total; q += BLOCK) {
        const uint32_t p = q >> 1;
        const uint32_t bin = L.sbin[p];
        reinterpret_cast<int4*>(out + (L.gofs[bin] + p))[q & 1] = L.stage[q];
    }
    L.cnt[t] += c;
    L.tcnt[t] = 0;
    __syncthreads();
}

__global__ __launch_bounds__(BLOCK) void k_part1_scatter(JoinFiles F, PartFiles P, int shift, uint32_t n_bins, uint32_t n_chunks,
                                                         const uint32_t* __restrict__ off, PartEntry* __restrict__ out,
                                                         unsigned long long* __restrict__ status)
{
    __shared__ ScatterLds L;
    const uint32_t t = threadIdx.x, chunk = blockIdx.x;
    L.cnt[t] = 0; L.tcnt[t] = 0;
    L.base[t] = t < n_bins ? off[(size_t)t * n_chunks + chunk] : 0u;
    __syncthreads();
    const int f = part_file_of(P, F.n, chunk);
    const uint32_t i0 = (chunk - P.chunk_first[f]) * PART_CHUNK, n = F.f[f].n_recs;
    const gci_rec* __restrict__ recs = F.f[f].d_recs;
    const uint64_t* __restrict__ noff = F.f[f].d_name_off;
    const uint64_t delta = F.f[f].name_delta;
    for (uint32_t tile = 0; tile < PART_CHUNK && i0 + tile < n; tile += SC_TILE) {
        PartEntry e[SC_PER];
        int d[SC_PER];
#pragma unroll
        for (int k = 0; k < SC_PER; k++) {
            const uint32_t i = i0 + tile + k * BLOCK + t;
            d[k] = -1;
            e[k].key = 0; e[k].idx = 0; e[k].meta = 0; e[k].start = e[k].end = e[k].qlen = 0; e[k].name_off16 = 0;
            if (i < n) {
                const gci_rec r = recs[i];
                if (part_takes(r)) {
                    const uint64_t at = noff[i] + delta;              // name bytes relative to the file's name base
                    if (at >> 36) part_refuse(status);
                    e[k].key = (r.name_hash & PART_KEY_MASK) | ((unsigned long long)r.name_len << 48) | ((unsigned long long)(at & 15ull) << 60);
                    e[k].idx = i;
                    e[k].meta = (uint32_t)r.contig | ((uint32_t)f << PART_CONTIG_BITS) | ((r.flags & GCI_REC_HQ) ? 1u << 30 : 0u) |
                                ((r.flags & GCI_REC_NAME16) ? 1u << 31 : 0u);
                    e[k].start = r.start; e[k].end = r.end; e[k].qlen = r.qlen;
                    e[k].name_off16 = (uint32_t)(at >> 4);
                    d[k] = (int)((uint32_t)((e[k].key & PART_KEY_MASK) >> shift) & (n_bins - 1));
                }
            }
        }
        scatter_tile(L, e, d, n_bins, out, (int)t);
    }
}

// Segment table of level 2 from the scanned level-1 histogram: seg[0 .. B1] = first entry of every level-1 bucket
// (+ the total), seg[B1 + 1 .. 2 B1 + 1] = exclusive scan of their chunk counts.  One workgroup.
__global__ __launch_bounds__(BLOCK) void k_part_mid(const uint32_t* __restrict__ off1, uint32_t n_bins, uint32_t n_chunks,
                                                    uint32_t* __restrict__ seg)
{
    __shared__ uint32_t s[257], c[257];
    const uint32_t t = threadIdx.x;
    if (t <= n_bins) s[t] = off1[(size_t)t * n_chunks];          // t == n_bins: the grand total the scan wrote behind the table
    if (t == 0 && n_bins == 256) s[256] = off1[(size_t)256 * n_chunks];
    __syncthreads();
    if (t == 0) {
        uint32_t acc = 0;
        for (uint32_t d = 0; d < n_bins; d++) { c[d] = acc; acc += (s[d + 1] - s[d] + PART_CHUNK - 1) / PART_CHUNK; }
        c[n_bins] = acc;
    }
    __syncthreads();
    if (t <= n_bins) { seg[t] = s[t]; seg[n_bins + 1 + t] = c[t]; }
    if (t == 0 && n_bins == 256) { seg[256] = s[256]; seg[513] = c[256]; }
}

// chunk w of level 2 -> (segment j, chunk k inside it); false when w is beyond the last chunk
__device__ __forceinline__ bool part2_chunk(const uint32_t* sS, const uint32_t* sC, uint32_t n_seg, uint32_t w, uint32_t& j,
                                            uint32_t& k, uint32_t& a, uint32_t& b, uint32_t& nch)
{
    if (w >= sC[n_seg]) return false;
    uint32_t lo = 0, hi = n_seg;                                 // first index with sC[idx] > w lies in (lo, hi]
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (sC[mid] <= w) lo = mid; else hi = mid; }
    j = lo; k = w - sC[j]; nch = sC[j + 1] - sC[j];
    a = sS[j] + k * PART_CHUNK;
    b = min(sS[j + 1], a + PART_CHUNK);
    return true;
}

__global__ __launch_bounds__(BLOCK) void k_part2_hist(const PartEntry* __restrict__ in, const uint32_t* __restrict__ seg,
                                                      uint32_t n_seg, int shift, uint32_t n_bins, uint32_t* __restrict__ hist)
{
    __shared__ uint32_t sS[257], sC[257], h[256];
    const uint32_t t = threadIdx.x, w = blockIdx.x;
    for (uint32_t i = t; i <= n_seg; i += BLOCK) { sS[i] = seg[i]; sC[i] = seg[n_seg + 1 + i]; }
    h[t] = 0;
    __syncthreads();
    uint32_t j, k, a, b, nch;
    if (!part2_chunk(sS, sC, n_seg, w, j, k, a, b, nch)) {        // unused tail of the table: zero for the scan
        if (t < n_bins) hist[(size_t)w * n_bins + t] = 0;
        return;
    }
    for (uint32_t i = a + t; i < b; i += BLOCK)
        atomicAdd(&h[(uint32_t)((in[i].key & PART_KEY_MASK) >> shift) & (n_bins - 1)], 1u);
    __syncthreads();
    if (t < n_bins) hist[(size_t)sC[j] * n_bins + (size_t)t * nch + k] = h[t];
}

__global__ __launch_bounds__(BLOCK) void k_part2_scatter(const PartEntry* __restrict__ in, const uint32_t* __restrict__ seg,
                                                         uint32_t n_seg, int shift, uint32_t n_bins,
                                                         const uint32_t* __restrict__ off, PartEntry* __restrict__ out)
{
    __shared__ ScatterLds L;
    __shared__ uint32_t sS[257], sC[257];
    const uint32_t t = threadIdx.x, w = blockIdx.x;
    for (uint32_t i = t; i <= n_seg; i += BLOCK) { sS[i] = seg[i]; sC[i] = seg[n_seg + 1 + i]; }
    L.cnt[t] = 0; L.tcnt[t] = 0;
    __syncthreads();
    uint32_t j, k, a, b, nch;
    if (!part2_chunk(sS, sC, n_seg, w, j, k, a, b, nch)) return;
    L.base[t] = t < n_bins ? off[(size_t)sC[j] * n_bins + (size_t)t * nch + k] : 0u;
    __syncthreads();
    for (uint32_t tile = a; tile < b; tile += SC_TILE) {
        PartEntry e[SC_PER];
        int d[SC_PER];
#pragma unroll
        for (int q = 0; q < SC_PER; q++) {
            const uint32_t i = tile + q * BLOCK + t;
            d[q] = -1;
            e[q].key = 0; e[q].idx = 0; e[q].meta = 0; e[q].start = e[q].end = e[q].qlen = 0; e[q].name_off16 = 0;
            if (i < b) {
                e[q] = load_entry(in + i);
                d[q] = (int)((uint32_t)((e[q].key & PART_KEY_MASK) >> shift) & (n_bins - 1));
            }
        }
        scatter_tile(L, e, d, n_bins, out, (int)t);
    }
}

// len bytes at pa against len bytes at pb, through naturally aligned dword loads re-aligned in registers (an
// unaligned vector load from global memory is ~100x slower here, and a byte loop with an early exit serialises its
// loads: both names sit in HBM at random places, so the nine dwords of a 32-byte round are requested together);
// nothing beyond the dword that holds the last byte of either name is touched.
__device__ __forceinline__ bool names_equal(const uint8_t* __restrict__ pa, const uint8_t* __restrict__ pb, uint32_t len)
{
    const uint32_t sa = (uint32_t)((uintptr_t)pa & 3u), sb = (uint32_t)((uintptr_t)pb & 3u);
    const uint32_t* __restrict__ qa = reinterpret_cast<const uint32_t*>(pa - sa);
    const uint32_t* __restrict__ qb = reinterpret_cast<const uint32_t*>(pb - sb);
    const uint32_t na = (sa + len + 3u) >> 2, nb = (sb + len + 3u) >> 2;      // dwords that hold bytes of the name
    uint32_t diff = 0;
    for (uint32_t w0 = 0; 4u * w0 < len; w0 += 8) {
        uint32_t a[9], b[9];
#pragma unroll
        for (int k = 0; k < 9; k++) {
            a[k] = w0 + k < na ? qa[w0 + k] : 0u;
            b[k] = w0 + k < nb ? qb[w0 + k] : 0u;
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint32_t w = w0 + k;
            if (4u * w < len) {
                uint32_t x = __builtin_amdgcn_alignbyte(a[k + 1], a[k], sa) ^ __builtin_amdgcn_alignbyte(b[k + 1], b[k], sb);
                const uint32_t left = len - 4u * w;
                if (left < 4u) x &= (1u << (8u * left)) - 1u;
                diff |= x;
            }
        }
    }
    return diff == 0;
}

// Two names that both carry GCI_REC_NAME16 (inside record pages / routed slots: at 0 or 4 mod 16, zero bytes behind them up
// to the next 16-byte boundary), of the same length and the same phase: equal iff the 16-byte pieces they lie in are equal
// from the name's first dword on -- three aligned 16-byte loads per name, all six requested together (the dword compare
// above issues a dozen loads per name and a second round trip from 33 bytes on; per wave that is 64 cache lines per load
// instruction: the address path, not the bytes, is what the comparisons cost).  len <= 44.
__device__ __forceinline__ bool names_equal16(const uint8_t* __restrict__ pa, const uint8_t* __restrict__ pb, uint32_t len)
{
    const uint32_t s = (uint32_t)((uintptr_t)pa & 15u);                  // == pb & 15, 0 or 4
    const uint4* __restrict__ qa = reinterpret_cast<const uint4*>(pa - s);
    const uint4* __restrict__ qb = reinterpret_cast<const uint4*>(pb - s);
    const uint32_t end = s + len;
    const uint4 z = make_uint4(0, 0, 0, 0);
    const uint4 a0 = qa[0], b0 = qb[0];
    const uint4 a1 = end > 16 ? qa[1] : z, b1 = end > 16 ? qb[1] : z;
    const uint4 a2 = end > 32 ? qa[2] : z, b2 = end > 32 ? qb[2] : z;
    uint32_t d = (s ? 0u : a0.x ^ b0.x) | (a0.y ^ b0.y) | (a0.z ^ b0.z) | (a0.w ^ b0.w);
    d |= (a1.x ^ b1.x) | (a1.y ^ b1.y) | (a1.z ^ b1.z) | (a1.w ^ b1.w);
    d |= (a2.x ^ b2.x) | (a2.y ^ b2.y) | (a2.z ^ b2.z) | (a2.w ^ b2.w);
    return d == 0;
}

struct PartJoinArgs {
    const PartEntry* in; const uint32_t* seg; const uint32_t* off2;
    uint32_t n_seg, n_bins2, slots, slot_shift;
    uint32_t no_verify;          // GCI_JOIN_NOVERIFY=1 (timing experiments: what the name comparisons cost; results not exact)
};

// Slot of entry e (index i inside its bucket) in the bucket's LDS table.  CLAIM: take an empty slot for a name not seen yet.
// exact == false: entries whose 40 tag bits (and, being in one bucket, all 48 hash bits) agree share a slot -- their names are
// compared afterwards (k_join_part).  exact == true: a tag match counts only if the full key and the name bytes equal those of
// the slot's claimant.  PART_NONE: table full (CLAIM) / name not in the table (cannot happen after an insert of the same entry).
template <bool CLAIM>
__device__ __forceinline__ uint32_t part_probe(unsigned long long* meta, uint32_t S, uint32_t slot_shift, const PartEntry& e, uint32_t i,
                                               const PartEntry* __restrict__ in, const JoinFiles& F, bool exact, uint16_t* used,
                                               uint32_t* s_used)
{
    const unsigned long long tag = e.key & 0xFFFFFFFFFFull;
    uint32_t slot = (uint32_t)((e.key & PART_KEY_MASK) >> slot_shift) & (S - 1);
    for (uint32_t probes = 0; probes < S; probes++) {
        unsigned long long m = meta[slot];
        if (m == SLOT_EMPTY) {
            if (!CLAIM) return PART_NONE;
            m = atomicCAS(meta + slot, SLOT_EMPTY, (tag << 24) | ((unsigned long long)i << 1));
            if (m == SLOT_EMPTY) {                               // claimed: this entry is the slot's reference name
                used[atomicAdd(s_used, 1u)] = (uint16_t)slot;
                return slot;
            }
        }
        if ((m >> 24) == tag) {
            if (!exact) return slot;
            const uint32_t c = (uint32_t)(m >> 1) & 0x7FFFFFu;
            if (c == i) return slot;
            const PartEntry ce = load_entry(in + c);
            if (((ce.key ^ e.key) & PART_CMP_MASK) == 0 &&
                names_equal(entry_name(F, e), entry_name(F, ce), (uint32_t)(e.key >> 48) & 0xFFFu))
                return slot;
        }
        slot = (slot + 1) & (S - 1);
    }
    return PART_NONE;
}

// The fold of fold_slot() (GCI.py:279-299) over the winners' ENTRIES: their intervals came along with the hashes, so the
// fold reads the bucket it has just streamed (L2) and nothing else.  FN == 0: any number of files, one by one.
template <int FN>
__device__ __forceinline__ bool fold_bucket_slot(const JoinFiles& F, const PartEntry* __restrict__ in, uint32_t slot,
                                                 const unsigned long long* last, bool high, double ovlp_percent,
                                                 const int32_t* __restrict__ contig_map, unsigned long long* __restrict__ status,
                                                 gci_ivl& o)
{
    constexpr int NF = FN ? FN : 1;
    const int Fn = FN ? FN : F.n;
    bool have = false, dead = false;                              // dead: the reference raised on this name
    int32_t contig = -1, s = 0, e = 0;
    const uint32_t cmask = (1u << PART_CONTIG_BITS) - 1u;
    if constexpr (FN != 0) {
        unsigned long long v[NF];
        int4 ra[NF], rb[NF];                                     // the winners' entries as two 16-byte halves, requested together
        bool comm = true;
#pragma unroll
        for (int f = 0; f < NF; f++) { v[f] = last[(size_t)slot * NF + f]; comm = comm && v[f] != 0; }
#pragma unroll
        for (int f = 0; f < NF; f++) {
            ra[f] = make_int4(0, 0, 0, 0); rb[f] = make_int4(0, 0, 0, 0);
            if (v[f]) {
                const int4* p = reinterpret_cast<const int4*>(in + (uint32_t)(v[f] & 0x7FFFFFull));
                ra[f] = p[0]; rb[f] = p[1];
            }
        }
        // entry: {key lo, key hi, idx, meta} {start, end, qlen, name offset}
        if (v[0] && (FN == 1 || high || comm)) { have = true; contig = (int32_t)((uint32_t)ra[0].w & cmask); s = rb[0].x; e = rb[0].y; }
#pragma unroll
        for (int f = 1; f < NF; f++) {                            // GCI.py:281-299 (no early exit: the loop stays unrolled)
            const int32_t r_contig = (int32_t)((uint32_t)ra[f].w & cmask), r_start = rb[f].x, r_end = rb[f].y, r_qlen = rb[f].z;
            const bool present = v[f] != 0 && !dead;
            if (present && have) {
                if (r_contig == contig) {
                    const int32_t ms = max(r_start, s), me = min(r_end, e);
                    const int64_t ovlp = (int64_t)me - (int64_t)ms;
                    if (r_qlen == 0) {                            // ZeroDivisionError at GCI.py:292
                        atomicMin(status, ((unsigned long long)F.f[f].d_recs[(uint32_t)ra[f].z].rec_idx << 8) | (unsigned)(-GCI_E_ZERO_DIV));
                        dead = true;
                    } else if ((double)ovlp / (double)r_qlen < ovlp_percent) have = false;
                    else { s = ms; e = me; }
                } else have = false;
            } else if (present && high) {
                have = true; contig = r_contig; s = r_start; e = r_end;
            }
        }
    } else {
        bool comm = true;
        for (int f = 0; f < Fn; f++) comm = comm && last[(size_t)slot * Fn + f] != 0;
        for (int f = 0; f < Fn && !dead; f++) {
            const unsigned long long v = last[(size_t)slot * Fn + f];
            if (!v) continue;
            if (f == 0) {                                         // file1 = entries of files[0] in high_qual | comm (GCI.py:279-280)
                if (Fn == 1 || high || comm) {
                    const PartEntry r = load_entry(in + (uint32_t)(v & 0x7FFFFFull));
                    have = true; contig = (int32_t)(r.meta & cmask); s = r.start; e = r.end;
                }
                continue;
            }
            if (!have && !high) continue;
            const PartEntry r = load_entry(in + (uint32_t)(v & 0x7FFFFFull));
            const int32_t r_contig = (int32_t)(r.meta & cmask);
            if (have) {
                if (r_contig == contig) {
                    const int32_t ms = max(r.start, s), me = min(r.end, e);
                    const int64_t ovlp = (int64_t)me - (int64_t)ms;
                    if (r.qlen == 0) {
                        atomicMin(status, ((unsigned long long)F.f[f].d_recs[r.idx].rec_idx << 8) | (unsigned)(-GCI_E_ZERO_DIV));
                        dead = true;
                    } else if ((double)ovlp / (double)r.qlen < ovlp_percent) have = false;
                    else { s = ms; e = me; }
                } else have = false;
            } else {
                have = true; contig = r_contig; s = r.start; e = r.end;
            }
        }
    }
    if (!have || dead) return false;
    if (contig_map) { contig = contig_map[contig]; if (contig < 0) return false; }
    o.contig = contig; o.start = s; o.end = e; o.pad = 0;
    return true;
}

// The fold (GCI.py:279-299) of one name by the thread that holds, in registers, the winning entry of the LOWEST file the
// name occurs in; what it needs of the later files' winners -- contig, start, end, qlen -- those left in LDS when they won.
template <int FN>
__device__ __forceinline__ bool fold_from_entry(const JoinFiles& F, const PartEntry* __restrict__ in, const PartEntry& mine, int f0,
                                                uint32_t slot, const unsigned long long* last, const int4* pay, bool high, double ovlp_percent,
                                                const int32_t* __restrict__ contig_map, unsigned long long* __restrict__ status, gci_ivl& o)
{
    const int Fn = FN ? FN : F.n;
    const uint32_t cmask = (1u << PART_CONTIG_BITS) - 1u;
    bool comm = f0 == 0;
    for (int f = f0 + 1; f < Fn; f++) comm = comm && last[(size_t)slot * Fn + f] != 0;
    // file1 = entries of files[0] whose name is in high_qual | comm (GCI.py:279-280); a name files[0] does not have enters
    // with the first later file that has it, if it is high-quality (GCI.py:298-299)
    if (!(f0 == 0 ? (Fn == 1 || high || comm) : high)) return false;
    int32_t contig = (int32_t)(mine.meta & cmask), s = mine.start, e = mine.end;
    bool have = true;
    for (int f = f0 + 1; f < Fn; f++) {
        const unsigned long long v = last[(size_t)slot * Fn + f];
        if (!v) continue;
        if (!have && !high) continue;
        const int4 r = pay[(size_t)slot * Fn + f];                           // {contig, start, end, qlen} of that file's winner
        const int32_t r_contig = r.x;
        if (have) {
            if (r_contig == contig) {
                const int32_t ms = max(r.y, s), me = min(r.z, e);
                const int64_t ovlp = (int64_t)me - (int64_t)ms;
                if (r.w == 0) {                                              // ZeroDivisionError at GCI.py:292
                    const uint32_t idx = in[(uint32_t)(v & 0x7FFFFFull)].idx;
                    atomicMin(status, ((unsigned long long)F.f[f].d_recs[idx].rec_idx << 8) | (unsigned)(-GCI_E_ZERO_DIV));
                    return false;
                }
                if ((double)ovlp / (double)r.w < ovlp_percent) have = false;
                else { s = ms; e = me; }
            } else have = false;
        } else {
            have = true; contig = r_contig; s = r.y; e = r.z;
        }
    }
    if (!have) return false;
    if (contig_map) { contig = contig_map[contig]; if (contig < 0) return false; }
    o.contig = contig; o.start = s; o.end = e; o.pad = 0;
    return true;
}

// One workgroup per final bucket.  LDS: meta[S] | cname[S] | last[S * F.n] | pay[S * F.n] | list of used slots.
//   meta  = tag (hash bits [39, 0]) << 24 | claimant (entry index inside the bucket) << 1 | high-quality bit; ~0 = empty
//   cname = where the claimant's name is: offset >> 4 | (offset & 15) << 32 | length << 36 | file << 48 | name16 << 52
//   last  = per (slot, file) the largest order key (contig, file position) + 1 seen, then WIN | index of the entry that has it
//   pay   = per (slot, file) what the fold needs of that entry: {contig, start, end, qlen}
// A bucket of up to PART_E * JB = 1024 entries (all but forged inputs) stays in REGISTERS from its one coalesced read to the
// fold: insert (LDS only) -> winners + every non-claimant's name against its claimant's (two independent random reads per
// entry, addresses from registers and LDS: nothing in front of them) -> fold.  Larger buckets re-read their entries per phase.
#ifndef PART_E
#define PART_E 2
#endif
#ifndef JB
#define JB 512
#endif
// JB threads per bucket workgroup, PART_E entries per thread in registers: PART_E * JB = 1024 = the most a table of 1024 slots holds
template <int FN>
__global__ __launch_bounds__(JB) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_join_part(JoinFiles F, PartJoinArgs A, double ovlp_percent,
                                                     const int32_t* __restrict__ contig_map, gci_ivl* __restrict__ sparse,
                                                     uint32_t* __restrict__ bucket_cnt, uint32_t* __restrict__ bucket_lo,
                                                     unsigned long long* __restrict__ status, const CountArgs cnt)
{
    extern __shared__ __attribute__((aligned(16))) unsigned long long sm[];
    __shared__ uint32_t wtot[JB / 64];
    __shared__ uint32_t s_used, s_bad;
    __shared__ int64_t s_len[COUNT_LDS], s_tf[COUNT_LDS];
    const uint32_t S = A.slots;
    const int Fn = FN ? FN : F.n;
    unsigned long long* meta = sm;
    unsigned long long* cname = sm + S;
    unsigned long long* last = sm + 2 * (size_t)S;
    int4* pay = reinterpret_cast<int4*>(sm + 2 * (size_t)S + (size_t)S * Fn);
    uint16_t* used = reinterpret_cast<uint16_t*>(sm + 2 * (size_t)S + 3 * (size_t)S * Fn);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const uint32_t j = blockIdx.x / A.n_bins2, d = blockIdx.x % A.n_bins2;
    const uint32_t cj = A.seg[A.n_seg + 1 + j], nch = A.seg[A.n_seg + 2 + j] - cj;
    if (nch == 0) { if (t == 0) bucket_cnt[blockIdx.x] = 0u; return; }      // empty level-1 bucket (uniform over the workgroup)
    const uint32_t lo = A.off2[(size_t)cj * A.n_bins2 + (size_t)d * nch], hi = A.off2[(size_t)cj * A.n_bins2 + (size_t)(d + 1) * nch];
    if (t == 0) { bucket_cnt[blockIdx.x] = 0u; bucket_lo[blockIdx.x] = lo; }
    if (lo == hi) return;
    const bool tables_in_lds = cnt.tile_cd && cnt.n_contigs <= COUNT_LDS;
    if (tables_in_lds) for (int c = t; c < cnt.n_contigs; c += JB) { s_len[c] = cnt.len[c]; s_tf[c] = cnt.tile_first[c]; }
    const PartEntry* __restrict__ in = A.in + lo;
    const uint32_t n = hi - lo;
    if (n >> 23) {                                               // claimant field: a bucket of 8 M entries and more (one name
        if (t == 0) part_refuse(status);                         // repeated): classic path
        return;
    }
    const uint32_t cmask = (1u << PART_CONTIG_BITS) - 1u;
    const bool in_regs = n <= PART_E * JB;
    PartEntry ent[PART_E];
    uint32_t slot_of[PART_E];
#pragma unroll
    for (int k = 0; k < PART_E; k++) {
        slot_of[k] = PART_NONE;
        const uint32_t i = (uint32_t)t + (uint32_t)k * JB;
        if (in_regs && i < n) ent[k] = load_entry(in + i);
        else { ent[k].key = 0; ent[k].idx = 0; ent[k].meta = 0; ent[k].start = ent[k].end = ent[k].qlen = 0; ent[k].name_off16 = 0; }
    }
    auto name_word = [](const PartEntry& e) -> unsigned long long {
        return (unsigned long long)e.name_off16 | ((e.key >> 60) << 32) | (((e.key >> 48) & 0xFFFull) << 36) |
               ((unsigned long long)((e.meta >> PART_CONTIG_BITS) & 15u) << 48) | ((unsigned long long)(e.meta >> 31) << 52);
    };
    // one entry through the insert; -> its slot
    auto insert = [&](const PartEntry& e, uint32_t i, bool exact) -> uint32_t {
        const uint32_t slot = part_probe<true>(meta, S, A.slot_shift, e, i, in, F, exact, used, &s_used);
        if (slot == PART_NONE) { part_refuse(status); return slot; }                  // more distinct names than slots
        if (((uint32_t)(meta[slot] >> 1) & 0x7FFFFFu) == i) cname[slot] = name_word(e);   // this entry is the claimant
        const int file = (int)((e.meta >> PART_CONTIG_BITS) & 15u);
        // order of dict insertion in the reference: contig by contig (header order), file order inside (GCI.py:269)
        const unsigned long long ord = (((unsigned long long)(e.meta & cmask) << 32) | e.idx) + 1ull;
        atomicMax(last + (size_t)slot * Fn + file, ord);
        if (Fn > 1 && (e.meta >> 30 & 1u)) atomicOr(meta + slot, 1ull);
        return slot;
    };
    // winner marking + (hash-trusting insert) the name against the claimant's
    auto settle = [&](const PartEntry& e, uint32_t i, uint32_t slot, bool exact) {
        const int file = (int)((e.meta >> PART_CONTIG_BITS) & 15u);
        const unsigned long long ord = (((unsigned long long)(e.meta & cmask) << 32) | e.idx) + 1ull;
        if (last[(size_t)slot * Fn + file] == ord) {
            last[(size_t)slot * Fn + file] = PART_WIN | i;
            pay[(size_t)slot * Fn + file] = make_int4((int)(e.meta & cmask), e.start, e.end, e.qlen);
        }
        if (!exact && !A.no_verify && ((uint32_t)(meta[slot] >> 1) & 0x7FFFFFu) != i) {
            const unsigned long long cw = cname[slot];
            const uint32_t len = (uint32_t)(e.key >> 48) & 0xFFFu;
            const uint8_t* cn = F.f[(cw >> 48) & 15u].d_name_base + (((cw & 0xFFFFFFFFull) << 4) | ((cw >> 32) & 15ull));
            const uint8_t* mn = entry_name(F, e);
            bool same;
            if (((uint32_t)(cw >> 36) & 0xFFFu) != len) same = false;
            else if ((e.meta >> 31) && ((cw >> 52) & 1ull) && len <= 44u && (((uintptr_t)mn ^ (uintptr_t)cn) & 15u) == 0)
                same = names_equal16(mn, cn, len);
            else same = names_equal(mn, cn, len);
            if (!same) s_bad = 1u;                                       // two names, one hash
        }
    };
    bool exact = false;
    for (;;) {
        for (uint32_t s = t; s < S; s += JB) meta[s] = SLOT_EMPTY;
        for (uint32_t s = t; s < S * (uint32_t)Fn; s += JB) last[s] = 0ull;
        if (t == 0) { s_used = 0; s_bad = 0; }
        __syncthreads();
        if (in_regs) {
#pragma unroll
            for (int k = 0; k < PART_E; k++) {
                const uint32_t i = (uint32_t)t + (uint32_t)k * JB;
                if (i < n) slot_of[k] = insert(ent[k], i, exact);
            }
        } else {
            for (uint32_t i = t; i < n; i += JB) insert(load_entry(in + i), i, exact);
        }
        __syncthreads();
        if (in_regs) {
#pragma unroll
            for (int k = 0; k < PART_E; k++) {
                const uint32_t i = (uint32_t)t + (uint32_t)k * JB;
                if (i < n && slot_of[k] != PART_NONE) settle(ent[k], i, slot_of[k], exact);
            }
        } else {
            for (uint32_t i = t; i < n; i += JB) {
                const PartEntry e = load_entry(in + i);
                const uint32_t slot = part_probe<false>(meta, S, A.slot_shift, e, i, in, F, exact, used, &s_used);
                if (slot != PART_NONE) settle(e, i, slot, exact);
            }
        }
        __syncthreads();
        if (exact || !s_bad) break;
        exact = true;                                            // a bucket with two names under one hash: redo it exactly
        __syncthreads();                                         // everybody has read s_bad before it is reset
    }
    // ---- fold: by the thread that holds the winner of the lowest file of a name; survivors appended with one returning
    // atomic per workgroup (per round of JB entries for the buckets that do not fit the registers) ---------------------
    auto fold_one = [&](const PartEntry& e, uint32_t i, uint32_t slot, gci_ivl& keep) -> bool {
        const int file = (int)((e.meta >> PART_CONTIG_BITS) & 15u);
        if (last[(size_t)slot * Fn + file] != (PART_WIN | i)) return false;
        for (int f = 0; f < file; f++) if (last[(size_t)slot * Fn + f] != 0) return false;      // an earlier file has the name
        return fold_from_entry<FN>(F, in, e, file, slot, last, pay, (meta[slot] & 1ull) != 0, ovlp_percent, contig_map, status, keep);
    };
    // A bucket's survivors go where its entries were counted: slot lo + rank of a buffer as long as the partition -- a name
    // survives at most once, so they fit -- and k_join_gather makes the dense output of them.  (One returning atomic per bucket
    // on the output counter was what bounded this kernel: 32 768 same-address atomics at ~12 ns each.)
    auto emit = [&](const gci_ivl& keep, uint32_t w) {
        sparse[lo + w] = keep;
        if (cnt.tile_cd) {
            const IvlSpan sp = tables_in_lds ? span_of(keep, cnt.flank, s_len, s_tf, cnt.n_contigs)
                                             : span_of(keep, cnt.flank, cnt.len, cnt.tile_first, cnt.n_contigs);
            if (sp.valid) count_span(sp, cnt.tile_cd);
        }
    };
    if (in_regs) {
        gci_ivl keep[PART_E];
        bool ok[PART_E];
        uint32_t mine = 0;
#pragma unroll
        for (int k = 0; k < PART_E; k++) {
            const uint32_t i = (uint32_t)t + (uint32_t)k * JB;
            ok[k] = i < n && slot_of[k] != PART_NONE && fold_one(ent[k], i, slot_of[k], keep[k]);
            mine += ok[k] ? 1u : 0u;
        }
        const uint32_t inc = wave_inclusive<uint32_t>(mine, lane);
        if (lane == 63) wtot[wave] = inc;
        __syncthreads();
        uint32_t pre = inc - mine, all = 0;
#pragma unroll
        for (int w = 0; w < JB / 64; w++) { if (w < wave) pre += wtot[w]; all += wtot[w]; }
        if (t == 0) bucket_cnt[blockIdx.x] = all;
        uint32_t w = pre;
#pragma unroll
        for (int k = 0; k < PART_E; k++) if (ok[k]) emit(keep[k], w++);
    } else {
        uint32_t run = 0;                                        // survivors of the rounds so far (uniform)
        for (uint32_t i0 = 0; i0 < n; i0 += JB) {
            const uint32_t i = i0 + t;
            gci_ivl keep;
            bool ok = false;
            if (i < n) {
                const PartEntry e = load_entry(in + i);
                const uint32_t slot = part_probe<false>(meta, S, A.slot_shift, e, i, in, F, exact, used, &s_used);
                ok = slot != PART_NONE && fold_one(e, i, slot, keep);
            }
            const unsigned long long bal = __ballot(ok);
            if (lane == 0) wtot[wave] = (uint32_t)__builtin_popcountll(bal);
            __syncthreads();
            uint32_t pre = (uint32_t)__builtin_popcountll(bal & ((1ull << lane) - 1ull)), all = 0;
#pragma unroll
            for (int w = 0; w < JB / 64; w++) { if (w < wave) pre += wtot[w]; all += wtot[w]; }
            if (ok) emit(keep, run + pre);
            run += all;
            __syncthreads();                                     // wtot is rewritten by the next round
        }
        if (t == 0) bucket_cnt[blockIdx.x] = run;
    }
}

// The dense output of the bucket join: bucket b's bucket_cnt[b] survivors from sparse[bucket_lo[b] ...] to out[off[b] ...]
// (off = exclusive scan of the counts, its total = the number of intervals).  One wave per bucket.
__global__ __launch_bounds__(BLOCK) void k_join_gather(const gci_ivl* __restrict__ sparse, const uint32_t* __restrict__ bucket_cnt,
                                                       const uint32_t* __restrict__ bucket_lo, const uint32_t* __restrict__ off,
                                                       uint32_t n_buckets, gci_ivl* __restrict__ out, uint32_t cap, uint32_t* __restrict__ n_out)
{
    const uint32_t b = blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (blockIdx.x == 0 && threadIdx.x == 0) *n_out = off[n_buckets];
    if (b >= n_buckets) return;
    const uint32_t c = bucket_cnt[b];
    if (!c) return;
    const gci_ivl* __restrict__ src = sparse + bucket_lo[b];
    const uint32_t o = off[b];
    for (uint32_t r = lane; r < c; r += 64) if (o + r < cap) out[o + r] = src[r];
}

static int name_join_partitioned(gci_ctx* ctx, const JoinFiles& F, uint64_t total, double ovlp_percent,
                                 const int32_t* d_contig_map, gci_ivl* d_out, uint32_t cap, uint32_t* d_n_out,
                                 uint64_t* d_status, const CountArgs& cnt, bool* done)
{
    *done = false;
    // slots per bucket table: what fits 64 KB of LDS with one order key and one 16-byte payload per file (two 512-thread
    // workgroups per CU, which is what their registers allow anyway)
    uint32_t S = 1024;
    { const char* e = getenv("GCI_JOIN_SLOTS"); if (e && atoi(e) >= 128 && atoi(e) <= 1024 && !(atoi(e) & (atoi(e) - 1))) S = (uint32_t)atoi(e); }   // (A/B)
    while (S > 128 && (size_t)S * (16 + 24 * (size_t)F.n) > 65536) S >>= 1;
    // buckets: a load of at most 0.63 in the worst case (every name distinct), 0.15 - 0.3 for two files of the same reads
    uint64_t nb = 256;
    while (nb * S * 5 < total * 8) nb <<= 1;
    if (nb > 65536) return GCI_OK;                                            // beyond two 8-bit levels: classic path
    int bits = 0;
    while ((1ull << bits) < nb) bits++;
    const int b1 = (bits + 1) / 2, b2 = bits - b1;
    const uint32_t B1 = 1u << b1, B2 = 1u << b2;
    int s_bits = 0;
    while ((1u << s_bits) < S) s_bits++;
    const int shift1 = 48 - b1, shift2 = 48 - b1 - b2, slot_shift = shift2 - s_bits;
    PartFiles P;
    uint32_t C1 = 0;
    for (int f = 0; f < F.n; f++) { P.chunk_first[f] = C1; C1 += (F.f[f].n_recs + PART_CHUNK - 1) / PART_CHUNK; }
    for (int f = F.n; f <= GCI_MAX_JOIN_FILES; f++) P.chunk_first[f] = C1;
    const uint32_t C2 = (uint32_t)(total / PART_CHUNK) + B1 + 1;
    const size_t n1 = (size_t)B1 * C1, n2 = (size_t)C2 * B2;
    GCI_TRY(gci_ensure(ctx, ctx->part_a, total * sizeof(PartEntry) + 16));
    GCI_TRY(gci_ensure(ctx, ctx->part_b, total * sizeof(PartEntry) + 16));
    GCI_TRY(gci_ensure(ctx, ctx->part_hist, (n1 + 1 + n2 + 1 + 2 * (size_t)B1 + 2) * 4 + 64));
    GCI_TRY(gci_ensure(ctx, ctx->part_blk, ((n1 > n2 ? n1 : n2) / TILE + 2) * 4));
    uint32_t* hist1 = (uint32_t*)ctx->part_hist.p;
    uint32_t* hist2 = hist1 + n1 + 1;
    uint32_t* seg = hist2 + n2 + 1;
    PartEntry* pa = (PartEntry*)ctx->part_a.p;
    PartEntry* pb = (PartEntry*)ctx->part_b.p;
    HIPCHK(hipMemsetAsync(d_status, 0xFF, 8, ctx->stream));
    {
        ProfScope _ps(ctx, GCI_PROF_PARTITION);
        hipLaunchKernelGGL(k_part1_hist, dim3(C1), dim3(BLOCK), 0, ctx->stream, F, P, shift1, B1, C1, hist1, d_n_out,
                           (unsigned long long*)d_status);
        LAUNCHCHK("k_part1_hist");
        int r = device_exclusive_scan<uint32_t, uint32_t>(ctx, hist1, hist1, (uint32_t*)ctx->part_blk.p, (int64_t)n1, true);
        if (r) return r;
        hipLaunchKernelGGL(k_part1_scatter, dim3(C1), dim3(BLOCK), 0, ctx->stream, F, P, shift1, B1, C1, (const uint32_t*)hist1, pa,
                           (unsigned long long*)d_status);
        LAUNCHCHK("k_part1_scatter");
        hipLaunchKernelGGL(k_part_mid, dim3(1), dim3(BLOCK), 0, ctx->stream, (const uint32_t*)hist1, B1, C1, seg);
        LAUNCHCHK("k_part_mid");
        hipLaunchKernelGGL(k_part2_hist, dim3(C2), dim3(BLOCK), 0, ctx->stream, (const PartEntry*)pa, (const uint32_t*)seg, B1,
                           shift2, B2, hist2);
        LAUNCHCHK("k_part2_hist");
        r = device_exclusive_scan<uint32_t, uint32_t>(ctx, hist2, hist2, (uint32_t*)ctx->part_blk.p, (int64_t)n2, true);
        if (r) return r;
        hipLaunchKernelGGL(k_part2_scatter, dim3(C2), dim3(BLOCK), 0, ctx->stream, (const PartEntry*)pa, (const uint32_t*)seg, B1,
                           shift2, B2, (const uint32_t*)hist2, pb);
        LAUNCHCHK("k_part2_scatter");
    }
    {
        ProfScope _ps(ctx, GCI_PROF_JOIN_PART);
        PartJoinArgs A;
        A.in = pb; A.seg = seg; A.off2 = hist2; A.n_seg = B1; A.n_bins2 = B2; A.slots = S; A.slot_shift = (uint32_t)slot_shift;
        { const char* nv = getenv("GCI_JOIN_NOVERIFY"); A.no_verify = nv && nv[0] == '1' ? 1u : 0u; }
        const size_t lds = (size_t)S * (16 + 24 * (size_t)F.n) + (size_t)S * 2;    // + the list of used slots
        const dim3 grid(B1 * B2), block(JB);
        const uint32_t NB = B1 * B2;
        GCI_TRY(gci_ensure(ctx, ctx->join_bucket, ((size_t)3 * NB + 2) * 4 + 64));
        uint32_t* bcnt = (uint32_t*)ctx->join_bucket.p;
        uint32_t* blo = bcnt + NB;
        uint32_t* boff = blo + NB;                                                 // NB + 1 entries
        gci_ivl* sparse = (gci_ivl*)pa;                                            // the level-1 copy of the entries is dead by now
        static_assert(sizeof(gci_ivl) <= sizeof(PartEntry), "the sparse intervals fit the partition buffer");
        const void* kfn = F.n == 1 ? (const void*)k_join_part<1> : F.n == 2 ? (const void*)k_join_part<2> : F.n == 3 ? (const void*)k_join_part<3>
                        : F.n == 4 ? (const void*)k_join_part<4> : (const void*)k_join_part<0>;
        HIPCHK(hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        switch (F.n) {
        case 1: hipLaunchKernelGGL((k_join_part<1>), grid, block, lds, ctx->stream, F, A, ovlp_percent, d_contig_map, sparse, bcnt, blo, (unsigned long long*)d_status, cnt); break;
        case 2: hipLaunchKernelGGL((k_join_part<2>), grid, block, lds, ctx->stream, F, A, ovlp_percent, d_contig_map, sparse, bcnt, blo, (unsigned long long*)d_status, cnt); break;
        case 3: hipLaunchKernelGGL((k_join_part<3>), grid, block, lds, ctx->stream, F, A, ovlp_percent, d_contig_map, sparse, bcnt, blo, (unsigned long long*)d_status, cnt); break;
        case 4: hipLaunchKernelGGL((k_join_part<4>), grid, block, lds, ctx->stream, F, A, ovlp_percent, d_contig_map, sparse, bcnt, blo, (unsigned long long*)d_status, cnt); break;
        default: hipLaunchKernelGGL((k_join_part<0>), grid, block, lds, ctx->stream, F, A, ovlp_percent, d_contig_map, sparse, bcnt, blo, (unsigned long long*)d_status, cnt); break;
        }
        LAUNCHCHK("k_join_part");
        int r2 = device_exclusive_scan<uint32_t, uint32_t>(ctx, bcnt, boff, (uint32_t*)ctx->part_blk.p, (int64_t)NB, true);
        if (r2) return r2;
        hipLaunchKernelGGL(k_join_gather, dim3((NB + BLOCK / 64 - 1) / (BLOCK / 64)), dim3(BLOCK), 0, ctx->stream, (const gci_ivl*)sparse,
                           (const uint32_t*)bcnt, (const uint32_t*)blo, (const uint32_t*)boff, NB, d_out, cap, d_n_out);
        LAUNCHCHK("k_join_gather");
    }
    *done = true;
    return GCI_OK;
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
    // genome scale: radix partition + per-bucket tables in LDS (GCI_JOIN=classic|partition overrides the size rule)
    {
        const bool force_part = ctx->join_mode == 2, force_classic = ctx->join_mode == 1;
        if (!force_classic && (force_part || total >= (1ull << 20)) && total > 0) {
            bool done = false;
            const int st = name_join_partitioned(ctx, F, total, ovlp_percent, d_contig_map, d_out, cap, d_n_out, d_status, cnt, &done);
            if (st != GCI_OK || done) return st;
        }
    }
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

// 0: by size (the radix-partitioned join from 2^20 records up), 1: always the classic table, 2: always partitioned.
// A context starts with what GCI_JOIN=classic|partition says (default 0).
extern "C" int gci_join_mode(gci_ctx* ctx, int mode)
{
    if (!ctx || mode < 0 || mode > 2) return GCI_E_INVALID;
    ctx->join_mode = mode;
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
    // large inputs: the build buckets its events by radix partition and counts on the way (gci_depth_build_begin)
    uint64_t total = 0;
    for (int f = 0; f < n_files; f++) total += h_files ? h_files[f].n_recs : 0;
    if (gci_evp_wanted(ctx, total)) {
        if (ctx->cd_state != 0) HIPCHK(hipMemsetAsync(ctx->tile_cd.p, 0, (size_t)(ctx->n_tiles + 1) * 8, ctx->stream));
        CountArgs none;
        memset(&none, 0, sizeof none);
        const int st = name_join_impl(ctx, h_files, n_files, ovlp_percent, d_contig_map, d_out, cap, d_n_out, d_status, none);
        if (st == GCI_OK) { ctx->cd_state = 1; ctx->counted_flank = flank; ctx->count_deferred = true; }
        return st;
    }
    if (ctx->n_tiles && ctx->cd_state != 0)
        HIPCHK(hipMemsetAsync(ctx->tile_cd.p, 0, (size_t)(ctx->n_tiles + 1) * 8, ctx->stream));
    ctx->cd_state = 2;
    ctx->count_deferred = false;
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
