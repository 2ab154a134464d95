// k_pages.hip -- N1, last step of the ingestion: the records of an inflated BAM stream (or of a heads stream) laid out
// as RECORD PAGES, the input format of the paged record filter (k_filter_pages.hip, gci_bam_filter_pages).
//
// Why a format of our own.  read_sam (/root/reference/GCI.py:146-169) never looks at SEQ / QUAL -- 98 % of a HiFi
// record -- and what it does look at sits at byte-granular offsets inside a stream whose records are found through an
// offset table: the filter over that stream (k_filter.hip) spends its time on three dependent round trips to memory
// (offset -> head -> CIGAR tail / aux) and on re-aligning every dword in registers.  Here the bytes the filter reads are
// copied ONCE, while the inflated stream is walked anyway, into fixed-size pages:
//
//   buffer = [page 0] ... [page n_pages - 1] [blob] [16 zero bytes]
//   page (page_bytes, a multiple of 4096):
//     +0  u32 n_recs | u32 first_rec (index of its first record in the call) | u32 used_bytes | u32 magic "GCP1"
//     +16 u16 dir[n_recs]: start of record j / 16;  records from 16 + align16(2 n_recs) on, each 16-byte aligned
//   record (size a multiple of 16, at most GCI_PAGE_MAX_REC = 1024 bytes):
//     +0  u32 size                       (BAM: block_size)
//     +4  refID, pos, l_read_name, mapq  (as in BAM)
//     +14 u16 kind                       (BAM: bin)       1 = CIGAR in the blob, 2 = whole record in the blob, 4 = malformed
//     +16 n_cigar_op, flag, l_seq        (as in BAM)
//     +24 u32 aux_len                    (BAM: next_refID)
//     +28 u64 blob offset, from the buffer start   (BAM: next_pos, tlen)
//     +36 read_name, zero padded so that the CIGAR starts at align16(36 + l_read_name); the CIGAR zero padded to 16
//         bytes (a zero word is "0M": it adds nothing to any total); the aux bytes behind it; zero padded to 16.
//   A record that does not fit 1024 bytes keeps its CIGAR words in the blob (kind 1: ONT reads, 10^3 - 10^5 operations,
//   summed chunk by chunk by k_cigar_chunks anyway; 16 bytes with the first operation stand in for them); one that still does not fit (a CG:B,I tag, long Z tags) leaves its
//   48-byte core in the page and its bytes -- the heads form: the record without SEQ / QUAL -- in the blob (kind 2).
//   No record straddles a page, nothing in a page needs the offset table, every CIGAR is 16-byte aligned: the filter
//   loads a page with one round of coalesced 16-byte loads and parses it out of LDS.
//
// Page assignment without a serial packing pass: with cost_i = size_i + 2 (the directory entry) and S = exclusive scan
// of the costs, record i goes to page floor(S_i / Q), Q = page_bytes - 1024 - 48 -- whatever starts inside a quantum
// fits its page, because only the last record can reach beyond it, by less than 1026 bytes.
#include "gci_ctx.hpp"

#define PG_MAX_REC GCI_PAGE_MAX_REC
#define PG_MAGIC 0x31504347u
#define PG_EXT 1u
#define PG_OVERSIZE 2u
#define PG_MALFORMED 4u

__device__ __forceinline__ uint32_t a16(uint32_t x) { return (x + 15u) & ~15u; }

__device__ __forceinline__ uint32_t pg_rd16(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }
__device__ __forceinline__ uint32_t pg_rd32(const uint8_t* p) { return pg_rd16(p) | (pg_rd16(p + 2) << 16); }

struct PgRec {
    uint32_t kind, size, blob;          // blob: bytes of the blob this record takes (a multiple of 16)
    uint32_t lrn, n_cig, aux_len;
    uint64_t aux_off;                   // in the source stream
    bool short_core;                    // fewer than 36 bytes of the record lie inside the stream
};

// What a record takes in the page format; the conditions under which the stream filter reports GCI_E_MALFORMED
// (k_filter.hip) make it a malformed stub here.
__device__ __forceinline__ PgRec pg_measure(const uint8_t* __restrict__ bam, uint64_t n_bytes, uint64_t off, bool has_seq)
{
    PgRec r;
    r.kind = PG_MALFORMED; r.size = 48; r.blob = 0; r.lrn = r.n_cig = r.aux_len = 0; r.aux_off = 0; r.short_core = false;
    if (off + 36 > n_bytes) { r.short_core = true; return r; }
    const uint8_t* p = bam + off;
    const int32_t block_size = (int32_t)pg_rd32(p);
    const uint32_t lrn = p[12], n_cig = pg_rd16(p + 16);
    const int32_t l_seq = (int32_t)pg_rd32(p + 20);
    const uint64_t rec_end = off + 4 + (uint64_t)(uint32_t)block_size;
    const uint64_t aux_off = off + 36 + lrn + 4ull * n_cig + (has_seq ? (((uint64_t)(uint32_t)l_seq + 1) >> 1) + (uint64_t)(uint32_t)l_seq : 0ull);
    if (block_size < 32 || rec_end > n_bytes || l_seq < 0 || aux_off > rec_end) return r;
    r.lrn = lrn; r.n_cig = n_cig; r.aux_off = aux_off;
    const uint64_t aux_len = rec_end - aux_off;
    const uint32_t cig_at = a16(36 + lrn);
    if (aux_len <= PG_MAX_REC && 4ull * n_cig <= PG_MAX_REC && a16(cig_at + a16(4 * n_cig) + (uint32_t)aux_len) <= PG_MAX_REC) {
        r.kind = 0; r.size = a16(cig_at + a16(4 * n_cig) + (uint32_t)aux_len); r.aux_len = (uint32_t)aux_len;
    } else if (aux_len <= PG_MAX_REC && a16(cig_at + 16 + (uint32_t)aux_len) <= PG_MAX_REC) {
        r.kind = PG_EXT; r.size = a16(cig_at + 16 + (uint32_t)aux_len); r.blob = a16(4 * n_cig); r.aux_len = (uint32_t)aux_len;
    } else {
        r.kind = PG_OVERSIZE; r.size = 48; r.aux_len = (uint32_t)aux_len;
        r.blob = (uint32_t)((36 + lrn + 4ull * n_cig + aux_len + 15ull) & ~15ull);
    }
    return r;
}

// page_first[k] = first record whose cost offset is >= k * Q (k = n_pages: n_rec)
__global__ __launch_bounds__(BLOCK) void k_pg_first(const unsigned long long* __restrict__ S, uint32_t n_rec, uint32_t n_pages, uint32_t Q,
                                                    uint32_t* __restrict__ page_first)
{
    const uint32_t k = blockIdx.x * BLOCK + threadIdx.x;
    if (k > n_pages) return;
    if (k == n_pages) { page_first[k] = n_rec; return; }
    const unsigned long long want = (unsigned long long)k * Q;
    uint32_t lo = 0, hi = n_rec;                                 // first i in [0, n_rec] with S[i] >= want
    while (lo < hi) { const uint32_t mid = lo + ((hi - lo) >> 1); if (S[mid] >= want) hi = mid; else lo = mid + 1; }
    page_first[k] = lo;
}

// the naturally aligned dword at address q of the stream [bam, end); bytes past the end read as zero
__device__ __forceinline__ uint32_t pg_ldw(const uint8_t* q, const uint8_t* end)
{
    if (q + 4 <= end) return *reinterpret_cast<const uint32_t*>(q);
    uint32_t w = 0;
    for (int b = 0; b < 4; b++) if (q + b < end) w |= (uint32_t)q[b] << (8 * b);
    return w;
}

// dword d of the `len` bytes that start at stream offset `src` (any alignment); bytes beyond len read as zero.  Two naturally
// aligned loads re-aligned in registers (an unaligned vector load from global memory is ~100x slower on gfx950).
__device__ __forceinline__ uint32_t pg_src_dword(const uint8_t* __restrict__ bam, uint64_t n_bytes, uint64_t src, uint32_t len, uint32_t d)
{
    const uint8_t* p = bam + src + 4ull * d;
    const uint32_t sh = (uint32_t)((uintptr_t)p & 3u);
    const uint8_t* end = bam + n_bytes;
    const uint32_t lo = pg_ldw(p - sh, end);
    const uint32_t hi = sh ? pg_ldw(p - sh + 4, end) : 0u;
    uint32_t w = __builtin_amdgcn_alignbyte(hi, lo, sh);
    const uint32_t left = len - 4u * d;
    if (left < 4u) w &= (1u << (8u * left)) - 1u;
    return w;
}

#ifndef PG_LANES
#define PG_LANES 8                     // lanes that copy one record into the page
#endif
struct PgArgs {
    const uint8_t* bam; uint64_t n_bytes; const uint64_t* rec_off; uint32_t n_rec; int has_seq;
    const unsigned long long* S; const unsigned long long* B; const uint32_t* page_first;
    uint32_t page_bytes; uint64_t blob_off; uint8_t* out;
};

// What the copy phase needs of a record, left in LDS by the thread that measured it: where its three runs of bytes lie in the stream
// (name + CIGAR directly behind the core, aux behind SEQ / QUAL) and where they go in the page.
struct PgMeta {
    uint64_t off, aux_off;
    uint16_t n_cig, aux_len, at;       // (aux_len <= 1024 for the kinds that are copied, n_cig is 16 bits in BAM, at < page_bytes <= 32768)
    uint8_t lrn, kind;
};
static_assert(sizeof(PgMeta) == 24, "PgMeta");
#define PG_META_MAX 128                // records measured at a time (a page of 24 KiB holds ~58 HiFi records; tiny records: several passes)

// pg_measure() from naturally aligned dword loads (the byte loads of the generic form are a dozen memory instructions per record):
// the 36 core bytes as 10 aligned dwords re-aligned in registers.
__device__ __forceinline__ PgRec pg_measure_fast(const uint8_t* __restrict__ bam, uint64_t n_bytes, uint64_t off, bool has_seq, uint32_t core[9])
{
    PgRec r;
    r.kind = PG_MALFORMED; r.size = 48; r.blob = 0; r.lrn = r.n_cig = r.aux_len = 0; r.aux_off = 0; r.short_core = false;
#pragma unroll
    for (int d = 0; d < 9; d++) core[d] = 0u;
    if (off + 36 > n_bytes) { r.short_core = true; return r; }
    const uint8_t* p = bam + off;
    const uint32_t sh = (uint32_t)((uintptr_t)p & 3u);
    const uint8_t* end = bam + n_bytes;
    uint32_t w[10];
#pragma unroll
    for (int d = 0; d < 10; d++) w[d] = (d < 9 || sh) ? pg_ldw(p - sh + 4 * d, end) : 0u;
#pragma unroll
    for (int d = 0; d < 9; d++) core[d] = __builtin_amdgcn_alignbyte(w[d + 1], w[d], sh);
    const int32_t block_size = (int32_t)core[0];
    const uint32_t lrn = core[3] & 0xFFu, n_cig = core[4] & 0xFFFFu;
    const int32_t l_seq = (int32_t)core[5];
    const uint64_t rec_end = off + 4 + (uint64_t)(uint32_t)block_size;
    const uint64_t aux_off = off + 36 + lrn + 4ull * n_cig + (has_seq ? (((uint64_t)(uint32_t)l_seq + 1) >> 1) + (uint64_t)(uint32_t)l_seq : 0ull);
    if (block_size < 32 || rec_end > n_bytes || l_seq < 0 || aux_off > rec_end) return r;
    r.lrn = lrn; r.n_cig = n_cig; r.aux_off = aux_off;
    const uint64_t aux_len = rec_end - aux_off;
    const uint32_t cig_at = a16(36 + lrn);
    if (aux_len <= PG_MAX_REC && 4ull * n_cig <= PG_MAX_REC && a16(cig_at + a16(4 * n_cig) + (uint32_t)aux_len) <= PG_MAX_REC) {
        r.kind = 0; r.size = a16(cig_at + a16(4 * n_cig) + (uint32_t)aux_len); r.aux_len = (uint32_t)aux_len;
    } else if (aux_len <= PG_MAX_REC && a16(cig_at + 16 + (uint32_t)aux_len) <= PG_MAX_REC) {
        r.kind = PG_EXT; r.size = a16(cig_at + 16 + (uint32_t)aux_len); r.blob = a16(4 * n_cig); r.aux_len = (uint32_t)aux_len;
    } else {
        r.kind = PG_OVERSIZE; r.size = 48; r.aux_len = (uint32_t)aux_len;
        r.blob = (uint32_t)((36 + lrn + 4ull * n_cig + aux_len + 15ull) & ~15ull);
    }
    return r;
}

__global__ __launch_bounds__(BLOCK) void k_pg_measure(const uint8_t* __restrict__ bam, uint64_t n_bytes, const uint64_t* __restrict__ rec_off,
                                                      uint32_t n_rec, int has_seq, uint32_t* __restrict__ cost, uint32_t* __restrict__ blob)
{
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n_rec) return;
    // (byte loads: pg_measure_fast's ten aligned dwords were measured SLOWER here -- 0.181 against 0.155 ms per quarter genome for the size
    //  pass: one lane per record, every lane its own cache line, and the dozen byte loads of a lane hit that one line)
    const PgRec r = pg_measure(bam, n_bytes, rec_off[i], has_seq != 0);
    cost[i] = r.size + 2u;
    blob[i] = r.blob;
}

// A record's three runs of bytes -- name, CIGAR (kind 0 only), aux -- into the page in LDS, by the PG_LANES lanes of a group.  The
// kernel is bound by the LATENCY of these loads (tools/hwtests/pages_ab.py: 0.60 of its 0.80 ms per quarter genome were this copy,
// and neither fewer load instructions -- one aligned dword per lane, the one behind it from the neighbour lane -- nor 4 / 16 lanes
// per record changed that: every step of the loop waited for its own loads before the next step's were issued).  So the three runs are
// ONE flat sequence of output dwords, and a lane asks for PG_UNROLL of them -- two aligned dwords each: the one that holds the first
// byte and the one behind it -- BEFORE it uses any: sixteen loads in flight per lane where there were two, a typical HiFi record
// (9 + 70 + 10 dwords) in two steps instead of thirteen (0.74 -> 0.44 ms).
#ifndef PG_UNROLL
#define PG_UNROLL 8
#endif
struct PgRun { uint64_t src; uint32_t len, n, at; };     // stream offset, bytes, dwords, offset in the record's page image

__device__ __forceinline__ void pg_copy_runs(const uint8_t* __restrict__ bam, const uint8_t* end, const PgRun r0, const PgRun r1, const PgRun r2,
                                             uint8_t* dst, uint32_t gl)
{
    const uint32_t total = r0.n + r1.n + r2.n;
    for (uint32_t f0 = 0; f0 < total; f0 += PG_UNROLL * PG_LANES) {
        uint32_t lo[PG_UNROLL], hi[PG_UNROLL], sh[PG_UNROLL], left[PG_UNROLL], where[PG_UNROLL];
#pragma unroll
        for (int u = 0; u < PG_UNROLL; u++) {
            const uint32_t f = f0 + u * PG_LANES + gl;
            const bool in0 = f < r0.n, in1 = f < r0.n + r1.n;
            const uint64_t src = in0 ? r0.src : in1 ? r1.src : r2.src;
            const uint32_t len = in0 ? r0.len : in1 ? r1.len : r2.len;
            const uint32_t d = in0 ? f : in1 ? f - r0.n : f - r0.n - r1.n;
            const uint32_t at = in0 ? r0.at : in1 ? r1.at : r2.at;
            const uint8_t* p = bam + src + 4ull * d;
            sh[u] = (uint32_t)((uintptr_t)p & 3u);
            left[u] = f < total ? len - 4u * d : 0u;
            where[u] = at + 4u * d;
            lo[u] = f < total ? pg_ldw(p - sh[u], end) : 0u;
            hi[u] = (f < total && sh[u]) ? pg_ldw(p - sh[u] + 4, end) : 0u;
        }
#pragma unroll
        for (int u = 0; u < PG_UNROLL; u++) {
            if (left[u]) {
                uint32_t w = __builtin_amdgcn_alignbyte(hi[u], lo[u], sh[u]);
                if (left[u] < 4u) w &= (1u << (8u * left[u])) - 1u;
                *reinterpret_cast<uint32_t*>(dst + where[u]) = w;
            }
        }
    }
}

#ifndef PG_BLOCK
#define PG_BLOCK BLOCK                 // threads of a k_pg_write workgroup
#endif
__global__ __launch_bounds__(PG_BLOCK) void k_pg_write(const PgArgs A)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t page[];
    // which records of the page leave something in the blob (nearly none of a HiFi file): the pass over the blob below looks
    // at those only -- measuring every record again there, one after the other by the whole workgroup, was 60 dependent trips
    // to memory per page and most of this kernel's time
    __shared__ uint16_t s_blob_rec[GCI_PAGE_MAX_BYTES / 48];
    __shared__ uint32_t s_n_blob;
    __shared__ PgMeta s_meta[PG_META_MAX];
    const uint32_t t = threadIdx.x, k = blockIdx.x;
    const uint32_t P = A.page_bytes;
    for (uint32_t i = t; i < P / 16; i += PG_BLOCK) reinterpret_cast<uint4*>(page)[i] = make_uint4(0, 0, 0, 0);
    if (t == 0) s_n_blob = 0;
    const uint32_t first = A.page_first[k], cnt = A.page_first[k + 1] - first;
    const unsigned long long S0 = cnt ? A.S[first] : 0ull;
    const uint32_t rec0 = 16u + a16(2u * cnt);
    const uint8_t* end = A.bam + A.n_bytes;
    __syncthreads();
    const uint32_t gl = t % PG_LANES, grp = t / PG_LANES;
    uint32_t used = rec0;
#ifdef PG_CUT_RECORDS                  // (measurement builds: zero fill + write-out only)
    for (uint32_t m0 = 0; m0 < 0; m0 += PG_META_MAX) {
#else
    for (uint32_t m0 = 0; m0 < cnt; m0 += PG_META_MAX) {
#endif
        const uint32_t mc = cnt - m0 < PG_META_MAX ? cnt - m0 : PG_META_MAX;
        // ---- phase A: one thread per record measures it (offset -> 10 aligned dwords: the only dependent trips to memory of the
        // page), leaves what the copy needs in LDS and writes the patched 36-byte core and the directory entry
        if (t < mc) {
            const uint32_t j = m0 + t, i = first + j;
            const uint64_t off = A.rec_off[i];
            const unsigned long long Si = A.S[i];
            uint32_t core[9];
            const PgRec r = pg_measure_fast(A.bam, A.n_bytes, off, A.has_seq != 0, core);
            const uint32_t at = rec0 + (uint32_t)(Si - S0) - 2u * j;
            const uint64_t blob_at = r.blob ? A.blob_off + A.B[i] : 0ull;
            if (r.blob) s_blob_rec[atomicAdd(&s_n_blob, 1u)] = (uint16_t)j;
            *reinterpret_cast<uint16_t*>(page + 16 + 2 * j) = (uint16_t)(at >> 4);
            core[0] = r.size;
            core[3] = (core[3] & 0xFFFFu) | (r.kind << 16);
            core[6] = r.aux_len; core[7] = (uint32_t)blob_at; core[8] = (uint32_t)(blob_at >> 32);
            if (r.kind == PG_MALFORMED) core[6] = core[7] = core[8] = 0u;
            uint32_t* dst = reinterpret_cast<uint32_t*>(page + at);
#pragma unroll
            for (int d = 0; d < 9; d++) dst[d] = core[d];
            PgMeta mt;
            mt.off = off; mt.aux_off = r.aux_off; mt.lrn = (uint8_t)r.lrn; mt.n_cig = (uint16_t)r.n_cig;
            mt.aux_len = (uint16_t)(r.aux_len <= PG_MAX_REC ? r.aux_len : 0u);
            mt.at = (uint16_t)at; mt.kind = (uint8_t)r.kind;
            s_meta[t] = mt;
        }
        __syncthreads();
        // ---- phase B: PG_LANES lanes per record copy its name, CIGAR and aux bytes; nothing here waits for anything but its own loads
#ifndef PG_CUT_COPY                    // (measurement builds, tools/hwtests/pages_ab.py: the kernel without its copy phase)
        for (uint32_t j0 = 0; j0 < mc; j0 += PG_BLOCK / PG_LANES) {
            const uint32_t jj = j0 + grp;
            if (jj >= mc) continue;
            const PgMeta mt = s_meta[jj];
            if (mt.kind == PG_MALFORMED || mt.kind == PG_OVERSIZE) continue;
            uint8_t* dst = page + mt.at;
            const uint32_t c = a16(36u + mt.lrn);
            PgRun r0, r1, r2;
            r0.src = mt.off + 36; r0.len = mt.lrn; r0.n = (mt.lrn + 3u) >> 2; r0.at = 36;
            r1.src = mt.off + 36 + mt.lrn; r1.at = c;
            if (mt.kind == 0) { r1.len = 4u * mt.n_cig; r1.n = mt.n_cig; }
            else { r1.len = mt.n_cig ? 4u : 0u; r1.n = mt.n_cig ? 1u : 0u; }       // kind 1: the first operation stays visible in the page
            r2.src = mt.aux_off; r2.len = mt.aux_len; r2.n = (mt.aux_len + 3u) >> 2;
            r2.at = c + (mt.kind == 0 ? a16(4u * mt.n_cig) : 16u);
            pg_copy_runs(A.bam, end, r0, r1, r2, dst, gl);
        }
#endif
        __syncthreads();                                           // (s_meta is reused by the next pass)
    }
    if (cnt) used = rec0 + (uint32_t)(A.S[first + cnt] - S0) - 2u * cnt;
    if (t == 0) {
        uint32_t* h = reinterpret_cast<uint32_t*>(page);
        h[0] = cnt; h[1] = first; h[2] = cnt ? used : 16u; h[3] = PG_MAGIC;
    }
    __syncthreads();
    uint4* g = reinterpret_cast<uint4*>(A.out + (uint64_t)k * P);
    for (uint32_t i = t; i < P / 16; i += PG_BLOCK) g[i] = reinterpret_cast<const uint4*>(page)[i];
    // what the page leaves in the blob: CIGAR words (kind 1) or heads-form records (kind 2), record after record, all threads
    const uint32_t n_blob = s_n_blob;                            // (written before the barrier above)
    for (uint32_t b = 0; b < n_blob; b++) {
        const uint32_t i = first + s_blob_rec[b];
        const uint64_t off = A.rec_off[i];
        const PgRec r = pg_measure(A.bam, A.n_bytes, off, A.has_seq != 0);         // (uniform over the workgroup)
        if (!r.blob) continue;
        uint32_t* o = reinterpret_cast<uint32_t*>(A.out + A.blob_off + A.B[i]);
        if (r.kind == PG_EXT) {
            for (uint32_t d = t; d < r.blob / 4; d += PG_BLOCK)
                o[d] = d < r.n_cig ? pg_src_dword(A.bam, A.n_bytes, off + 36 + r.lrn, 4 * r.n_cig, d) : 0u;
        } else {
            // heads form: core (block_size shortened), name + CIGAR (contiguous in the stream), aux (behind SEQ / QUAL there)
            const uint32_t head = 36 + r.lrn + 4 * r.n_cig, total = head + r.aux_len;
            for (uint32_t d = t; d < r.blob / 4; d += PG_BLOCK) {
                uint32_t w = 0;
                const uint32_t b0 = 4 * d;
                if (b0 + 4 <= head) w = pg_src_dword(A.bam, A.n_bytes, off, head, d);
                else if (b0 >= head) { if (b0 < total) w = pg_src_dword(A.bam, A.n_bytes, r.aux_off + (b0 - head), total - b0, 0); }
                else {                                             // the dword that holds the seam
                    for (uint32_t b = 0; b < 4 && b0 + b < total; b++) {
                        const uint32_t x = b0 + b;
                        w |= (uint32_t)(x < head ? A.bam[off + x] : A.bam[r.aux_off + (x - head)]) << (8 * b);
                    }
                }
                if (d == 0) w = total - 4;
                o[d] = w;
            }
        }
    }
}

extern "C" int gci_bam_pages_size(gci_ctx* ctx, const uint8_t* d_stream, uint64_t n_bytes, const uint64_t* d_rec_off, uint32_t n_rec,
                                  int has_seq, uint32_t page_bytes, uint64_t* h_out)
{
    if (!ctx || !h_out || (n_rec && (!d_stream || !d_rec_off))) return GCI_E_INVALID;
    if (page_bytes < 8192 || page_bytes > GCI_PAGE_MAX_BYTES || (page_bytes & 4095u)) return GCI_E_INVALID;
    h_out[0] = h_out[1] = h_out[2] = 0;
    ctx->pg_n_rec = n_rec; ctx->pg_page_bytes = page_bytes; ctx->pg_n_pages = 0; ctx->pg_blob_off = 0; ctx->pg_blob_bytes = 0;
    if (n_rec == 0) { h_out[1] = 16; return GCI_OK; }
    const uint32_t Q = page_bytes - PG_MAX_REC - 48;
    GCI_TRY(gci_ensure(ctx, ctx->pg_cost, ((size_t)n_rec + 1) * 4 * 2));
    GCI_TRY(gci_ensure(ctx, ctx->pg_scan, ((size_t)n_rec + 1) * 8 * 2));
    GCI_TRY(gci_ensure(ctx, ctx->blk_u64, (size_t)(n_rec / TILE + 2) * 8));
    uint32_t* cost = (uint32_t*)ctx->pg_cost.p;
    uint32_t* blob = cost + n_rec + 1;
    unsigned long long* S = (unsigned long long*)ctx->pg_scan.p;
    unsigned long long* B = S + n_rec + 1;
    int r;
    {
        ProfScope _ps(ctx, GCI_PROF_PAGES_SIZE);
        hipLaunchKernelGGL(k_pg_measure, dim3((n_rec + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, ctx->stream, d_stream, n_bytes, d_rec_off, n_rec,
                           has_seq, cost, blob);
        LAUNCHCHK("k_pg_measure");
        r = device_exclusive_scan<uint32_t, unsigned long long>(ctx, cost, S, (unsigned long long*)ctx->blk_u64.p, n_rec, true);
        if (r) return r;
        r = device_exclusive_scan<uint32_t, unsigned long long>(ctx, blob, B, (unsigned long long*)ctx->blk_u64.p, n_rec, true);
        if (r) return r;
    }
    unsigned long long tail[2], b_total;
    HIPCHK(hipMemcpyAsync(tail, S + n_rec - 1, 16, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(&b_total, B + n_rec, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    const uint64_t n_pages = tail[0] / Q + 1;
    if (n_pages > 0xFFFFFFF0ull) return GCI_E_INVALID;
    GCI_TRY(gci_ensure(ctx, ctx->pg_first, (size_t)(n_pages + 1) * 4));
    hipLaunchKernelGGL(k_pg_first, dim3((uint32_t)((n_pages + 1 + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, ctx->stream, S, n_rec,
                       (uint32_t)n_pages, Q, (uint32_t*)ctx->pg_first.p);
    LAUNCHCHK("k_pg_first");
    ctx->pg_n_pages = (uint32_t)n_pages;
    ctx->pg_blob_off = n_pages * page_bytes;
    ctx->pg_blob_bytes = b_total;
    h_out[0] = n_pages;
    h_out[1] = n_pages * page_bytes + b_total + 16;
    h_out[2] = ctx->pg_blob_off;
    return GCI_OK;
}

extern "C" int gci_bam_pages_write(gci_ctx* ctx, const uint8_t* d_stream, uint64_t n_bytes, const uint64_t* d_rec_off, uint32_t n_rec,
                                   int has_seq, uint8_t* d_out, uint64_t cap)
{
    if (!ctx || !d_out || (n_rec && (!d_stream || !d_rec_off))) return GCI_E_INVALID;
    if (n_rec != ctx->pg_n_rec) return GCI_E_INVALID;                       // not the input gci_bam_pages_size measured
    if (n_rec == 0) { HIPCHK(hipMemsetAsync(d_out, 0, cap < 16 ? cap : 16, ctx->stream)); return GCI_OK; }
    unsigned long long* S = (unsigned long long*)ctx->pg_scan.p;
    unsigned long long* B = S + n_rec + 1;
    // (size of the blob: read back by the size call; the caller allocated what that call said)
    PgArgs A;
    A.bam = d_stream; A.n_bytes = n_bytes; A.rec_off = d_rec_off; A.n_rec = n_rec; A.has_seq = has_seq;
    A.S = S; A.B = B; A.page_first = (const uint32_t*)ctx->pg_first.p; A.page_bytes = ctx->pg_page_bytes;
    A.blob_off = ctx->pg_blob_off; A.out = d_out;
    const uint64_t need = ctx->pg_blob_off + ctx->pg_blob_bytes + 16;        // = h_out[1] of the size call
    if (cap < need) return GCI_E_CAPACITY;
    {
        ProfScope _ps(ctx, GCI_PROF_PAGES_WRITE);
        hipLaunchKernelGGL(k_pg_write, dim3(ctx->pg_n_pages), dim3(PG_BLOCK), ctx->pg_page_bytes, ctx->stream, A);
        LAUNCHCHK("k_pg_write");
    }
    // the 16 zero bytes directly behind the blob (k_cigar_chunks fetches whole 16-byte pieces)
    HIPCHK(hipMemsetAsync(d_out + need - 16, 0, 16, ctx->stream));
    return GCI_OK;
}
