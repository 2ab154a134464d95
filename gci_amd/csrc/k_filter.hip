// k_filter.hip -- K1: BAM record filter (read_sam, /root/reference/GCI.py:146-169).
//
// The decode is bound by vector-memory address processing, not by bytes: a wave that parses 16 scattered
// records with 4-byte loads issues ~47 load instructions that each touch 16 different cache lines.  So the
// record is STAGED: four lanes per record pull its first 128 bytes (core, name, first CIGAR words) and the
// first 128 bytes of its aux block into LDS with 16-byte loads (four load instructions per wave), and the
// parsing -- core fields, NUL search, name hash, the aux walk for NM / CG, the first CIGAR words -- runs out
// of LDS (gfx950 serves unaligned ds_read_b32 / b64; checked by tools/hwtests/lds_unaligned.hip).
//
//   fast path (k_bam_filter), 16 records per wave:
//     flag / MAPQ tests (GCI.py:152-156) -> query_name (first NUL) + 64-bit name hash -> first NM tag
//     (bam_aux_get semantics) -> CIGAR base totals (get_cigar_stats, GCI.py:157-162) in four per-lane register
//     sums, added over the four lanes of the record -> the two IEEE f64 divisions of GCI.py:165 -> 32-byte
//     compact record.
//   slow path (slow_record, at the end of the same kernel), one wave per record, everything from global memory with
//     naturally aligned loads only: htslib's CG:B,I restore (CIGARs of more than 65535 operations),
//     records whose NM tag does not show up in the staged part of the aux block, names longer than the
//     staged head.  Same decisions, same status codes.
// SEQ and QUAL are skipped by pointer arithmetic and never touched.
#include "gci_ctx.hpp"
#include <stdlib.h>

#ifndef G
#define G 4                      // lanes per record on the fast path (2 or 4)
#endif
#define NSLOT 10                 // op codes 0..8 (M I D N S H P = X) + one slot for everything else
#define LONG_OPS 512u
#ifndef HEAD
#define HEAD 112                 // staged bytes from the record start: core, name (up to 72 bytes on the fast path) and the first
                                 // CIGAR words -- with the 16 bytes of alignment slack eight 16-byte pieces, two per lane.  224
                                 // staged more of the CIGAR and was 8 % slower (more staging loads per wave); 352: +41 %
#endif
#ifndef AUXB
#define AUXB 112                 // staged bytes from the aux start
#endif
#define HEADP (HEAD + 16)        // staged from the 16-byte boundary below the record start
#define AUXP (AUXB + 16)
#define ROW (HEADP + AUXP)
#ifndef KB
#define KB 128                    // threads per workgroup of the fast path (KB / G records)
#endif
#ifndef NUP
#define NUP 8                     // 16-byte pieces of the CIGAR tail a lane requests together: a HiFi record's share of the
                                  // tail in one round trip (3: +5 % time; 10 spills registers)
#endif
#ifndef K1_WAVES
#define K1_WAVES 6
#endif
// sum / min over the G lanes of a record
#if G == 4
#define GRP_STEPS(op) do { op(1); op(2); } while (0)
#else
#define GRP_STEPS(op) do { op(1); } while (0)
#endif

#define PIECES 8                  // 16-byte pieces per lane of one CIGAR chunk
#define CHUNK_DW (64 * PIECES * 4)   // aligned dwords (= CIGAR ops) per chunk: 2048 ops, 8 KiB

// A parsed record whose CIGAR totals are still to be summed.  Its op array is cut at multiples of CHUNK_DW aligned
// dwords into chunks; k_cigar_chunks gives every chunk to one wave and stores the chunk's class sums, k_cigar_finish
// adds an item's chunk sums and takes the decision.  No atomics, no fences on that path.
struct LongItem {
    uint64_t ops_off;            // byte offset of its CIGAR words in the stream
    uint32_t n_ops;
    uint32_t rec;
    int64_t nm;                  // INT64_MAX: no NM tag, INT64_MIN: NM of a non-integer type
    int32_t pos, contig, l_seq, n_cigar_field;
    uint32_t mapq;
    uint32_t n_chunks;           // 0: refused (queue full), nothing to finish
    uint32_t chunk_base;         // its chunks are queue entries chunk_base .. chunk_base + n_chunks - 1
    uint32_t pad;
};

struct ChunkSums { unsigned long long v[5]; };       // bases per class: M/=/X, I, D, N, S

struct LongQueue {
    uint32_t* next_counters;     // the counters of the NEXT call (the other half of a ping-pong pair): zeroed by this one
    unsigned long long* status_in;    // what k_bam_filter and k_bam_filter_slow report into (k_cigar_chunks publishes it)
    unsigned long long* next_status;  // ... of the next call: set to "no error" by this one
    uint32_t* n_slow;
    unsigned long long* n_long;  // (long items << 32) | chunks: one atomic hands out both
    uint32_t* slow_list;
    LongItem* items;
    unsigned long long* chunks;  // (item << 32) | chunk index; ~0 = hole left by a refused item
    ChunkSums* sums;             // per queue entry
    uint32_t cap_items, cap_chunks;
    uint32_t has_seq;            // 0: heads stream (records without SEQ / QUAL, gci_bam_heads)
};

__device__ __forceinline__ void report(unsigned long long* status, uint32_t rec, int code)
{
    atomicMin(status, ((unsigned long long)rec << 8) | (unsigned long long)(uint8_t)(-code));
}

__device__ __forceinline__ bool has_zero_byte(uint32_t x) { return ((x - 0x01010101u) & ~x & 0x80808080u) != 0; }

// The decision of GCI.py:163-168; fills r and returns a gci_status (GCI_OK also when filtered).  With M = bases under M, =
// and X together, the op totals enter it only as S, den1 = M + I + S, den2 = M + I + D and rlen = M + D + N
// (mm = NM - (I + D), GCI.py:164, appears as M - mm = den2 - NM).
// (double)a / (double)b  <=  c  (le) or  >=  c  (!le), b != 0: the reference's IEEE f64 division and comparison (GCI.py:165),
// bit for bit.  An f64 division is ~40 instructions a wave executes for all its lanes, and almost no record is anywhere near a
// threshold: a single-precision quotient (relative error < 1e-6) decides whenever it is further than 1e-4 from the threshold --
// rounding is monotonic, so the f64 quotient falls on the same side -- and only the records in between take the division.
__device__ __forceinline__ bool ratio_cmp(int64_t a, int64_t b, double c, bool le)
{
    if (((a < 0 ? -a : a) | b) >> 30 == 0) {                     // both convert to f32 through the 32-bit path
        const float x = (float)(int32_t)a * __builtin_amdgcn_rcpf((float)(int32_t)b);
        const float cf = (float)c, m = 1e-4f * fmaxf(1.0f, fabsf(x));
        if (c == c && fabs(c) < 1e30) {                          // (a NaN or absurd threshold: the exact comparison below)
            if (x < cf - m) return le;
            if (x > cf + m) return !le;
        }
    }
    const double q = (double)a / (double)b;
    return le ? q <= c : q >= c;
}

__device__ __forceinline__ int decide4(gci_rec& r, int64_t S, int64_t den1, int64_t den2, int64_t rlen, bool have_nm,
                                       bool nm_bad_type, int64_t NM, int32_t pos, int32_t contig, int32_t l_seq,
                                       uint32_t n_cigar_field, int mapq, int mq_cutoff, double clip_percent,
                                       double iden_percent)
{
    if (!have_nm) return GCI_E_NO_NM;                                              // get_tag('NM'): KeyError
    if (nm_bad_type) return GCI_E_BAD_NM_TYPE;
    if (den1 == 0) return GCI_E_ZERO_DIV;
    // Python's `and` short-circuits: the identity division only runs when the clip test passed
    if (!ratio_cmp(S, den1, clip_percent, true)) return GCI_OK;                    // GCI.py:165
    if (den2 == 0) return GCI_E_ZERO_DIV;
    if (!ratio_cmp(den2 - NM, den2, iden_percent, false)) return GCI_OK;
    if (n_cigar_field == 0) return GCI_E_NO_END;
    r.contig = contig;
    r.start = pos;
    r.end = (int32_t)((int64_t)pos + (rlen > 0 ? rlen : 1));                        // bam_endpos
    r.qlen = l_seq;                                                                // query_length
    r.flags = GCI_REC_PASS | (mapq >= mq_cutoff ? GCI_REC_HQ : 0);                 // GCI.py:166-168
    return GCI_OK;
}

__device__ __forceinline__ int decide(gci_rec& r, int64_t M, int64_t I, int64_t D, int64_t N, int64_t S, bool have_nm,
                                      bool nm_bad_type, int64_t NM, int32_t pos, int32_t contig, int32_t l_seq,
                                      uint32_t n_cigar_field, int mapq, int mq_cutoff, double clip_percent,
                                      double iden_percent)
{
    return decide4(r, S, M + I + S, M + I + D, M + D + N, have_nm, nm_bad_type, NM, pos, contig, l_seq, n_cigar_field, mapq, mq_cutoff,
                   clip_percent, iden_percent);
}

// integer value of an NM tag whose type byte is at t (value follows); false if the type is not an integer
template <typename P>
__device__ __forceinline__ bool nm_value(P t, int64_t& NM)
{
    uint32_t w = 0;
    switch (t[0]) {
    case 'c': NM = (int8_t)t[1]; return true;
    case 'C': NM = t[1]; return true;
    case 's': w = (uint32_t)t[1] | ((uint32_t)t[2] << 8); NM = (int16_t)w; return true;
    case 'S': w = (uint32_t)t[1] | ((uint32_t)t[2] << 8); NM = w; return true;
    case 'i': w = (uint32_t)t[1] | ((uint32_t)t[2] << 8) | ((uint32_t)t[3] << 16) | ((uint32_t)t[4] << 24); NM = (int32_t)w; return true;
    case 'I': w = (uint32_t)t[1] | ((uint32_t)t[2] << 8) | ((uint32_t)t[3] << 16) | ((uint32_t)t[4] << 24); NM = w; return true;
    default: return false;
    }
}

// the 16 bytes at aligned offset `at`, zero past the end of the stream (the last, partial chunk)
__device__ __forceinline__ uint4 load16_tail(const uint8_t* __restrict__ bam, uint64_t at, uint64_t n_bytes)
{
    uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 16; i++) if (at + i < n_bytes) w[i >> 2] |= (uint32_t)bam[at + i] << (8 * (i & 3));
    return make_uint4(w[0], w[1], w[2], w[3]);
}

// 16 bytes global -> LDS from a 16-byte aligned stream offset; bytes past the end of the stream read as zero
__device__ __forceinline__ void stage16(uint8_t* dst, const uint8_t* __restrict__ bam, uint64_t at, uint64_t n_bytes)
{
    uint4 v = make_uint4(0, 0, 0, 0);
    if (at + 16 <= n_bytes) v = *reinterpret_cast<const uint4*>(bam + at);
    else if (at < n_bytes) {
        v = load16_tail(bam, at, n_bytes);
    }
    *reinterpret_cast<uint4*>(dst) = v;
}

__device__ __forceinline__ uint32_t lds_u32(const uint8_t* p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }

// lanes of one wave exchange data through LDS: the LDS unit executes a wave's operations in order, so only the
// compiler has to be kept from moving accesses across this point
__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Queue a parsed record for chunked CIGAR summing.  Called by W consecutive lanes (li = index among them) that hold
// the same item -- W == G: any subset of a wave's record groups, their requests combined into one atomic; W == 64: a
// whole wave with one record.  False when the queues are full (only possible with overlapping record offsets).
template <int W>
__device__ __forceinline__ bool enqueue_long(const LongQueue& q, LongItem it, int li)
{
    const uint32_t d0 = (uint32_t)(it.ops_off & 15ull) >> 2;       // first op's dword inside its 16-byte piece
    const uint32_t nch = (uint32_t)(((uint64_t)d0 + it.n_ops + CHUNK_DW - 1) / CHUNK_DW);
    uint32_t slot, base;
    if (W == 64) {
        unsigned long long old = 0;
        if (li == 0) old = atomicAdd(q.n_long, (1ull << 32) | nch);
        slot = (uint32_t)__shfl((int)(old >> 32), 0, 64);
        base = (uint32_t)__shfl((int)(uint32_t)old, 0, 64);
    } else {
        const int lane = threadIdx.x & 63;
        const unsigned long long leaders = __ballot(li == 0);      // the first lane of every group that is here
        uint32_t items_below = 0, chunks_below = 0, chunks_all = 0;
        for (unsigned long long m = leaders; m; m &= m - 1ull) {
            const int l = __builtin_ctzll(m);
            const uint32_t v = (uint32_t)__shfl((int)nch, l, 64);
            if (l < (lane & ~(W - 1))) { items_below++; chunks_below += v; }
            chunks_all += v;
        }
        const int first = __builtin_ctzll(leaders);
        unsigned long long old = 0;
        if (lane == first) old = atomicAdd(q.n_long, ((unsigned long long)__builtin_popcountll(leaders) << 32) | chunks_all);
        slot = (uint32_t)__shfl((int)(old >> 32), first, 64) + items_below;
        base = (uint32_t)__shfl((int)(uint32_t)old, first, 64) + chunks_below;
    }
    const bool ok = slot < q.cap_items && (uint64_t)base + nch <= q.cap_chunks;
    if (li == 0 && slot < q.cap_items) {
        it.n_chunks = ok ? nch : 0u; it.chunk_base = base; it.pad = 0;
        q.items[slot] = it;
    }
    for (uint32_t c = li; c < nch; c += W)
        if ((uint64_t)base + c < q.cap_chunks) q.chunks[base + c] = ok ? ((unsigned long long)slot << 32) | c : ~0ull;
    return ok;
}

// ---- long CIGARs: one WAVE per queued record sums the op array (16-byte aligned loads, four chunk pairs in flight)
// and takes the decision; everything else about the record was parsed by the fast path ------------------------------

__device__ __forceinline__ void cigar_totals_wave(const uint8_t* __restrict__ bam, uint64_t n_bytes, uint64_t p0,
                                                  uint32_t n_ops, int lane, int64_t (&tot)[NSLOT])
{
    long long sum[NSLOT];
#pragma unroll
    for (int k = 0; k < NSLOT; k++) sum[k] = 0;
    const uint32_t sh = (uint32_t)(p0 & 15ull), d = sh >> 2, b = sh & 3u;
    const uint64_t a0 = p0 & ~15ull, lim = n_bytes & ~15ull;
    for (uint32_t c0 = lane; 4ull * c0 < n_ops; c0 += 256) {
        uint4 lo[4], hi[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t c = c0 + 64u * u;
            const uint64_t a = a0 + 16ull * c;
            const bool in = 4ull * c < n_ops;
            lo[u] = !in ? make_uint4(0, 0, 0, 0) : a < lim ? *reinterpret_cast<const uint4*>(bam + a) : load16_tail(bam, a, n_bytes);
            hi[u] = !in ? make_uint4(0, 0, 0, 0) : a + 16 < lim ? *reinterpret_cast<const uint4*>(bam + a + 16) : load16_tail(bam, a + 16, n_bytes);
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t c = c0 + 64u * u;
            if (4ull * c >= n_ops) continue;
            const uint32_t w[8] = {lo[u].x, lo[u].y, lo[u].z, lo[u].w, hi[u].x, hi[u].y, hi[u].z, hi[u].w};
            const uint32_t m = min(4u, n_ops - 4u * c);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t x0 = d == 0 ? w[j] : d == 1 ? w[j + 1] : d == 2 ? w[j + 2] : w[j + 3];
                const uint32_t x1 = d == 0 ? w[j + 1] : d == 1 ? w[j + 2] : d == 2 ? w[j + 3] : w[j + 4];
                const uint32_t v = __builtin_amdgcn_alignbyte(x1, x0, b);
                const uint32_t op = v & 0xFu;
                const long long len = (uint32_t)j < m ? (long long)(v >> 4) : 0;
#pragma unroll
                for (int q = 0; q < NSLOT - 1; q++) sum[q] += op == (uint32_t)q ? len : 0;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < NSLOT - 1; k++) tot[k] = wave_sum<long long>(sum[k]);
    tot[NSLOT - 1] = 0;
}

// ---- slow path: one WAVE per queued record, everything read from global memory --------------------------------
// Only naturally aligned global loads: bytes for the scalar fields and the aux walk, 16-byte aligned chunks
// (re-aligned in registers) for the CIGAR.

__device__ __forceinline__ uint32_t rd16b(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }
__device__ __forceinline__ uint32_t rd32b(const uint8_t* p) { return rd16b(p) | (rd16b(p + 2) << 16); }

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
        const uint8_t sub = p[0];
        const int64_t n = rd32b(p + 1);
        const int64_t es = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2
                         : (sub == 'i' || sub == 'I' || sub == 'f') ? 4 : -1;
        return es < 0 ? -1 : 5 + n * es;
    }
    default: return -1;
    }
}

// One record, one wave (all 64 lanes): called by k_bam_filter for the records its staged window could not answer.
// QUEUE_ALL (the paged filter): every CIGAR goes to the chunk queue, so that the rare path's registers (ten 64-bit sums,
// eight 16-byte pieces in flight) do not set the occupancy of a kernel that lives on its waves per SIMD.
template <bool QUEUE_ALL = false>
__device__ __forceinline__ void slow_record(
    const uint8_t* __restrict__ bam, uint64_t n_bytes, const uint64_t off,
    const int32_t* __restrict__ ref_sel, uint32_t rec, int lane, const LongQueue& lq, int mq_cutoff, double clip_percent,
    double iden_percent, uint32_t rec_idx_base, gci_rec* __restrict__ out)
{
    // the fast path has already validated the record's bounds and passed its flag / MAPQ tests;
    // every lane parses the scalar part redundantly (same addresses: one transaction per load)
    const uint8_t* p = bam + off;
    const int32_t block_size = (int32_t)rd32b(p);
    const int32_t ref_id = (int32_t)rd32b(p + 4);
    const int32_t pos = (int32_t)rd32b(p + 8);
    const uint32_t l_read_name = p[12];
    const int mapq = p[13];
    const uint32_t n_cigar = rd16b(p + 16);
    const int32_t l_seq = (int32_t)rd32b(p + 20);
    const int32_t contig = ref_sel[ref_id];
    const uint8_t* name = p + 36;
    const uint8_t* rec_end = p + 4 + block_size;
    const uint8_t* cig = name + l_read_name;
    const uint8_t* aux = cig + 4 * (uint64_t)n_cigar + (lq.has_seq ? (((uint64_t)(uint32_t)l_seq + 1) >> 1) + (uint64_t)(uint32_t)l_seq : 0ull);
    // query_name and its hash, lanes striding over bytes / words
    uint32_t nul = l_read_name;
    for (uint32_t i = lane; i < l_read_name; i += 64) if (name[i] == 0) { nul = i; break; }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) nul = min(nul, (uint32_t)__shfl_xor((int)nul, m, 64));
    const uint32_t name_len = nul;
    uint64_t acc = 0;
    for (uint32_t k = lane; k * 8 < name_len; k += 64) {
        uint64_t w = 0;
        for (int b = 0; b < 8; b++) if (k * 8 + b < name_len) w |= (uint64_t)name[k * 8 + b] << (8 * b);
        acc += gci_hash_word(w, k);
    }
    acc = (uint64_t)wave_sum<long long>((long long)acc);
    // first NM, first CG (bam_aux_get semantics)
    const uint8_t* nm_p = nullptr;
    const uint8_t* cg_p = nullptr;
    for (const uint8_t* q = aux; q + 3 <= rec_end;) {
        const int64_t sz = aux_value_size(q + 3, rec_end, q[2]);
        if (sz < 0 || q + 3 + sz > rec_end) break;
        if (q[0] == 'N' && q[1] == 'M' && !nm_p) nm_p = q + 2;
        if (q[0] == 'C' && q[1] == 'G' && !cg_p) cg_p = q + 2;
        q += 3 + sz;
    }
    int64_t NM = 0;
    const bool nm_bad = nm_p ? !nm_value(nm_p, NM) : false;
    // htslib moves a >65535-op CIGAR back from CG:B,I when op0 == <l_seq>S
    const uint8_t* ops = cig;
    uint32_t n_ops = n_cigar;
    if (n_cigar > 0 && pos >= 0) {
        const uint32_t op0 = rd32b(cig);
        if ((op0 & 0xF) == 4 && (op0 >> 4) == (uint32_t)l_seq && cg_p && cg_p[0] == 'B' && (cg_p[1] == 'I' || cg_p[1] == 'i')) {
            const uint32_t cg_len = rd32b(cg_p + 2);
            if (cg_len >= n_cigar && cg_len < (1u << 29)) { ops = cg_p + 6; n_ops = cg_len; }
        }
    }
    gci_rec r;
    r.name_hash = gci_hash_finish(acc, name_len); r.contig = -1; r.start = 0; r.end = 0; r.qlen = 0;
    r.rec_idx = rec + rec_idx_base; r.mapq = (uint8_t)mapq; r.flags = 0; r.name_len = (uint16_t)name_len;
    if (n_ops > (QUEUE_ALL ? 0u : LONG_OPS)) {   // summed chunk by chunk like the fast path's long CIGARs
        LongItem li;
        li.ops_off = (uint64_t)(ops - bam); li.n_ops = n_ops; li.rec = rec;
        li.nm = nm_p ? (nm_bad ? INT64_MIN : NM) : INT64_MAX;
        li.pos = pos; li.contig = contig; li.l_seq = l_seq; li.n_cigar_field = (int32_t)n_cigar;
        li.mapq = (uint32_t)mapq;
        if (lane == 0) out[rec] = r;
        if (!enqueue_long<64>(lq, li, lane) && lane == 0) report(lq.status_in, rec, GCI_E_CAPACITY);
        return;
    }
    int64_t tot[NSLOT];
    if constexpr (QUEUE_ALL) {
#pragma unroll
        for (int k = 0; k < NSLOT; k++) tot[k] = 0;         // no operations at all
    } else cigar_totals_wave(bam, n_bytes, (uint64_t)(ops - bam), n_ops, lane, tot);
    if (lane == 0) {
        const int st = decide(r, tot[0] + tot[7] + tot[8], tot[1], tot[2], tot[3], tot[4], nm_p != nullptr, nm_bad, NM, pos,
                              contig, l_seq, n_cigar, mapq, mq_cutoff, clip_percent, iden_percent);
        if (st != GCI_OK) report(lq.status_in, rec, st);
        out[rec] = r;
    }

}

// 80 VGPRs (the compiler settles on 83 by itself) and 9 KB of LDS: 6 waves per SIMD instead of 5, -3.5 us on chr19
__global__ __launch_bounds__(KB) __attribute__((amdgpu_waves_per_eu(K1_WAVES, K1_WAVES))) void k_bam_filter(
    const uint8_t* __restrict__ bam, uint64_t n_bytes, const uint64_t* __restrict__ rec_off, uint32_t n_rec,
    const int32_t* __restrict__ ref_sel, int32_t n_ref, int map_qual, int mq_cutoff, double clip_percent,
    double iden_percent, uint32_t rec_idx_base, gci_rec* __restrict__ out, unsigned long long* __restrict__ status,
    const LongQueue lq
    )
{
#define TR(i) do {} while (0)
    TR(0);
    if (blockIdx.x == 0 && threadIdx.x < 4) lq.next_counters[threadIdx.x] = 0u;    // nobody reads that set during this call
    if (blockIdx.x == 0 && threadIdx.x == 4) *lq.next_status = ~0ull;
    __shared__ __attribute__((aligned(16))) uint8_t stage[KB / G][ROW];
    __shared__ uint8_t aux_sz[256];                                         // fixed value size per aux type, 0 = other
    const int t = threadIdx.x;
    const int gl = t & (G - 1), grp = t / G;
    const uint32_t rec = (uint32_t)((uint64_t)blockIdx.x * (KB / G) + grp);
    uint8_t* row = stage[grp];

    // ---- independent loads first: the record offset, then (one round trip later) its head ------------------------
    bool live = rec < n_rec;
    const uint64_t off = live ? rec_off[rec] : 0;
    for (int i = t; i < 256; i += KB) {
        const char c = (char)i;
        aux_sz[i] = (c == 'A' || c == 'c' || c == 'C') ? 1 : (c == 's' || c == 'S') ? 2 : (c == 'i' || c == 'I' || c == 'f') ? 4 : 0;
    }
    if (live && off + 36 > n_bytes) {
        if (gl == 0) {
            gci_rec r; r.name_hash = 0; r.contig = -1; r.start = r.end = r.qlen = 0; r.rec_idx = rec + rec_idx_base;
            r.mapq = 0; r.flags = 0; r.name_len = 0;
            report(lq.status_in, rec, GCI_E_MALFORMED); out[rec] = r;
        }
        live = false;
    }
    if (live) {
        const uint64_t a0 = off & ~15ull;                       // aligned loads; the record starts at row[off & 15]
#pragma unroll
        for (int i = 0; i < (HEADP / 16 + G - 1) / G; i++) {
            const int c = gl + G * i;
            if (c < HEADP / 16) stage16(row + 16 * c, bam, a0 + 16ull * c, n_bytes);
        }
    }
    TR(1);
    __syncthreads();                                            // the two small tables; (also covers the staged heads)
    TR(2);
    // The fast path of one record (its four lanes); true when the staged window cannot answer and the record has to be
    // parsed from global memory.  Written as a lambda so that no lane LEAVES the kernel here: the slow records of a wave
    // are handled below by all 64 of its lanes.
    auto fast_path = [&]() -> bool {

    // ---- core fields (GCI.py:152-156) ---------------------------------------------------------------------------
    gci_rec r;
    r.name_hash = 0; r.contig = -1; r.start = 0; r.end = 0; r.qlen = 0; r.rec_idx = rec + rec_idx_base; r.mapq = 0;
    r.flags = 0; r.name_len = 0;
    const uint8_t* hd = row + (off & 15ull);          // record byte k is hd[k]
    const int32_t block_size = (int32_t)lds_u32(hd);
    const int32_t ref_id = (int32_t)lds_u32(hd + 4);
    const int32_t pos = (int32_t)lds_u32(hd + 8);
    const uint32_t w12 = lds_u32(hd + 12), w16 = lds_u32(hd + 16);
    const uint32_t l_read_name = w12 & 0xFF;
    const int mapq = (w12 >> 8) & 0xFF;
    const uint32_t n_cigar = w16 & 0xFFFF;
    const uint32_t flag = w16 >> 16;
    const int32_t l_seq = (int32_t)lds_u32(hd + 20);
    const uint64_t rec_end = off + 4 + (uint64_t)(uint32_t)block_size;
    // a heads stream (gci_bam_heads) holds the records without their SEQ / QUAL bytes; l_seq keeps its value
    const uint64_t aux_off = off + 36 + l_read_name + 4ull * n_cigar +
                             (lq.has_seq ? (((uint64_t)(uint32_t)l_seq + 1) >> 1) + (uint64_t)(uint32_t)l_seq : 0ull);
    r.mapq = (uint8_t)mapq;
    if (block_size < 32 || rec_end > n_bytes || l_seq < 0 || aux_off > rec_end) {
        if (gl == 0) { report(lq.status_in, rec, GCI_E_MALFORMED); out[rec] = r; }
        return false;
    }
    if (ref_id < 0 || ref_id >= n_ref || (flag & (0x4u | 0x100u | 0x800u)) || mapq < map_qual) {
        if (gl == 0) out[rec] = r;
        return false;
    }
    TR(8);
    // ---- second (and last) dependent round trip: the refID -> selected-contig entry, the aux head and the CIGAR
    // words behind the staged head.  (The table is not kept in LDS: the waves per SIMD are what this kernel lives on.)
    const int32_t contig = ref_sel[ref_id];
    const uint32_t cig_at = 36 + l_read_name;                       // byte offset of the CIGAR in the record
    {
        const uint64_t a0 = aux_off & ~15ull;
#pragma unroll
        for (int i = 0; i < (AUXP / 16 + G - 1) / G; i++) {
            const int c = gl + G * i;
            if (c < AUXP / 16) stage16(row + HEADP + 16 * c, bam, a0 + 16ull * c, n_bytes);
        }
    }
    TR(9);
    // fetch(contig=target) only ever yields records of selected contigs (GCI.py:151, 260)
    if (contig < 0) {
        if (gl == 0) out[rec] = r;
        return false;
    }
    const bool odd_name = cig_at + 4 > HEAD;                    // first CIGAR word must lie inside the staged head
    const bool long_cigar = n_cigar > LONG_OPS;                // its totals are computed by k_cigar_chunks
    const bool odd = odd_name || long_cigar;
    const uint32_t staged_ops = odd ? 0u : min(n_cigar, (HEAD - cig_at) / 4u);
    // CIGAR base totals (get_cigar_stats()[0], GCI.py:157-162) of this lane's share of the ops, in registers -- not per op
    // code but as the four sums the decision uses (decide4): S, den1 = M + I + S, den2 = M + I + D, rlen = M + D + N.  Which
    // of them an op code feeds is a nibble of a constant (bit 0: S, 1: den1, 2: den2, 3: rlen; M, = and X feed the last three).
    unsigned long long sS = 0, sQ = 0, sA = 0, sR = 0;
    auto add_op = [&](uint32_t v) {
        const uint32_t nib = (uint32_t)(0x0000000EE0038C6Eull >> (4u * (v & 0xFu)));
        const uint32_t len = v >> 4;
        sS += len & (0u - (nib & 1u));
        sQ += len & (0u - ((nib >> 1) & 1u));
        sA += len & (0u - ((nib >> 2) & 1u));
        sR += len & (0u - ((nib >> 3) & 1u));
    };
    if (!odd) {
        // CIGAR words behind the staged head, straight from global memory.  Unaligned vector loads from global
        // memory are served, but ~100x slower than aligned ones on gfx950 (tools/exp_k1_trace.py), and a CIGAR
        // starts at 36 + l_read_name: so each lane walks a CONTIGUOUS quarter of the tail with 16-byte ALIGNED
        // loads and re-aligns in registers (v_alignbyte), carrying the previous chunk over.
        const uint32_t n_tail = n_cigar - staged_ops;
        const uint32_t per = (n_tail + G - 1) / G;
        const uint32_t k0 = staged_ops + gl * per, k1 = min(n_cigar, k0 + per);
        if (k0 < k1) {
            const uint64_t p0 = off + cig_at + 4ull * k0;                   // byte offset of this lane's first op
            const uint32_t sh = (uint32_t)(p0 & 15ull), d = sh >> 2, b = sh & 3u, cnt = k1 - k0;
            const uint64_t a = p0 & ~15ull;
            const uint64_t lim = n_bytes & ~15ull;                           // aligned chunks fully inside the stream
            auto ld = [&](uint64_t at) { return at < lim ? *reinterpret_cast<const uint4*>(bam + at) : load16_tail(bam, at, n_bytes); };
            // chunk c holds aligned dwords 4c .. 4c + 3; this lane's op k starts in dword d + k.  Every dword of a
            // chunk is decoded (static register indexing) and the ops outside [0, cnt) are predicated off.
            auto take = [&](const uint4& lo, uint32_t next_x, uint32_t c) {
                const uint32_t w[5] = {lo.x, lo.y, lo.z, lo.w, next_x};
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const uint32_t v = __builtin_amdgcn_alignbyte(w[j + 1], w[j], b);
                    if (4u * c + j - d < cnt) add_op(v);
                }
            };
            // the first NUP chunks are requested together (with four lanes per record a lane's share of a HiFi tail is
            // ~16 ops: one round trip); longer tails continue with one new chunk at a time
            const uint32_t nch = (sh + 4u * cnt + 15u) >> 4;
            const uint4 z = make_uint4(0, 0, 0, 0);
            uint4 cu[NUP];
#pragma unroll
            for (int i = 0; i < NUP; i++) cu[i] = (uint32_t)i < nch ? ld(a + 16ull * i) : z;
#pragma unroll
            for (int i = 0; i + 1 < NUP; i++) if ((uint32_t)i < nch) take(cu[i], cu[i + 1].x, (uint32_t)i);
            uint4 lo = cu[NUP - 1];
            for (uint32_t c = NUP - 1; c < nch; c++) {
                const uint4 hi = c + 1 < nch ? ld(a + 16ull * (c + 1)) : z;
                take(lo, hi.x, c);
                lo = hi;
            }
        }
        TR(10);
        // staged part of the CIGAR (GCI.py:157-162: get_cigar_stats()[0], base totals per op code)
        for (uint32_t k = gl; k < staged_ops; k += G) {
            add_op(lds_u32(hd + cig_at + 4 * k));
        }
    }
    TR(3);
    wave_lds_fence();                                               // aux head visible to the 4 lanes of the record
    TR(4);

    // ---- anything the staged window cannot answer goes to the slow path -------------------------------------------
    bool slow = odd_name;
    // htslib's long-CIGAR placeholder: op0 == <l_seq>S and a CG:B,I tag somewhere in the aux block
    if (!slow && n_cigar > 0 && pos >= 0) {
        const uint32_t op0 = lds_u32(hd + cig_at);
        if ((op0 & 0xF) == 4 && (op0 >> 4) == (uint32_t)l_seq) slow = true;      // (rare: let the slow path look for CG)
    }
    // ---- aux walk in the staged window: first NM (bam_aux_get semantics) ------------------------------------------
    const uint32_t aux_len = (uint32_t)min((uint64_t)AUXB, rec_end - aux_off);
    const uint8_t* ax = row + HEADP + (aux_off & 15ull);
    bool have_nm = false, nm_bad = false, walked_all = false;
    int64_t NM = 0;
    if (!slow) {
        uint32_t q = 0;
        for (;;) {
            if (q + 3 > aux_len) { walked_all = aux_off + q + 3 > rec_end; break; }          // no further complete tag header
            const uint32_t w = lds_u32(ax + q);                                          // tag[2] | type | first value byte
            const uint32_t tag = w & 0xFFFF;
            const uint8_t ty = (uint8_t)(w >> 16);
            uint32_t sz = aux_sz[ty];
            if (tag == (uint32_t)('N' | ('M' << 8))) {
                if (sz == 0 || q + 3 + sz > aux_len) break;                              // odd type / value not staged: slow path
                have_nm = nm_value(ax + q + 2, NM);
                nm_bad = !have_nm; have_nm = true;
                break;
            }
            if (sz == 0) {
                // variable-size values, decided here only if they end inside the staged window
                if (ty == 'Z' || ty == 'H') {
                    uint32_t e = q + 3;
                    while (e < aux_len && ax[e]) e++;
                    if (e >= aux_len) break;
                    sz = e - (q + 3) + 1;
                } else if (ty == 'B' && q + 8 <= aux_len) {
                    const uint8_t sub = ax[q + 3];
                    const uint32_t es = aux_sz[sub] == 1 && sub != 'A' ? 1u : (sub == 's' || sub == 'S') ? 2u
                                      : (sub == 'i' || sub == 'I' || sub == 'f') ? 4u : 0u;
                    const uint32_t cnt = lds_u32(ax + q + 4);
                    if (es == 0 || cnt > AUXB) break;
                    sz = 5 + cnt * es;
                } else break;
            }
            if (q + 3 + sz > aux_len) break;                                             // window end
            q += 3 + sz;
        }
        if (!have_nm && !walked_all) slow = true;          // NM (if any) lies beyond what was staged
    }
    if (slow) return true;

    TR(5);
    // ---- query_name: bytes up to the first NUL, 64-bit hash of its 8-byte words -----------------------------------
    const uint8_t* name = hd + 36;
    uint32_t nul = l_read_name;
    for (uint32_t i = gl * 4; i < l_read_name; i += 4 * G) {
        if (has_zero_byte(lds_u32(name + i))) {
            for (uint32_t b = i; b < i + 4 && b < l_read_name; b++) if (name[b] == 0) { nul = min(nul, b); break; }
            break;
        }
    }
#define STEP_MIN(m) nul = min(nul, (uint32_t)__shfl_xor((int)nul, m, G))
    GRP_STEPS(STEP_MIN);
#undef STEP_MIN
    const uint32_t name_len = nul;
    uint64_t acc = 0;
    for (uint32_t k = gl; k * 8 < name_len; k += G) {
        const uint32_t b0 = k * 8;
        uint64_t w;
        __builtin_memcpy(&w, name + b0, 8);                                        // stays inside the staged row
        const uint32_t keep = name_len - b0;                                       // bytes of this word inside the name
        if (keep < 8) w &= (1ull << (8 * keep)) - 1ull;
        acc += gci_hash_word(w, k);
    }
#define STEP_ACC(m) acc += (uint64_t)__shfl_xor((long long)acc, m, G)
    GRP_STEPS(STEP_ACC);
#undef STEP_ACC
    r.name_hash = gci_hash_finish(acc, name_len);
    r.name_len = (uint16_t)name_len;

    TR(6);
    if (long_cigar) {           // parse is complete: hand only the CIGAR totals + decision to k_cigar_chunks
        LongItem it;
        it.ops_off = off + cig_at; it.n_ops = n_cigar; it.rec = rec;
        it.nm = have_nm ? (nm_bad ? INT64_MIN : NM) : INT64_MAX;                   // sentinels: bad type / absent
        it.pos = pos; it.contig = contig; it.l_seq = l_seq; it.n_cigar_field = (int32_t)n_cigar;
        it.mapq = (uint32_t)mapq;
        if (gl == 0) out[rec] = r;                                                  // name hash / length are final
        if (!enqueue_long<G>(lq, it, gl) && gl == 0) report(lq.status_in, rec, GCI_E_CAPACITY);
        return false;
    }
    // the four lanes' shares -> every lane of the group holds the record's totals
#define GRP_SUM(x) do { x += (unsigned long long)__shfl_xor((long long)x, 1, G); if (G == 4) x += (unsigned long long)__shfl_xor((long long)x, 2, G); } while (0)
    GRP_SUM(sS); GRP_SUM(sQ); GRP_SUM(sA); GRP_SUM(sR);
#undef GRP_SUM
    if (gl != 0) return false;  // the rest is scalar per record
    const int st = decide4(r, (int64_t)sS, (int64_t)sQ, (int64_t)sA, (int64_t)sR, have_nm, nm_bad, NM, pos, contig, l_seq,
                           n_cigar, mapq, mq_cutoff, clip_percent, iden_percent);
    if (st != GCI_OK) report(lq.status_in, rec, st);
    out[rec] = r;
    TR(7);
    return false;
    };
    const bool is_slow = live && fast_path();
    // ---- slow records of this wave, one after the other, by the whole wave (rare: NM beyond the staged window, htslib's
    // CG:B,I long-CIGAR restore, names of 73 bytes and more); no second kernel launch for them
    for (unsigned long long m = __ballot(is_slow && gl == 0); m; m &= m - 1ull) {
        const uint32_t rs = (uint32_t)__shfl((int)rec, __builtin_ctzll(m), 64);
        slow_record(bam, n_bytes, rec_off[rs], ref_sel, rs, t & 63, lq, mq_cutoff, clip_percent, iden_percent, rec_idx_base, out);
    }
}

// ---- long CIGARs, chunk by chunk: one wave per chunk of CHUNK_DW aligned dwords.  A lane owns PIECES 16-byte
// pieces (all requested up front), takes the dword after each piece from its neighbour, re-aligns with
// v_alignbyte and adds the op lengths to five class sums; the wave stores the chunk's sums.

__device__ __forceinline__ unsigned long long wave_sum_u40(unsigned long long v)        // v < 2^40 in every lane
{
    const uint32_t lo = wave_sum<uint32_t>((uint32_t)v & 0xFFFFu);
    const uint32_t hi = wave_sum<uint32_t>((uint32_t)(v >> 16));
    return (unsigned long long)lo + ((unsigned long long)hi << 16);
}

__global__ __launch_bounds__(BLOCK) void k_cigar_chunks(const uint8_t* __restrict__ bam, uint64_t n_bytes, const LongQueue lq,
                                                        unsigned long long* __restrict__ status)
{
    // Publish what the two kernels before this one reported (they are complete, this kernel reports nothing and the
    // one after it reports into `status` directly): the caller's word is written without a memset launch in front.
    if (blockIdx.x == 0 && threadIdx.x == 0) *status = *lq.status_in;
    const int lane = threadIdx.x & 63;
    const uint32_t n = (uint32_t)min((unsigned long long)(uint32_t)*lq.n_long, (unsigned long long)lq.cap_chunks);
    const uint32_t waves = gridDim.x * (BLOCK / 64);
    for (uint32_t wi = blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6); wi < n; wi += waves) {
        const unsigned long long e = lq.chunks[wi];
        if (e == ~0ull) continue;
        const LongItem* it = lq.items + (uint32_t)(e >> 32);
        const uint32_t ci = (uint32_t)e;
        const uint64_t p0 = it->ops_off;
        const uint32_t n_ops = it->n_ops;
        const uint32_t d0 = (uint32_t)(p0 & 15ull) >> 2, b = (uint32_t)p0 & 3u;
        // op k of the record starts in aligned dword d0 + k (counted from the 16-byte boundary below p0);
        // this chunk owns the ops that start in its dwords [r0, r1)
        const uint64_t g0 = (uint64_t)ci * CHUNK_DW;
        const uint64_t abase = (p0 & ~15ull) + g0 * 4ull;
        const uint32_t r0 = ci == 0 ? d0 : 0u;
        const uint32_t r1 = (uint32_t)min((uint64_t)CHUNK_DW, (uint64_t)d0 + n_ops - g0);
        uint4 pc[PIECES];
#pragma unroll
        for (int u = 0; u < PIECES; u++) {
            const uint32_t c = lane + 64u * u;
            const uint64_t at = abase + 16ull * c;
            const bool need = 4u * c < r1 + 1u && 4u * c + 4u > r0;           // + 1: the dword that completes the last op
            pc[u] = !need ? make_uint4(0, 0, 0, 0) : at + 16 <= n_bytes ? *reinterpret_cast<const uint4*>(bam + at)
                                                                          : load16_tail(bam, at, n_bytes);
        }
        uint32_t ext = 0;                                                       // first dword of the next chunk
        if (r1 == CHUNK_DW && b != 0) {
            const uint64_t at = abase + 4ull * CHUNK_DW;
            ext = at + 4 <= n_bytes ? *reinterpret_cast<const uint32_t*>(bam + at) : load16_tail(bam, at, n_bytes).x;
        }
        unsigned long long wide[5] = {0ull, 0ull, 0ull, 0ull, 0ull};
        uint32_t s[5] = {0u, 0u, 0u, 0u, 0u};                                   // at most 16 lengths < 2^28 each
#pragma unroll
        for (int u = 0; u < PIECES; u++) {
            uint32_t w4 = (uint32_t)__shfl((int)pc[u].x, (lane + 1) & 63, 64);
            const uint32_t first_next = u + 1 < PIECES ? (uint32_t)__builtin_amdgcn_readfirstlane((int)pc[(u + 1) % PIECES].x) : ext;
            if (lane == 63) w4 = first_next;
            const uint32_t w[5] = {pc[u].x, pc[u].y, pc[u].z, pc[u].w, w4};
            const uint32_t li0 = 4u * (lane + 64u * u);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t v = __builtin_amdgcn_alignbyte(w[j + 1], w[j], b);
                const uint32_t len = (li0 + j - r0) < (r1 - r0) ? v >> 4 : 0u;     // unsigned: r0 <= li < r1
                const uint32_t op = v & 0xFu;
                s[0] += len & (0u - ((0x181u >> op) & 1u));                        // M, =, X
                s[1] += op == 1u ? len : 0u;
                s[2] += op == 2u ? len : 0u;
                s[3] += op == 3u ? len : 0u;
                s[4] += op == 4u ? len : 0u;
            }
            if ((u & 3) == 3) {
#pragma unroll
                for (int k = 0; k < 5; k++) { wide[k] += s[k]; s[k] = 0u; }
            }
        }
        ChunkSums out;
#pragma unroll
        for (int k = 0; k < 5; k++) out.v[k] = wave_sum_u40(wide[k]);
        if (lane == 0) lq.sums[wi] = out;
    }
}

__global__ __launch_bounds__(BLOCK) void k_cigar_finish(const LongQueue lq, int mq_cutoff, double clip_percent,
                                                        double iden_percent, gci_rec* __restrict__ out,
                                                        unsigned long long* __restrict__ status)
{
    const uint32_t n = (uint32_t)min(*lq.n_long >> 32, (unsigned long long)lq.cap_items);
    for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK) {
        const LongItem x = lq.items[i];
        if (x.n_chunks == 0) continue;
        int64_t t[5] = {0, 0, 0, 0, 0};
        for (uint32_t c = 0; c < x.n_chunks; c++) {
            const ChunkSums cs = lq.sums[x.chunk_base + c];
#pragma unroll
            for (int k = 0; k < 5; k++) t[k] += (int64_t)cs.v[k];
        }
        gci_rec r = out[x.rec];
        const int st = decide(r, t[0], t[1], t[2], t[3], t[4], x.nm != INT64_MAX, x.nm == INT64_MIN, x.nm, x.pos, x.contig,
                              x.l_seq, (uint32_t)x.n_cigar_field, (int)(x.mapq & 0xFFu), mq_cutoff, clip_percent, iden_percent);
        if (st != GCI_OK) report(status, x.rec, st);
        if (r.flags) r.flags |= (uint8_t)(x.mapq >> 8);                 // (GCI_REC_NAME16 of a record queued by the paged filter)
        out[x.rec] = r;
    }
}

// =====================================================================================================================
// The paged record filter (round 3).  Input: RECORD PAGES (k_pages.hip: the bytes read_sam looks at, copied once into
// fixed-size pages while the inflated stream is walked; every record and every CIGAR 16-byte aligned, a small directory
// per page).  One workgroup per page: ONE round of coalesced 16-byte loads brings the page into LDS -- no offset table,
// no second and third dependent round trip, no re-aligning of dwords in registers -- and four lanes per record parse it
// there: core fields, NUL search + name hash, the whole aux walk (bam_aux_get semantics, CG:B,I restore included), the
// CIGAR as aligned ds_read_b128 pieces.  Same decisions, same 32-byte records, same status word as k_bam_filter.
//   kind 1 (CIGAR in the blob: ONT) -> the chunk queue of k_cigar_chunks, which reads the blob;
//   kind 2 (the whole record in the blob: CG:B,I with 65536+ operations, kilobytes of tags) -> slow_record over the blob.
#define PGK 256                    // threads per workgroup: 64 records per pass
#define PG_KIND_EXT 1u
#define PG_KIND_OVERSIZE 2u
#define PG_KIND_MALFORMED 4u
#define PG_MAGIC 0x31504347u
#define PG_REF_LDS 1024

struct PagesArgs {
    const uint8_t* buf; uint64_t total_bytes; uint32_t page_bytes, n_pages, n_rec;
    const int32_t* ref_sel; int32_t n_ref; int map_qual, mq_cutoff; double clip_percent, iden_percent;
    uint32_t rec_idx_base; gci_rec* out; uint64_t* name_off;
};

#ifndef PG_LPR
#define PG_LPR 4                   // lanes per record in the lean pass (a power of two)
#endif
#ifndef PG_WAVES
#define PG_WAVES 5                 // 96 VGPRs, no spills: 0.260 ms against 0.290 (6: 80 VGPRs, 6 spilled) and 0.288 (4) on one box
#endif
__global__ __launch_bounds__(PGK) __attribute__((amdgpu_waves_per_eu(PG_WAVES, PG_WAVES))) void k_bam_filter_pages(const PagesArgs A, const LongQueue lq)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t page[];
    __shared__ uint8_t aux_sz[256];                                         // fixed value size per aux type, 0 = other
    __shared__ uint8_t aux_kind[256];
    __shared__ uint16_t s_full[GCI_PAGE_MAX_BYTES / 48];                    // records pass A leaves to pass B
    __shared__ uint32_t s_n_full;
    __shared__ int32_t s_ref_sel[PG_REF_LDS];                               // refID -> selected contig, when the table is small:
    const int t = threadIdx.x;                                              // a gather from global memory in the middle of the
    const bool ref_in_lds = A.n_ref <= PG_REF_LDS;                          // parse would be a second round trip per page
    if (ref_in_lds) for (int i = t; i < A.n_ref; i += PGK) s_ref_sel[i] = A.ref_sel[i];
    if (t == 0) s_n_full = 0;
    if (blockIdx.x == 0 && t < 4) lq.next_counters[t] = 0u;                 // nobody reads that set during this call
    if (blockIdx.x == 0 && t == 4) *lq.next_status = ~0ull;
    const uint32_t P = A.page_bytes;
    const uint64_t page_at = (uint64_t)blockIdx.x * P;
    {
        // the whole page, requested up front (P is a multiple of 4096 = one 16-byte piece per thread and round)
        const uint4* __restrict__ src = reinterpret_cast<const uint4*>(A.buf + page_at);
        const uint32_t rounds = P / (16u * PGK);
        for (uint32_t r0 = 0; r0 < rounds; r0 += 8) {
            uint4 v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = r0 + u < rounds ? src[(r0 + u) * PGK + t] : make_uint4(0, 0, 0, 0);
#pragma unroll
            for (int u = 0; u < 8; u++) if (r0 + u < rounds) reinterpret_cast<uint4*>(page)[(r0 + u) * PGK + t] = v[u];
        }
    }
    for (int i = t; i < 256; i += PGK) {
        const char c = (char)i;
        aux_sz[i] = (c == 'A' || c == 'c' || c == 'C') ? 1 : (c == 's' || c == 'S') ? 2 : (c == 'i' || c == 'I' || c == 'f') ? 4 : 0;
        // for the lean path: size | signed << 3 | integer << 4
        aux_kind[i] = aux_sz[i] | ((c == 'c' || c == 's' || c == 'i') ? 8 : 0) | ((c == 'c' || c == 'C' || c == 's' || c == 'S' || c == 'i' || c == 'I') ? 16 : 0);
    }
    __syncthreads();
    const uint32_t* ph = reinterpret_cast<const uint32_t*>(page);
    uint32_t n_recs = ph[0];
    const uint32_t first_rec = ph[1];
    if (ph[3] != PG_MAGIC || n_recs > (P - 16u) / 50u || (uint64_t)first_rec + n_recs > A.n_rec) {       // not a page of ours
        if (t == 0) report(lq.status_in, first_rec < A.n_rec ? first_rec : 0u, GCI_E_MALFORMED);
        n_recs = 0;
    }
    // Pass A: the lean path over every record of the page, PG_LPR lanes per record.  What a wave spends per record outside the
    // CIGAR loop (header, tag walk, name hash, decision, output) is the same number of instructions whether 16 or 32 records
    // share the wave, and the kernel is bound by instruction issue (profiles/r03t_pmc_summary.txt): fewer lanes per record,
    // fewer instructions per page.  A record the lean path does not finish is listed for pass B.
    {
        // (with fewer than four lanes per record not every wave has records: which waves do rotates with the page, so that the
        // pages resident on a CU keep all four SIMDs busy -- a workgroup's waves go to the SIMDs in a fixed order)
        const int tv = PG_LPR == 4 ? t : (int)(((uint32_t)t + 64u * (blockIdx.x & 3u)) & (PGK - 1u));
        const int gl = tv % PG_LPR;
        const uint32_t lgrp = (uint32_t)tv / PG_LPR;
        for (uint32_t j0 = 0; j0 < n_recs; j0 += PGK / PG_LPR) {
            const uint32_t j = j0 + lgrp;
            const uint32_t rec = first_rec + j;
            // ---- the lean path: what nearly every record needs and nothing else, straight-line.  A record with anything unusual
            // about it -- not inline (kind != 0), the long-CIGAR placeholder, a NUL inside its name, NM behind a Z / H / B tag or of
            // an odd type or missing, an operation of 2^24 bases, a zero denominator, a quotient within 1e-4 of a threshold -- is
            // left to pass B's fast_path(), which knows every case (1); a record that is done returns 0.
            auto lean_path = [&]() -> int {
                uint32_t base = (uint32_t)reinterpret_cast<const uint16_t*>(page + 16)[j] << 4;
                if (base > P - 48u) return 1;
                const uint8_t* hd = page + base;
                const uint4 c0 = *reinterpret_cast<const uint4*>(hd);
                const uint4 c1 = *reinterpret_cast<const uint4*>(hd + 16);
                const uint32_t size = c0.x, kind = c0.w >> 16, l_read_name = c0.w & 0xFFu, n_cigar = c1.x & 0xFFFFu, flag = c1.x >> 16;
                const int32_t ref_id = (int32_t)c0.y, pos = (int32_t)c0.z, l_seq = (int32_t)c1.y;
                const int mapq = (int)((c0.w >> 8) & 0xFFu);
                const uint32_t aux_len = c1.z;
                const uint32_t cig_at = (36u + l_read_name + 15u) & ~15u;
                const uint32_t aux_at = cig_at + ((4u * n_cigar + 15u) & ~15u);
                if (kind != 0 || l_read_name == 0 || n_cigar == 0 || (uint64_t)aux_at + aux_len > size || base + size > P) return 1;
                gci_rec r;
                r.name_hash = 0; r.contig = -1; r.start = 0; r.end = 0; r.qlen = 0; r.rec_idx = rec + A.rec_idx_base;
                r.mapq = (uint8_t)mapq; r.flags = 0; r.name_len = 0;
                if (gl == 0 && A.name_off) A.name_off[rec] = page_at + base + 36u;
                int32_t contig = -1;
                if (!(ref_id < 0 || ref_id >= A.n_ref || (flag & (0x4u | 0x100u | 0x800u)) || mapq < A.map_qual))      // GCI.py:152-156
                    contig = ref_in_lds ? s_ref_sel[ref_id] : A.ref_sel[ref_id];
                if (contig < 0) {                                    // filtered, or not on a selected contig (GCI.py:151, 260)
                    if (gl == 0) A.out[rec] = r;
                    return 0;
                }
                const uint32_t op0 = lds_u32(hd + cig_at);
                if (pos >= 0 && (op0 & 0xFu) == 4u && (op0 >> 4) == (uint32_t)l_seq) return 1;        // htslib's long-CIGAR placeholder?
                // CIGAR totals (see fast_path)
                uint32_t sS = 0, sQ = 0, sA = 0, sR = 0, big = 0;
                auto add_op = [&](uint32_t v) {
                    const uint32_t op = v & 0xFu, len = v >> 4;
                    sS = __umul24(len, __builtin_amdgcn_ubfe(0x010u, op, 1)) + sS;
                    sQ = __umul24(len, __builtin_amdgcn_ubfe(0x193u, op, 1)) + sQ;
                    sA = __umul24(len, __builtin_amdgcn_ubfe(0x187u, op, 1)) + sA;
                    sR = __umul24(len, __builtin_amdgcn_ubfe(0x18Du, op, 1)) + sR;
                };
                {
                    const uint8_t* cg = hd + cig_at;
                    for (uint32_t p = gl; 4u * p < n_cigar; p += PG_LPR) {
                        const uint4 v = *reinterpret_cast<const uint4*>(cg + 16u * p);
                        big |= v.x | v.y | v.z | v.w;
                        add_op(v.x); add_op(v.y); add_op(v.z); add_op(v.w);
                    }
                }
                // first NM among fixed-size tags
                const uint8_t* ax = hd + aux_at;
                uint32_t q = 0, nmk = 0;
                for (;;) {
                    if (q + 3 > aux_len) return 1;
                    const uint32_t w = lds_u32(ax + q);
                    const uint32_t k = aux_kind[(w >> 16) & 0xFFu];
                    if ((k & 7u) == 0 || q + 3 + (k & 7u) > aux_len) return 1;
                    if ((w & 0xFFFFu) == (uint32_t)('N' | ('M' << 8))) { nmk = k; break; }
                    q += 3 + (k & 7u);
                }
                if (!(nmk & 16u)) return 1;                          // NM of a non-integer type
                const uint32_t nv = lds_u32(ax + q + 3), nbits = 8u * (nmk & 7u);
                const int64_t NM = nbits == 32u ? ((nmk & 8u) ? (int64_t)(int32_t)nv : (int64_t)nv)
                                                : ((nmk & 8u) ? (int64_t)__builtin_amdgcn_sbfe((int)nv, 0, nbits) : (int64_t)__builtin_amdgcn_ubfe(nv, 0, nbits));
                // query_name: l_read_name - 1 bytes and the NUL -- if there is no NUL before it
                const uint8_t* name = hd + 36;
                const uint32_t name_len = l_read_name - 1u;
                bool odd = name[name_len] != 0;
                uint64_t acc = 0;
                for (uint32_t k = gl; k * 8 < name_len; k += PG_LPR) {
                    const uint32_t b0 = k * 8;
                    uint64_t w;
                    __builtin_memcpy(&w, name + b0, 8);
                    const uint32_t keep = name_len - b0;
                    if (keep < 8) w &= (1ull << (8 * keep)) - 1ull;
                    const uint64_t full = keep < 8 ? w | (~0ull << (8 * keep)) : w;                 // bytes past the name count as non-zero
                    odd = odd || (((full - 0x0101010101010101ull) & ~full & 0x8080808080808080ull) != 0);
                    acc += gci_hash_word(w, k);
                }
#pragma unroll
                for (int m = 1; m < PG_LPR; m <<= 1) acc += (uint64_t)__shfl_xor((long long)acc, m, PG_LPR);
                big |= odd ? 0xF0000000u : 0u;                       // (one exchange for both reasons to leave)
#pragma unroll
                for (int m = 1; m < PG_LPR; m <<= 1) big |= (uint32_t)__shfl_xor((int)big, m, PG_LPR);
                if (big >> 28) return 1;
#define PG_SUM32(x) do { _Pragma("unroll") for (int m = 1; m < PG_LPR; m <<= 1) x += (uint32_t)__shfl_xor((int)x, m, PG_LPR); } while (0)
                PG_SUM32(sS); PG_SUM32(sQ); PG_SUM32(sA); PG_SUM32(sR);
#undef PG_SUM32
                // the decision of GCI.py:163-168 where single precision settles it (ratio_cmp): den1 = M + I + S, den2 = M + I + D
                const int64_t num2 = (int64_t)sA - NM;
                if (sQ == 0 || sA == 0 || (sQ >> 30) || (sA >> 30) || num2 > (1 << 30) || num2 < -(1 << 30)) return 1;
                const float x1 = (float)sS * __builtin_amdgcn_rcpf((float)sQ), c1f = (float)A.clip_percent;
                const float x2 = (float)(int32_t)num2 * __builtin_amdgcn_rcpf((float)sA), c2f = (float)A.iden_percent;
                const float m1 = 1e-4f * fmaxf(1.0f, fabsf(x1)), m2 = 1e-4f * fmaxf(1.0f, fabsf(x2));
                if (!(fabs(A.clip_percent) < 1e30) || !(fabs(A.iden_percent) < 1e30)) return 1;
                if (!(x1 < c1f - m1 || x1 > c1f + m1)) return 1;     // too close to the clip threshold for single precision
                r.name_hash = gci_hash_finish(acc, name_len);
                r.name_len = (uint16_t)name_len;
                if (x1 < c1f - m1) {                                 // clip test passed: the identity test decides
                    if (!(x2 < c2f - m2 || x2 > c2f + m2)) return 1;
                    if (x2 > c2f + m2) {
                        r.contig = contig; r.start = pos;
                        r.end = (int32_t)((int64_t)pos + (sR > 0 ? (int64_t)sR : 1));
                        r.qlen = l_seq;
                        r.flags = GCI_REC_PASS | (mapq >= A.mq_cutoff ? GCI_REC_HQ : 0) | GCI_REC_NAME16;
                    }
                }
                if (gl == 0) A.out[rec] = r;
                return 0;
            };

            if (j < n_recs && lean_path() != 0 && gl == 0) s_full[atomicAdd(&s_n_full, 1u)] = (uint16_t)j;
        }
    }
    __syncthreads();
    // Pass B: the records left over (every ONT record, the odd HiFi one), four lanes each, with the path that knows every case
    const uint32_t n_full = s_n_full;
    const int gl = t & 3, grp = t >> 2;
    for (uint32_t q0 = 0; q0 < n_full; q0 += PGK / 4) {
        const uint32_t qi = q0 + grp;
        const bool on = qi < n_full;
        const uint32_t j = on ? s_full[qi] : 0u;
        const uint32_t rec = first_rec + j;
        uint64_t slow_off = 0;
        auto fast_path = [&]() -> bool {
            uint32_t base = (uint32_t)reinterpret_cast<const uint16_t*>(page + 16)[j] << 4;
            if (base > P - 48u) base = P - 48u;                                    // (never, in a page of ours)
            const uint8_t* hd = page + base;
            const uint4 c0 = *reinterpret_cast<const uint4*>(hd);                  // size | refID | pos | l_read_name, mapq, kind
            const uint4 c1 = *reinterpret_cast<const uint4*>(hd + 16);             // n_cigar, flag | l_seq | aux_len | blob offset lo
            const uint32_t blob_hi = *reinterpret_cast<const uint32_t*>(hd + 32);
            const uint32_t size = c0.x, kind = c0.w >> 16, l_read_name = c0.w & 0xFFu, n_cigar = c1.x & 0xFFFFu, flag = c1.x >> 16;
            const int32_t ref_id = (int32_t)c0.y, pos = (int32_t)c0.z, l_seq = (int32_t)c1.y;
            const int mapq = (int)((c0.w >> 8) & 0xFFu);
            const uint32_t aux_len = c1.z;
            const uint64_t blob_at = (uint64_t)c1.w | ((uint64_t)blob_hi << 32);
            gci_rec r;
            r.name_hash = 0; r.contig = -1; r.start = 0; r.end = 0; r.qlen = 0; r.rec_idx = rec + A.rec_idx_base;
            r.mapq = (uint8_t)mapq; r.flags = 0; r.name_len = 0;
            if (gl == 0 && A.name_off) A.name_off[rec] = (kind == PG_KIND_OVERSIZE ? blob_at : page_at + base) + 36u;
            const bool ext = kind == PG_KIND_EXT;
            const uint32_t cig_at = (36u + l_read_name + 15u) & ~15u;
            const uint32_t aux_at = cig_at + (ext ? 16u : (4u * n_cigar + 15u) & ~15u);
            if ((kind & PG_KIND_MALFORMED) || (kind != PG_KIND_OVERSIZE && ((uint64_t)aux_at + aux_len > size || base + size > P))) {
                if (gl == 0) { report(lq.status_in, rec, GCI_E_MALFORMED); A.out[rec] = r; }
                return false;
            }
            if (ref_id < 0 || ref_id >= A.n_ref || (flag & (0x4u | 0x100u | 0x800u)) || mapq < A.map_qual) {   // GCI.py:152-156
                if (gl == 0) A.out[rec] = r;
                return false;
            }
            const int32_t contig = ref_in_lds ? s_ref_sel[ref_id] : A.ref_sel[ref_id];
            // fetch(contig=target) only ever yields records of selected contigs (GCI.py:151, 260)
            if (contig < 0) {
                if (gl == 0) A.out[rec] = r;
                return false;
            }
            if (kind == PG_KIND_OVERSIZE) {
                if (blob_at + 36 > A.total_bytes) { if (gl == 0) { report(lq.status_in, rec, GCI_E_MALFORMED); A.out[rec] = r; } return false; }
                slow_off = blob_at;
                return true;
            }
            // ---- CIGAR base totals (get_cigar_stats()[0], GCI.py:157-162) as the four sums the decision uses (decide4):
            // S, den1 = M + I + S, den2 = M + I + D, rlen = M + D + N (M, = and X together); which of them an op code feeds
            // is one bit of a constant per sum.  The CIGAR is 16-byte aligned and zero padded to 16 bytes (0 = "0M": feeds
            // nothing), so a lane takes whole ds_read_b128 pieces without a mask; lengths below 2^24 (checked on the way: `big`)
            // go through v_mad_u32_u24 into 32-bit sums -- at most 64 operations per lane, no overflow.
            uint32_t sS = 0, sQ = 0, sA = 0, sR = 0, big = 0;
            auto add_op = [&](uint32_t v) {
                const uint32_t op = v & 0xFu, len = v >> 4;
                sS = __umul24(len, __builtin_amdgcn_ubfe(0x010u, op, 1)) + sS;
                sQ = __umul24(len, __builtin_amdgcn_ubfe(0x193u, op, 1)) + sQ;
                sA = __umul24(len, __builtin_amdgcn_ubfe(0x187u, op, 1)) + sA;
                sR = __umul24(len, __builtin_amdgcn_ubfe(0x18Du, op, 1)) + sR;
            };
            if (!ext) {
                const uint8_t* cg = hd + cig_at;
                for (uint32_t p = gl; 4u * p < n_cigar; p += 4) {
                    const uint4 v = *reinterpret_cast<const uint4*>(cg + 16u * p);
                    big |= v.x | v.y | v.z | v.w;
                    add_op(v.x); add_op(v.y); add_op(v.z); add_op(v.w);
                }
            }
            // ---- aux walk (bam_aux_get semantics): first NM; first CG when the CIGAR is htslib's long-CIGAR placeholder
            const uint8_t* ax = hd + aux_at;
            const uint32_t op0 = lds_u32(hd + cig_at);
            const bool want_cg = n_cigar > 0 && pos >= 0 && (op0 & 0xFu) == 4u && (op0 >> 4) == (uint32_t)l_seq;
            bool have_nm = false, nm_bad = false;
            int64_t NM = 0;
            uint32_t cg_q = 0xFFFFFFFFu;
            // Four lanes walk the same tags, and a wave walks as long as its slowest record: the loop that finds NM among
            // fixed-size tags (A c C s S i I f: what aligners write around it) is kept to one tag word, one table byte and a
            // handful of instructions per tag; a Z / H / B value in front of NM, or the CG search, takes the general loop below
            uint32_t q = 0;
            bool general = want_cg;
            while (!general && q + 3 <= aux_len) {
                const uint32_t w = lds_u32(ax + q);                                // tag[2] | type | first value byte
                const uint32_t sz = aux_sz[(w >> 16) & 0xFFu];
                if (sz == 0) { general = true; break; }                            // a variable-size (or unknown) type: general loop
                if (q + 3 + sz > aux_len) break;                                   // value cut off: bam_aux_get stops here
                if ((w & 0xFFFFu) == (uint32_t)('N' | ('M' << 8))) { have_nm = true; nm_bad = !nm_value(ax + q + 2, NM); break; }
                q += 3 + sz;
            }
            for (; general && q + 3 <= aux_len;) {
                const uint32_t w = lds_u32(ax + q);                                // tag[2] | type | first value byte
                const uint32_t tag = w & 0xFFFFu;
                const uint8_t ty = (uint8_t)(w >> 16);
                uint32_t sz = aux_sz[ty];
                if (sz == 0) {
                    if (ty == 'Z' || ty == 'H') {
                        uint32_t e = q + 3;
                        while (e < aux_len && ax[e]) e++;
                        if (e >= aux_len) break;
                        sz = e - (q + 3) + 1;
                    } else if (ty == 'B') {
                        if (q + 8 > aux_len) break;
                        const uint8_t sub = ax[q + 3];
                        const uint32_t es = (sub == 'c' || sub == 'C') ? 1u : (sub == 's' || sub == 'S') ? 2u
                                          : (sub == 'i' || sub == 'I' || sub == 'f') ? 4u : 0u;
                        const uint64_t s64 = 5ull + (uint64_t)lds_u32(ax + q + 4) * es;
                        if (es == 0 || s64 > aux_len) break;
                        sz = (uint32_t)s64;
                    } else break;
                }
                if (q + 3 + sz > aux_len) break;
                if (tag == (uint32_t)('N' | ('M' << 8)) && !have_nm) {
                    have_nm = true;
                    nm_bad = !nm_value(ax + q + 2, NM);
                    if (!want_cg || cg_q != 0xFFFFFFFFu) break;
                }
                if (want_cg && tag == (uint32_t)('C' | ('G' << 8)) && cg_q == 0xFFFFFFFFu) {
                    cg_q = q + 2;
                    if (have_nm) break;
                }
                q += 3 + sz;
            }
            // htslib moves a long CIGAR back from CG:B,I when op0 == <l_seq>S (rare: the sums are redone over the tag's payload)
            bool restored = false;
            if (cg_q != 0xFFFFFFFFu && ax[cg_q] == 'B' && (ax[cg_q + 1] == 'I' || ax[cg_q + 1] == 'i')) {
                const uint32_t cg_len = lds_u32(ax + cg_q + 2);
                if (cg_len >= n_cigar && cg_len < (1u << 29)) {
                    restored = true;
                    sS = sQ = sA = sR = 0; big = 0;
                    for (uint32_t k = gl; k < cg_len; k += 4) { const uint32_t v = lds_u32(ax + cg_q + 6 + 4 * k); big |= v; add_op(v); }
                }
            }
            // ---- query_name: bytes up to the first NUL, 64-bit hash of its 8-byte words -----------------------------------
            const uint8_t* name = hd + 36;
            uint32_t nul = l_read_name;
            for (uint32_t i = gl * 4; i < l_read_name; i += 16) {
                if (has_zero_byte(lds_u32(name + i))) {
                    for (uint32_t b = i; b < i + 4 && b < l_read_name; b++) if (name[b] == 0) { nul = min(nul, b); break; }
                    break;
                }
            }
            nul = min(nul, (uint32_t)__shfl_xor((int)nul, 1, 4));
            nul = min(nul, (uint32_t)__shfl_xor((int)nul, 2, 4));
            const uint32_t name_len = nul;
            uint64_t acc = 0;
            for (uint32_t k = gl; k * 8 < name_len; k += 4) {
                const uint32_t b0 = k * 8;
                uint64_t w;
                __builtin_memcpy(&w, name + b0, 8);                                    // stays inside the page (zero padding behind)
                const uint32_t keep = name_len - b0;                                   // bytes of this word inside the name
                if (keep < 8) w &= (1ull << (8 * keep)) - 1ull;
                acc += gci_hash_word(w, k);
            }
            acc += (uint64_t)__shfl_xor((long long)acc, 1, 4);
            acc += (uint64_t)__shfl_xor((long long)acc, 2, 4);
            r.name_hash = gci_hash_finish(acc, name_len);
            r.name_len = (uint16_t)name_len;
            // in a page: at 16 k + 4, NUL and zero padding up to the CIGAR's 16-byte boundary
            const uint8_t name16 = name_len + 1u == l_read_name ? GCI_REC_NAME16 : 0;
            // an operation of 2^24 bases and more (its sums above would be wrong): the exact chunk path takes the CIGAR -- where
            // it lies in the pages buffer, 16-byte aligned
            big |= (uint32_t)__shfl_xor((int)big, 1, 4);
            big |= (uint32_t)__shfl_xor((int)big, 2, 4);
            const bool wide = (big >> 28) != 0 && !restored;
            if ((ext && !restored) || wide) {   // the parse is complete: only the CIGAR totals + decision are left to k_cigar_chunks / _finish
                LongItem it;
                it.ops_off = ext ? blob_at : page_at + base + cig_at; it.n_ops = n_cigar; it.rec = rec;
                it.nm = have_nm ? (nm_bad ? INT64_MIN : NM) : INT64_MAX;
                it.pos = pos; it.contig = contig; it.l_seq = l_seq; it.n_cigar_field = (int32_t)n_cigar;
                it.mapq = (uint32_t)mapq | ((uint32_t)name16 << 8);
                r.flags = 0;
                if (gl == 0) A.out[rec] = r;
                if (it.ops_off + 4ull * n_cigar > A.total_bytes) { if (gl == 0) report(lq.status_in, rec, GCI_E_MALFORMED); return false; }
                if (!enqueue_long<4>(lq, it, gl) && gl == 0) report(lq.status_in, rec, GCI_E_CAPACITY);
                return false;
            }
            unsigned long long tS = sS, tQ = sQ, tA = sA, tR = sR;
            if ((big >> 28) != 0) {            // a CG:B,I payload with such an operation: exact sums, one operation at a time
                tS = tQ = tA = tR = 0;
                const uint32_t cg_len = lds_u32(ax + cg_q + 2);
                for (uint32_t k = gl; k < cg_len; k += 4) {
                    const uint32_t v = lds_u32(ax + cg_q + 6 + 4 * k), op = v & 0xFu;
                    const unsigned long long len = v >> 4;
                    tS += len & (0ull - ((0x010u >> op) & 1u)); tQ += len & (0ull - ((0x193u >> op) & 1u));
                    tA += len & (0ull - ((0x187u >> op) & 1u)); tR += len & (0ull - ((0x18Du >> op) & 1u));
                }
            }
#define PG_SUM(x) do { x += (unsigned long long)__shfl_xor((long long)x, 1, 4); x += (unsigned long long)__shfl_xor((long long)x, 2, 4); } while (0)
            PG_SUM(tS); PG_SUM(tQ); PG_SUM(tA); PG_SUM(tR);
#undef PG_SUM
            if (gl != 0) return false;
            const int st = decide4(r, (int64_t)tS, (int64_t)tQ, (int64_t)tA, (int64_t)tR, have_nm, nm_bad, NM, pos, contig, l_seq,
                                   n_cigar, mapq, A.mq_cutoff, A.clip_percent, A.iden_percent);
            if (st != GCI_OK) report(lq.status_in, rec, st);
            if (r.flags) r.flags |= name16;
            A.out[rec] = r;
            return false;
        };
        const bool is_slow = on && fast_path();
        // records whose bytes live in the blob (kind 2), one after the other, by the whole wave
        for (unsigned long long m = __ballot(is_slow && gl == 0); m; m &= m - 1ull) {
            const int l = __builtin_ctzll(m);
            const uint32_t rs = (uint32_t)__shfl((int)rec, l, 64);
            const uint64_t so = (uint64_t)__shfl((long long)slow_off, l, 64);
            slow_record<true>(A.buf, A.total_bytes, so, A.ref_sel, rs, t & 63, lq, A.mq_cutoff, A.clip_percent, A.iden_percent, A.rec_idx_base, A.out);
        }
    }
}

// Scratch of one K1 call: [2 x (n_slow u32, pad, n_long u64)][2 x status u64][pad][slow_list u32 x n_rec (+pad)][long items]
// [chunk queue][chunk sums].  A queued item has at least min_item_bytes of CIGAR of its own inside the long_bytes that can
// hold such CIGARs, and a chunk covers 4 * CHUNK_DW bytes: that bounds both queues.
static int k1_prepare(gci_ctx* ctx, uint32_t n_rec, uint64_t long_bytes, uint32_t min_item_bytes, bool has_seq, LongQueue& lq,
                      uint64_t extra_items = 0)
{
    const size_t list_bytes = ((size_t)n_rec * 4 + 15) & ~(size_t)15;
    const uint64_t by_bytes = long_bytes / min_item_bytes + 1 + extra_items;
    const uint32_t cap_items = (uint32_t)(by_bytes < n_rec ? by_bytes : n_rec);
    const uint64_t cap_chunks64 = long_bytes / (4ull * CHUNK_DW) + 2ull * cap_items + 1;
    if (cap_chunks64 > 0xFFFFFFFFull) return GCI_E_INVALID;
    const uint32_t cap_chunks = (uint32_t)cap_chunks64;
    const size_t cap_before = ctx->long_items.cap;
    GCI_TRY(gci_ensure(ctx, ctx->long_items, 64 + list_bytes + (size_t)cap_items * sizeof(LongItem) +
                                             (size_t)cap_chunks * (8 + sizeof(ChunkSums))));
    // two sets of counters [n_slow u32, pad, n_long u64], used alternately: the fast kernel of one call zeroes the set of
    // the next, so no memset is launched per call (a fill costs a whole dependent launch, ~4.6 us)
    if (ctx->long_items.cap != cap_before) {
        HIPCHK(hipMemsetAsync(ctx->long_items.p, 0, 32, ctx->stream));
        HIPCHK(hipMemsetAsync((uint8_t*)ctx->long_items.p + 32, 0xFF, 16, ctx->stream));       // the two status words: no error
        ctx->k1_parity = 0;
    }
    const uint32_t par = ctx->k1_parity;
    lq.n_slow = (uint32_t*)ctx->long_items.p + 4 * par;
    lq.next_counters = (uint32_t*)ctx->long_items.p + 4 * (par ^ 1u);
    lq.status_in = (unsigned long long*)((uint8_t*)ctx->long_items.p + 32) + par;
    lq.next_status = (unsigned long long*)((uint8_t*)ctx->long_items.p + 32) + (par ^ 1u);
    lq.n_long = (unsigned long long*)(lq.n_slow + 2);
    lq.slow_list = (uint32_t*)ctx->long_items.p + 16;
    lq.items = (LongItem*)((uint8_t*)ctx->long_items.p + 64 + list_bytes);
    lq.chunks = (unsigned long long*)((uint8_t*)lq.items + (size_t)cap_items * sizeof(LongItem));
    lq.sums = (ChunkSums*)(lq.chunks + cap_chunks);
    lq.cap_items = cap_items; lq.cap_chunks = cap_chunks;
    lq.has_seq = has_seq ? 1u : 0u;
    return GCI_OK;
}

// the two kernels behind the fast one: long CIGARs chunk by chunk, then their decisions
static int k1_finish(gci_ctx* ctx, const uint8_t* d_bam, uint64_t n_bytes, const LongQueue& lq, int mq_cutoff, double clip_percent,
                     double iden_percent, gci_rec* d_out, uint64_t* d_status)
{
    hipLaunchKernelGGL(k_cigar_chunks, dim3(lq.cap_chunks < 8192u ? (lq.cap_chunks + 3) / 4 : 2048u), dim3(BLOCK), 0, ctx->stream,
                       d_bam, n_bytes, lq, (unsigned long long*)d_status);
    LAUNCHCHK("k_cigar_chunks");
    hipLaunchKernelGGL(k_cigar_finish, dim3(lq.cap_items < 65536u ? (lq.cap_items + BLOCK - 1) / BLOCK : 256u), dim3(BLOCK), 0,
                       ctx->stream, lq, mq_cutoff, clip_percent, iden_percent, d_out, (unsigned long long*)d_status);
    LAUNCHCHK("k_cigar_finish");
    return GCI_OK;
}

static int bam_filter_impl(gci_ctx* ctx, const uint8_t* d_bam, uint64_t n_bytes, const uint64_t* d_rec_off,
                           uint32_t n_rec, const int32_t* d_ref_sel, int32_t n_ref, int map_qual, int mq_cutoff,
                           double clip_percent, double iden_percent, uint32_t rec_idx_base, gci_rec* d_out,
                           uint64_t* d_status, bool has_seq)
{
    if (!ctx || !d_out || !d_status || (n_rec && (!d_bam || !d_rec_off || !d_ref_sel))) return GCI_E_INVALID;
    LongQueue lq;
    GCI_TRY(k1_prepare(ctx, n_rec, n_bytes, 4u * LONG_OPS, has_seq, lq));
    if (n_rec == 0) { HIPCHK(hipMemsetAsync(d_status, 0xFF, 8, ctx->stream)); return GCI_OK; }
    ctx->k1_parity ^= 1u;                   // the fast kernel below zeroes the other set for the next call
    ProfScope _ps(ctx, GCI_PROF_BAM_FILTER);
    const uint32_t per_block = KB / G;
    hipLaunchKernelGGL(k_bam_filter, dim3((n_rec + per_block - 1) / per_block), dim3(KB), 0, ctx->stream, d_bam, n_bytes,
                       d_rec_off, n_rec, d_ref_sel, n_ref, map_qual, mq_cutoff, clip_percent, iden_percent, rec_idx_base,
                       d_out, (unsigned long long*)d_status, lq
                       );
    LAUNCHCHK("k_bam_filter");
    return k1_finish(ctx, d_bam, n_bytes, lq, mq_cutoff, clip_percent, iden_percent, d_out, d_status);
}

// The paged record filter: same outputs and status word as gci_bam_filter; d_name_off (nullable) receives, per record,
// the offset of its query name inside the pages buffer (the join's d_name_off with d_name_base = d_pages, name_delta = 0).
extern "C" int gci_bam_filter_pages(gci_ctx* ctx, const uint8_t* d_pages, uint64_t total_bytes, uint32_t page_bytes, uint32_t n_pages,
                                    uint32_t n_rec, const int32_t* d_ref_sel, int32_t n_ref, int map_qual, int mq_cutoff,
                                    double clip_percent, double iden_percent, uint32_t rec_idx_base, gci_rec* d_out,
                                    uint64_t* d_name_off, uint64_t* d_status)
{
    if (!ctx || !d_out || !d_status || (n_rec && (!d_pages || !d_ref_sel))) return GCI_E_INVALID;
    if (page_bytes < 8192 || page_bytes > GCI_PAGE_MAX_BYTES || (page_bytes & 4095u)) return GCI_E_INVALID;
    if ((uint64_t)n_pages * page_bytes + 16 > total_bytes && n_pages) return GCI_E_INVALID;
    LongQueue lq;
    // What is queued: every passing kind-1 / kind-2 record (its CIGAR lies in the blob, in a piece of its own of a multiple of 16
    // bytes -- a record goes there for its TAGS as well: 60 operations and 900 bytes of MD / cs / SA is kind 1 with a 240-byte
    // piece, so the blob bounds their NUMBER by bytes / 16, not by bytes / 512 as round 3 assumed), a kind-1 record without any
    // operation (no piece), and an inline record with an operation of 2^24 bases or more ("wide": exact sums only in the chunk
    // path).  The last two kinds do not occur in real files; one in 64 records + 1024 may be one before the call refuses
    // (GCI_E_CAPACITY in the status word, never a silent loss).
    const uint64_t blob_bytes = total_bytes - (uint64_t)n_pages * page_bytes;
    GCI_TRY(k1_prepare(ctx, n_rec, blob_bytes, 16u, false, lq, (uint64_t)n_rec / 64 + 1024));
    if (n_rec == 0 || n_pages == 0) { HIPCHK(hipMemsetAsync(d_status, 0xFF, 8, ctx->stream)); return GCI_OK; }
    ctx->k1_parity ^= 1u;
    ProfScope _ps(ctx, GCI_PROF_BAM_FILTER);
    PagesArgs A;
    A.buf = d_pages; A.total_bytes = total_bytes; A.page_bytes = page_bytes; A.n_pages = n_pages; A.n_rec = n_rec;
    A.ref_sel = d_ref_sel; A.n_ref = n_ref; A.map_qual = map_qual; A.mq_cutoff = mq_cutoff;
    A.clip_percent = clip_percent; A.iden_percent = iden_percent; A.rec_idx_base = rec_idx_base; A.out = d_out; A.name_off = d_name_off;
    hipLaunchKernelGGL(k_bam_filter_pages, dim3(n_pages), dim3(PGK), page_bytes, ctx->stream, A, lq);
    LAUNCHCHK("k_bam_filter_pages");
    return k1_finish(ctx, d_pages, total_bytes, lq, mq_cutoff, clip_percent, iden_percent, d_out, d_status);
}

extern "C" int gci_bam_filter(gci_ctx* ctx, const uint8_t* d_bam, uint64_t n_bytes, const uint64_t* d_rec_off,
                              uint32_t n_rec, const int32_t* d_ref_sel, int32_t n_ref, int map_qual, int mq_cutoff,
                              double clip_percent, double iden_percent, uint32_t rec_idx_base, gci_rec* d_out,
                              uint64_t* d_status)
{
    return bam_filter_impl(ctx, d_bam, n_bytes, d_rec_off, n_rec, d_ref_sel, n_ref, map_qual, mq_cutoff, clip_percent,
                           iden_percent, rec_idx_base, d_out, d_status, true);
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
