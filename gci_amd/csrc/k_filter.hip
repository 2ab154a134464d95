// k_filter.hip -- K1: BAM record filter (read_sam, /root/reference/GCI.py:146-169).
#include "gci_ctx.hpp"

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

