/*
 * oracle/gci_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C, single-threaded restatement of the per-record and per-base loops of the
 * reference's hot path (/root/reference/GCI.py), used only by tests/, by
 * __graft_entry__.smoke() and by bench.py's cpu_baseline leg as the checker / the timed CPU
 * baseline.  Nothing under gci_amd/ may import, link or call it.
 *
 * Every function restates the reference's own algorithm literally (slice add per interval,
 * flag-driven run scan, one decimal per line) -- deliberately NOT the difference-array /
 * prefix-sum / stream-compaction formulation the HIP kernels use, so that agreement between
 * the two is evidence and not tautology.
 *
 * Parity status: R6-R13 are pinned by the reference's own example/MH63.* triple and by golden
 * vectors produced by the unmodified reference in the build container (tools/make_golden.py).
 * R1 (BAM decode) is "parity unpinned" at the pysam/htslib boundary: htslib is
 * not in /root/reference, so its documented behaviour (SAM/BAM spec 1.6; htslib sam.c
 * bam_tag2cigar / bam_endpos, pysam AlignedSegment.get_cigar_stats / reference_end /
 * query_length, versions unpinned by README.md:33) is restated here and anchored on
 * hand-assembled records in tests/test_bam_decode.py.
 */
#include <stdint.h>
#include <stddef.h>
#include <string.h>

#define ORC_OK 0
#define ORC_E_NO_NM (-3)        /* reference: KeyError at GCI.py:163 */
#define ORC_E_ZERO_DIV (-4)     /* reference: ZeroDivisionError at GCI.py:165 */
#define ORC_E_BAD_NM_TYPE (-5)  /* NM present but not an integer type */
#define ORC_E_NO_END (-6)       /* reference_end is None (n_cigar == 0): TypeError at GCI.py:305 */
#define ORC_E_MALFORMED (-7)

static int32_t rd_i32(const uint8_t *p) { int32_t v; memcpy(&v, p, 4); return v; }
static uint32_t rd_u32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }
static uint16_t rd_u16(const uint8_t *p) { uint16_t v; memcpy(&v, p, 2); return v; }

/* Size in bytes of an aux value of type `t` starting at p (p points at the value); -1 on error. */
static int64_t aux_value_size(const uint8_t *p, const uint8_t *end, uint8_t t)
{
    switch (t) {
    case 'A': case 'c': case 'C': return 1;
    case 's': case 'S': return 2;
    case 'i': case 'I': case 'f': return 4;
    case 'Z': case 'H': {
        const uint8_t *q = p;
        while (q < end && *q) q++;
        if (q >= end) return -1;
        return (q - p) + 1;
    }
    case 'B': {
        if (p + 5 > end) return -1;
        uint8_t sub = p[0];
        int64_t n = rd_u32(p + 1);
        int64_t es;
        switch (sub) {
        case 'c': case 'C': es = 1; break;
        case 's': case 'S': es = 2; break;
        case 'i': case 'I': case 'f': es = 4; break;
        default: return -1;
        }
        return 5 + n * es;
    }
    default: return -1;
    }
}

/* First aux field with the given two-letter tag (bam_aux_get semantics); returns pointer to the
 * type byte or NULL. */
static const uint8_t *aux_find(const uint8_t *aux, const uint8_t *end, char a, char b)
{
    const uint8_t *p = aux;
    while (p + 3 <= end) {
        uint8_t t = p[2];
        int64_t sz = aux_value_size(p + 3, end, t);
        if (sz < 0 || p + 3 + sz > end) return NULL;
        if (p[0] == (uint8_t)a && p[1] == (uint8_t)b) return p + 2;
        p += 3 + sz;
    }
    return NULL;
}

/*
 * R1: the record filter of read_sam (GCI.py:146-169), over every record of an inflated BAM
 * stream (the union of all fetch() chunks of a contig == all records carrying that refID).
 *
 * ref_sel[refID] = index of the contig among the selected targets, or -1 (not selected).
 * Per record outputs (arrays of n_rec):
 *   pass[i]   1 iff the record reaches GCI.py:166
 *   hq[i]     1 iff pass and mapq >= mq_cutoff            (GCI.py:167-168)
 *   contig[i], start[i], end[i], qlen[i]                  the tuple stored at GCI.py:166
 *   name_off[i], name_len[i]                              query_name bytes in the stream
 * Returns ORC_OK, or a negative status with *bad_rec = index of the first record at which the
 * reference would have raised.
 */
int orc_bam_filter(const uint8_t *bam, uint64_t n_bytes, const uint64_t *rec_off, uint32_t n_rec,
                   const int32_t *ref_sel, int32_t n_ref,
                   int map_qual, int mq_cutoff, double clip_percent, double iden_percent,
                   uint8_t *pass, uint8_t *hq, int32_t *contig, int32_t *start, int32_t *end,
                   int32_t *qlen, uint64_t *name_off, uint32_t *name_len, uint32_t *bad_rec, int has_seq)
{
    for (uint32_t i = 0; i < n_rec; i++) {
        pass[i] = 0; hq[i] = 0; contig[i] = -1; start[i] = 0; end[i] = 0; qlen[i] = 0;
        uint64_t off = rec_off[i];
        if (off + 36 > n_bytes) { *bad_rec = i; return ORC_E_MALFORMED; }
        const uint8_t *r = bam + off;
        int32_t block_size = rd_i32(r);
        if (block_size < 32 || off + 4 + (uint64_t)block_size > n_bytes) { *bad_rec = i; return ORC_E_MALFORMED; }
        int32_t ref_id = rd_i32(r + 4);
        int32_t pos = rd_i32(r + 8);
        uint32_t l_read_name = r[12];
        int mapq = r[13];
        uint32_t n_cigar = rd_u16(r + 16);
        uint32_t flag = rd_u16(r + 18);
        int32_t l_seq = rd_i32(r + 20);
        const uint8_t *name = r + 36;
        const uint8_t *rec_end = r + 4 + block_size;
        const uint8_t *cig = name + l_read_name;
        /* has_seq == 0: a heads stream (include/gci_hip.h, gci_bam_heads): the same records with the SEQ and QUAL
         * bytes cut out (l_seq keeps its value) -- read_sam never looks at them (GCI.py:146-169) */
        const uint8_t *aux = cig + 4 * (uint64_t)n_cigar + (has_seq ? ((uint64_t)l_seq + 1) / 2 + (uint64_t)l_seq : 0);
        if (aux > rec_end) { *bad_rec = i; return ORC_E_MALFORMED; }
        uint32_t nl = 0;
        while (nl < l_read_name && name[nl]) nl++;
        name_off[i] = off + 36;
        name_len[i] = nl;

        /* fetch(contig=target): only records on a selected contig are ever seen (GCI.py:151,260) */
        if (ref_id < 0 || ref_id >= n_ref || ref_sel[ref_id] < 0) continue;
        /* GCI.py:152-156 */
        if (flag & 0x4) continue;
        if (flag & 0x100) continue;
        if (flag & 0x800) continue;
        if (mapq < map_qual) continue;

        /* htslib restores a CIGAR parked in the CG:B,I tag before pysam sees the record */
        const uint8_t *ops = cig;
        uint64_t n_ops = n_cigar;
        if (n_cigar > 0 && pos >= 0) {
            uint32_t op0 = rd_u32(cig);
            if ((op0 & 0xF) == 4 && (int64_t)(op0 >> 4) == (int64_t)l_seq) {
                const uint8_t *cg = aux_find(aux, rec_end, 'C', 'G');
                if (cg && cg[0] == 'B' && (cg[1] == 'I' || cg[1] == 'i')) {
                    uint32_t cg_len = rd_u32(cg + 2);
                    if (cg_len >= n_cigar && cg_len < (1u << 29)) { ops = cg + 6; n_ops = cg_len; }
                }
            }
        }
        /* get_cigar_stats()[0]: base totals per op code (GCI.py:157-162) */
        int64_t tot[16];
        memset(tot, 0, sizeof tot);
        for (uint64_t k = 0; k < n_ops; k++) {
            uint32_t v = rd_u32(ops + 4 * k);
            tot[v & 0xF] += (int64_t)(v >> 4);
        }
        int64_t M = tot[0], I = tot[1], D = tot[2], S = tot[4], EQ = tot[7], X = tot[8];
        /* get_tag('NM') (GCI.py:163) */
        const uint8_t *nmp = aux_find(aux, rec_end, 'N', 'M');
        if (!nmp) { *bad_rec = i; return ORC_E_NO_NM; }
        int64_t NM;
        switch (nmp[0]) {
        case 'c': NM = (int8_t)nmp[1]; break;
        case 'C': NM = nmp[1]; break;
        case 's': NM = (int16_t)rd_u16(nmp + 1); break;
        case 'S': NM = rd_u16(nmp + 1); break;
        case 'i': NM = rd_i32(nmp + 1); break;
        case 'I': NM = rd_u32(nmp + 1); break;
        default: *bad_rec = i; return ORC_E_BAD_NM_TYPE;
        }
        int64_t mm = NM - (I + D);                                    /* GCI.py:164 */
        int64_t den1 = M + EQ + X + I + S, den2 = M + EQ + X + I + D;
        if (den1 == 0) { *bad_rec = i; return ORC_E_ZERO_DIV; }
        /* Python `and` short-circuits: the second division only happens if the first test passes */
        if (!((double)S / (double)den1 <= clip_percent)) continue;     /* GCI.py:165 */
        if (den2 == 0) { *bad_rec = i; return ORC_E_ZERO_DIV; }
        if (!((double)(M + EQ + X - mm) / (double)den2 >= iden_percent)) continue;
        if (n_cigar == 0) { *bad_rec = i; return ORC_E_NO_END; }
        int64_t rlen = tot[0] + tot[2] + tot[3] + tot[7] + tot[8];    /* M D N = X consume reference */
        pass[i] = 1;
        contig[i] = ref_sel[ref_id];
        start[i] = pos;
        end[i] = (int32_t)((int64_t)pos + (rlen > 0 ? rlen : 1));       /* bam_endpos */
        qlen[i] = l_seq;                                               /* query_length == l_qseq */
        if (mapq >= mq_cutoff) hq[i] = 1;
    }
    return ORC_OK;
}

/* Python slice-bound normalisation for a sequence of length L (what numpy applies to
 * depths[target][start:end+1] at GCI.py:306 and depths[target][a:b] = 0 at GCI.py:328). */
static int64_t slice_bound(int64_t v, int64_t L)
{
    if (v < 0) { v += L; if (v < 0) v = 0; }
    else if (v > L) v = L;
    return v;
}

/* R6: for each surviving interval, depths[start+fl : end-fl+1] += 1 (GCI.py:302-306). */
void orc_depth_build(int64_t *depth, int64_t L, const int64_t *s, const int64_t *e, uint64_t n, int64_t flank)
{
    for (uint64_t i = 0; i < n; i++) {
        int64_t a = slice_bound(s[i] + flank, L);
        int64_t b = slice_bound(e[i] - flank + 1, L);
        for (int64_t p = a; p < b; p++) depth[p] += 1;
    }
}

/* R8: depths[a:b] = 0 (GCI.py:328). */
void orc_zero_range(int64_t *depth, int64_t L, int64_t a, int64_t b)
{
    a = slice_bound(a, L);
    b = slice_bound(b, L);
    for (int64_t p = a; p < b; p++) depth[p] = 0;
}

/* R9: max(hifi, nano) per base (GCI.py:350). */
void orc_max2(const int64_t *a, const int64_t *b, int64_t L, int64_t *out)
{
    for (int64_t i = 0; i < L; i++) out[i] = a[i] > b[i] ? a[i] : b[i];
}

/* R10: collapse_depth_range for one contig (GCI.py:369-390), flag for flag.
 * Writes (start, end) pairs; returns the number of pairs found (may exceed cap: caller retries). */
uint64_t orc_collapse(const int64_t *depth, int64_t chr_len, double leftmost, double rightmost,
                      int64_t flank_len, int64_t start_pos, int64_t *pairs, uint64_t cap)
{
    uint64_t n = 0;
    int start_flag = 0, end_flag = 1;
    int64_t start = 0;
    /* depth_list[flank_len : chr_len - flank_len] with Python slice normalisation */
    int64_t lo = slice_bound(flank_len, chr_len), hi = slice_bound(chr_len - flank_len, chr_len);
    for (int64_t p = lo, i = 0; p < hi; p++, i++) {
        double d = (double)depth[p];
        if (leftmost < d && d <= rightmost) {
            if (start_flag == 0) { start = i + flank_len; start_flag = 1; end_flag = 0; }
            if (i == chr_len - flank_len * 2 - 1) {
                if (n < cap) { pairs[2 * n] = start + start_pos; pairs[2 * n + 1] = i + flank_len + 1 + start_pos; }
                n++;
            }
        } else {
            if (end_flag == 0) {
                if (i > flank_len) {
                    if (n < cap) { pairs[2 * n] = start + start_pos; pairs[2 * n + 1] = i + flank_len + start_pos; }
                    n++;
                }
                end_flag = 1;
                start_flag = 0;
            }
        }
    }
    return n;
}

/* R7: one decimal per line (GCI.py:115-117).  `out` must hold 21 bytes per element worst case;
 * returns bytes written. */
uint64_t orc_depth_text(const int64_t *depth, int64_t L, uint8_t *out)
{
    uint8_t *p = out;
    char tmp[24];
    for (int64_t i = 0; i < L; i++) {
        int64_t v = depth[i];
        uint64_t u = v < 0 ? (uint64_t)(-v) : (uint64_t)v;
        int k = 0;
        do { tmp[k++] = (char)('0' + u % 10); u /= 10; } while (u);
        if (v < 0) *p++ = '-';
        while (k) *p++ = (uint8_t)tmp[--k];
        *p++ = '\n';
    }
    return (uint64_t)(p - out);
}

/* R15: integer sum feeding np.mean (GCI.py:862-868). */
int64_t orc_sum(const int64_t *depth, int64_t L)
{
    int64_t s = 0;
    for (int64_t i = 0; i < L; i++) s += depth[i];
    return s;
}

/* Test helper (inverse of orc_depth_text): parse the decimal lines of ONE contig body (no '>'
 * line) into int64.  Returns the number of values parsed, or -1 on a malformed byte. */
int64_t orc_parse_depth_lines(const uint8_t *txt, uint64_t n, int64_t *out, int64_t cap)
{
    int64_t k = 0, v = 0;
    int have = 0, neg = 0;
    for (uint64_t i = 0; i < n; i++) {
        uint8_t c = txt[i];
        if (c >= '0' && c <= '9') { v = v * 10 + (c - '0'); have = 1; }
        else if (c == '-' && !have) neg = 1;
        else if (c == '\n') {
            if (have) { if (k < cap) out[k] = neg ? -v : v; k++; }
            v = 0; have = 0; neg = 0;
        } else return -1;
    }
    if (have) { if (k < cap) out[k] = neg ? -v : v; k++; }
    return k;
}
