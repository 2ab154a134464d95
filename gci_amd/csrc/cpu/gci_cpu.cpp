// gci_cpu.cpp -- libgci_cpu.so: the function seams of include/gci_hip.h on HOST memory and host threads
// ("same header compiled twice"; 8(d)(ii): the CPU baseline behind the same C-ABI).
//
// Same signatures, same structs, same status words as libgci_hip.so for the seam set
//   gci_bam_filter   gci_name_join   gci_depth_build   gci_gap_mask   gci_max2   gci_issue_scan(_windows)
//   gci_depth_text_size / _write   gci_depth_sum   gci_range_sums
// and the context / layout / memory calls around them.  "d_" pointers are host pointers here (gci_malloc is an aligned
// malloc, gci_memcpy_* a memcpy, gci_sync a no-op): a host written against the header runs on either library.  What a GPU
// makes worthwhile -- record pages, the partitioned join, the tile build with its fused by-products, the BGZF / DEFLATE
// kernels, the PAF filter, the multi-GPU routing -- is not restated here.
//
// The code is this repository's own statement of the reference's rules (citations into /root/reference/GCI.py), written for
// threads and caches; it shares nothing with oracle/ (the tests hold the two against each other).
#include "../../../include/gci_hip.h"
#include "../gci_common.h"

#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <string>
#include <thread>
#include <vector>

struct gci_ctx {
    int threads = 1;
    bool heads = false;                     // gci_bam_filter reads a HEADS stream (records without SEQ / QUAL, gci_bam_heads)
    std::string err;
    int32_t n_contigs = 0;
    std::vector<int64_t> len, off;
    int64_t total = 0;
};

namespace {

// fn(t, n_threads) on every thread
template <typename F>
void on_threads(int threads, F fn)
{
    if (threads <= 1) { fn(0, 1); return; }
    std::vector<std::thread> pool;
    pool.reserve((size_t)threads - 1);
    for (int t = 1; t < threads; t++) pool.emplace_back([&fn, t, threads] { fn(t, threads); });
    fn(0, threads);
    for (auto& th : pool) th.join();
}

// [0, n) in blocks of `grain`, handed out by a counter: fn(lo, hi)
template <typename F>
void parallel_blocks(int threads, uint64_t n, uint64_t grain, F fn)
{
    if (n == 0) return;
    std::atomic<uint64_t> next{0};
    on_threads(threads, [&](int, int) {
        for (;;) {
            const uint64_t lo = next.fetch_add(grain);
            if (lo >= n) break;
            fn(lo, std::min(n, lo + grain));
        }
    });
}

inline void report(std::atomic<uint64_t>& status, uint32_t rec, int code)
{
    const uint64_t v = ((uint64_t)rec << 8) | (uint64_t)(uint8_t)(-code);
    uint64_t cur = status.load();
    while (v < cur && !status.compare_exchange_weak(cur, v)) {}
}

inline uint32_t rd16(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }
inline uint32_t rd32(const uint8_t* p) { return rd16(p) | (rd16(p + 2) << 16); }

// size of an aux value of type t at p; -1: malformed / past the end of the record
int64_t aux_value_size(const uint8_t* p, const uint8_t* end, uint8_t t)
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
        const int64_t n = rd32(p + 1);
        const int64_t es = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : (sub == 'i' || sub == 'I' || sub == 'f') ? 4 : -1;
        return es < 0 ? -1 : 5 + n * es;
    }
    default: return -1;
    }
}

bool nm_value(const uint8_t* t, int64_t& NM)
{
    switch (t[0]) {
    case 'c': NM = (int8_t)t[1]; return true;
    case 'C': NM = t[1]; return true;
    case 's': NM = (int16_t)rd16(t + 1); return true;
    case 'S': NM = rd16(t + 1); return true;
    case 'i': NM = (int32_t)rd32(t + 1); return true;
    case 'I': NM = rd32(t + 1); return true;
    default: return false;
    }
}

uint64_t name_hash(const uint8_t* name, uint32_t len)
{
    uint64_t acc = 0;
    for (uint32_t k = 0; k * 8 < len; k++) {
        uint64_t w = 0;
        const uint32_t n = std::min(8u, len - 8 * k);
        memcpy(&w, name + 8 * k, n);
        acc += gci_hash_word(w, k);
    }
    return gci_hash_finish(acc, len);
}

// read_sam on one record (GCI.py:146-169); the record's compact form into r, GCI_OK also when it is filtered
int filter_record(const uint8_t* bam, uint64_t n_bytes, uint64_t off, const int32_t* ref_sel, int32_t n_ref, int map_qual, int mq_cutoff,
                  double clip_percent, double iden_percent, bool has_seq, gci_rec& r)
{
    if (off + 36 > n_bytes) return GCI_E_MALFORMED;
    const uint8_t* p = bam + off;
    const int32_t block_size = (int32_t)rd32(p), ref_id = (int32_t)rd32(p + 4), pos = (int32_t)rd32(p + 8);
    const uint32_t l_read_name = p[12];
    const int mapq = p[13];
    const uint32_t n_cigar = rd16(p + 16), flag = rd16(p + 18);
    const int32_t l_seq = (int32_t)rd32(p + 20);
    const uint64_t rec_end = off + 4 + (uint64_t)(uint32_t)block_size;
    const uint64_t aux_off = off + 36 + l_read_name + 4ull * n_cigar + (has_seq ? (((uint64_t)(uint32_t)l_seq + 1) >> 1) + (uint64_t)(uint32_t)l_seq : 0ull);
    r.mapq = (uint8_t)mapq;
    if (block_size < 32 || rec_end > n_bytes || l_seq < 0 || aux_off > rec_end) return GCI_E_MALFORMED;
    // mapped, primary, MAPQ (GCI.py:152-156); fetch(contig = target) only yields records of selected contigs (:151, :260)
    if (ref_id < 0 || ref_id >= n_ref || (flag & (0x4u | 0x100u | 0x800u)) || mapq < map_qual) return GCI_OK;
    const int32_t contig = ref_sel[ref_id];
    if (contig < 0) return GCI_OK;
    const uint8_t* name = p + 36;
    const uint8_t* end = bam + rec_end;
    uint32_t name_len = 0;
    while (name_len < l_read_name && name[name_len]) name_len++;
    r.name_hash = name_hash(name, name_len);
    r.name_len = (uint16_t)name_len;
    // the first NM and the first CG tag (bam_aux_get)
    const uint8_t* nm_p = nullptr;
    const uint8_t* cg_p = nullptr;
    for (const uint8_t* q = bam + aux_off; q + 3 <= end;) {
        const int64_t sz = aux_value_size(q + 3, end, q[2]);
        if (sz < 0 || q + 3 + sz > end) break;
        if (q[0] == 'N' && q[1] == 'M' && !nm_p) nm_p = q + 2;
        if (q[0] == 'C' && q[1] == 'G' && !cg_p) cg_p = q + 2;
        q += 3 + sz;
    }
    int64_t NM = 0;
    const bool nm_bad = nm_p ? !nm_value(nm_p, NM) : false;
    // htslib puts a CIGAR of more than 65535 operations back from CG:B,I when op0 is <l_seq>S
    const uint8_t* ops = name + l_read_name;
    uint32_t n_ops = n_cigar;
    if (n_cigar > 0 && pos >= 0) {
        const uint32_t op0 = rd32(ops);
        if ((op0 & 0xF) == 4 && (op0 >> 4) == (uint32_t)l_seq && cg_p && cg_p[0] == 'B' && (cg_p[1] == 'I' || cg_p[1] == 'i')) {
            const uint32_t cg_len = rd32(cg_p + 2);
            if (cg_len >= n_cigar && cg_len < (1u << 29)) { ops = cg_p + 6; n_ops = cg_len; }
        }
    }
    // base totals per operation (get_cigar_stats()[0], GCI.py:157-162): M, = and X together
    int64_t M = 0, I = 0, D = 0, N = 0, S = 0;
    for (uint32_t k = 0; k < n_ops; k++) {
        const uint32_t v = rd32(ops + 4ull * k);
        const int64_t len = v >> 4;
        switch (v & 0xF) {
        case 0: case 7: case 8: M += len; break;
        case 1: I += len; break;
        case 2: D += len; break;
        case 3: N += len; break;
        case 4: S += len; break;
        default: break;
        }
    }
    if (!nm_p) return GCI_E_NO_NM;                                          // get_tag('NM'): KeyError
    if (nm_bad) return GCI_E_BAD_NM_TYPE;
    const int64_t den1 = M + I + S, den2 = M + I + D, rlen = M + D + N;
    if (den1 == 0) return GCI_E_ZERO_DIV;
    if (!((double)S / (double)den1 <= clip_percent)) return GCI_OK;         // GCI.py:165; `and` short-circuits
    if (den2 == 0) return GCI_E_ZERO_DIV;
    if (!((double)(den2 - NM) / (double)den2 >= iden_percent)) return GCI_OK;
    if (n_cigar == 0) return GCI_E_NO_END;
    r.contig = contig;
    r.start = pos;
    r.end = (int32_t)((int64_t)pos + (rlen > 0 ? rlen : 1));               // bam_endpos
    r.qlen = l_seq;
    r.flags = GCI_REC_PASS | (mapq >= mq_cutoff ? GCI_REC_HQ : 0);         // GCI.py:166-168
    return GCI_OK;
}

int need_layout(gci_ctx* ctx) { return !ctx ? GCI_E_INVALID : ctx->n_contigs <= 0 ? GCI_E_NO_LAYOUT : GCI_OK; }

int issue_scan(gci_ctx* ctx, const int32_t* depth, const gci_window* win, uint32_t n_win, double lo, double hi, uint64_t* keys, uint32_t cap,
               uint32_t* n_keys)
{
    if (!depth || !n_keys || (cap && !keys)) return GCI_E_INVALID;
    std::atomic<uint32_t> n{0};
    // a window in pieces of 4 M elements: a piece reports the boundaries inside it, its neighbours' edges are matched up by looking
    // one element back
    struct Piece { uint32_t w; int64_t lo, hi; };
    std::vector<Piece> pieces;
    for (uint32_t w = 0; w < n_win; w++)
        for (int64_t a = win[w].begin; a < win[w].end; a += (int64_t)1 << 22) pieces.push_back({w, a, std::min(win[w].end, a + ((int64_t)1 << 22))});
    parallel_blocks(ctx->threads, pieces.size(), 1, [&](uint64_t a, uint64_t) {
        const Piece pc = pieces[a];
        const gci_window W = win[pc.w];
        auto emit = [&](bool is_end, int64_t at) {
            const uint32_t k = n.fetch_add(1);
            if (k < cap) keys[k] = ((uint64_t)pc.w << 33) | ((uint64_t)(at - W.begin) << 1) | (is_end ? 1u : 0u);
        };
        auto in_range = [&](int64_t i) { const double d = (double)depth[i]; return lo < d && d <= hi; };
        bool in = pc.lo > W.begin ? in_range(pc.lo - 1) : false;
        for (int64_t i = pc.lo; i < pc.hi; i++) {
            const bool now = in_range(i);
            if (now != in) { emit(in, i); in = now; }
        }
        if (in && pc.hi == W.end) emit(true, W.end);
    });
    *n_keys = n.load();
    return GCI_OK;
}

uint32_t decimal_width(int32_t v)
{
    uint32_t w = v < 0 ? 1u : 0u;
    uint64_t a = v < 0 ? (uint64_t)(-(int64_t)v) : (uint64_t)v;
    do { w++; a /= 10; } while (a);
    return w;
}

}  // namespace

extern "C" {

int gci_abi_version(void) { return GCI_ABI_VERSION; }

int gci_ctx_create(int, void*, int, gci_ctx** out)
{
    if (!out) return GCI_E_INVALID;
    gci_ctx* c = new (std::nothrow) gci_ctx;
    if (!c) return GCI_E_NOMEM;
    const char* e = getenv("GCI_CPU_THREADS");
    const int hw = (int)std::thread::hardware_concurrency();
    c->threads = e && atoi(e) > 0 ? atoi(e) : (hw > 0 ? hw : 1);
    *out = c;
    return GCI_OK;
}
int gci_ctx_destroy(gci_ctx* ctx) { delete ctx; return GCI_OK; }
int gci_sync(gci_ctx* ctx) { return ctx ? GCI_OK : GCI_E_INVALID; }
/* libgci_cpu.so only.  "threads": host threads of the calls on this context (value 0 asks, > 0 sets); "heads": != 0 makes
 * gci_bam_filter read a heads stream (the records without SEQ / QUAL: gci_bam_heads).  Returns the value in force, -1: no such option. */
int gci_cpu_option(gci_ctx* ctx, const char* name, int value)
{
    if (!ctx || !name) return -1;
    if (!strcmp(name, "threads")) { if (value > 0) ctx->threads = value; return ctx->threads; }
    if (!strcmp(name, "heads")) { ctx->heads = value != 0; return ctx->heads ? 1 : 0; }
    return -1;
}
const char* gci_strerror(int status)
{
    switch (status) {
    case GCI_OK: return "ok";
    case GCI_E_INVALID: return "invalid argument";
    case GCI_E_HIP: return "HIP runtime error";
    case GCI_E_NO_NM: return "record has no NM tag";
    case GCI_E_ZERO_DIV: return "division by zero";
    case GCI_E_BAD_NM_TYPE: return "NM tag is not an integer";
    case GCI_E_NO_END: return "record has no reference end";
    case GCI_E_MALFORMED: return "malformed record";
    case GCI_E_CAPACITY: return "output buffer too small";
    case GCI_E_NOMEM: return "out of memory";
    case GCI_E_NO_LAYOUT: return "gci_layout_set has not been called";
    default: return "unknown status";
    }
}
const char* gci_last_error(gci_ctx* ctx) { return ctx ? ctx->err.c_str() : ""; }
int gci_malloc(gci_ctx* ctx, size_t bytes, void** out)
{
    if (!ctx || !out) return GCI_E_INVALID;
    void* p = nullptr;
    if (posix_memalign(&p, 64, bytes ? bytes : 64) != 0) return GCI_E_NOMEM;
    *out = p;
    return GCI_OK;
}
int gci_free(gci_ctx* ctx, void* p) { if (!ctx) return GCI_E_INVALID; free(p); return GCI_OK; }
int gci_memcpy_h2d(gci_ctx* ctx, void* dst, const void* src, size_t bytes) { if (!ctx) return GCI_E_INVALID; memcpy(dst, src, bytes); return GCI_OK; }
int gci_memcpy_d2h(gci_ctx* ctx, void* dst, const void* src, size_t bytes) { if (!ctx) return GCI_E_INVALID; memcpy(dst, src, bytes); return GCI_OK; }
int gci_memset(gci_ctx* ctx, void* dst, int byte, size_t bytes) { if (!ctx) return GCI_E_INVALID; memset(dst, byte, bytes); return GCI_OK; }

/* ---- layout: as libgci_hip.so lays a track out (contigs at multiples of GCI_TILE elements), so that tracks are interchangeable ---- */
int gci_layout_set(gci_ctx* ctx, int32_t n_contigs, const int64_t* h_lengths)
{
    if (!ctx || n_contigs <= 0 || !h_lengths) return GCI_E_INVALID;
    ctx->len.assign(h_lengths, h_lengths + n_contigs);
    ctx->off.assign((size_t)n_contigs, 0);
    int64_t at = 0;
    for (int32_t c = 0; c < n_contigs; c++) {
        if (h_lengths[c] < 0) return GCI_E_INVALID;
        ctx->off[(size_t)c] = at;
        at += (h_lengths[c] + GCI_TILE - 1) / GCI_TILE * GCI_TILE;
    }
    ctx->total = at > 0 ? at : GCI_TILE;
    ctx->n_contigs = n_contigs;
    return GCI_OK;
}
int64_t gci_layout_total(gci_ctx* ctx) { return ctx ? ctx->total : 0; }
int gci_layout_offsets(gci_ctx* ctx, int64_t* h_offsets)
{
    const int st = need_layout(ctx);
    if (st) return st;
    if (!h_offsets) return GCI_E_INVALID;
    memcpy(h_offsets, ctx->off.data(), (size_t)ctx->n_contigs * sizeof(int64_t));
    return GCI_OK;
}

/* ---- R1 ---------------------------------------------------------------------------------------------------------------- */
uint64_t gci_name_hash(const uint8_t* h_name, uint32_t len) { return name_hash(h_name, len); }

int gci_decode_status(uint64_t word, uint32_t* rec_idx)
{
    if (word == ~0ull) { if (rec_idx) *rec_idx = 0; return GCI_OK; }
    if (rec_idx) *rec_idx = (uint32_t)(word >> 8);
    return -(int)(word & 0xFFu);
}

int gci_bam_filter(gci_ctx* ctx, const uint8_t* bam, uint64_t n_bytes, const uint64_t* rec_off, uint32_t n_rec, const int32_t* ref_sel,
                   int32_t n_ref, int map_qual, int mq_cutoff, double clip_percent, double iden_percent, uint32_t rec_idx_base, gci_rec* out,
                   uint64_t* status)
{
    if (!ctx || !status || (n_rec && (!bam || !rec_off || !ref_sel || !out))) return GCI_E_INVALID;
    std::atomic<uint64_t> st{~0ull};
    parallel_blocks(ctx->threads, n_rec, 4096, [&](uint64_t lo, uint64_t hi) {
        for (uint64_t i = lo; i < hi; i++) {
            gci_rec r;
            r.name_hash = 0; r.contig = -1; r.start = 0; r.end = 0; r.qlen = 0; r.rec_idx = (uint32_t)i + rec_idx_base; r.mapq = 0; r.flags = 0;
            r.name_len = 0;
            const int code = filter_record(bam, n_bytes, rec_off[i], ref_sel, n_ref, map_qual, mq_cutoff, clip_percent, iden_percent, !ctx->heads, r);
            if (code != GCI_OK) report(st, (uint32_t)i, code);
            out[i] = r;
        }
    });
    *status = st.load();
    return GCI_OK;
}

/* ---- R5: the cross-file join (GCI.py:272-301), name by name ------------------------------------------------------------------ */
int gci_name_join(gci_ctx* ctx, const gci_join_file* files, int n_files, double ovlp_percent, const int32_t* contig_map, gci_ivl* out,
                  uint32_t cap, uint32_t* n_out, uint64_t* status)
{
    if (!ctx || !files || n_files < 1 || n_files > GCI_MAX_JOIN_FILES || !n_out || !status || (cap && !out)) return GCI_E_INVALID;
    struct Ref { uint64_t hash; const uint8_t* name; uint32_t pos; uint16_t len; uint8_t file; uint8_t hq; const gci_rec* rec; };
    // the passing records of every file (read_sam keeps nothing else, GCI.py:166), cut into parts by their hash: a name lives in one part
    const int parts = std::max(1, std::min(256, ctx->threads * 4));
    std::vector<std::vector<Ref>> by_part((size_t)parts);
    {
        std::vector<std::vector<std::vector<Ref>>> local((size_t)ctx->threads, std::vector<std::vector<Ref>>((size_t)parts));
        for (int f = 0; f < n_files; f++) {
            const gci_join_file& F = files[f];
            if (F.n_recs && (!F.d_recs || !F.d_name_base || !F.d_name_off)) return GCI_E_INVALID;
            std::atomic<uint64_t> next{0};
            on_threads(ctx->threads, [&](int t, int) {
                auto& mine = local[(size_t)t];
                for (;;) {
                    const uint64_t lo = next.fetch_add(8192);
                    if (lo >= F.n_recs) break;
                    const uint64_t hi = std::min<uint64_t>(F.n_recs, lo + 8192);
                    for (uint64_t i = lo; i < hi; i++) {
                        const gci_rec& r = F.d_recs[i];
                        if (!(r.flags & GCI_REC_PASS)) continue;
                        mine[(size_t)((r.name_hash >> 33) % (uint64_t)parts)].push_back(
                            Ref{r.name_hash, F.d_name_base + F.d_name_off[i] + F.name_delta, (uint32_t)i, r.name_len, (uint8_t)f,
                                (uint8_t)((r.flags & GCI_REC_HQ) ? 1 : 0), &r});
                    }
                }
            });
        }
        on_threads(ctx->threads, [&](int t, int nt) {
            for (int p = t; p < parts; p += nt) {
                size_t n = 0;
                for (auto& l : local) n += l[(size_t)p].size();
                by_part[(size_t)p].reserve(n);
                for (auto& l : local) by_part[(size_t)p].insert(by_part[(size_t)p].end(), l[(size_t)p].begin(), l[(size_t)p].end());
            }
        });
    }
    std::atomic<uint32_t> n{0};
    std::atomic<uint64_t> st{~0ull};
    parallel_blocks(ctx->threads, (uint64_t)parts, 1, [&](uint64_t p, uint64_t) {
        auto& v = by_part[(size_t)p];
        // same name together (hash, then the bytes), files in order, positions in order
        std::sort(v.begin(), v.end(), [](const Ref& a, const Ref& b) {
            if (a.hash != b.hash) return a.hash < b.hash;
            if (a.len != b.len) return a.len < b.len;
            const int c = memcmp(a.name, b.name, a.len);
            if (c) return c < 0;
            if (a.file != b.file) return a.file < b.file;
            return a.pos < b.pos;
        });
        for (size_t a = 0; a < v.size();) {
            size_t b = a + 1;
            while (b < v.size() && v[b].hash == v[a].hash && v[b].len == v[a].len && memcmp(v[b].name, v[a].name, v[a].len) == 0) b++;
            // per file the LAST record of the name (dict semantics, GCI.py:166, 269); high quality if ANY passing record of it was (:167-168)
            const gci_rec* last[GCI_MAX_JOIN_FILES] = {nullptr};
            bool hq = false;
            int present = 0;
            for (size_t k = a; k < b; k++) { if (!last[v[k].file]) present++; last[v[k].file] = v[k].rec; hq = hq || v[k].hq; }
            bool have = false;
            int32_t contig = 0, s = 0, e = 0;
            if (n_files == 1) { have = true; contig = last[0]->contig; s = last[0]->start; e = last[0]->end; }
            else {
                const bool in_final = hq || present == n_files;                         // GCI.py:277-279
                if (last[0] && in_final) { have = true; contig = last[0]->contig; s = last[0]->start; e = last[0]->end; }
                for (int f = 1; f < n_files; f++) {
                    const gci_rec* r = last[f];
                    if (!r) continue;
                    if (have) {
                        if (r->contig == contig) {
                            const int64_t ovlp = (int64_t)std::min(r->end, e) - (int64_t)std::max(r->start, s);
                            if (r->qlen == 0) { report(st, r->rec_idx, GCI_E_ZERO_DIV); have = false; }     // ZeroDivisionError, GCI.py:292
                            else if ((double)ovlp / (double)r->qlen < ovlp_percent) have = false;
                            else { s = std::max(r->start, s); e = std::min(r->end, e); }
                        } else have = false;                                             // GCI.py:296-297
                    } else if (hq) { have = true; contig = r->contig; s = r->start; e = r->end; }   // GCI.py:298-299
                }
            }
            if (have) {
                const int32_t c = contig_map ? contig_map[contig] : contig;
                if (c >= 0) {
                    const uint32_t k = n.fetch_add(1);
                    if (k < cap) out[k] = gci_ivl{c, s, e, 0};
                }
            }
            a = b;
        }
    });
    *n_out = n.load();
    *status = st.load();
    return GCI_OK;
}

/* ---- R6: depths[c][s + flank : e - flank + 1] += 1 (GCI.py:302-306), NumPy slice semantics ----------------------------------- */
int gci_depth_build(gci_ctx* ctx, const gci_ivl* ivl, const uint32_t* d_n, uint32_t max_n, int flank, int32_t* depth)
{
    const int st = need_layout(ctx);
    if (st) return st;
    if (!depth || (max_n && !ivl)) return GCI_E_INVALID;
    const uint32_t n = d_n ? std::min(*d_n, max_n) : max_n;
    // the track as a difference array (atomic +1 / -1 at the slice's ends), then a running sum per contig, blocks side by side
    parallel_blocks(ctx->threads, (uint64_t)ctx->total, (uint64_t)1 << 22, [&](uint64_t lo, uint64_t hi) { memset(depth + lo, 0, (hi - lo) * sizeof(int32_t)); });
    parallel_blocks(ctx->threads, n, 1 << 16, [&](uint64_t lo, uint64_t hi) {
        for (uint64_t i = lo; i < hi; i++) {
            const gci_ivl v = ivl[i];
            if (v.contig < 0 || v.contig >= ctx->n_contigs) continue;
            const int64_t L = ctx->len[(size_t)v.contig], base = ctx->off[(size_t)v.contig];
            const int64_t a = gci_slice_bound((int64_t)v.start + flank, L), b = gci_slice_bound((int64_t)v.end - flank + 1, L);
            if (a >= b) continue;
            __atomic_fetch_add(&depth[base + a], 1, __ATOMIC_RELAXED);
            if (b < L) __atomic_fetch_add(&depth[base + b], -1, __ATOMIC_RELAXED);
        }
    });
    struct Blk { int64_t lo, hi; int32_t c; int64_t sum; };
    std::vector<Blk> blocks;
    const int64_t B = (int64_t)1 << 22;
    for (int32_t c = 0; c < ctx->n_contigs; c++)
        for (int64_t a = 0; a < ctx->len[(size_t)c]; a += B) blocks.push_back({ctx->off[(size_t)c] + a, ctx->off[(size_t)c] + std::min(ctx->len[(size_t)c], a + B), c, 0});
    parallel_blocks(ctx->threads, blocks.size(), 1, [&](uint64_t k, uint64_t) {
        int64_t s = 0;
        for (int64_t i = blocks[k].lo; i < blocks[k].hi; i++) s += depth[i];
        blocks[k].sum = s;
    });
    std::vector<int64_t> carry(blocks.size(), 0);
    for (size_t k = 1; k < blocks.size(); k++) carry[k] = blocks[k].c == blocks[k - 1].c ? carry[k - 1] + blocks[k - 1].sum : 0;
    parallel_blocks(ctx->threads, blocks.size(), 1, [&](uint64_t k, uint64_t) {
        int64_t run = carry[k];
        for (int64_t i = blocks[k].lo; i < blocks[k].hi; i++) { run += depth[i]; depth[i] = (int32_t)run; }
    });
    return GCI_OK;
}

/* ---- R8: depths[c][a:b] = 0 (GCI.py:324-328) --------------------------------------------------------------------------------- */
int gci_gap_mask(gci_ctx* ctx, int32_t* depth, const gci_ivl* gaps, uint32_t n_gaps)
{
    const int st = need_layout(ctx);
    if (st) return st;
    if (!depth || (n_gaps && !gaps)) return GCI_E_INVALID;
    parallel_blocks(ctx->threads, n_gaps, 1, [&](uint64_t i, uint64_t) {
        const gci_ivl g = gaps[i];
        if (g.contig < 0 || g.contig >= ctx->n_contigs) return;
        const int64_t L = ctx->len[(size_t)g.contig];
        const int64_t a = gci_slice_bound(g.start, L), b = gci_slice_bound(g.end, L);
        if (a < b) memset(depth + ctx->off[(size_t)g.contig] + a, 0, (size_t)(b - a) * sizeof(int32_t));
    });
    return GCI_OK;
}

/* ---- R9: max(h[i], n[i]) per base (GCI.py:350) ---------------------------------------------------------------------------------- */
int gci_max2(gci_ctx* ctx, const int32_t* a, const int32_t* b, int32_t* out)
{
    const int st = need_layout(ctx);
    if (st) return st;
    if (!a || !b || !out) return GCI_E_INVALID;
    parallel_blocks(ctx->threads, (uint64_t)ctx->total, (uint64_t)1 << 22, [&](uint64_t lo, uint64_t hi) {
        for (uint64_t i = lo; i < hi; i++) out[i] = std::max(a[i], b[i]);
    });
    return GCI_OK;
}

/* ---- R10: the run boundaries of collapse_depth_range (GCI.py:369-390), keys as libgci_hip.so emits them ----------------------------- */
int gci_issue_scan_windows(gci_ctx* ctx, const int32_t* depth, const gci_window* h_windows, uint32_t n_windows, double lo, double hi,
                           uint64_t* keys, uint32_t cap, uint32_t* n_keys)
{
    if (!ctx || (n_windows && !h_windows)) return GCI_E_INVALID;
    return issue_scan(ctx, depth, h_windows, n_windows, lo, hi, keys, cap, n_keys);
}

int gci_issue_scan(gci_ctx* ctx, const int32_t* depth, double lo, double hi, int flank, uint64_t* keys, uint32_t cap, uint32_t* n_keys)
{
    const int st = need_layout(ctx);
    if (st) return st;
    std::vector<gci_window> win((size_t)ctx->n_contigs);
    for (int32_t c = 0; c < ctx->n_contigs; c++) {
        const int64_t L = ctx->len[(size_t)c], base = ctx->off[(size_t)c];
        win[(size_t)c] = L > 2 * (int64_t)flank ? gci_window{base + flank, base + L - flank} : gci_window{base, base};
    }
    return issue_scan(ctx, depth, win.data(), (uint32_t)win.size(), lo, hi, keys, cap, n_keys);
}

/* ---- R7: f'{depth}\n' per base (GCI.py:115-117), contig after contig, without the '>' lines --------------------------------------- */
int gci_depth_text_size(gci_ctx* ctx, const int32_t* depth, uint64_t* contig_off)
{
    const int st = need_layout(ctx);
    if (st) return st;
    if (!depth || !contig_off) return GCI_E_INVALID;
    std::vector<std::atomic<uint64_t>> bytes((size_t)ctx->n_contigs);
    for (auto& b : bytes) b = 0;
    struct Blk { int64_t lo, hi; int32_t c; };
    std::vector<Blk> blocks;
    for (int32_t c = 0; c < ctx->n_contigs; c++)
        for (int64_t a = 0; a < ctx->len[(size_t)c]; a += (int64_t)1 << 22)
            blocks.push_back({ctx->off[(size_t)c] + a, ctx->off[(size_t)c] + std::min(ctx->len[(size_t)c], a + ((int64_t)1 << 22)), c});
    parallel_blocks(ctx->threads, blocks.size(), 1, [&](uint64_t k, uint64_t) {
        uint64_t s = 0;
        for (int64_t i = blocks[k].lo; i < blocks[k].hi; i++) s += decimal_width(depth[i]) + 1u;
        bytes[(size_t)blocks[k].c] += s;
    });
    uint64_t at = 0;
    for (int32_t c = 0; c < ctx->n_contigs; c++) { contig_off[c] = at; at += bytes[(size_t)c].load(); }
    contig_off[ctx->n_contigs] = at;
    return GCI_OK;
}

int gci_depth_text_write(gci_ctx* ctx, const int32_t* depth, uint8_t* out, uint64_t cap)
{
    const int st = need_layout(ctx);
    if (st) return st;
    if (!depth || !out) return GCI_E_INVALID;
    // sizes per block first, so that every block knows where its lines go
    struct Blk { int64_t lo, hi; uint64_t bytes, at; };
    std::vector<Blk> blocks;
    for (int32_t c = 0; c < ctx->n_contigs; c++)
        for (int64_t a = 0; a < ctx->len[(size_t)c]; a += (int64_t)1 << 20)
            blocks.push_back({ctx->off[(size_t)c] + a, ctx->off[(size_t)c] + std::min(ctx->len[(size_t)c], a + ((int64_t)1 << 20)), 0, 0});
    parallel_blocks(ctx->threads, blocks.size(), 1, [&](uint64_t k, uint64_t) {
        uint64_t s = 0;
        for (int64_t i = blocks[k].lo; i < blocks[k].hi; i++) s += decimal_width(depth[i]) + 1u;
        blocks[k].bytes = s;
    });
    uint64_t at = 0;
    for (auto& b : blocks) { b.at = at; at += b.bytes; }
    if (at > cap) return GCI_E_CAPACITY;
    parallel_blocks(ctx->threads, blocks.size(), 1, [&](uint64_t k, uint64_t) {
        uint8_t* w = out + blocks[k].at;
        for (int64_t i = blocks[k].lo; i < blocks[k].hi; i++) {
            const int32_t v = depth[i];
            uint64_t a = v < 0 ? (uint64_t)(-(int64_t)v) : (uint64_t)v;
            const uint32_t wd = decimal_width(v);
            if (v < 0) w[0] = '-';
            for (uint32_t d = wd; d-- > (v < 0 ? 1u : 0u);) { w[d] = (uint8_t)('0' + a % 10); a /= 10; }
            w[wd] = '\n';
            w += wd + 1;
        }
    });
    return GCI_OK;
}

/* ---- R15: the numerator of np.mean (GCI.py:862-868) ------------------------------------------------------------------------------ */
int gci_depth_sum(gci_ctx* ctx, const int32_t* depth, int64_t* sums)
{
    const int st = need_layout(ctx);
    if (st) return st;
    if (!depth || !sums) return GCI_E_INVALID;
    std::vector<std::atomic<int64_t>> acc((size_t)ctx->n_contigs);
    for (auto& a : acc) a = 0;
    struct Blk { int64_t lo, hi; int32_t c; };
    std::vector<Blk> blocks;
    for (int32_t c = 0; c < ctx->n_contigs; c++)
        for (int64_t a = 0; a < ctx->len[(size_t)c]; a += (int64_t)1 << 22)
            blocks.push_back({ctx->off[(size_t)c] + a, ctx->off[(size_t)c] + std::min(ctx->len[(size_t)c], a + ((int64_t)1 << 22)), c});
    parallel_blocks(ctx->threads, blocks.size(), 1, [&](uint64_t k, uint64_t) {
        int64_t s = 0;
        for (int64_t i = blocks[k].lo; i < blocks[k].hi; i++) s += depth[i];
        acc[(size_t)blocks[k].c] += s;
    });
    for (int32_t c = 0; c < ctx->n_contigs; c++) sums[c] = acc[(size_t)c].load();
    return GCI_OK;
}

/* ---- N3: window sums (sliding_window_average_depth, GCI.py:660-705) ---------------------------------------------------------------- */
int gci_range_sums(gci_ctx* ctx, const int32_t* depth, const int64_t* ranges, uint64_t n_ranges, int64_t* sums)
{
    if (!ctx || !depth || (n_ranges && (!ranges || !sums))) return GCI_E_INVALID;
    parallel_blocks(ctx->threads, n_ranges, 64, [&](uint64_t lo, uint64_t hi) {
        for (uint64_t r = lo; r < hi; r++) {
            int64_t s = 0;
            for (int64_t i = ranges[2 * r]; i < ranges[2 * r + 1]; i++) s += depth[i];
            sums[r] = s;
        }
    });
    return GCI_OK;
}

}  // extern "C"

/* ---- N2: the depth text as gzip members (write_depth incl. its gzip, GCI.py:99-143) ------------------------------------------------------
 * The format libgci_hip.so writes (k_deflate.hip), token for token: a tile of 4096 bases is one fixed-Huffman block -- per run of n
 * equal depths the line's literals and matches of distance = line width over the other (n - 1) lines, never leaving one or two
 * bytes behind a match -- closed by an empty stored block (byte alignment); 64 tiles are one member:
 *     1f 8b 08 00 00000000 00 ff | tile 0 | ... | tile 63 | 03 00 | CRC-32 | ISIZE
 * Here the text of a tile is written out (12 KB) and its CRC taken byte by byte: no GF(2) algebra to agree with. */
namespace {
struct BitW {
    uint8_t* out = nullptr;          // nullptr: count only
    uint64_t acc = 0, total = 0;
    uint32_t nb = 0;
    void put(uint32_t bits, uint32_t n)
    {
        acc |= (uint64_t)bits << nb;
        nb += n;
        total += n;
        while (nb >= 8u) { if (out) *out++ = (uint8_t)acc; acc >>= 8; nb -= 8u; }
    }
    void align() { if (nb) put(0u, 8u - nb); }
};
inline uint32_t rev(uint32_t v, int bits) { uint32_t r = 0; for (int i = 0; i < bits; i++) r |= ((v >> i) & 1u) << (bits - 1 - i); return r; }
inline void put_literal(BitW& o, uint32_t byte) { o.put(rev(0x30u + byte, 8), 8u); }                     // bytes < 144
inline void put_match(BitW& o, uint32_t len, uint32_t dist)                                              // 3 <= len <= 258, 2 <= dist <= 12
{
    if (len == 258u) o.put(rev(0xC5u, 8), 8u);
    else {
        const uint32_t t = len - 3u;
        uint32_t e = 0;
        if (t >= 8u) { uint32_t lg = 0; while ((t >> (lg + 1u)) != 0u) lg++; e = lg - 2u; }
        const uint32_t sym = 257u + 4u * e + (e ? (t >> e) : t);
        if (sym <= 279u) o.put(rev(sym - 256u, 7), 7u); else o.put(rev(0xC0u + (sym - 280u), 8), 8u);
        if (e) o.put(t & ((1u << e) - 1u), e);
    }
    uint32_t code, eb, ev;
    if (dist <= 4u) { code = dist - 1u; eb = 0; ev = 0; }
    else if (dist <= 8u) { code = 4u + ((dist - 5u) >> 1); eb = 1; ev = (dist - 5u) & 1u; }
    else { code = 6u + ((dist - 9u) >> 2); eb = 2; ev = (dist - 9u) & 3u; }
    o.put(rev(code, 5), 5u);
    if (eb) o.put(ev, eb);
}
const uint32_t* crc_table()
{
    static uint32_t T[256];
    static const bool made = [] {
        for (uint32_t i = 0; i < 256; i++) { uint32_t c = i; for (int k = 0; k < 8; k++) c = (c >> 1) ^ ((c & 1u) ? 0xEDB88320u : 0u); T[i] = c; }
        return true;
    }();
    (void)made;
    return T;
}
// one tile: its block into o, its text's bytes into the running CRC register / length
void deflate_tile(const int32_t* d, uint32_t n, BitW& o, uint32_t& crc_reg, uint32_t& text_len)
{
    if (n == 0) return;
    const uint32_t* T = crc_table();
    o.put(2u, 3u);                                                          // BFINAL = 0, BTYPE = 01
    for (uint32_t i = 0; i < n;) {
        uint32_t j = i + 1;
        while (j < n && d[j] == d[i]) j++;
        const uint32_t reps = j - i;
        char line[16];
        const int w = snprintf(line, sizeof line, "%u\n", (uint32_t)d[i]);
        for (uint32_t r = 0; r < reps; r++)
            for (int k = 0; k < w; k++) crc_reg = T[(crc_reg ^ (uint8_t)line[k]) & 0xFFu] ^ (crc_reg >> 8);
        text_len += reps * (uint32_t)w;
        for (int k = 0; k < w; k++) put_literal(o, (uint8_t)line[k]);
        uint32_t rest = (reps - 1u) * (uint32_t)w;
        while (rest >= 3u) {
            uint32_t len = rest < 258u ? rest : 258u;
            if (rest - len != 0u && rest - len < 3u) len = rest - 3u;       // never leave 1 or 2 bytes behind
            put_match(o, len, (uint32_t)w);
            rest -= len;
        }
        for (uint32_t k = 0; k < rest; k++) put_literal(o, (uint8_t)line[k]);   // (two lines of two bytes: the second as literals)
        i = j;
    }
    o.put(0u, 7u);                                                          // end of block
    o.put(0u, 3u);                                                          // empty stored block
    o.align();
    o.put(0x0000u, 16u);
    o.put(0xFFFFu, 16u);
}
constexpr uint32_t DEF_TILE = 4096, DEF_MEMBER_TILES = 64;
}  // namespace

extern "C" {

int gci_depth_deflate_from_build(gci_ctx* ctx, const int32_t* depth)
{
    (void)ctx; (void)depth;
    return GCI_E_INVALID;                              // this library's build keeps no run lists: the pair always walks the track
}

int gci_depth_deflate_size(gci_ctx* ctx, const int32_t* depth, const uint64_t* member_elem, const uint32_t* member_n, uint32_t n_members,
                           uint32_t* tile_bytes, uint32_t* member_bytes, uint32_t* member_crc, uint32_t* member_isize)
{
    if (!ctx || (n_members && (!depth || !member_elem || !member_n || !tile_bytes || !member_bytes || !member_crc || !member_isize)))
        return GCI_E_INVALID;
    parallel_blocks(ctx->threads, n_members, 1, [&](uint64_t m, uint64_t) {
        uint32_t reg = 0xFFFFFFFFu, len = 0, sum = 0;
        for (uint32_t t = 0; t < DEF_MEMBER_TILES; t++) {
            const uint32_t first = t * DEF_TILE, n_all = member_n[m];
            const uint32_t n = n_all > first ? std::min(DEF_TILE, n_all - first) : 0u;
            BitW o;
            deflate_tile(depth + member_elem[m] + first, n, o, reg, len);
            tile_bytes[m * DEF_MEMBER_TILES + t] = (uint32_t)(o.total >> 3);
            sum += (uint32_t)(o.total >> 3);
        }
        member_bytes[m] = sum + 20u;
        member_crc[m] = reg ^ 0xFFFFFFFFu;
        member_isize[m] = len;
    });
    return GCI_OK;
}

int gci_depth_deflate_write(gci_ctx* ctx, const int32_t* depth, const uint64_t* member_elem, const uint32_t* member_n, uint32_t n_members,
                            const uint32_t* tile_bytes, const uint32_t* member_crc, const uint32_t* member_isize, const uint64_t* member_out,
                            uint8_t* out, uint64_t cap)
{
    if (!ctx || (n_members && (!depth || !member_elem || !member_n || !tile_bytes || !member_crc || !member_isize || !member_out || !out)))
        return GCI_E_INVALID;
    std::atomic<int> bad{0};
    parallel_blocks(ctx->threads, n_members, 1, [&](uint64_t m, uint64_t) {
        uint32_t total = 20u;
        for (uint32_t t = 0; t < DEF_MEMBER_TILES; t++) total += tile_bytes[m * DEF_MEMBER_TILES + t];
        if (member_out[m] + total > cap) { bad = 1; return; }
        uint8_t* h = out + member_out[m];
        const uint8_t head[10] = {0x1F, 0x8B, 8, 0, 0, 0, 0, 0, 0, 0xFF};
        memcpy(h, head, 10);
        uint8_t* w = h + 10;
        for (uint32_t t = 0; t < DEF_MEMBER_TILES; t++) {
            const uint32_t first = t * DEF_TILE, n_all = member_n[m];
            const uint32_t n = n_all > first ? std::min(DEF_TILE, n_all - first) : 0u;
            BitW o;
            o.out = w;
            uint32_t reg = 0, len = 0;
            deflate_tile(depth + member_elem[m] + first, n, o, reg, len);
            w += tile_bytes[m * DEF_MEMBER_TILES + t];
        }
        const uint32_t c = member_crc[m], z = member_isize[m];
        const uint8_t tail[10] = {0x03, 0x00, (uint8_t)c, (uint8_t)(c >> 8), (uint8_t)(c >> 16), (uint8_t)(c >> 24),
                                  (uint8_t)z, (uint8_t)(z >> 8), (uint8_t)(z >> 16), (uint8_t)(z >> 24)};
        memcpy(w, tail, 10);
    });
    return bad ? GCI_E_CAPACITY : GCI_OK;
}

}  // extern "C"
