// host_io.cpp -- host-side helpers of libgci_hip.so for the containers either side of the GPU path
// (SURVEY.md section 8f, N1 / N2): parallel BGZF inflate, the BAM record-offset chase and parallel gzip
// framing of the depth text.  Plain C++ threads + zlib; no GPU work here.  The reference reaches these
// layers through pysam/htslib (GCI.py:150-151) and Python's gzip (GCI.py:111).
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <zlib.h>

#include <atomic>
#include <thread>
#include <vector>

#include "../../include/gci_hip.h"

namespace {

struct Block { uint64_t pos, size, isize, out; };

// walk the BSIZE chain of a BGZF byte string; returns a status
int scan(const uint8_t* raw, uint64_t n, std::vector<Block>& blocks, uint64_t& total)
{
    uint64_t pos = 0;
    total = 0;
    while (pos < n) {
        if (n - pos < 18) return GCI_E_MALFORMED;
        const uint8_t* h = raw + pos;
        if (h[0] != 0x1F || h[1] != 0x8B || h[2] != 8 || !(h[3] & 4)) return GCI_E_MALFORMED;
        const uint32_t xlen = h[10] | (h[11] << 8);
        if (pos + 12 + xlen > n) return GCI_E_MALFORMED;
        int64_t bsize = -1;
        for (uint32_t p = 12; p + 4 <= 12 + xlen;) {
            const uint32_t slen = h[p + 2] | (h[p + 3] << 8);
            if (h[p] == 66 && h[p + 1] == 67 && slen == 2 && p + 6 <= 12 + xlen) bsize = h[p + 4] | (h[p + 5] << 8);
            p += 4 + slen;
        }
        if (bsize < 0) return GCI_E_MALFORMED;
        const uint64_t size = (uint64_t)bsize + 1;
        if (size < 12 + xlen + 8 || pos + size > n) return GCI_E_MALFORMED;
        uint32_t isize;
        memcpy(&isize, h + size - 4, 4);
        blocks.push_back({pos, size, isize, total});
        total += isize;
        pos += size;
    }
    return GCI_OK;
}

// Where the walk reads the file from.  MemSrc: the byte string itself.  FdSrc: pread() on a descriptor -- a header every
// ~22 KB of a memory-mapped 77 GB file is one page fault per member (3.4 M of them per whole-genome BAM, 1.4 s on 16 threads that
// contend for the one address space, while the threads that stage the file's bytes fault on it too); a pread of 32 bytes costs
// a system call and no page-table entry.  A thread's FdSrc keeps the 32 bytes it read last: the ISIZE of a member (its last
// four bytes) and the header of the next one arrive in one call.
struct MemSrc {
    const uint8_t* raw; uint64_t n;
    bool fetch(uint64_t pos, uint32_t len, uint8_t* dst) { if (pos > n || n - pos < len) return false; memcpy(dst, raw + pos, len); return true; }
    // the first 0x1F in [lo, hi) or ~0
    uint64_t find_magic(uint64_t lo, uint64_t hi)
    {
        const uint8_t* q = (const uint8_t*)memchr(raw + lo, 0x1F, (size_t)(hi - lo));
        return q ? (uint64_t)(q - raw) : ~0ull;
    }
};
struct FdSrc {
    int fd; uint64_t n;
    uint64_t cpos = ~0ull; uint32_t clen = 0; uint8_t cache[32];
    std::vector<uint8_t> win;
    bool fetch(uint64_t pos, uint32_t len, uint8_t* dst)
    {
        if (pos > n || n - pos < len) return false;
        if (cpos != ~0ull && pos >= cpos && pos + len <= cpos + clen) { memcpy(dst, cache + (pos - cpos), len); return true; }
        if (len <= sizeof(cache)) {
            const uint32_t want = (uint32_t)(n - pos < sizeof(cache) ? n - pos : sizeof(cache));
            if (!read_all(pos, want, cache)) return false;
            cpos = pos; clen = want;
            memcpy(dst, cache, len);
            return true;
        }
        return read_all(pos, len, dst);
    }
    bool read_all(uint64_t pos, uint32_t len, uint8_t* dst)
    {
        uint32_t got = 0;
        while (got < len) {
            const ssize_t r = pread(fd, dst + got, len - got, (off_t)(pos + got));
            if (r <= 0) return false;
            got += (uint32_t)r;
        }
        return true;
    }
    uint64_t find_magic(uint64_t lo, uint64_t hi)
    {
        const uint32_t W = 1u << 18;
        win.resize(W);
        for (uint64_t a = lo; a < hi; a += W) {
            const uint32_t len = (uint32_t)(hi - a < W ? hi - a : W);
            if (!read_all(a, len, win.data())) return ~0ull;
            const uint8_t* q = (const uint8_t*)memchr(win.data(), 0x1F, len);
            if (q) return a + (uint64_t)(q - win.data());
        }
        return ~0ull;
    }
};

// member_at() through a source
template <typename Src>
bool member_via(Src& src, uint64_t pos, uint64_t& size, uint32_t& isize)
{
    const uint64_t n = src.n;
    if (pos > n || n - pos < 18) return false;
    uint8_t h[18];
    if (!src.fetch(pos, 18, h)) return false;
    if (h[0] != 0x1F || h[1] != 0x8B || h[2] != 8 || !(h[3] & 4)) return false;
    const uint32_t xlen = h[10] | (h[11] << 8);
    if (pos + 12 + xlen > n) return false;
    int64_t bsize = -1;
    if (xlen == 6) {                                             // what every BGZF writer makes: the one BC subfield
        if (h[12] == 66 && h[13] == 67 && (h[14] | (h[15] << 8)) == 2) bsize = h[16] | (h[17] << 8);
    } else {
        std::vector<uint8_t> x(12 + (size_t)xlen);
        if (!src.fetch(pos, 12 + xlen, x.data())) return false;
        for (uint32_t p = 12; p + 4 <= 12 + xlen;) {
            const uint32_t slen = x[p + 2] | (x[p + 3] << 8);
            if (x[p] == 66 && x[p + 1] == 67 && slen == 2 && p + 6 <= 12 + xlen) bsize = x[p + 4] | (x[p + 5] << 8);
            p += 4 + slen;
        }
    }
    if (bsize < 0) return false;
    size = (uint64_t)bsize + 1;
    if (size < 12 + xlen + 8 || pos + size > n) return false;
    uint8_t t[4];
    if (!src.fetch(pos + size - 4, 4, t)) return false;
    memcpy(&isize, t, 4);
    return true;
}

// The same table by several threads.  The chain is serial -- a member's BSIZE says where the next one starts -- and at
// genome size it is a page fault every ~27 KB over tens of GB (0.65 s for 64.5 GB, 25 ms at chr19 with the device waiting for
// it).  So the byte string is cut into ranges; the thread of a range looks for the first position in it at which a member
// header stands AND from which the chain holds for four members (the 16 fixed bytes of a BGZF header plus three more such
// headers exactly where the BSIZEs say: deflate output does not produce that by chance), walks from there to the end of its
// range, and the ranges are stitched: a range's walk must END exactly where the next range's walk STARTED -- if it does not
// (a header-like pattern inside member data), the serial walk goes on through that range.  Same table as scan().
// limit < n: only the members that START in front of `limit` (the table of the beginning of a file, wanted before the rest).
template <typename Src>
int scan_mt_via(const Src& proto, int threads, std::vector<Block>& blocks, uint64_t& total, uint64_t limit = ~0ull)
{
    const uint64_t n_all = proto.n;
    const uint64_t n = limit < n_all ? limit : n_all;            // the ranges cut [0, n); members are read up to n_all
    uint64_t min_range = 32ull << 20;
    if (const char* e = getenv("GCI_BGZF_RANGE")) { const long long v = atoll(e); if (v >= 64) min_range = (uint64_t)v; }   // (tests: small files)
    uint64_t R = n / min_range;
    if (threads < 1) threads = 1;
    if (R < 1) R = 1;
    if (R > (uint64_t)threads * 4) R = (uint64_t)threads * 4;
    struct Range { uint64_t start = ~0ull, end = 0; std::vector<Block> blocks; bool bad = false; };
    std::vector<Range> rg(R);
    std::atomic<uint64_t> next{0};
    auto work = [&]() {
        Src src = proto;                                         // (a thread's own read cache)
        for (;;) {
            const uint64_t k = next.fetch_add(1);
            if (k >= R) return;
            const uint64_t lo = n * k / R, hi = n * (k + 1) / R;
            uint64_t pos = lo;
            if (k) {                                             // the first member that starts in [lo, hi)
                pos = ~0ull;
                for (uint64_t p = lo; p < hi && p + 18 <= n_all; p++) {
                    p = src.find_magic(p, hi);
                    if (p == ~0ull) break;
                    uint64_t c = p, size;
                    uint32_t isz;
                    int good = 0;
                    while (good < 4 && c < n_all && member_via(src, c, size, isz)) { c += size; good++; }
                    if (good == 4 || (good > 0 && c == n_all)) { pos = p; break; }
                }
                if (pos == ~0ull) { rg[k].start = ~0ull; continue; }        // no member starts in this range (one huge gap?)
            }
            rg[k].start = pos;
            uint64_t size;
            uint32_t isz;
            while (pos < hi) {
                if (!member_via(src, pos, size, isz)) { rg[k].bad = true; break; }
                rg[k].blocks.push_back({pos, size, isz, 0});
                pos += size;
            }
            rg[k].end = pos;
        }
    };
    std::vector<std::thread> pool;
    const int T = (uint64_t)threads < R ? threads : (int)R;
    for (int t = 1; t < T; t++) pool.emplace_back(work);
    work();
    for (auto& th : pool) th.join();
    // stitch
    Src src = proto;
    blocks.clear();
    total = 0;
    uint64_t pos = 0;
    for (uint64_t k = 0; k < R; k++) {
        const uint64_t hi = n * (k + 1) / R;
        if (pos >= hi) continue;                                 // the previous range's last member reaches beyond this range
        if (!rg[k].bad && rg[k].start == pos) {
            for (const Block& b : rg[k].blocks) { blocks.push_back({b.pos, b.size, b.isize, total}); total += b.isize; }
            pos = rg[k].end;
            continue;
        }
        while (pos < hi) {                                       // this range started on something else: walk it serially
            uint64_t size;
            uint32_t isz;
            if (!member_via(src, pos, size, isz)) return GCI_E_MALFORMED;
            blocks.push_back({pos, size, isz, total});
            total += isz;
            pos += size;
        }
    }
    return (n < n_all ? pos >= n : pos == n) ? GCI_OK : GCI_E_MALFORMED;
}

int scan_mt(const uint8_t* raw, uint64_t n, int threads, std::vector<Block>& blocks, uint64_t& total)
{
    uint64_t min_range = 32ull << 20;
    if (const char* e = getenv("GCI_BGZF_RANGE")) { const long long v = atoll(e); if (v >= 64) min_range = (uint64_t)v; }
    if (threads < 2 || n / min_range < 2) return scan(raw, n, blocks, total);
    return scan_mt_via(MemSrc{raw, n}, threads, blocks, total);
}

template <typename F>
void parallel_for(uint64_t n, int threads, F f)
{
    if (threads < 1) threads = 1;
    if ((uint64_t)threads > n) threads = (int)(n ? n : 1);
    std::atomic<uint64_t> next{0};
    const uint64_t grain = n / ((uint64_t)threads * 8) + 1;
    auto worker = [&]() {
        for (;;) {
            const uint64_t a = next.fetch_add(grain);
            if (a >= n) return;
            const uint64_t b = a + grain < n ? a + grain : n;
            for (uint64_t i = a; i < b; i++) f(i);
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < threads; t++) pool.emplace_back(worker);
    worker();
    for (auto& th : pool) th.join();
}

}  // namespace

// Total inflated size of a BGZF byte string (sum of the ISIZE fields) and its number of members.
extern "C" int gci_bgzf_scan(const uint8_t* h_raw, uint64_t n_raw, uint64_t* n_blocks, uint64_t* inflated_bytes)
{
    if (!h_raw && n_raw) return GCI_E_INVALID;
    std::vector<Block> blocks;
    uint64_t total = 0;
    const int st = scan(h_raw, n_raw, blocks, total);
    if (st) return st;
    if (n_blocks) *n_blocks = blocks.size();
    if (inflated_bytes) *inflated_bytes = total;
    return GCI_OK;
}

// Member table of a BGZF byte string: h_pos[i] = byte offset of member i, h_isize[i] = its inflated size
// (h_pos[n_blocks] = n_raw when cap allows).  Lets a host stream a large file chunk by chunk.
extern "C" int gci_bgzf_blocks(const uint8_t* h_raw, uint64_t n_raw, uint64_t* h_pos, uint64_t* h_isize, uint64_t cap,
                               uint64_t* n_blocks)
{
    if ((!h_raw && n_raw) || !h_pos || !h_isize) return GCI_E_INVALID;
    std::vector<Block> blocks;
    uint64_t total = 0;
    const int st = scan(h_raw, n_raw, blocks, total);
    if (st) return st;
    if (n_blocks) *n_blocks = blocks.size();
    if (blocks.size() > cap) return GCI_E_CAPACITY;
    for (size_t i = 0; i < blocks.size(); i++) { h_pos[i] = blocks[i].pos; h_isize[i] = blocks[i].isize; }
    if (blocks.size() < cap) h_pos[blocks.size()] = n_raw;
    return GCI_OK;
}

// The member table in one pass over the byte string, by `threads` threads (scan_mt): a handle the caller reads and frees.
struct gci_bgzf_table { std::vector<Block> blocks; uint64_t total = 0, n_raw = 0; };

extern "C" int gci_bgzf_table_build(const uint8_t* h_raw, uint64_t n_raw, int threads, gci_bgzf_table** out)
{
    if ((!h_raw && n_raw) || !out) return GCI_E_INVALID;
    *out = nullptr;
    gci_bgzf_table* t = new (std::nothrow) gci_bgzf_table();
    if (!t) return GCI_E_NOMEM;
    t->n_raw = n_raw;
    const int st = scan_mt(h_raw, n_raw, threads, t->blocks, t->total);
    if (st) { delete t; return st; }
    *out = t;
    return GCI_OK;
}

// The table of the members that start in front of byte `limit` of the string (t's byte length = where the last of them ends): what
// a host needs to put the first run of a large file on the device while the rest of the table is still being made.
extern "C" int gci_bgzf_table_build_prefix(const uint8_t* h_raw, uint64_t n_raw, uint64_t limit, int threads, gci_bgzf_table** out)
{
    if ((!h_raw && n_raw) || !out) return GCI_E_INVALID;
    *out = nullptr;
    gci_bgzf_table* t = new (std::nothrow) gci_bgzf_table();
    if (!t) return GCI_E_NOMEM;
    const int st = scan_mt_via(MemSrc{h_raw, n_raw}, threads, t->blocks, t->total, limit);
    if (st) { delete t; return st; }
    t->n_raw = t->blocks.empty() ? 0 : t->blocks.back().pos + t->blocks.back().size;
    *out = t;
    return GCI_OK;
}

// The same table read through a file descriptor (pread) instead of a mapping of the file: see FdSrc.
extern "C" int gci_bgzf_table_build_fd(int fd, uint64_t n_raw, int threads, gci_bgzf_table** out)
{
    if (fd < 0 || !out) return GCI_E_INVALID;
    *out = nullptr;
    gci_bgzf_table* t = new (std::nothrow) gci_bgzf_table();
    if (!t) return GCI_E_NOMEM;
    t->n_raw = n_raw;
    FdSrc src;
    src.fd = fd; src.n = n_raw;
    const int st = scan_mt_via(src, threads, t->blocks, t->total);
    if (st) { delete t; return st; }
    *out = t;
    return GCI_OK;
}

extern "C" uint64_t gci_bgzf_table_count(const gci_bgzf_table* t) { return t ? t->blocks.size() : 0; }

// h_pos: count + 1 entries (the last = the length of the byte string); h_isize: count entries
extern "C" int gci_bgzf_table_export(const gci_bgzf_table* t, uint64_t* h_pos, uint64_t* h_isize)
{
    if (!t || !h_pos || (!h_isize && !t->blocks.empty())) return GCI_E_INVALID;
    for (size_t i = 0; i < t->blocks.size(); i++) { h_pos[i] = t->blocks[i].pos; h_isize[i] = t->blocks[i].isize; }
    h_pos[t->blocks.size()] = t->n_raw;
    return GCI_OK;
}

extern "C" int gci_bgzf_table_free(gci_bgzf_table* t) { delete t; return GCI_OK; }

// Inflate every member into h_out (capacity cap >= the size gci_bgzf_scan reported), members in parallel.
extern "C" int gci_bgzf_inflate(const uint8_t* h_raw, uint64_t n_raw, uint8_t* h_out, uint64_t cap, int threads,
                                int check_crc)
{
    if ((!h_raw && n_raw) || (!h_out && cap)) return GCI_E_INVALID;
    std::vector<Block> blocks;
    uint64_t total = 0;
    int st = scan(h_raw, n_raw, blocks, total);
    if (st) return st;
    if (total > cap) return GCI_E_CAPACITY;
    std::atomic<int> err{GCI_OK};
    parallel_for(blocks.size(), threads, [&](uint64_t i) {
        const Block& b = blocks[i];
        if (b.isize == 0) return;
        const uint8_t* h = h_raw + b.pos;
        const uint32_t xlen = h[10] | (h[11] << 8);
        z_stream zs;
        memset(&zs, 0, sizeof zs);
        if (inflateInit2(&zs, -15) != Z_OK) { err = GCI_E_NOMEM; return; }
        zs.next_in = const_cast<Bytef*>(h + 12 + xlen);
        zs.avail_in = (uInt)(b.size - 12 - xlen - 8);
        zs.next_out = h_out + b.out;
        zs.avail_out = (uInt)b.isize;
        const int r = inflate(&zs, Z_FINISH);
        const bool ok = r == Z_STREAM_END && zs.total_out == b.isize;
        inflateEnd(&zs);
        if (!ok) { err = GCI_E_MALFORMED; return; }
        if (check_crc) {
            uint32_t crc;
            memcpy(&crc, h + b.size - 8, 4);
            if ((uint32_t)crc32(0L, h_out + b.out, (uInt)b.isize) != crc) err = GCI_E_MALFORMED;
        }
    });
    return err.load();
}

// BAM header end + record offsets of an inflated stream: the one serial step of the decode.
// h_offs may be NULL (count only).  *first_record = byte offset of the first record.
extern "C" int gci_bam_record_offsets(const uint8_t* h_stream, uint64_t n, uint64_t* h_offs, uint64_t cap,
                                      uint64_t* n_rec, uint64_t* first_record)
{
    if (!h_stream || n < 12 || memcmp(h_stream, "BAM\1", 4) != 0) return GCI_E_MALFORMED;
    int32_t l_text, n_ref;
    memcpy(&l_text, h_stream + 4, 4);
    uint64_t p = 8 + (uint64_t)(uint32_t)l_text;
    if (l_text < 0 || p + 4 > n) return GCI_E_MALFORMED;
    memcpy(&n_ref, h_stream + p, 4);
    p += 4;
    for (int32_t i = 0; i < n_ref; i++) {
        int32_t l_name;
        if (p + 4 > n) return GCI_E_MALFORMED;
        memcpy(&l_name, h_stream + p, 4);
        if (l_name < 0) return GCI_E_MALFORMED;
        p += 4 + (uint64_t)l_name + 4;
        if (p > n) return GCI_E_MALFORMED;
    }
    if (first_record) *first_record = p;
    uint64_t k = 0;
    while (p < n) {
        int32_t bs;
        if (p + 4 > n) return GCI_E_MALFORMED;
        memcpy(&bs, h_stream + p, 4);
        if (bs < 32 || p + 4 + (uint64_t)bs > n) return GCI_E_MALFORMED;
        if (h_offs) { if (k >= cap) return GCI_E_CAPACITY; h_offs[k] = p; }
        k++;
        p += 4 + (uint64_t)bs;
    }
    if (n_rec) *n_rec = k;
    return GCI_OK;
}

// Record offsets inside one CHUNK of an inflated stream (a whole number of records is not guaranteed): starts at
// byte `start`, stops before the first record that is not completely inside [0, n).  *consumed = offset of that
// record (== n when the chunk ends on a record boundary): the caller carries h_buf[consumed:] into the next chunk.
extern "C" int gci_bam_chunk_offsets(const uint8_t* h_buf, uint64_t n, uint64_t start, uint64_t* h_offs, uint64_t cap,
                                     uint64_t* n_rec, uint64_t* consumed)
{
    if ((!h_buf && n) || !n_rec || !consumed || start > n) return GCI_E_INVALID;
    uint64_t p = start, k = 0;
    while (p + 4 <= n) {
        int32_t bs;
        memcpy(&bs, h_buf + p, 4);
        if (bs < 32) return GCI_E_MALFORMED;
        if (p + 4 + (uint64_t)bs > n) break;
        if (h_offs) { if (k >= cap) return GCI_E_CAPACITY; h_offs[k] = p; }
        k++;
        p += 4 + (uint64_t)bs;
    }
    *n_rec = k;
    *consumed = p;
    return GCI_OK;
}

// gzip-frame `n` bytes of text as members of `chunk` input bytes each, compressed in parallel at `level`.
// h_out must hold gci_gzip_bound(n, chunk) bytes; members are written back to back, *n_out = total bytes.
extern "C" uint64_t gci_gzip_bound(uint64_t n, uint64_t chunk)
{
    if (chunk == 0) chunk = 1;
    const uint64_t members = n / chunk + 1;
    return n + n / 1000 + members * 64 + 64;
}

extern "C" int gci_gzip_members(const uint8_t* h_text, uint64_t n, uint64_t chunk, int level, int threads,
                                uint8_t* h_out, uint64_t cap, uint64_t* n_out)
{
    if ((!h_text && n) || !h_out || !n_out || chunk == 0 || chunk > 0x7fffffffULL) return GCI_E_INVALID;
    const uint64_t members = (n + chunk - 1) / chunk;
    std::vector<std::vector<uint8_t>> parts(members);
    std::atomic<int> err{GCI_OK};
    parallel_for(members, threads, [&](uint64_t i) {
        const uint64_t a = i * chunk, len = (a + chunk < n ? chunk : n - a);
        z_stream zs;
        memset(&zs, 0, sizeof zs);
        if (deflateInit2(&zs, level, Z_DEFLATED, 31, 8, Z_DEFAULT_STRATEGY) != Z_OK) { err = GCI_E_NOMEM; return; }
        std::vector<uint8_t>& o = parts[i];
        o.resize(deflateBound(&zs, (uLong)len) + 32);
        zs.next_in = const_cast<Bytef*>(h_text + a);
        zs.avail_in = (uInt)len;
        zs.next_out = o.data();
        zs.avail_out = (uInt)o.size();
        const int r = deflate(&zs, Z_FINISH);
        if (r != Z_STREAM_END) err = GCI_E_CAPACITY;
        o.resize(zs.total_out);
        deflateEnd(&zs);
    });
    if (err.load()) return err.load();
    uint64_t total = 0;
    for (auto& o : parts) total += o.size();
    if (total > cap) return GCI_E_CAPACITY;
    uint64_t p = 0;
    for (auto& o : parts) { memcpy(h_out + p, o.data(), o.size()); p += o.size(); }
    *n_out = total;
    return GCI_OK;
}

// ---- N4 (second half, host): the PAF filter (filter(), GCI.py:211-254) in native code -----------------------------------
// Same arithmetic in the same order as the reference: identity = nmatch / alnlen (IEEE double), blocks accumulate per
// (query, target) in file order and ACROSS files (the reference creates its table once, outside the per-file loop), per
// query the target with the largest (mean identity * covered / qlen, target name) wins, its interval is the longest
// merged target block (leftmost on ties).  Output per file: one compact record + name per query seen so far, in first-
// appearance order -- what gci_name_join takes.
#include <algorithm>
#include <string>
#include <unordered_map>

namespace {

// one alignment line that passed the filter, in file order
struct PafHit { const uint8_t* qn; uint32_t qn_len; int32_t t; int64_t qlen, qs, qe, ts, te; double identity; uint32_t hq; uint32_t query; };
struct PafQueryRef { uint64_t name_off; uint32_t name_len; bool hq; };
struct PafEmit { uint32_t query; int32_t target; int64_t s, e, qlen; };

// union of closed-touching blocks: covered length and the longest merged block (leftmost on ties)
void merge_span(std::vector<std::pair<int64_t, int64_t>>& p, int64_t& covered, int64_t& bs, int64_t& be)
{
    std::sort(p.begin(), p.end());
    covered = 0;
    int64_t best = -1;
    bs = be = 0;
    int64_t lo = p[0].first, hi = p[0].second;
    for (size_t i = 1; i <= p.size(); i++) {
        if (i < p.size() && hi >= p[i].first) { hi = std::max(hi, p[i].second); continue; }
        covered += hi - lo;
        if (hi - lo > best) { best = hi - lo; bs = lo; be = hi; }
        if (i < p.size()) { lo = p[i].first; hi = p[i].second; }
    }
}

bool is_space(uint8_t c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == 0x0b || c == 0x0c; }

// Python's int() on a column: optional blanks, optional sign, digits with single underscores between them
bool parse_int(const uint8_t* a, const uint8_t* b, int64_t& v)
{
    while (a < b && is_space(*a)) a++;
    while (b > a && is_space(b[-1])) b--;
    bool neg = false;
    if (a < b && (*a == '+' || *a == '-')) { neg = *a == '-'; a++; }
    if (a >= b) return false;
    uint64_t x = 0;
    bool prev_digit = false;
    for (; a < b; a++) {
        if (*a == '_') {
            if (!prev_digit || a + 1 >= b || a[1] < '0' || a[1] > '9') return false;
            prev_digit = false;
            continue;
        }
        if (*a < '0' || *a > '9') return false;
        prev_digit = true;
        if (x > (0x7fffffffffffffffULL - (*a - '0')) / 10) return false;
        x = x * 10 + (*a - '0');
    }
    v = neg ? -(int64_t)x : (int64_t)x;
    return true;
}

}  // namespace

struct gci_paf {
    std::vector<PafQueryRef> queries;       // first-appearance order
    std::vector<uint8_t> names;             // their names back to back
    std::vector<std::vector<PafEmit>> per_file;
};

namespace {

// What one thread found in its share of a file's lines
struct PafSlice {
    std::vector<PafHit> hits;
    uint64_t lines = 0;
    int status = GCI_OK;
    uint64_t err_line = 0;                  // 1-based inside the slice
};

inline bool line_starts_at(const uint8_t* base, const uint8_t* p)     // universal newlines: \n, \r, \r\n
{
    if (p == base) return true;
    return p[-1] == '\n' || (p[-1] == '\r' && *p != '\n');
}

// the lines that START in [lo, hi): tokenise, filter (GCI.py:218-239), keep what passes
void paf_scan(const uint8_t* base, const uint8_t* lo, const uint8_t* hi, const uint8_t* end,
              const std::unordered_map<std::string, int32_t>& tmap, int map_qual, int mq_cutoff, double iden_percent, PafSlice& out)
{
    const uint8_t* p = lo;
    while (p < hi && !line_starts_at(base, p)) p++;
    std::string key;
    while (p < hi) {
        // one line, stripped like str.strip()
        const uint8_t* le = p;
        while (le < end && *le != '\n' && *le != '\r') le++;
        const uint8_t* next = le;
        if (next < end) next += (*next == '\r' && next + 1 < end && next[1] == '\n') ? 2 : 1;
        const uint8_t *a = p, *b = le;
        p = next;
        out.lines++;
        while (a < b && is_space(*a)) a++;
        while (b > a && is_space(b[-1])) b--;
        const uint8_t* col[13];
        int nc = 0;
        col[0] = a;
        for (const uint8_t* q = a; q < b && nc < 12; q++) if (*q == '\t') col[++nc] = q + 1;
        auto col_end = [&](int k) { const uint8_t* q = col[k]; while (q < b && *q != '\t') q++; return q; };
        if (nc < 5) { out.status = GCI_E_MALFORMED; out.err_line = out.lines; return; }                 // col[5]: IndexError
        key.assign((const char*)col[5], (size_t)(col_end(5) - col[5]));
        const auto ti = tmap.find(key);
        if (ti == tmap.end()) continue;
        int64_t v[12] = {0};
        bool ok = nc >= 11;
        for (int k : {1, 2, 3, 7, 8, 9, 10, 11}) ok = ok && parse_int(col[k], col_end(k), v[k]);
        if (!ok) { out.status = GCI_E_MALFORMED; out.err_line = out.lines; return; }
        if (v[10] == 0) { out.status = GCI_E_ZERO_DIV; out.err_line = out.lines; return; }                 // nmatch / alnlen
        const double identity = (double)v[9] / (double)v[10];
        if (v[11] >= map_qual && identity >= iden_percent)
            out.hits.push_back(PafHit{col[0], (uint32_t)(col_end(0) - col[0]), ti->second, v[1], v[2], v[3], v[7], v[8], identity,
                                      v[11] >= mq_cutoff ? 1u : 0u, 0u});
    }
}

}  // namespace

// threads: host threads for the tokenising pass and the per-query arithmetic (0 = one per hardware thread, at most 32)
extern "C" int gci_paf_filter(const uint8_t* const* h_files, const uint64_t* n_bytes, int n_files, const char* const* targets,
                              int n_targets, int map_qual, int mq_cutoff, double iden_percent, int threads, gci_paf** out,
                              uint64_t* err_line)
{
    if (!out || n_files < 0 || (n_files && (!h_files || !n_bytes)) || (n_targets && !targets)) return GCI_E_INVALID;
    if (threads <= 0) { threads = (int)std::thread::hardware_concurrency(); threads = threads < 1 ? 1 : threads > 32 ? 32 : threads; }
    gci_paf* R = new (std::nothrow) gci_paf();
    if (!R) return GCI_E_NOMEM;
    std::unordered_map<std::string, int32_t> tmap;
    for (int t = 0; t < n_targets; t++) tmap.emplace(targets[t], t);
    // all hits of all files so far, in file order (the reference never resets its table between files), and the
    // queries in first-appearance order: open addressing on the 64-bit name hash, names verified
    std::vector<PafHit> hits;
    std::vector<const uint8_t*> qname_ptr;
    std::vector<uint64_t> slot_hash;
    std::vector<uint32_t> slot_query;
    size_t n_slots = 1024;
    slot_hash.assign(n_slots, 0); slot_query.assign(n_slots, 0xFFFFFFFFu);
    auto grow = [&]() {
        const size_t m = n_slots * 2;
        std::vector<uint64_t> h2(m, 0); std::vector<uint32_t> q2(m, 0xFFFFFFFFu);
        for (size_t i = 0; i < n_slots; i++) if (slot_query[i] != 0xFFFFFFFFu) {
            size_t k = (size_t)(slot_hash[i] ^ (slot_hash[i] >> 29)) & (m - 1);
            while (q2[k] != 0xFFFFFFFFu) k = (k + 1) & (m - 1);
            h2[k] = slot_hash[i]; q2[k] = slot_query[i];
        }
        slot_hash.swap(h2); slot_query.swap(q2); n_slots = m;
    };
    int status = GCI_OK;
    for (int f = 0; f < n_files && status == GCI_OK; f++) {
        const uint8_t* base = h_files[f];
        const uint8_t* end = base + n_bytes[f];
        // ---- pass 1 (parallel): every thread the lines that start in its byte range
        const uint64_t n_slices = n_bytes[f] < (1u << 16) ? 1 : (uint64_t)threads * 4;
        std::vector<PafSlice> slices(n_slices);
        parallel_for(n_slices, threads, [&](uint64_t i) {
            paf_scan(base, base + n_bytes[f] * i / n_slices, base + n_bytes[f] * (i + 1) / n_slices, end, tmap, map_qual, mq_cutoff,
                     iden_percent, slices[i]);
        });
        uint64_t lines_before = 0;
        for (const PafSlice& sl : slices) {                      // the first offending line in file order, as the reference's loop
            if (sl.status != GCI_OK) { status = sl.status; if (err_line) *err_line = lines_before + sl.err_line; break; }
            lines_before += sl.lines;
        }
        if (status != GCI_OK) break;
        // ---- pass 2 (serial, cheap): query index of every hit
        const size_t first_new = hits.size();
        for (const PafSlice& sl : slices) hits.insert(hits.end(), sl.hits.begin(), sl.hits.end());
        for (size_t i = first_new; i < hits.size(); i++) {
            PafHit& h = hits[i];
            if ((R->queries.size() + 1) * 2 > n_slots) grow();
            const uint64_t hv = gci_name_hash(h.qn, h.qn_len) | 1ull;                 // never 0
            size_t k = (size_t)(hv ^ (hv >> 29)) & (n_slots - 1);
            for (;;) {
                const uint32_t q = slot_query[k];
                if (q == 0xFFFFFFFFu) {
                    slot_hash[k] = hv; slot_query[k] = (uint32_t)R->queries.size();
                    h.query = (uint32_t)R->queries.size();
                    R->queries.push_back(PafQueryRef{0, h.qn_len, false});
                    qname_ptr.push_back(h.qn);
                    break;
                }
                if (slot_hash[k] == hv && R->queries[q].name_len == h.qn_len && memcmp(qname_ptr[q], h.qn, h.qn_len) == 0) { h.query = q; break; }
                k = (k + 1) & (n_slots - 1);
            }
            if (h.hq) R->queries[h.query].hq = true;
        }
        // hits of a query side by side, file order kept (counting sort by query)
        const size_t nq = R->queries.size();
        std::vector<uint64_t> start(nq + 1, 0);
        for (const PafHit& h : hits) start[h.query + 1]++;
        for (size_t q = 0; q < nq; q++) start[q + 1] += start[q];
        std::vector<uint32_t> order(hits.size());
        {
            std::vector<uint64_t> cur(start.begin(), start.end() - 1);
            for (size_t i = 0; i < hits.size(); i++) order[cur[hits[i].query]++] = (uint32_t)i;
        }
        // ---- pass 3 (parallel over queries): GCI.py:241-254
        std::vector<PafEmit> emit(nq);
        std::atomic<int> qstatus{GCI_OK};
        const uint64_t q_slices = nq < 4096 ? 1 : (uint64_t)threads * 8;
        parallel_for(q_slices, threads, [&](uint64_t si) {
            std::vector<std::pair<int64_t, int64_t>> pairs;
            std::vector<int32_t> seen;
            for (size_t q = nq * si / q_slices; q < nq * (si + 1) / q_slices; q++) {
                bool have = false;
                double best_rank = 0;
                const char* best_name = nullptr;
                PafEmit best{(uint32_t)q, -1, 0, 0, 0};
                seen.clear();
                for (uint64_t a = start[q]; a < start[q + 1]; a++) {
                    const int32_t t = hits[order[a]].t;
                    bool dup = false;
                    for (int32_t x : seen) dup = dup || x == t;
                    if (dup) continue;
                    seen.push_back(t);
                    // this target's blocks, in file order
                    pairs.clear();
                    int64_t qlen = 0;
                    double total = 0.0;
                    uint64_t n_aln = 0;
                    for (uint64_t c = a; c < start[q + 1]; c++) {
                        const PafHit& x = hits[order[c]];
                        if (x.t != t) continue;
                        if (n_aln == 0) qlen = x.qlen;                                  // qlen of the first block
                        pairs.emplace_back(x.qs, x.qe);
                        total = total + x.identity;                                     // file order, as sum() does
                        n_aln++;
                    }
                    int64_t covered, s0, e0;
                    merge_span(pairs, covered, s0, e0);
                    if (qlen == 0) { qstatus = GCI_E_ZERO_DIV; break; }
                    const double rank = total / (double)n_aln * ((double)covered / (double)qlen);
                    const char* name = targets[t];
                    if (!have || rank > best_rank || (rank == best_rank && strcmp(name, best_name) > 0)) {
                        pairs.clear();
                        for (uint64_t c = a; c < start[q + 1]; c++) {
                            const PafHit& x = hits[order[c]];
                            if (x.t == t) pairs.emplace_back(x.ts, x.te);
                        }
                        merge_span(pairs, covered, s0, e0);
                        have = true; best_rank = rank; best_name = name;
                        best = PafEmit{(uint32_t)q, t, s0, e0, qlen};
                    }
                }
                emit[q] = best;
            }
        });
        if (qstatus.load() != GCI_OK) { status = qstatus.load(); break; }
        R->per_file.push_back(std::move(emit));
    }
    if (status != GCI_OK) { delete R; return status; }
    // the result owns its names
    uint64_t off = 0;
    for (size_t q = 0; q < R->queries.size(); q++) { R->queries[q].name_off = off; off += R->queries[q].name_len; }
    R->names.resize(off);
    for (size_t q = 0; q < R->queries.size(); q++)
        if (R->queries[q].name_len) memcpy(R->names.data() + R->queries[q].name_off, qname_ptr[q], R->queries[q].name_len);
    *out = R;
    return GCI_OK;
}

extern "C" uint64_t gci_paf_count(const gci_paf* r, int file)
{
    return r && file >= 0 && (size_t)file < r->per_file.size() ? r->per_file[file].size() : 0;
}

extern "C" uint64_t gci_paf_name_bytes(const gci_paf* r, int file)
{
    if (!r || file < 0 || (size_t)file >= r->per_file.size()) return 0;
    uint64_t n = 0;
    for (const PafEmit& e : r->per_file[file]) n += r->queries[e.query].name_len;
    return n;
}

// h_recs: gci_paf_count() records; h_names: gci_paf_name_bytes() bytes; h_name_off: count + 1 offsets into h_names
extern "C" int gci_paf_export(const gci_paf* r, int file, gci_rec* h_recs, uint8_t* h_names, uint64_t* h_name_off)
{
    if (!r || file < 0 || (size_t)file >= r->per_file.size() || !h_name_off) return GCI_E_INVALID;
    const auto& em = r->per_file[file];
    if (em.size() && (!h_recs || !h_names)) {
        bool any_bytes = false;
        for (const PafEmit& e : em) any_bytes = any_bytes || r->queries[e.query].name_len != 0;
        if (!h_recs || (any_bytes && !h_names)) return GCI_E_INVALID;
    }
    uint64_t off = 0;
    for (size_t i = 0; i < em.size(); i++) {
        const PafEmit& e = em[i];
        const PafQueryRef& Q = r->queries[e.query];
        const uint8_t* qn = r->names.data() + Q.name_off;
        for (int64_t x : {e.s, e.e, e.qlen}) if (x > 0x7fffffffLL || x < -0x80000000LL) return GCI_E_INVALID;
        if (Q.name_len > 0xFFFF) return GCI_E_INVALID;
        gci_rec rec;
        memset(&rec, 0, sizeof rec);
        rec.name_hash = gci_name_hash(qn, Q.name_len);
        rec.contig = e.target; rec.start = (int32_t)e.s; rec.end = (int32_t)e.e; rec.qlen = (int32_t)e.qlen;
        rec.rec_idx = (uint32_t)i; rec.mapq = 0;
        rec.flags = (uint8_t)(GCI_REC_PASS | (Q.hq ? GCI_REC_HQ : 0));
        rec.name_len = (uint16_t)Q.name_len;
        h_recs[i] = rec;
        h_name_off[i] = off;
        if (Q.name_len) memcpy(h_names + off, qn, Q.name_len);
        off += Q.name_len;
    }
    h_name_off[em.size()] = off;
    return GCI_OK;
}

extern "C" int gci_paf_free(gci_paf* r) { delete r; return GCI_OK; }

// ---- N1: BGZF file -> heads stream --------------------------------------------------------------------------------
// The record filter reads a record's fixed part, name, CIGAR and aux block; SEQ and QUAL (98 % of a HiFi record) are
// skipped by pointer arithmetic (GCI.py:146-169 never looks at them).  gci_bam_heads inflates the file group by group
// into three rotating buffers (worker threads, member-parallel), walks the block_size chain of the group inflated
// before (the one serial step, on the calling thread, overlapped with the inflate of the next group) and copies
// everything but SEQ / QUAL of each record into one compact "heads stream" (workers again):
//     [BAM header, verbatim][record 0 without SEQ/QUAL][record 1 without SEQ/QUAL] ...
// block_size of an emitted record is its new length - 4, l_seq keeps its value.  A record whose fields contradict its
// block_size (l_seq < 0, or name + CIGAR + SEQ + QUAL longer than the record) is emitted as its 36 fixed bytes with
// l_seq = -1: the filter kernel reports it as GCI_E_MALFORMED with its index, as it would on the full stream.
// The inflated stream is never held as a whole: host memory is 3 groups + the heads (about 400 B per HiFi record).
#include <sys/mman.h>

#include <condition_variable>
#include <memory>
#include <mutex>
#include <new>

namespace {

struct HeadTask { const uint8_t* src; uint64_t dst; uint32_t head, aux_src, aux; int32_t bad; };

// compact form of one complete record at p (block_size already validated >= 32): lengths only
inline void head_shape(const uint8_t* p, HeadTask& t)
{
    int32_t bs, l_seq;
    memcpy(&bs, p, 4);
    memcpy(&l_seq, p + 20, 4);
    const uint32_t l_name = p[12];
    const uint32_t n_cig = p[16] | (p[17] << 8);
    const uint64_t head = 36ull + l_name + 4ull * n_cig;
    const uint64_t seq = l_seq < 0 ? 0 : (((uint64_t)(uint32_t)l_seq + 1) >> 1) + (uint64_t)(uint32_t)l_seq;
    const uint64_t total = 4ull + (uint64_t)(uint32_t)bs;
    t.src = p;
    if (l_seq < 0 || head + seq > total) { t.bad = 1; t.head = 36; t.aux_src = 0; t.aux = 0; return; }
    t.bad = 0;
    t.head = (uint32_t)head;
    t.aux_src = (uint32_t)(head + seq);
    t.aux = (uint32_t)(total - head - seq);
}

inline void head_copy(uint8_t* out, const HeadTask& t)
{
    uint8_t* d = out + t.dst;
    memcpy(d, t.src, t.head);
    if (t.aux) memcpy(d + t.head, t.src + t.aux_src, t.aux);
    const int32_t bs = (int32_t)(t.head + t.aux - 4);
    memcpy(d, &bs, 4);
    if (t.bad) { const int32_t m1 = -1; memcpy(d + 20, &m1, 4); }
}

// length of a complete BAM header at p, or 0 when [p, p+n) does not hold all of it yet, or -1 when it is not one
int64_t bam_header_len(const uint8_t* p, uint64_t n)
{
    if (n < 4) return 0;
    if (memcmp(p, "BAM\1", 4) != 0) return -1;
    if (n < 12) return 0;
    int32_t l_text, n_ref;
    memcpy(&l_text, p + 4, 4);
    if (l_text < 0) return -1;
    uint64_t q = 8 + (uint64_t)l_text;
    if (q + 4 > n) return 0;
    memcpy(&n_ref, p + q, 4);
    if (n_ref < 0) return -1;
    q += 4;
    for (int32_t i = 0; i < n_ref; i++) {
        int32_t l_name;
        if (q + 4 > n) return 0;
        memcpy(&l_name, p + q, 4);
        if (l_name < 0) return -1;
        q += 4 + (uint64_t)l_name + 4;
        if (q > n) return 0;
    }
    return (int64_t)q;
}

// all-thread rendezvous.  Blocking, not spinning: under a CPU quota (containers) 64 spinning threads burn the quota of
// the whole process and get it throttled.
struct Barrier {
    std::mutex m;
    std::condition_variable cv;
    uint32_t count = 0, gen = 0;
    const uint32_t n;
    explicit Barrier(uint32_t n_) : n(n_) {}
    void wait()
    {
        std::unique_lock<std::mutex> lk(m);
        const uint32_t g = gen;
        if (++count == n) {
            count = 0;
            gen++;
            lk.unlock();
            cv.notify_all();
        } else {
            cv.wait(lk, [&] { return gen != g; });
        }
    }
};

bool inflate_member(const uint8_t* raw, const Block& b, uint8_t* out, int check_crc)
{
    if (b.isize == 0) return true;
    const uint8_t* h = raw + b.pos;
    const uint32_t xlen = h[10] | (h[11] << 8);
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (inflateInit2(&zs, -15) != Z_OK) return false;
    zs.next_in = const_cast<Bytef*>(h + 12 + xlen);
    zs.avail_in = (uInt)(b.size - 12 - xlen - 8);
    zs.next_out = out;
    zs.avail_out = (uInt)b.isize;
    const int r = inflate(&zs, Z_FINISH);
    bool ok = r == Z_STREAM_END && zs.total_out == b.isize;
    inflateEnd(&zs);
    if (ok && check_crc) {
        uint32_t crc;
        memcpy(&crc, h + b.size - 8, 4);
        ok = (uint32_t)crc32(0L, out, (uInt)b.isize) == crc;
    }
    return ok;
}

struct Group { size_t lo, hi; uint64_t bytes; };

}  // namespace

struct gci_heads {
    uint8_t* stream = nullptr;        // mmap'ed, reserved = inflated size of the file (an upper bound), touched = n_bytes
    uint64_t reserved = 0, n_bytes = 0, first_record = 0;
    std::vector<uint64_t> offs;
    ~gci_heads() { if (stream) munmap(stream, reserved); }
};

extern "C" int gci_bam_heads(const uint8_t* h_raw, uint64_t n_raw, int threads, uint64_t group_bytes, int check_crc,
                             gci_heads** out)
{
    if ((!h_raw && n_raw) || !out) return GCI_E_INVALID;
    *out = nullptr;
    std::vector<Block> blocks;
    uint64_t total = 0;
    const int st = scan(h_raw, n_raw, blocks, total);
    if (st) return st;
    if (group_bytes == 0) group_bytes = 16ull << 20;
    // groups of whole members, at most group_bytes each (a member inflates to <= 64 KiB; at least one per group)
    std::vector<Group> groups;
    uint64_t cap = 0;
    for (size_t i = 0; i < blocks.size();) {
        size_t j = i;
        uint64_t acc = 0;
        while (j < blocks.size() && (j == i || acc + blocks[j].isize <= group_bytes)) acc += blocks[j++].isize;
        groups.push_back({i, j, acc});
        cap = acc > cap ? acc : cap;
        i = j;
    }
    gci_heads* H = new (std::nothrow) gci_heads;
    if (!H) return GCI_E_NOMEM;
    H->reserved = total + 64;
    void* m = mmap(nullptr, H->reserved, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (m == MAP_FAILED) { delete H; return GCI_E_NOMEM; }
    H->stream = (uint8_t*)m;
    std::unique_ptr<uint8_t[]> bufs[3];                        // not zero-filled: first touched by the inflating threads
    for (auto& b : bufs) {
        b.reset(new (std::nothrow) uint8_t[cap + 64]);
        if (!b) { delete H; return GCI_E_NOMEM; }
    }

    const int n_groups = (int)groups.size();
    if (threads < 2) threads = 2;                               // the chaser + at least one worker
    const int n_workers = threads - 1;
    Barrier bar((uint32_t)n_workers + 1);
    std::atomic<int> err{GCI_OK};
    std::vector<HeadTask> tasks[2];                            // tasks[g & 1] = records of group g that lie inside its buffer
    const int n_steps = n_groups + 2;
    std::unique_ptr<std::atomic<uint64_t>[]> next_member(new std::atomic<uint64_t>[n_steps]);
    std::unique_ptr<std::atomic<uint64_t>[]> next_task(new std::atomic<uint64_t>[n_steps]);
    for (int s = 0; s < n_steps; s++) { next_member[s].store(0); next_task[s].store(0); }

    // step s: the workers inflate group s (into buffer s % 3) and copy the heads of group s - 2 (out of buffer
    // (s - 2) % 3) while the caller chases group s - 1 (buffer (s - 1) % 3); one rendezvous per step
    auto worker = [&]() {
        for (int s = 0; s < n_steps; s++) {
            if (s < n_groups && err.load(std::memory_order_relaxed) == GCI_OK) {
                const Group& g = groups[s];
                uint8_t* buf = bufs[s % 3].get();
                for (;;) {
                    const uint64_t a = next_member[s].fetch_add(4);
                    if (a >= g.hi - g.lo) break;
                    for (uint64_t k = a; k < a + 4 && k < g.hi - g.lo; k++) {
                        const Block& b = blocks[g.lo + k];
                        if (!inflate_member(h_raw, b, buf + (b.out - blocks[g.lo].out), check_crc)) err = GCI_E_MALFORMED;
                    }
                }
            }
            if (s >= 2) {
                const std::vector<HeadTask>& T = tasks[s & 1];
                for (;;) {
                    const uint64_t a = next_task[s].fetch_add(64);
                    if (a >= T.size()) break;
                    for (uint64_t k = a; k < a + 64 && k < T.size(); k++) head_copy(H->stream, T[k]);
                }
            }
            bar.wait();
        }
    };
    std::vector<std::thread> pool;
    for (int t = 0; t < n_workers; t++) pool.emplace_back(worker);

    // ---- the chaser
    std::vector<uint8_t> carry;                                 // a header or record that straddles groups, assembled here
    bool in_header = true;
    uint64_t w = 0;                                             // bytes of heads stream assigned so far
    int chase_err = GCI_OK;
    auto emit_direct = [&](const uint8_t* p) {                   // a record assembled in `carry`: copied here and now
        HeadTask t;
        head_shape(p, t);
        t.dst = w;
        head_copy(H->stream, t);
        H->offs.push_back(w);
        w += t.head + t.aux;
    };
    auto chase = [&](const uint8_t* D, uint64_t n, std::vector<HeadTask>& T) {
        uint64_t p = 0;
        if (in_header) {
            const uint8_t* src = D;
            uint64_t len = n;
            if (!carry.empty()) { carry.insert(carry.end(), D, D + n); src = carry.data(); len = carry.size(); }
            const int64_t hl = bam_header_len(src, len);
            if (hl < 0) { chase_err = GCI_E_MALFORMED; return; }
            if (hl == 0) { if (carry.empty()) carry.assign(D, D + n); return; }
            memcpy(H->stream, src, (size_t)hl);
            w = H->first_record = (uint64_t)hl;
            in_header = false;
            if (!carry.empty()) {                               // header longer than a group (rare): go on inside `carry`
                std::vector<uint8_t> rest(carry.begin() + hl, carry.end());
                carry.clear();
                uint64_t q = 0;
                while (q + 4 <= rest.size()) {
                    int32_t bs;
                    memcpy(&bs, rest.data() + q, 4);
                    if (bs < 32) { chase_err = GCI_E_MALFORMED; return; }
                    if (q + 4 + (uint64_t)bs > rest.size()) break;
                    emit_direct(rest.data() + q);
                    q += 4 + (uint64_t)bs;
                }
                carry.assign(rest.begin() + q, rest.end());
                return;
            }
            p = (uint64_t)hl;
        } else if (!carry.empty()) {
            // complete the straddling record from the front of this group
            while (p < n) {
                if (carry.size() < 4) { carry.push_back(D[p++]); continue; }
                int32_t bs;
                memcpy(&bs, carry.data(), 4);
                if (bs < 32) { chase_err = GCI_E_MALFORMED; return; }
                const uint64_t need = 4ull + (uint64_t)bs - carry.size();
                const uint64_t take = need < n - p ? need : n - p;
                carry.insert(carry.end(), D + p, D + p + take);
                p += take;
                if (take == need) { emit_direct(carry.data()); carry.clear(); break; }
            }
            if (!carry.empty()) return;                          // the group ended inside the record
        }
        while (p + 4 <= n) {
            int32_t bs;
            memcpy(&bs, D + p, 4);
            if (bs < 32) { chase_err = GCI_E_MALFORMED; return; }
            if (p + 4 + (uint64_t)bs > n) break;
            HeadTask t;
            head_shape(D + p, t);
            t.dst = w;
            T.push_back(t);
            H->offs.push_back(w);
            w += t.head + t.aux;
            p += 4 + (uint64_t)bs;
        }
        carry.assign(D + p, D + n);
    };

    for (int s = 0; s < n_steps; s++) {
        if (s >= 1 && s <= n_groups) {
            std::vector<HeadTask>& T = tasks[(s - 1) & 1];
            T.clear();
            if (!chase_err && err.load() == GCI_OK) chase(bufs[(s - 1) % 3].get(), groups[s - 1].bytes, T);
            if (chase_err) { err = chase_err; T.clear(); }
        } else if (s > n_groups) {
            tasks[(s - 1) & 1].clear();
        }
        bar.wait();
    }
    for (auto& th : pool) th.join();
    int rc = err.load();
    if (rc == GCI_OK && (in_header || !carry.empty())) rc = GCI_E_MALFORMED;       // no header / truncated last record
    if (rc != GCI_OK) { delete H; return rc; }
    H->n_bytes = w;
    *out = H;
    return GCI_OK;
}

extern "C" uint64_t gci_bam_heads_bytes(const gci_heads* h) { return h ? h->n_bytes : 0; }
extern "C" uint64_t gci_bam_heads_count(const gci_heads* h) { return h ? h->offs.size() : 0; }
extern "C" uint64_t gci_bam_heads_first(const gci_heads* h) { return h ? h->first_record : 0; }
extern "C" const uint8_t* gci_bam_heads_stream(const gci_heads* h) { return h ? h->stream : nullptr; }
extern "C" const uint64_t* gci_bam_heads_offsets(const gci_heads* h) { return h ? h->offs.data() : nullptr; }
extern "C" int gci_bam_heads_free(gci_heads* h) { delete h; return GCI_OK; }

// ---- FASTA record index (host): byte offsets of every '>' that begins a line, in file order ---------------------------
// (SeqIO.parse at GCI.py:30 / :940 yields one record per such line.)  memchr over `threads` slices of the text.
extern "C" int gci_fasta_titles(const uint8_t* h_text, uint64_t n, int threads, uint64_t* h_pos, uint64_t cap, uint64_t* n_pos)
{
    if ((!h_text && n) || !n_pos || (!h_pos && cap)) return GCI_E_INVALID;
    if (threads < 1) threads = 1;
    const uint64_t slices = n < (1u << 20) ? 1 : (uint64_t)threads * 4;
    std::vector<std::vector<uint64_t>> found(slices);
    parallel_for(slices, threads, [&](uint64_t i) {
        const uint64_t a = n * i / slices, b = n * (i + 1) / slices;
        const uint8_t* p = h_text + a;
        while (p < h_text + b) {
            p = (const uint8_t*)memchr(p, '>', (size_t)(h_text + b - p));
            if (!p) break;
            if (p == h_text || p[-1] == '\n') found[i].push_back((uint64_t)(p - h_text));
            p++;
        }
    });
    uint64_t k = 0;
    for (auto& f : found) k += f.size();
    *n_pos = k;
    if (k > cap) return h_pos ? GCI_E_CAPACITY : GCI_OK;
    k = 0;
    for (auto& f : found) for (uint64_t v : f) h_pos[k++] = v;
    return GCI_OK;
}
