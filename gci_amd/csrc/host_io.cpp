// host_io.cpp -- host-side helpers of libgci_hip.so for the containers either side of the GPU path
// (SURVEY.md section 8f, N1 / N2): parallel BGZF inflate, the BAM record-offset chase and parallel gzip
// framing of the depth text.  Plain C++ threads + zlib; no GPU work here.  The reference reaches these
// layers through pysam/htslib (GCI.py:150-151) and Python's gzip (GCI.py:111).
#include <stdint.h>
#include <string.h>
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
