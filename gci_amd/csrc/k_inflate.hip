// k_inflate.hip -- N1 on the GPU: BGZF members inflated by the device and the BAM record walk without its serial chain
// (what pysam / htslib do for the reference at GCI.py:150-151).
//
// gci_bgzf_inflate_device: one wave per BGZF member (a gzip member of at most 64 KiB of payload, RFC 1951 / 1952).  DEFLATE
// is a serial bit stream, so the parallelism is across members (56 k for a chr19 40x HiFi file, millions for a genome); inside
// a member the wave decodes symbol by symbol (every lane holds the same state, so the LDS reads are broadcasts) and the
// lanes share what is parallel: building the decode tables, copying a match, moving finished output to HBM.
//   LDS per wave (= workgroup): a 32 KiB sliding window of the output (DEFLATE's maximum distance; a half is written
//   to global memory with aligned 16-byte stores as soon as it is complete), 4 KiB of the member's input (refilled
//   half by half, so the 32-bit bit-buffer refills never leave LDS), a 10-bit primary table for the literal / length code
//   and an 8-bit one for distances; longer codes (rare) are decoded bit by bit from the canonical first-code tables.
//   CRC-32 of the output is verified per member (as htslib does): every lane the CRC of its share of a 16 KiB half,
//   the shares concatenated with the GF(2) rule crc(A||B) = crc(A) * x^(8|B|) + crc(B).
//
// gci_bam_record_offsets_device: the offsets of the records of an inflated BAM stream WITHOUT walking the block_size
// chain serially (137 k dependent loads of ~1 us each at chr19): every byte position is tested for "a record could
// start here" (block_size, refID, pos, l_read_name, l_seq, next_refID, next_pos consistent with the format -- SEQ / QUAL
// bytes never pass), each candidate's successor (offset + 4 + block_size) is looked up among the candidates, and the
// candidates reachable from the first record are found by pointer doubling.  The chain must end exactly at the end of
// the stream (or, for a chunk, in a record that runs past it); otherwise -- a record the strict test rejects -- the call
// reports GCI_E_MALFORMED and the caller takes the host path.
#include "gci_ctx.hpp"
#include <stdlib.h>

namespace {

#define INF_WIN 32768u
#define INF_HALF 16384u
#define INF_IN 2048u
#define LIT_BITS 10
#define DIST_BITS 8

__constant__ uint16_t c_len_base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
__constant__ uint8_t c_len_extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
__constant__ uint16_t c_dist_base[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073,
                                         4097, 6145, 8193, 12289, 16385, 24577};
__constant__ uint8_t c_dist_extra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
__constant__ uint8_t c_clen_order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

constexpr uint32_t CRC_POLY = 0xEDB88320u;      // reflected: bit 31 holds x^0

__device__ __forceinline__ uint32_t gf_mul(uint32_t a, uint32_t b)   // a * b mod P over GF(2), reflected representation
{
    uint32_t r = 0;
    for (int i = 0; i < 32; i++) {
        r ^= (a & 0x80000000u) ? b : 0u;
        a <<= 1;
        b = (b >> 1) ^ ((b & 1u) ? CRC_POLY : 0u);
    }
    return r;
}

// Canonical Huffman code of one alphabet: per length the first code and the index of its first symbol in `sorted`
// (RFC 1951 3.2.2), and a primary table indexed by the next PRIMARY bits of the stream (codes are packed starting from
// their most significant bit, i.e. bit reversed in the LSB-first bit buffer): entry = symbol | length << 9, 0 = the code is
// longer than PRIMARY bits (or unused).
struct Canon { uint16_t count[16], first_code[16], first_idx[16], next[16]; };

__device__ __forceinline__ uint32_t bit_reverse(uint32_t v, int n) { return __brev(v) >> (32 - n); }

// Every lane runs this with the same arguments (lane 0 writes): lens[0 .. n) in LDS -> canon, sorted symbols, primary table.
// false: over-subscribed code.  (An incomplete code is accepted as zlib accepts it for a single distance code.)
template <int PRIMARY>
__device__ bool build_code(const uint8_t* lens, int n, Canon& cn, uint16_t* sorted, uint16_t* table, int lane)
{
    for (int i = lane; i < (1 << PRIMARY); i += 64) table[i] = 0;
    if (lane < 16) { cn.count[lane] = 0; }
    __syncthreads();
    if (lane == 0) {
        for (int i = 0; i < n; i++) cn.count[lens[i]]++;
        cn.count[0] = 0;
        uint32_t code = 0, idx = 0;
        for (int l = 1; l < 16; l++) {
            code = (code + cn.count[l - 1]) << 1;
            cn.first_code[l] = (uint16_t)code;
            cn.first_idx[l] = (uint16_t)idx;
            idx += cn.count[l];
        }
    }
    __syncthreads();
    // over-subscription: sum count[l] * 2^(15 - l) must not exceed 2^15
    uint32_t left = 1u << 15;
    bool ok = true;
    for (int l = 1; l < 16; l++) { const uint32_t need = (uint32_t)cn.count[l] << (15 - l); if (need > left) ok = false; else left -= need; }
    if (!ok) return false;
    if (lane == 0) {
        for (int l = 0; l < 16; l++) cn.next[l] = 0;
        for (int s = 0; s < n; s++) {
            const int l = lens[s];
            if (!l) continue;
            const uint32_t rank = cn.next[l]++;
            sorted[cn.first_idx[l] + rank] = (uint16_t)s;
            if (l <= PRIMARY) {
                const uint32_t rev = bit_reverse(cn.first_code[l] + rank, l);
                for (uint32_t k = rev; k < (1u << PRIMARY); k += 1u << l) table[k] = (uint16_t)(s | (l << 9));
            }
        }
    }
    __syncthreads();
    return true;
}

struct Bits {
    unsigned long long bb;      // bit buffer, next bit = bit 0
    int bn;                     // valid bits
    uint32_t ip;                // next payload byte to take (offset inside the payload)
};

}  // namespace

// One workgroup = one wave = one member.
__global__ __launch_bounds__(64) void k_bgzf_inflate(const uint8_t* __restrict__ raw, const uint64_t* __restrict__ member_pos,
                                                     const uint64_t* __restrict__ out_off, uint32_t n_members, uint8_t* __restrict__ out,
                                                     uint64_t out_cap, int check_crc, unsigned long long* __restrict__ status)
{
    __shared__ __attribute__((aligned(16))) uint8_t win[INF_WIN];
    __shared__ __attribute__((aligned(16))) uint8_t inb[INF_IN + 16];
    __shared__ uint16_t lit_tab[1 << LIT_BITS], dist_tab[1 << DIST_BITS];
    __shared__ uint16_t lit_sorted[288], dist_sorted[32];
    __shared__ uint8_t lens[384];                                 // [0, 19): code-length code; [32, 32 + 286 + 30): both alphabets
    __shared__ Canon lit_cn, dist_cn, cl_cn;
    __shared__ uint16_t cl_tab[1 << 7], cl_sorted[19];
    __shared__ uint32_t crc_tab[16];                              // CRC-32 a nibble at a time (LDS is needed for the window)
    const int lane = threadIdx.x;
    const uint32_t m = blockIdx.x;
    if (m >= n_members) return;
    const uint64_t pos = member_pos[m], pos_next = member_pos[m + 1];
    const uint64_t o0 = out_off[m];
    const uint32_t isize = (uint32_t)(out_off[m + 1] - o0);
    auto fail = [&](int code) __attribute__((always_inline)) { if (lane == 0) atomicMin(status, ((unsigned long long)m << 8) | (unsigned long long)(uint8_t)(-code)); };
    if (pos_next < pos + 26 || o0 + isize > out_cap || isize > 65536u) { fail(GCI_E_MALFORMED); return; }
    const uint32_t xlen = (uint32_t)raw[pos + 10] | ((uint32_t)raw[pos + 11] << 8);
    const uint64_t pay0 = pos + 12 + xlen;
    if (pay0 + 8 > pos_next || raw[pos] != 0x1f || raw[pos + 1] != 0x8b || raw[pos + 2] != 8) { fail(GCI_E_MALFORMED); return; }
    const uint32_t pay_len = (uint32_t)(pos_next - 8 - pay0);
    const uint8_t* __restrict__ pay = raw + pay0;
    if (check_crc && lane < 16) {
        uint32_t c = (uint32_t)lane;
        for (int k = 0; k < 4; k++) c = (c >> 1) ^ ((c & 1u) ? CRC_POLY : 0u);
        crc_tab[lane] = c;
    }
    // ---- input window: inb holds payload bytes [in_base, in_base + INF_IN) ---------------------------------------------
    uint32_t in_base = 0;
    auto load_in = [&](uint32_t from, uint32_t to_slot, uint32_t n) __attribute__((always_inline)) {          // payload[from .. from + n) -> inb[to_slot ..]
        for (uint32_t i = lane; i < n; i += 64) inb[to_slot + i] = from + i < pay_len ? pay[from + i] : 0;
    };
    load_in(0, 0, INF_IN + 16);
    __syncthreads();
    Bits B;
    B.bb = 0; B.bn = 0; B.ip = 0;
    bool bad = false;
    // at least 32 valid bits (zero bits past the end of the payload: an over-read is caught by the ip check at the end)
    auto need32 = [&]() __attribute__((always_inline)) {
        if (B.bn < 32) {
            if (B.ip - in_base > INF_IN - 8) {                                 // slide: the upper half moves down, a new half comes in
                const uint32_t keep_from = (B.ip - in_base) & ~15u;            // 16-byte granules keep the copies aligned
                for (uint32_t i = lane * 16u; keep_from + i < INF_IN + 16; i += 64 * 16u)
                    *reinterpret_cast<uint4*>(inb + i) = *reinterpret_cast<const uint4*>(inb + keep_from + i);
                __syncthreads();
                const uint32_t have = INF_IN + 16 - keep_from;
                in_base += keep_from;
                load_in(in_base + have, have, keep_from);
                __syncthreads();
            }
            uint32_t w;
            __builtin_memcpy(&w, inb + (B.ip - in_base), 4);
            B.bb |= (unsigned long long)w << B.bn;
            B.ip += 4; B.bn += 32;
        }
    };
    auto take = [&](int n) __attribute__((always_inline)) -> uint32_t { const uint32_t v = (uint32_t)(B.bb & ((1ull << n) - 1ull)); B.bb >>= n; B.bn -= n; return v; };
    // one symbol of a code: primary table, else bit by bit along the canonical first codes (codes of PRIMARY + 1 .. 15 bits)
    auto decode = [&](const uint16_t* table, int primary, const Canon& cn, const uint16_t* sorted) __attribute__((always_inline)) -> int {
        const uint32_t e = table[(uint32_t)B.bb & ((1u << primary) - 1u)];
        if (e) { const int l = (int)(e >> 9); B.bb >>= l; B.bn -= l; return (int)(e & 0x1FFu); }
        uint32_t code = 0;
        for (int l = 1; l < 16; l++) {
            code = (code << 1) | (uint32_t)((B.bb >> (l - 1)) & 1ull);
            const uint32_t rel = code - cn.first_code[l];
            if (l > primary && cn.count[l] && rel < cn.count[l]) { B.bb >>= l; B.bn -= l; return (int)sorted[cn.first_idx[l] + rel]; }
        }
        return -1;
    };
    uint32_t op = 0;                                            // bytes of output so far
    uint32_t flushed = 0;                                       // ... of which in global memory
    uint32_t crc = 0;                                           // finalised CRC-32 of the flushed output
    uint8_t* __restrict__ dst = out + o0;
    // a complete part [flushed, upto) of the window -> global memory (+ its CRC); parts end at INF_HALF boundaries or at the end
    auto flush = [&](uint32_t upto) __attribute__((always_inline)) {
        __syncthreads();
        const uint32_t n = upto - flushed;
        // head bytes up to the next 16-byte boundary of the DESTINATION, then aligned 16-byte stores, then the tail
        const uint64_t d0 = (uint64_t)(uintptr_t)(dst + flushed);
        uint32_t head = (uint32_t)((16 - (d0 & 15)) & 15);
        if (head > n) head = n;
        for (uint32_t i = lane; i < head; i += 64) dst[flushed + i] = win[(flushed + i) & (INF_WIN - 1)];
        const uint32_t body = (n - head) & ~15u;
        for (uint32_t i = lane * 16u; i < body; i += 64 * 16u) {
            const uint32_t s = flushed + head + i;
            uint32_t w[4];                                                     // (a part never wraps around the window: it ends at a half boundary)
#pragma unroll
            for (int k = 0; k < 4; k++) __builtin_memcpy(&w[k], win + ((s & (INF_WIN - 1)) + 4 * k), 4);
            *reinterpret_cast<uint4*>(dst + s) = make_uint4(w[0], w[1], w[2], w[3]);
        }
        for (uint32_t i = head + body + lane; i < n; i += 64) dst[flushed + i] = win[(flushed + i) & (INF_WIN - 1)];
        if (check_crc && n) {
            // lane l: the CRC register of its share, started at 0 (pure polynomial remainder); shares concatenate as
            // reg(A||B) = reg(A) * x^(8|B|) + reg(B); the running value `crc` is kept finalised-free the same way
            const uint32_t per = (n + 63) / 64;
            const uint32_t a = min(n, (uint32_t)lane * per), b = min(n, a + per);
            uint32_t c = 0;
            for (uint32_t i = a; i < b; i++) {
                c ^= win[(flushed + i) & (INF_WIN - 1)];
                c = crc_tab[c & 0xFu] ^ (c >> 4);
                c = crc_tab[c & 0xFu] ^ (c >> 4);
            }
            // x^(8 len) for len = b - a, by square and multiply on x^8
            uint32_t xp = 0x80000000u, base = 0x00800000u;                    // x^0, x^8 (reflected)
            for (uint32_t e = b - a; e; e >>= 1) { if (e & 1u) xp = gf_mul(xp, base); base = gf_mul(base, base); }
            // ordered product over the lanes (tree): (c, x) . (c', x') = (c * x' + c', x * x')
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t oc = (uint32_t)__shfl_down((int)c, d, 64), ox = (uint32_t)__shfl_down((int)xp, d, 64);
                if ((lane & (2 * d - 1)) == 0) { c = gf_mul(c, ox) ^ oc; xp = gf_mul(xp, ox); }
            }
            const uint32_t part_c = (uint32_t)__shfl((int)c, 0, 64), part_x = (uint32_t)__shfl((int)xp, 0, 64);
            crc = gf_mul(crc, part_x) ^ part_c;
        }
        flushed = upto;
        __syncthreads();
    };
    auto emit_literal = [&](uint32_t v) __attribute__((always_inline)) {
        if (lane == 0) win[op & (INF_WIN - 1)] = (uint8_t)v;
        op++;
        if ((op & (INF_HALF - 1)) == 0) flush(op);
    };
    // ---- blocks ------------------------------------------------------------------------------------------------------
    for (bool last = false; !last && !bad;) {
        need32();
        last = take(1) != 0;
        const uint32_t type = take(2);
        if (type == 0) {                                                     // stored
            take(B.bn & 7);                                                  // to the byte boundary
            need32();
            const uint32_t len = take(16), nlen = take(16);
            if ((len ^ 0xFFFFu) != nlen || op + len > isize) { bad = true; break; }
            // the bytes still in the bit buffer first, then straight from the payload
            uint32_t done = 0;
            while (done < len && B.bn >= 8) { emit_literal(take(8)); done++; }
            if (done < len) {                                                // the bit buffer is empty (it held whole bytes): from the payload
                B.bb = 0; B.bn = 0;
                while (done < len) {
                    const uint32_t room = INF_HALF - (op & (INF_HALF - 1));  // up to the next half boundary
                    const uint32_t n = min(len - done, room);
                    if (B.ip + n > pay_len) { bad = true; break; }
                    for (uint32_t i = lane; i < n; i += 64) win[(op + i) & (INF_WIN - 1)] = pay[B.ip + i];
                    B.ip += n; op += n; done += n;
                    if ((op & (INF_HALF - 1)) == 0) flush(op);
                }
                // the input window no longer matches ip: reload it
                in_base = B.ip & ~15u;
                __syncthreads();
                load_in(in_base, 0, INF_IN + 16);
                __syncthreads();
            }
            continue;
        }
        if (type == 3) { bad = true; break; }
        if (type == 1) {                                                     // fixed code (RFC 1951 3.2.6)
            for (int i = lane; i < 288; i += 64) lens[32 + i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8;
            for (int i = lane; i < 30; i += 64) lens[32 + 288 + i] = 5;
            __syncthreads();
            if (!build_code<LIT_BITS>(lens + 32, 288, lit_cn, lit_sorted, lit_tab, lane) ||
                !build_code<DIST_BITS>(lens + 32 + 288, 30, dist_cn, dist_sorted, dist_tab, lane)) { bad = true; break; }
        } else {                                                             // dynamic code (3.2.7)
            need32();
            const int hlit = (int)take(5) + 257, hdist = (int)take(5) + 1, hclen = (int)take(4) + 4;
            if (hlit > 286 || hdist > 30) { bad = true; break; }
            if (lane < 19) lens[lane] = 0;
            __syncthreads();
            for (int i = 0; i < hclen; i++) { need32(); const uint32_t v = take(3); if (lane == 0) lens[c_clen_order[i]] = (uint8_t)v; }
            __syncthreads();
            if (!build_code<7>(lens, 19, cl_cn, cl_sorted, cl_tab, lane)) { bad = true; break; }
            // the code lengths of both alphabets, run-length coded; lens[32 ..] so that lens[0 .. 19) stays the code-length code
            int n = 0, prev = 0;
            uint8_t* ll = lens + 32;
            while (n < hlit + hdist) {
                need32();
                const int sym = decode(cl_tab, 7, cl_cn, cl_sorted);
                if (sym < 0) { bad = true; break; }
                int rep = 1, val = sym;
                if (sym == 16) { if (n == 0) { bad = true; break; } val = prev; rep = 3 + (int)take(2); }
                else if (sym == 17) { val = 0; rep = 3 + (int)take(3); }
                else if (sym == 18) { val = 0; rep = 11 + (int)take(7); }
                if (n + rep > hlit + hdist) { bad = true; break; }
                for (int i = lane; i < rep; i += 64) ll[n + i] = (uint8_t)val;
                n += rep; prev = val;
            }
            if (bad) break;
            __syncthreads();
            if (ll[256] == 0) { bad = true; break; }                          // no end-of-block code
            if (!build_code<LIT_BITS>(ll, hlit, lit_cn, lit_sorted, lit_tab, lane) ||
                !build_code<DIST_BITS>(ll + hlit, hdist, dist_cn, dist_sorted, dist_tab, lane)) { bad = true; break; }
        }
        // ---- symbols of the block ------------------------------------------------------------------------------------------
        for (;;) {
            need32();
            const int sym = decode(lit_tab, LIT_BITS, lit_cn, lit_sorted);
            if (sym < 0 || sym > 285) { bad = true; break; }
            if (sym < 256) {
                if (op >= isize) { bad = true; break; }
                emit_literal((uint32_t)sym);
                continue;
            }
            if (sym == 256) break;
            const uint32_t len = c_len_base[sym - 257] + take(c_len_extra[sym - 257]);
            need32();
            const int ds = decode(dist_tab, DIST_BITS, dist_cn, dist_sorted);
            if (ds < 0 || ds > 29) { bad = true; break; }
            const uint32_t dist = c_dist_base[ds] + take(c_dist_extra[ds]);
            if (dist > op || op + len > isize) { bad = true; break; }
            // the match, at most up to the next half boundary at a time; byte i comes from i mod dist when it overlaps
            __syncthreads();
            uint32_t done = 0;
            while (done < len) {
                const uint32_t room = INF_HALF - (op & (INF_HALF - 1));
                const uint32_t n = min(len - done, room);
                for (uint32_t i0 = 0; i0 < n; i0 += 64) {
                    const uint32_t i = i0 + lane;
                    uint8_t v = 0;
                    // a chunk of 64 bytes may only read what is already written: split at multiples of dist when dist < 64
                    if (i < n) v = win[(op - dist + ((done + i) % dist) - done) & (INF_WIN - 1)];
                    __syncthreads();
                    if (i < n) win[(op + i) & (INF_WIN - 1)] = v;
                    __syncthreads();
                }
                op += n; done += n;
                if ((op & (INF_HALF - 1)) == 0) flush(op);
            }
        }
    }
    if (!bad && op != isize) bad = true;
    if (!bad && B.ip - (uint32_t)(B.bn >> 3) > pay_len) bad = true;            // the stream ran past the payload
    if (bad) { fail(GCI_E_MALFORMED); return; }
    if (flushed < op) flush(op);
    if (check_crc) {
        // `crc` is the remainder of the message polynomial; the CRC-32 of n bytes is that of the message with the first 32
        // bits complemented, complemented: crc32(M) = reg_ff(M) ^ ~0 with reg_ff(M) = reg_0(M) ^ (0xFFFFFFFF * x^(8n))
        uint32_t xp = 0x80000000u, base = 0x00800000u;
        for (uint32_t e = isize; e; e >>= 1) { if (e & 1u) xp = gf_mul(xp, base); base = gf_mul(base, base); }
        const uint32_t got = crc ^ gf_mul(0xFFFFFFFFu, xp) ^ 0xFFFFFFFFu;
        const uint64_t tp = pos_next - 8;
        const uint32_t want = (uint32_t)raw[tp] | ((uint32_t)raw[tp + 1] << 8) | ((uint32_t)raw[tp + 2] << 16) | ((uint32_t)raw[tp + 3] << 24);
        if (got != want) fail(GCI_E_MALFORMED);
    }
}

extern "C" int gci_bgzf_inflate_device(gci_ctx* ctx, const uint8_t* d_raw, const uint64_t* d_member_pos, const uint64_t* d_out_off,
                                       uint32_t n_members, uint8_t* d_out, uint64_t out_cap, int check_crc, uint64_t* d_status)
{
    if (!ctx || !d_status || (n_members && (!d_raw || !d_member_pos || !d_out_off || !d_out))) return GCI_E_INVALID;
    HIPCHK(hipMemsetAsync(d_status, 0xFF, 8, ctx->stream));
    if (n_members) {
        hipLaunchKernelGGL(k_bgzf_inflate, dim3(n_members), dim3(64), 0, ctx->stream, d_raw, d_member_pos, d_out_off, n_members, d_out,
                           out_cap, check_crc, (unsigned long long*)d_status);
        LAUNCHCHK("k_bgzf_inflate");
    }
    return GCI_OK;
}

// =====================================================================================================================
// record offsets without the serial chain
// =====================================================================================================================
namespace {

// "a BAM record could start at p" -- every field a well-formed record constrains (SAM spec 4.2)
__device__ __forceinline__ bool plausible(const uint8_t* c /* 36 bytes */, int32_t n_ref)
{
    int32_t w[9];
    __builtin_memcpy(w, c, 36);
    const int32_t block_size = w[0], ref_id = w[1], pos = w[2], l_seq = w[5], next_ref = w[6], next_pos = w[7];
    const uint32_t l_read_name = (uint32_t)w[3] & 0xFFu, n_cigar = (uint32_t)w[4] & 0xFFFFu;
    if (block_size < 32 || ref_id < -1 || ref_id >= n_ref || pos < -1 || l_read_name < 1 || l_seq < 0 || next_ref < -1 || next_ref >= n_ref ||
        next_pos < -1)
        return false;
    const uint64_t need = 32ull + l_read_name + 4ull * n_cigar + (((uint64_t)(uint32_t)l_seq + 1) >> 1) + (uint64_t)(uint32_t)l_seq;
    return need <= (uint64_t)(uint32_t)block_size;
}

#define CAND_HALO 48
__global__ __launch_bounds__(BLOCK) void k_rec_candidates(const uint8_t* __restrict__ s, uint64_t lo, uint64_t n, int32_t n_ref,
                                                          const uint32_t* __restrict__ tile_off, uint32_t* __restrict__ tile_count,
                                                          uint64_t* __restrict__ cand)
{
    __shared__ __attribute__((aligned(16))) uint8_t buf[TILE + CAND_HALO];
    __shared__ uint32_t wtot[BLOCK / 64];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const uint64_t base = lo + (uint64_t)blockIdx.x * TILE;
    for (uint32_t i = t; i < TILE + CAND_HALO; i += BLOCK) buf[i] = base + i < n ? s[base + i] : 0xFF;
    __syncthreads();
    uint32_t mask = 0;
    for (int i = 0; i < 16; i++) {
        const uint64_t p = base + (uint64_t)t * 16 + i;
        if (p + 36 <= n && plausible(buf + t * 16 + i, n_ref)) mask |= 1u << i;
    }
    const uint32_t cnt = (uint32_t)__builtin_popcount(mask);
    const uint32_t inc = wave_inclusive<uint32_t>(cnt, lane);
    if (lane == 63) wtot[wave] = inc;
    __syncthreads();
    if (!cand) {                                                             // counting pass
        if (t == 0) tile_count[blockIdx.x] = wtot[0] + wtot[1] + wtot[2] + wtot[3];
        return;
    }
    uint32_t w = tile_off[blockIdx.x] + inc - cnt;
    for (int k = 0; k < wave; k++) w += wtot[k];
    for (uint32_t m = mask; m; m &= m - 1) cand[w++] = base + (uint64_t)t * 16 + (uint32_t)__builtin_ctz(m);
}

#define NODE_END 0xFFFFFFFEu          // the record ends exactly at the end of the stream
#define NODE_TAIL 0xFFFFFFFDu         // the record runs past the end of the stream (a chunk's partial last record)
#define NODE_HEAD 0xFFFFFFFCu         // the record is complete and fewer than 36 bytes follow it (a chunk's partial record head)
#define NODE_NONE 0xFFFFFFFFu         // its successor is not a candidate: not on the chain (or the chain is broken there)
__global__ __launch_bounds__(BLOCK) void k_rec_link(const uint8_t* __restrict__ s, uint64_t n, const uint64_t* __restrict__ cand, uint32_t nc,
                                                    uint32_t* __restrict__ next)
{
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= nc) return;
    const uint64_t p = cand[i];
    int32_t bs;
    __builtin_memcpy(&bs, s + p, 4);                                         // (a byte-wise copy: p has any alignment)
    const uint64_t q = p + 4 + (uint64_t)(uint32_t)bs;
    uint32_t r = NODE_NONE;
    if (q == n) r = NODE_END;
    else if (q > n) r = NODE_TAIL;
    else {
        uint32_t lo = i + 1, hi = nc;                                        // first candidate >= q
        while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (cand[mid] < q) lo = mid + 1; else hi = mid; }
        if (lo < nc && cand[lo] == q) r = lo;
        else if (q + 36 > n) r = NODE_HEAD;                                   // fewer than 36 bytes left: a partial record head
    }
    next[i] = r;
}

// jump_out[i] = jump_in[jump_in[i]] (terminal values stay)
__global__ __launch_bounds__(BLOCK) void k_rec_double(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint32_t nc)
{
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= nc) return;
    const uint32_t a = in[i];
    out[i] = a < nc ? in[a] : a;
}

// marked nodes mark their 2^k-th successor
__global__ __launch_bounds__(BLOCK) void k_rec_mark(const uint32_t* __restrict__ jump, uint32_t* __restrict__ mark, uint32_t nc)
{
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= nc || !mark[i]) return;
    const uint32_t a = jump[i];
    if (a < nc) mark[a] = 1u;
}

__global__ __launch_bounds__(BLOCK) void k_rec_emit(const uint8_t* __restrict__ s, const uint64_t* __restrict__ cand,
                                                    const uint32_t* __restrict__ mark, const uint32_t* __restrict__ pos,
                                                    const uint32_t* __restrict__ next, uint32_t nc, uint64_t n, uint64_t* __restrict__ offs,
                                                    uint64_t cap, uint64_t* __restrict__ result)
{
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= nc || !mark[i]) return;
    const uint32_t nx = next[i];
    // the last record of the chain decides how the walk ended: result[1] = bytes consumed, result[2] = 0 ok / 1 broken
    if (nx == NODE_TAIL) { result[1] = cand[i]; return; }                    // partial last record: not emitted
    if (pos[i] < cap) offs[pos[i]] = cand[i];
    if (nx == NODE_END) result[1] = n;
    else if (nx == NODE_HEAD) { int32_t bs; __builtin_memcpy(&bs, s + cand[i], 4); result[1] = cand[i] + 4 + (uint64_t)(uint32_t)bs; }
    else if (nx == NODE_NONE) { result[1] = cand[i]; result[2] = 1; }
}

__global__ __launch_bounds__(BLOCK) void k_rec_flags(const uint32_t* __restrict__ mark, const uint32_t* __restrict__ next, uint32_t nc,
                                                     uint32_t* __restrict__ flag)
{
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i < nc) flag[i] = (mark[i] && next[i] != NODE_TAIL) ? 1u : 0u;
}

}  // namespace

// d_result (device, 3 x uint64): [0] = number of records, [1] = bytes consumed (n_bytes when the last record ends there, else
// the offset of the partial last record), [2] = 0 ok / 1 the chain broke at offset [1] (GCI_E_MALFORMED on the host side).
// d_offs: up to `cap` offsets; [0] may exceed cap (call again with more room).  first_record: offset of the first record
// (the end of the BAM header), which must pass the strict test itself.
extern "C" int gci_bam_record_offsets_device(gci_ctx* ctx, const uint8_t* d_stream, uint64_t n_bytes, uint64_t first_record, int32_t n_ref,
                                             uint64_t* d_offs, uint64_t cap, uint64_t* d_result)
{
    if (!ctx || !d_result || (n_bytes && !d_stream) || first_record > n_bytes || (cap && !d_offs)) return GCI_E_INVALID;
    hipStream_t st = ctx->stream;
    {
        const uint64_t init[3] = {0, first_record, 0};
        GCI_TRY(gci_upload_small(ctx, d_result, init, sizeof init));
    }
    if (n_bytes - first_record < 36) return GCI_OK;                            // no complete record head: everything is tail
    const uint64_t span = n_bytes - first_record;
    const uint64_t n_tiles64 = (span + TILE - 1) / TILE;
    if (n_tiles64 > 0x7fffffffULL) return GCI_E_INVALID;
    const uint32_t n_tiles = (uint32_t)n_tiles64;
    GCI_TRY(gci_ensure(ctx, ctx->part_hist, (size_t)(n_tiles + 2) * 4));
    GCI_TRY(gci_ensure(ctx, ctx->part_blk, (size_t)(n_tiles / TILE + 2) * 4));
    uint32_t* d_tile = (uint32_t*)ctx->part_hist.p;
    hipLaunchKernelGGL(k_rec_candidates, dim3(n_tiles), dim3(BLOCK), 0, st, d_stream, first_record, n_bytes, n_ref, (const uint32_t*)nullptr,
                       d_tile, (uint64_t*)nullptr);
    LAUNCHCHK("k_rec_candidates(count)");
    int r = device_exclusive_scan<uint32_t, uint32_t>(ctx, d_tile, d_tile, (uint32_t*)ctx->part_blk.p, (int64_t)n_tiles, true);
    if (r) return r;
    uint32_t nc = 0;
    HIPCHK(hipMemcpyAsync(&nc, d_tile + n_tiles, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (nc == 0) { const uint64_t res[3] = {0, first_record, 1}; return gci_upload_small(ctx, d_result, res, sizeof res); }
    // candidates (u64) | next, jump a, jump b, mark, flag, pos (u32 each)
    GCI_TRY(gci_ensure(ctx, ctx->part_a, (size_t)nc * 8 + 64));
    GCI_TRY(gci_ensure(ctx, ctx->part_b, (size_t)(nc + 1) * 4 * 6 + 64));
    uint64_t* d_cand = (uint64_t*)ctx->part_a.p;
    uint32_t* d_next = (uint32_t*)ctx->part_b.p;
    uint32_t* d_ja = d_next + (nc + 1);
    uint32_t* d_jb = d_ja + (nc + 1);
    uint32_t* d_mark = d_jb + (nc + 1);
    uint32_t* d_flag = d_mark + (nc + 1);
    uint32_t* d_pos = d_flag + (nc + 1);
    hipLaunchKernelGGL(k_rec_candidates, dim3(n_tiles), dim3(BLOCK), 0, st, d_stream, first_record, n_bytes, n_ref, (const uint32_t*)d_tile,
                       (uint32_t*)nullptr, d_cand);
    LAUNCHCHK("k_rec_candidates(write)");
    const dim3 grid((nc + BLOCK - 1) / BLOCK);
    hipLaunchKernelGGL(k_rec_link, grid, dim3(BLOCK), 0, st, d_stream, n_bytes, (const uint64_t*)d_cand, nc, d_next);
    LAUNCHCHK("k_rec_link");
    // The first record must be the first candidate.  Marks spread from it by pointer doubling: the 2^k-step successor
    // tables are built bottom up and kept on the device one after the other would need K tables; instead the marks are
    // spread bottom up as well -- after round k every node within 2^(k+1) - 1 steps of a marked node is marked, because
    // round k marks the 2^k-th successors of ALL nodes marked so far (distances 0 .. 2^k - 1 from node 0 become
    // 0 .. 2^(k+1) - 1).
    HIPCHK(hipMemsetAsync(d_mark, 0, (size_t)nc * 4, st));
    {
        uint64_t first_cand = 0;
        HIPCHK(hipMemcpyAsync(&first_cand, d_cand, 8, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        if (first_cand != first_record) { const uint64_t res[3] = {0, first_record, 1}; return gci_upload_small(ctx, d_result, res, sizeof res); }
        const uint32_t one = 1;
        GCI_TRY(gci_upload_small(ctx, d_mark, &one, 4));
    }
    HIPCHK(hipMemcpyAsync(d_ja, d_next, (size_t)nc * 4, hipMemcpyDeviceToDevice, st));
    uint32_t* jin = d_ja;
    uint32_t* jout = d_jb;
    for (uint64_t reach = 1; reach < (uint64_t)nc; reach <<= 1) {
        hipLaunchKernelGGL(k_rec_mark, grid, dim3(BLOCK), 0, st, (const uint32_t*)jin, d_mark, nc);
        hipLaunchKernelGGL(k_rec_double, grid, dim3(BLOCK), 0, st, (const uint32_t*)jin, jout, nc);
        uint32_t* t = jin; jin = jout; jout = t;
    }
    hipLaunchKernelGGL(k_rec_mark, grid, dim3(BLOCK), 0, st, (const uint32_t*)jin, d_mark, nc);
    LAUNCHCHK("k_rec_mark");
    hipLaunchKernelGGL(k_rec_flags, grid, dim3(BLOCK), 0, st, (const uint32_t*)d_mark, (const uint32_t*)d_next, nc, d_flag);
    LAUNCHCHK("k_rec_flags");
    GCI_TRY(gci_ensure(ctx, ctx->part_blk, (size_t)(nc / TILE + 2) * 4));
    r = device_exclusive_scan<uint32_t, uint32_t>(ctx, d_flag, d_pos, (uint32_t*)ctx->part_blk.p, (int64_t)nc, true);
    if (r) return r;
    hipLaunchKernelGGL(k_rec_emit, grid, dim3(BLOCK), 0, st, d_stream, (const uint64_t*)d_cand, (const uint32_t*)d_mark, (const uint32_t*)d_pos,
                       (const uint32_t*)d_next, nc, n_bytes, d_offs, cap, d_result);
    LAUNCHCHK("k_rec_emit");
    // the count: d_pos[nc] (u32) -> d_result[0] (u64)
    uint32_t n_rec = 0;
    HIPCHK(hipMemcpyAsync(&n_rec, d_pos + nc, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    const uint64_t n64 = n_rec;
    GCI_TRY(gci_upload_small(ctx, d_result, &n64, 8));
    return GCI_OK;
}
