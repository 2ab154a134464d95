/*
 * gci_hip.h -- C ABI of libgci_hip.so: the MI355X (gfx950) implementation of GCI's
 * alignment-filter -> per-base-depth -> issue-scan hot path.
 *
 * The reference (yeeus/GCI, a single pure-Python file) has no FFI; the seams below are the
 * function seams of its hot path, one export per seam so each is parity-testable against
 * the matching reference function (citations are into /root/reference/GCI.py):
 *
 *   gci_bam_filter      read_sam                      GCI.py:146-169  (+ fan-out 257-270)
 *   gci_name_join       filter(), cross-file join     GCI.py:272-301  (+ dict "last wins" 166, 269)
 *   gci_depth_build     filter(), slice += 1          GCI.py:201-208, 302-306
 *   gci_gap_mask        merge_gaps_depths             GCI.py:315-329
 *   gci_max2            merge_two_type_depth          GCI.py:350
 *   gci_issue_scan*     collapse_depth_range          GCI.py:356-390
 *   gci_depth_text_*    write_depth (text body)       GCI.py:110-117
 *   gci_depth_sum       np.mean numerator             GCI.py:862-868
 *   gci_range_sums      sliding_window_average_depth  GCI.py:660-705 (window sums)
 *   gci_fasta_n_scan    get_Ns_ref                    GCI.py:27-35
 *   gci_paf_filter_device   filter(), PAF path        GCI.py:211-254  (+ helpers 49-61, 64-96)
 *
 * Conventions
 *   - every export returns int: GCI_OK (0) or a negative gci_status; nothing throws, exits,
 *     prints or touches files;
 *   - the caller owns every buffer (torch / hipMalloc / gci_malloc); the library owns only
 *     the scratch inside its gci_ctx;
 *   - pointers named d_* are DEVICE pointers, h_* are HOST pointers;
 *   - all work is enqueued on the ctx stream and is asynchronous unless stated; gci_sync()
 *     waits for it.  A ctx is not thread-safe; distinct ctxs are independent;
 *   - plain C structs with fixed layout, little endian.
 *
 * Track layout ("depth track"): one int32 per base, all selected contigs in ONE buffer.
 * Contig c starts at element gci_layout_offsets()[c], a multiple of GCI_TILE (4096 elements,
 * 16 KiB), so tiles never straddle contigs and every vector access is 16-byte aligned.
 * Elements between the end of a contig and the next multiple of GCI_TILE are padding (always 0).
 */
#ifndef GCI_HIP_H
#define GCI_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GCI_ABI_VERSION 1
#define GCI_TILE 4096          /* elements per tile of a depth track */
#define GCI_MAX_JOIN_FILES 16

typedef enum gci_status {
    GCI_OK = 0,
    GCI_E_INVALID = -1,        /* bad argument */
    GCI_E_HIP = -2,            /* a HIP runtime call failed: see gci_last_error() */
    GCI_E_NO_NM = -3,          /* a record that reaches GCI.py:163 has no NM tag (reference: KeyError) */
    GCI_E_ZERO_DIV = -4,       /* zero denominator at GCI.py:165 / 292 (reference: ZeroDivisionError) */
    GCI_E_BAD_NM_TYPE = -5,    /* NM tag present but not an integer type */
    GCI_E_NO_END = -6,         /* reference_end would be None (n_cigar_op == 0) */
    GCI_E_MALFORMED = -7,      /* record runs past the end of the stream */
    GCI_E_CAPACITY = -8,       /* output buffer too small: grow and call again */
    GCI_E_NOMEM = -9,
    GCI_E_NO_LAYOUT = -10      /* gci_layout_set() has not been called */
} gci_status;

/* Compact alignment record: what read_sam keeps per record (GCI.py:166-168), 32 bytes. */
typedef struct gci_rec {
    uint64_t name_hash;        /* gci_name_hash() of the query name */
    int32_t contig;            /* index among the SELECTED contigs (not the BAM refID), -1 if none */
    int32_t start;             /* reference_start */
    int32_t end;               /* reference_end */
    int32_t qlen;              /* query_length */
    uint32_t rec_idx;          /* rec_idx_base + index of the record in the K1 call (informational) */
    uint8_t mapq;
    uint8_t flags;             /* GCI_REC_PASS | GCI_REC_HQ */
    uint16_t name_len;         /* bytes, without NUL */
} gci_rec;
#define GCI_REC_PASS 1u        /* record reaches GCI.py:166 */
#define GCI_REC_HQ 2u          /* ... and mapq >= mq_cutoff (GCI.py:167) */
#define GCI_REC_NAME16 4u      /* where its query name lies (gci_join_file): at an address = 0 or 4 (mod 16), followed by zero
                                * bytes up to the next 16-byte boundary -- names inside record pages (gci_bam_filter_pages) and
                                * routed name slots (gci_route_records).  The join then compares two such names as whole 16-byte
                                * pieces (three loads per name instead of a dozen) */

/* Interval on a selected contig, 16 bytes. */
typedef struct gci_ivl {
    int32_t contig;
    int32_t start;
    int32_t end;
    int32_t pad;
} gci_ivl;

/* One input of the join: compact records + where the query-name bytes of the record at POSITION i of d_recs
 * live:  d_name_base + d_name_off[i] + name_delta  (BAM: base = the inflated stream, off = record offsets,
 * delta = 36;  packed names / PAF: base = a names blob, off = blob offsets, delta = 0).  Records must be in file
 * order: among records of one file with the same name the LAST position wins (dict semantics, GCI.py:166, 269). */
typedef struct gci_join_file {
    const gci_rec* d_recs;
    uint32_t n_recs;
    uint32_t name_delta;
    const uint8_t* d_name_base;
    const uint64_t* d_name_off;
} gci_join_file;

/* Scan window over a track, in flat element coordinates of the track buffer. */
typedef struct gci_window {
    int64_t begin;
    int64_t end;
} gci_window;

typedef struct gci_ctx gci_ctx;

/* ---- context ------------------------------------------------------------------------------ */
int gci_abi_version(void);
/* own_stream == 0: enqueue on `stream`, a hipStream_t of the caller (e.g. torch's current stream;
 * NULL is the device's default stream).  own_stream != 0: `stream` is ignored and the ctx creates
 * (and later destroys) a non-blocking stream of its own.
 * Environment, read here: GCI_FORCE_DENSE=1 makes the depth build take its dense (difference array in LDS) path for
 * every tile instead of only for tiles with many events -- same results, for testing and A/B timing. */
int gci_ctx_create(int device, void* stream, int own_stream, gci_ctx** out);
int gci_ctx_destroy(gci_ctx* ctx);
int gci_sync(gci_ctx* ctx);
const char* gci_strerror(int status);
const char* gci_last_error(gci_ctx* ctx);
/* Convenience for hosts without a device allocator of their own (cgo, JNI ...). */
int gci_malloc(gci_ctx* ctx, size_t bytes, void** d_out);
int gci_free(gci_ctx* ctx, void* d_ptr);
int gci_memcpy_h2d(gci_ctx* ctx, void* d_dst, const void* h_src, size_t bytes);   /* synchronous */
int gci_memcpy_d2h(gci_ctx* ctx, void* h_dst, const void* d_src, size_t bytes);   /* synchronous */
int gci_memset(gci_ctx* ctx, void* d_dst, int byte, size_t bytes);                /* async */

/* ---- device memory, streams and events for a host without a tensor library (k_hbm.hip; round 6) -------------------------------
 * What the single-GPU command line holds its HBM buffers with (gci_amd/hbm.py: a caching allocator, stream-ordered like the
 * one it replaces, over gci_dev_malloc) -- `import torch` costs half a second of a 4 s run and is needed only for
 * torch.distributed (--gpus N).  Not tied to a context: `device` is the HIP device index, `stream` / `event` are hipStream_t /
 * hipEvent_t handles (a stream made here can be handed to gci_ctx_create with own_stream = 0).  Every call returns a gci_status;
 * gci_dev_last_error() is the calling thread's last HIP message.  gci_dev_malloc returns GCI_E_NOMEM when the device is full
 * (the caller frees what it caches and tries again).
 * gci_dev_memcpy_async: kind 1 = host -> device, 2 = device -> host, 3 = device -> device.
 * The three element-wise helpers are what the host side does to buffers it cannot touch: out[i] = in[i] + delta (name offsets
 * of the runs of a file put behind one another), recs[i].flags &= mask (GCI_REC_NAME16 dropped when names are packed), and the
 * exclusive prefix sum of the .depth.gz member sizes (uint32 in, uint64 out, n + 1 entries; one workgroup). */
const char* gci_dev_last_error(void);
int gci_dev_count(int* n_out);
int gci_dev_malloc(int device, size_t bytes, void** d_out);
int gci_dev_free(int device, void* d_ptr);
/* Device memory of this library -- gci_dev_malloc, gci_malloc and every context's scratch -- comes from an ARENA (k_hbm.hip): small
 * and medium blocks are cut from slabs (256 MiB, then GCI_ARENA_SLAB_MB = 1024 each), a block of 64 MiB or more that nothing free
 * can hold is a driver allocation of its own, and whatever is given back coalesces and serves later requests -- the memory of one
 * phase of a run becomes the next phase's without the driver being asked again.  That matters because a driver allocation costs by
 * the GB on this chip (the kernel driver clears VRAM it does not know to be clean: 35 - 45 ms per GB measured).  Slabs stay until
 * the process ends; GCI_ARENA=0 turns the arena off.
 * gci_dev_reserve: slabs for at least `bytes` in all, now (for a host that wants that cost in a place of its choosing).
 * gci_dev_arena_info: bytes in slabs, bytes handed out, number of slabs. */
int gci_dev_reserve(int device, uint64_t bytes, uint64_t* reserved_out);
int gci_dev_arena_info(int device, uint64_t* reserved, uint64_t* in_use, uint32_t* n_slabs);
int gci_dev_mem_info(int device, uint64_t* free_bytes, uint64_t* total_bytes);
int gci_dev_sync(int device);
int gci_dev_host_alloc(int device, size_t bytes, void** h_out);      /* pinned */
int gci_dev_host_free(int device, void* h_ptr);
int gci_dev_stream_create(int device, void** out);                   /* non-blocking */
int gci_dev_stream_destroy(int device, void* stream);
int gci_dev_stream_sync(int device, void* stream);
int gci_dev_event_create(int device, int timing, void** out);
int gci_dev_event_destroy(int device, void* event);
int gci_dev_event_record(int device, void* event, void* stream);
int gci_dev_event_sync(int device, void* event);
int gci_dev_event_elapsed_ms(int device, void* event_a, void* event_b, double* ms);
int gci_dev_stream_wait_event(int device, void* stream, void* event);
int gci_dev_memcpy_async(int device, void* dst, const void* src, size_t bytes, int kind, void* stream);
int gci_dev_memset_async(int device, void* d_dst, int byte, size_t bytes, void* stream);
int gci_dev_i64_add(int device, const int64_t* d_in, uint64_t n, int64_t delta, int64_t* d_out, void* stream);
int gci_dev_rec_flags_and(int device, gci_rec* d_recs, uint64_t n, uint32_t mask, void* stream);
int gci_dev_u32_scan_u64(int device, const uint32_t* d_in, uint32_t n, uint64_t* d_out, void* stream);

/* ---- optional per-kernel timing with HIP events on the ctx stream ------------------------------
 * gci_profile_enable(mask): bit k enables an event pair around every launch of kernel id k.
 * gci_profile_read(): synchronises, folds finished event pairs in and returns the accumulated
 * milliseconds / launch count of one kernel id (reset != 0 clears that id afterwards). */
enum {
    GCI_PROF_BAM_FILTER = 0, GCI_PROF_JOIN_INSERT, GCI_PROF_JOIN_FOLD, GCI_PROF_DEPTH_DIFF, GCI_PROF_SCAN_TILES,
    GCI_PROF_DEPTH_SCAN, GCI_PROF_GAP_MASK, GCI_PROF_MAX2, GCI_PROF_ISSUE_SCAN, GCI_PROF_TEXT_COUNT,
    GCI_PROF_TEXT_WRITE, GCI_PROF_DEPTH_SUM, GCI_PROF_MEMSET, GCI_PROF_TILE_PASS1, GCI_PROF_TILE_DENSE, GCI_PROF_PARTITION,
    GCI_PROF_JOIN_PART, GCI_PROF_PAGES_SIZE, GCI_PROF_PAGES_WRITE, GCI_PROF_COUNT
};
int gci_profile_enable(gci_ctx* ctx, int mask);
int gci_profile_read(gci_ctx* ctx, int kernel_id, double* total_ms, uint64_t* launches, int reset);
const char* gci_profile_name(int kernel_id);

/* ---- layout (GCI.py:201-208: contig table of the first BAM, optionally restricted by --chrs) */
int gci_layout_set(gci_ctx* ctx, int32_t n_contigs, const int64_t* h_lengths);
int64_t gci_layout_total(gci_ctx* ctx);                    /* elements of one track buffer */
int gci_layout_offsets(gci_ctx* ctx, int64_t* h_offsets);  /* n_contigs entries */

/* ---- R1: record filter ---------------------------------------------------------------------
 * d_bam: inflated BAM stream; d_rec_off[i]: byte offset of record i's block_size word;
 * d_ref_sel[refID]: index among the selected contigs or -1.  Writes one gci_rec per input
 * record (flags == 0 for filtered records); out[i].rec_idx = rec_idx_base + i (a rank that
 * decodes a slice of a file passes the slice's first record index).  *d_status (uint64, device) receives
 * min over failing records of (rec_idx << 8 | -status), or UINT64_MAX if none failed;
 * decode with gci_decode_status(). */
int gci_bam_filter(gci_ctx* ctx, const uint8_t* d_bam, uint64_t n_bytes, const uint64_t* d_rec_off,
                   uint32_t n_rec, const int32_t* d_ref_sel, int32_t n_ref, int map_qual, int mq_cutoff,
                   double clip_percent, double iden_percent, uint32_t rec_idx_base, gci_rec* d_out,
                   uint64_t* d_status);
/* ---- R1 over RECORD PAGES (round 3): the layout the record filter is fastest on --------------------------------
 * read_sam (GCI.py:146-169) looks at ~400 of a HiFi record's 27 000 bytes.  gci_bam_pages_size / _write copy exactly
 * those bytes -- once, on the device, as the last step of the ingestion that walks the inflated stream anyway -- into
 * fixed-size pages in which every record and every CIGAR is 16-byte aligned and nothing needs an offset table:
 *
 *   buffer = [page 0] ... [page n_pages - 1] [blob] [16 zero bytes]
 *   page   = u32 n_recs | u32 first_rec | u32 used_bytes | u32 magic "GCP1" | u16 dir[n_recs] (record start / 16) | records
 *   record = the BAM record without SEQ / QUAL, its fixed 36 bytes kept except: block_size -> size of the record in the
 *            page (a multiple of 16, <= GCI_PAGE_MAX_REC); bin -> kind; next_refID -> aux_len; next_pos, tlen -> blob
 *            offset (u64, from the buffer start).  read_name at +36, zero padded so that the CIGAR starts at
 *            align16(36 + l_read_name); the CIGAR zero padded to 16 bytes; the aux bytes behind it.
 *            kind 1: the CIGAR words are in the blob (a record that does not fit: ONT), 16 bytes with the first
 *            operation stand in for them; kind 2: core only, the record's bytes (heads form) are in the blob (CG:B,I
 *            long-CIGAR records, kilobytes of tags); kind 4: a record the stream filter reports GCI_E_MALFORMED for.
 *   Record i goes to page floor(S_i / (page_bytes - GCI_PAGE_MAX_REC - 48)), S = exclusive scan of (size + 2).
 *
 * gci_bam_pages_size: d_stream / d_rec_off as for gci_bam_filter (has_seq != 0), or a HEADS STREAM (has_seq == 0: gci_bam_heads
 *   below -- the BAM header followed by every record without its SEQ and QUAL bytes, block_size shortened accordingly, l_seq
 *   unchanged: the bytes read_sam never looks at, GCI.py:146-169, are 98 % of a HiFi record);
 *   h_out[0] = n_pages, h_out[1] = bytes of the buffer to allocate, h_out[2] = offset of the blob.  Synchronises.
 * gci_bam_pages_write: fills d_out (cap >= h_out[1]) for the input the size call measured.
 * gci_bam_filter_pages: the filter over such a buffer -- same records, same status word as gci_bam_filter over the
 *   stream the pages were made from; d_name_off (nullable, n_rec entries) receives where every record's query name lies
 *   in the buffer (gci_join_file: d_name_base = d_pages, d_name_off, name_delta = 0). */
#define GCI_PAGE_MAX_REC 1024
#define GCI_PAGE_MAX_BYTES 32768
#define GCI_PAGE_BYTES_DEFAULT 24576
int gci_bam_pages_size(gci_ctx* ctx, const uint8_t* d_stream, uint64_t n_bytes, const uint64_t* d_rec_off, uint32_t n_rec,
                       int has_seq, uint32_t page_bytes, uint64_t* h_out);
int gci_bam_pages_write(gci_ctx* ctx, const uint8_t* d_stream, uint64_t n_bytes, const uint64_t* d_rec_off, uint32_t n_rec,
                        int has_seq, uint8_t* d_out, uint64_t cap);
int gci_bam_filter_pages(gci_ctx* ctx, const uint8_t* d_pages, uint64_t total_bytes, uint32_t page_bytes, uint32_t n_pages,
                         uint32_t n_rec, const int32_t* d_ref_sel, int32_t n_ref, int map_qual, int mq_cutoff,
                         double clip_percent, double iden_percent, uint32_t rec_idx_base, gci_rec* d_out,
                         uint64_t* d_name_off, uint64_t* d_status);
int gci_decode_status(uint64_t status_word, uint32_t* rec_idx);   /* -> gci_status */
/* The name hash K1 uses, for hosts that build gci_rec themselves (PAF path). */
uint64_t gci_name_hash(const uint8_t* h_name, uint32_t len);
/* Pack the query names of `n` records into a dense blob (multi-GPU exchange):
 * d_out_off[i] = byte offset of name i (exclusive scan of name_len), d_out_off[n] = total. */
int gci_pack_names(gci_ctx* ctx, const gci_join_file* h_file, uint8_t* d_out_names, uint64_t cap,
                   uint64_t* d_out_off);

/* ---- R5: cross-file join --------------------------------------------------------------------
 * h_files in reference order (PAF files, then BAM files, each in command-line order).
 * One file: the dict semantics of GCI.py:166/269 only (a repeated name keeps its LAST record).
 * d_contig_map (nullable): remaps rec.contig -> output contig, entries < 0 drop the interval
 * (multi-GPU: keep only the contigs this rank owns).  Output order is unspecified.
 * *d_n_out may exceed cap: then nothing beyond cap was written, grow and retry. */
int gci_name_join(gci_ctx* ctx, const gci_join_file* h_files, int n_files, double ovlp_percent,
                  const int32_t* d_contig_map, gci_ivl* d_out, uint32_t cap, uint32_t* d_n_out,
                  uint64_t* d_status);
/* Which implementation gci_name_join / gci_name_join_count use: 0 = by size (an open-addressing table in HBM below 2^20
 * records, the radix-partitioned join with per-bucket tables in LDS from there up), 1 = always the table, 2 = always
 * partitioned.  Same results either way.  The partitioned join reports GCI_E_CAPACITY (record 0) in *d_status for inputs its
 * 32-byte entries cannot hold (a bucket with more distinct names than its table has slots -- forged hashes only --, a name of
 * 4 KiB and more, name bytes beyond 64 GiB from d_name_base): the caller then sets mode 1 and joins again. */
int gci_join_mode(gci_ctx* ctx, int mode);
/* The same join, fused with the first pass of the depth build: the kernel that emits an interval also counts it into
 * the per-tile tables of the layout (flank = the build's --flank-len), so a gci_depth_build_begin over exactly this
 * output with opts.counted = 1 skips that pass (one dependent launch and one read of the intervals less).  Needs
 * gci_layout_set; if *d_n_out exceeds cap, call it again with more room before building.  From 2^20 records up the
 * build buckets its events by radix partition and counts on the way (GCI_EVENTS=atomic|radix overrides): the call is then
 * the plain join, and opts.counted = 1 still says "these are the intervals of that join". */
int gci_name_join_count(gci_ctx* ctx, const gci_join_file* h_files, int n_files, double ovlp_percent,
                        const int32_t* d_contig_map, gci_ivl* d_out, uint32_t cap, uint32_t* d_n_out,
                        uint64_t* d_status, int flank);

/* Cross-rank name check for contig-sharded runs (exact).  gci_hash_bucket sorts the 64-bit name hashes of the
 * passing records into n_parts buckets by (hash >> 33) % n_parts; a bucket is part_cap + 1 uint64 words:
 * [0] = number of hashes the sender had for it (> part_cap: overflow), [1..] = the hashes.  After ONE all-to-all of
 * the buckets, gci_hash_conflicts ADDS to *d_n_conflicts the number of hashes that arrived from more than one
 * source bucket (+1 per overflowing bucket).  Zero on every rank => no query name is shared between ranks, so
 * each rank's local join equals the global one.
 * d_next_out (may be NULL): a second bucket array the caller alternates with d_out.  When given, the call zeroes ITS
 * count words for the next call and relies on d_out's count words being zero already (zero-filled at allocation, then
 * kept so by this rule) -- no clearing launch per call.  NULL: d_out's count words are cleared by the call itself. */
int gci_hash_bucket(gci_ctx* ctx, const gci_rec* d_recs, uint32_t n, uint32_t n_parts, uint32_t part_cap,
                    uint64_t* d_out, uint64_t* d_next_out);
int gci_hash_conflicts(gci_ctx* ctx, const uint64_t* d_buckets, uint32_t n_parts, uint32_t part_cap,
                       uint32_t* d_n_conflicts);

/* ---- multi-GPU: the name-hash-sharded join ---------------------------------------------------------
 * Contigs are sharded over the ranks, the join (GCI.py:272-301) is by read name and independent per name: every passing
 * record goes to rank (name_hash >> 33) % n_parts, each rank joins the names it owns (gci_name_join over what arrived),
 * the surviving intervals go to the rank that owns their contig.  These calls fill / finish the fixed-shape buckets of the
 * three all-to-alls (the host side does the collectives: RCCL through torch.distributed, gci_amd/shard.py):
 *   gci_route_records   one file's records -> n_parts buckets of (cap + 1) gci_rec slots (slot 0: header, name_hash = number
 *                       of records routed to the part, flags = 0; then the records in FILE ORDER -- the routing is stable)
 *                       and n_parts * cap name slots of name_slot bytes (a multiple of 16, at least the longest name; zero padded)
 *   gci_route_seal_records   on the receiving side: slots beyond a bucket's count get flags = 0, so the whole receive buffer
 *                       is one gci_rec array in file order (source rank after source rank) for gci_name_join; the name of
 *                       slot d * (cap + 1) + 1 + k is at (d * cap + k) * name_slot of the received names
 *   gci_route_intervals intervals (contig = index among ALL selected contigs) -> buckets by d_owner[contig] (< 0: dropped);
 *                       slot 0 of a bucket = {contig = -1, start = count}
 *   gci_route_seal_intervals  receiving side: contig -> d_cmap[contig] (the rank's track layout), -1 beyond the count: the
 *                       buffer of n_parts * (cap + 1) intervals goes to gci_depth_build_* as it is
 * A count beyond cap or a name longer than name_slot sets *d_status to GCI_E_CAPACITY (grow the buckets). */
int gci_route_records(gci_ctx* ctx, const gci_join_file* h_file, uint32_t n_parts, uint32_t cap, gci_rec* d_out_recs,
                      uint8_t* d_out_names, uint32_t name_slot, uint64_t* d_status);
int gci_route_seal_records(gci_ctx* ctx, gci_rec* d_recs, uint32_t n_parts, uint32_t cap, uint64_t* d_status);
int gci_route_intervals(gci_ctx* ctx, const gci_ivl* d_ivl, const uint32_t* d_n, uint32_t max_n, const int32_t* d_owner,
                        int32_t n_contigs, uint32_t n_parts, uint32_t cap, gci_ivl* d_out, uint64_t* d_status);
int gci_route_seal_intervals(gci_ctx* ctx, gci_ivl* d_ivl, uint32_t n_parts, uint32_t cap, const int32_t* d_cmap,
                             int32_t n_contigs, uint64_t* d_status);

/* ---- R6: depth build ------------------------------------------------------------------------
 * depth[c][start+flank : end-flank+1] += 1 with Python/NumPy slice semantics, for n intervals
 * (n read from *d_n if d_n != NULL, clamped to max_n).  Overwrites the whole track. */
int gci_depth_build(gci_ctx* ctx, const gci_ivl* d_ivl, const uint32_t* d_n, uint32_t max_n, int flank,
                    int32_t* d_depth);

/* Fused form of the depth build (same result as gci_depth_build) that also delivers, from the
 * same pass and without re-reading the track from HBM, the things the reference computes from the
 * freshly built depths: the decimal text of write_depth (GCI.py:115-117), the per-contig sums behind
 * np.mean (GCI.py:862-868) and -- valid only when no gap mask will follow -- the run boundaries of
 * collapse_depth_range (GCI.py:369-390; same key format as gci_issue_scan).
 *   begin : buckets the intervals per tile, computes the requested by-products.  After it (and a
 *           sync) d_contig_text_off[n_contigs] tells the caller how large the text buffer must be.
 *   finish: writes the track and, if d_text != NULL, the text (requires want_text at begin). */
typedef struct gci_build_opts {
    int flank;                    /* --flank-len of the depth build */
    int want_text;                /* compute text offsets in begin so that finish can render text */
    uint64_t* d_contig_text_off;  /* n_contigs + 1 entries; required if want_text */
    int64_t* d_sums;              /* n_contigs entries or NULL */
    uint32_t* d_n_keys;           /* NULL: no fused issue scan */
    uint64_t* d_keys;
    uint32_t key_cap;
    int issue_flank;
    double lo, hi;
    int counted;                  /* 1: the intervals are exactly what the last gci_name_join_count emitted (same flank):
                                     the per-tile counting pass has been done there */
    int want_runs;                /* 1: finish also keeps, in the context, the constant-depth runs of every 4096-base tile as it
                                     wrote them; a gci_depth_deflate_size / _write pair over the SAME d_depth, announced by
                                     gci_depth_deflate_from_build(), then takes the runs from there instead of reading
                                     the track (members that start on a tile boundary: every contig does), once.  gci_gap_mask / gci_max2 / gci_two_type_tail, a new build and
                                     gci_layout_set drop the lists; a caller that writes the track by other means
                                     between the build and the deflate calls must not set this.  (Occupies what was
                                     padding: sizeof(gci_build_opts) is unchanged.) */
} gci_build_opts;
int gci_depth_build_begin(gci_ctx* ctx, const gci_ivl* d_ivl, const uint32_t* d_n, uint32_t max_n,
                          const gci_build_opts* h_opts);
int gci_depth_build_finish(gci_ctx* ctx, int32_t* d_depth, uint8_t* d_text, uint64_t text_cap);

/* ---- R8 / R9 -------------------------------------------------------------------------------- */
int gci_gap_mask(gci_ctx* ctx, int32_t* d_depth, const gci_ivl* d_gaps, uint32_t n_gaps);
int gci_max2(gci_ctx* ctx, const int32_t* d_a, const int32_t* d_b, int32_t* d_out);

/* The tail of a two-read-type run in one pass over the two tracks (GCI.py:1014-1024): N-run masks of both (GCI.py:324-328; h_gaps
 * in contig coordinates with Python's slice rules, HOST memory), their per-base maximum (GCI.py:350) into d_out, and the
 * issue-run boundary keys of all three tracks (GCI.py:369-390): d_keys = 3 x cap keys, d_n_keys = 3 counters -- per track the
 * keys gci_issue_scan(track, lo, hi, flank) gives.  d_a / d_b are masked in place.  d_sums (nullable): 3 x n_contigs sums of depth
 * (the masked a, the masked b, the maximum: what gci_depth_sum gives for each -- the numerator of np.mean, GCI.py:862-868).
 * 12 bytes per base instead of the 24 of gci_gap_mask x 2 + gci_max2 + gci_issue_scan x 3 (+ 12 for three gci_depth_sum). */
int gci_two_type_tail(gci_ctx* ctx, int32_t* d_a, int32_t* d_b, int32_t* d_out, const gci_ivl* h_gaps, uint32_t n_gaps,
                      double lo, double hi, int flank, uint64_t* d_keys, uint32_t cap, uint32_t* d_n_keys, int64_t* d_sums);

/* ---- R10: issue scan ------------------------------------------------------------------------
 * Finds every maximal run of `lo < depth <= hi` inside each window.  Emits unordered 64-bit
 * keys  (window << 33) | (rel << 1) | is_end,  rel = run start - window.begin for a start key,
 * run end (exclusive) - window.begin for an end key; sorted ascending they read
 * start,end,start,end... per window.  The reference's `i > flank_len` drop rule and
 * coordinate shifts are applied by the host.  gci_issue_scan uses one window per contig:
 * [flank, L - flank). */
int gci_issue_scan(gci_ctx* ctx, const int32_t* d_depth, double lo, double hi, int flank,
                   uint64_t* d_keys, uint32_t cap, uint32_t* d_n_keys);
int gci_issue_scan_windows(gci_ctx* ctx, const int32_t* d_depth, const gci_window* h_windows,
                           uint32_t n_windows, double lo, double hi, uint64_t* d_keys, uint32_t cap,
                           uint32_t* d_n_keys);

/* ---- R7: depth text -------------------------------------------------------------------------
 * size: d_contig_off[c] = byte offset of contig c's lines in the text, d_contig_off[n_contigs] =
 * total bytes (no '>' lines: the host writes those).  write: fills d_out (cap >= total). */
int gci_depth_text_size(gci_ctx* ctx, const int32_t* d_depth, uint64_t* d_contig_off);
int gci_depth_text_write(gci_ctx* ctx, const int32_t* d_depth, uint8_t* d_out, uint64_t cap);

/* ---- R7 / N2: the depth text as gzip members, written by the GPU ------------------------------------------------
 * write_depth (GCI.py:99-143) sends f'{depth}\n' per base through gzip; these two calls emit the same payload as
 * DEFLATE without ever materialising the text: per constant-depth run the line's literals + length/distance pairs
 * (distance = line length) in the fixed Huffman code, one block per 4096-base tile closed by an empty stored block,
 * 64 tiles (one wave) = one gzip member with its CRC-32 and ISIZE (CRC of the virtual text by GF(2) polynomial
 * arithmetic over the runs).  Any int32 track (fresh, gap-masked, two-type) with depths >= 0.
 * A member m covers d_member_n[m] <= 64 * 4096 consecutive bases starting at element d_member_elem[m] (a multiple
 * of 4; the caller cuts members so that none spans two contigs and writes the '>contig' lines as members of its own).
 *   size:  d_tile_bytes[64 m + t], d_member_bytes[m] (header and trailer included), d_member_crc[m], d_member_isize[m]
 *   write: d_member_out[m] = byte offset of member m in d_out (exclusive scan of d_member_bytes by the caller). */
/* The run lists a build kept (gci_build_opts.want_runs) are used by the NEXT size / write pair only when the caller says so, here:
 * "d_depth is the track that build wrote, and nothing -- no call of this library through any context, no other code -- has
 * written it since".  The library cannot see writes that go around this context (another context's gci_gap_mask, the caller's
 * own kernels, an allocator handing the address to a new buffer), so a pointer match alone is never taken as that statement.
 * Returns GCI_E_INVALID when this context holds no lists for d_depth; the pair that follows consumes the lists (one shot). */
int gci_depth_deflate_from_build(gci_ctx* ctx, const int32_t* d_depth);
int gci_depth_deflate_size(gci_ctx* ctx, const int32_t* d_depth, const uint64_t* d_member_elem, const uint32_t* d_member_n,
                           uint32_t n_members, uint32_t* d_tile_bytes, uint32_t* d_member_bytes, uint32_t* d_member_crc,
                           uint32_t* d_member_isize);
int gci_depth_deflate_write(gci_ctx* ctx, const int32_t* d_depth, const uint64_t* d_member_elem, const uint32_t* d_member_n,
                            uint32_t n_members, const uint32_t* d_tile_bytes, const uint32_t* d_member_crc,
                            const uint32_t* d_member_isize, const uint64_t* d_member_out, uint8_t* d_out, uint64_t cap);

/* ---- R15 ------------------------------------------------------------------------------------ */
int gci_depth_sum(gci_ctx* ctx, const int32_t* d_depth, int64_t* d_sums /* n_contigs */);

/* ---- N3: window sums for the `-p` numeric front-end (sliding_window_average_depth, GCI.py:660-705) ----------
 * d_ranges holds n_ranges pairs (begin, end) of TRACK element indices (contig offset + position); d_sums[r] = sum of
 * the depths in [begin, end).  The caller derives the ranges from the zero-depth runs (gci_issue_scan_windows with
 * lo = -1, hi = 0) and the window size: the reference restarts its window at every zero-depth base. */
int gci_range_sums(gci_ctx* ctx, const int32_t* d_depth, const int64_t* d_ranges, uint64_t n_ranges, int64_t* d_sums);

/* ---- N4 (first half): N runs of the assembly, get_Ns_ref (GCI.py:27-35) ------------------------------------------
 * d_text: the FASTA file's bytes.  d_body: n_records pairs (begin, end) of byte offsets, the body of every record
 * (behind its title line, up to the next '>' line), sorted.  Output: d_tile_kept[k] = bytes of the 4096-byte tile k
 * that count as sequence (not a line end, '\r' or blank, not in a title line), and keys = (byte offset << 1) | e for
 * every base where a run of N / n begins (e = 0) or that follows the last base of a run (e = 1); *d_n_keys may exceed
 * cap (nothing is written beyond it: call again with more room).  A run that reaches the end of a record has no end
 * key.  The host turns byte offsets into sequence coordinates with the prefix sum of d_tile_kept. */
int gci_fasta_n_scan(gci_ctx* ctx, const uint8_t* d_text, uint64_t n_bytes, const int64_t* d_body, uint32_t n_records,
                     uint32_t* d_tile_kept, uint64_t* d_keys, uint32_t cap, uint32_t* d_n_keys);

/* ---- host-side container helpers (no GPU work) ---------------------------------
 * The reference reaches BGZF / BAM through pysam/htslib (GCI.py:150-151) and writes gzip through Python's gzip
 * module (GCI.py:111).  All pointers are HOST pointers.
 *   gci_bgzf_scan            member count and total inflated size (sum of ISIZE) of a BGZF byte string
 *   gci_bgzf_inflate         inflate every member into h_out, members in parallel on `threads` host threads
 *   gci_bam_record_offsets   end of the BAM header and the byte offset of every record (the one serial step of
 *                            the decode); h_offs may be NULL to count only
 *   gci_gzip_members         gzip-frame text as independent members of `chunk` input bytes, compressed in
 *                            parallel (any multi-member gzip whose payload equals the text is a valid .depth.gz)
 *   gci_bgzf_blocks / gci_bam_chunk_offsets   the same for a host that streams a large file chunk by chunk: the
 *                            member table, and the record offsets of one chunk with the partial tail reported
 *   gci_bam_heads            BGZF file bytes -> heads stream + record offsets in one pipelined pass (replaces
 *                            pysam's AlignmentFile + fetch, GCI.py:150-151, for a host that feeds
 *                            gci_bam_pages_* with has_seq = 0): groups of `group_bytes` (0 = 16 MiB) of inflated members rotate
 *                            through three buffers -- worker threads inflate group g while the caller's thread walks
 *                            the block_size chain of group g-1 and the workers copy the heads of group g-2 -- so the
 *                            inflated stream (27 KB per HiFi record) never exists as a whole and only ~400 B per
 *                            record cross PCIe.  A record whose l_seq / name / CIGAR lengths contradict its
 *                            block_size is emitted as its 36 fixed bytes with l_seq = -1 (the filter reports
 *                            GCI_E_MALFORMED with its index, as on the full stream); block_size < 32, a bad
 *                            member or a truncated last record return GCI_E_MALFORMED here.
 *                            gci_bam_heads_stream / _offsets stay valid until gci_bam_heads_free. */
/*   gci_fasta_titles         byte offsets of every '>' that begins a line (one per record SeqIO.parse yields,
 *                            GCI.py:30 / :940), in file order; h_pos may be NULL with cap 0 to count only */
int gci_fasta_titles(const uint8_t* h_text, uint64_t n, int threads, uint64_t* h_pos, uint64_t cap, uint64_t* n_pos);
typedef struct gci_heads gci_heads;
int gci_bam_heads(const uint8_t* h_raw, uint64_t n_raw, int threads, uint64_t group_bytes, int check_crc, gci_heads** out);
uint64_t gci_bam_heads_bytes(const gci_heads* h);          /* length of the heads stream */
uint64_t gci_bam_heads_count(const gci_heads* h);          /* records */
uint64_t gci_bam_heads_first(const gci_heads* h);          /* length of the BAM header = offset of record 0 */
const uint8_t* gci_bam_heads_stream(const gci_heads* h);
const uint64_t* gci_bam_heads_offsets(const gci_heads* h); /* offset of every record's block_size word */
int gci_bam_heads_free(gci_heads* h);
int gci_bgzf_scan(const uint8_t* h_raw, uint64_t n_raw, uint64_t* n_blocks, uint64_t* inflated_bytes);
int gci_bgzf_blocks(const uint8_t* h_raw, uint64_t n_raw, uint64_t* h_pos, uint64_t* h_isize, uint64_t cap,
                    uint64_t* n_blocks);
/* The member table in ONE pass by `threads` host threads (the chain of BSIZE fields is walked range by range and stitched: a
 * page fault per member is what the walk costs, 0.65 s for 64.5 GB on one thread): a handle; _export fills count + 1 offsets
 * (the last = n_raw) and count ISIZEs. */
typedef struct gci_bgzf_table gci_bgzf_table;
int gci_bgzf_table_build(const uint8_t* h_raw, uint64_t n_raw, int threads, gci_bgzf_table** out);
/* Only the members that start in front of byte `limit` (the table's byte length = the end of the last of them): the first run of
 * a large file can be on its way to the device while the table of the rest is made. */
int gci_bgzf_table_build_prefix(const uint8_t* h_raw, uint64_t n_raw, uint64_t limit, int threads, gci_bgzf_table** out);
/* The same table read with pread() through a descriptor of the file (n_raw = its size) instead of a mapping of it: one system
 * call per member and no page-table entry -- on a freshly mapped 77 GB file the page faults of the walk are 1.4 s on 16 threads,
 * in the address space the threads that stage the file's bytes fault in as well. */
int gci_bgzf_table_build_fd(int fd, uint64_t n_raw, int threads, gci_bgzf_table** out);
uint64_t gci_bgzf_table_count(const gci_bgzf_table* t);
int gci_bgzf_table_export(const gci_bgzf_table* t, uint64_t* h_pos, uint64_t* h_isize);
int gci_bgzf_table_free(gci_bgzf_table* t);
int gci_bam_chunk_offsets(const uint8_t* h_buf, uint64_t n, uint64_t start, uint64_t* h_offs, uint64_t cap,
                          uint64_t* n_rec, uint64_t* consumed);
int gci_bgzf_inflate(const uint8_t* h_raw, uint64_t n_raw, uint8_t* h_out, uint64_t cap, int threads, int check_crc);
int gci_bam_record_offsets(const uint8_t* h_stream, uint64_t n, uint64_t* h_offs, uint64_t cap, uint64_t* n_rec,
                           uint64_t* first_record);
uint64_t gci_gzip_bound(uint64_t n, uint64_t chunk);
int gci_gzip_members(const uint8_t* h_text, uint64_t n, uint64_t chunk, int level, int threads, uint8_t* h_out,
                     uint64_t cap, uint64_t* n_out);

/* ---- N4 (second half, host): the PAF filter of filter(), GCI.py:211-254 --------------------------------------------
 * h_files[i] / n_bytes[i]: the bytes of PAF file i, in command-line order.  targets: the selected contigs (index =
 * gci_rec.contig).  On success *out holds, per file, one compact record + name for every query seen so far in first-
 * appearance order (the reference's block table is created once, outside its per-file loop); GCI_REC_HQ marks the
 * names of the reference's high_qual set as it stands after the last file.  GCI_E_MALFORMED / GCI_E_ZERO_DIV: a line
 * the reference would raise IndexError / ValueError / ZeroDivisionError on; *err_line = its 1-based number.
 * Lines are tokenised and filtered on `threads` host threads (byte ranges cut at line starts), queries are numbered in
 * first-appearance order by one pass over the surviving lines, and the per-query arithmetic runs in parallel again. */
typedef struct gci_paf gci_paf;
int gci_paf_filter(const uint8_t* const* h_files, const uint64_t* n_bytes, int n_files, const char* const* targets,
                   int n_targets, int map_qual, int mq_cutoff, double iden_percent, int threads /* 0 = automatic */,
                   gci_paf** out, uint64_t* err_line);
uint64_t gci_paf_count(const gci_paf* r, int file);
uint64_t gci_paf_name_bytes(const gci_paf* r, int file);
int gci_paf_export(const gci_paf* r, int file, gci_rec* h_recs, uint8_t* h_names, uint64_t* h_name_off /* count + 1 */);
int gci_paf_free(gci_paf* r);

/* ---- R3 / N4: the same PAF filter on the GPU (k_paf.hip) ----------------------------------------------------------------
 * d_text: the bytes of all PAF files back to back in ONE device buffer, h_file_end[i] = end offset of file i (host array).
 * Line starts, the tokeniser (str.strip + split, int() of the eight numeric columns, target lookup, identity and the
 * mapq / identity filter), the grouping of the surviving lines by query and the per-query scoring (GCI.py:241-254) all run
 * on the device; results and errors are those of gci_paf_filter (*err_line = 1-based line of the offending file, 0 for an
 * error raised while scoring).  The call synchronises.  Per file the handle holds one compact record per query seen so
 * far (order unspecified) and the byte offset of its name INSIDE d_text: join with d_name_base = d_text, name_delta = 0.
 * gci_paf_dev_export copies them into the caller's device buffers (gci_paf_dev_count entries each), on the ctx stream. */
typedef struct gci_paf_dev gci_paf_dev;
int gci_paf_filter_device(gci_ctx* ctx, const uint8_t* d_text, const uint64_t* h_file_end, int n_files, const char* const* targets,
                          int n_targets, int map_qual, int mq_cutoff, double iden_percent, gci_paf_dev** out, uint64_t* err_line);
uint64_t gci_paf_dev_count(const gci_paf_dev* r, int file);
int gci_paf_dev_export(const gci_paf_dev* r, int file, gci_rec* d_recs, uint64_t* d_name_off);
int gci_paf_dev_free(gci_paf_dev* r);

/* The same filter in two halves, for runs that shard a PAF file over the GPUs of a node by BYTE RANGE (each rank tokenises the
 * lines of its range; a query's lines may lie in several ranges, and GCI.py:241-254 scores a query over all of them -- in
 * file order, over all files so far -- so the hits travel to the rank that owns the query name before they are scored):
 *   gci_paf_hits_device   stage A: the lines of this rank's ranges (d_text: the ranges of the files back to back, cut at line
 *                         starts) that pass GCI.py:218-239, as gci_paf_hit in line order; errors as gci_paf_filter_device,
 *                         *err_line counted from the start of the range (sharded callers fall back to the whole files on
 *                         any error, where the reference's exception comes out exactly)
 *   gci_route_hits        hits -> n_parts buckets of (cap + 1) slots by (qhash >> 33) % n_parts, stable (slot 0: header,
 *                         qhash = count), query names into slots of name_slot bytes -- as gci_route_records
 *   gci_paf_score_device  stage B: the hits of the queries this rank owns, file after file (h_hits_upto), every file in line
 *                         order (source rank after source rank: the ranges ascend with the rank), qn_off relative to d_names
 *                         -> the handle of gci_paf_filter_device */
typedef struct gci_paf_hit {
    uint64_t qn_off;          /* where the query name lies (stage A: in d_text; stage B: in d_names) */
    uint64_t qhash;           /* gci_name_hash of the query name */
    int64_t qlen, qs, qe, ts, te;
    double identity;          /* nmatch / alnlen, IEEE f64 */
    uint32_t qn_len;
    int32_t t;                /* target: index among the selected contigs */
    uint32_t hq;              /* mapq >= mq_cutoff */
    uint32_t slot;            /* scratch of stage B */
} gci_paf_hit;
#define GCI_PAF_HIT_BYTES 80
typedef struct gci_paf_hits gci_paf_hits;
int gci_paf_hits_device(gci_ctx* ctx, const uint8_t* d_text, const uint64_t* h_file_end, int n_files, const char* const* targets,
                        int n_targets, int map_qual, int mq_cutoff, double iden_percent, gci_paf_hits** out, uint64_t* err_line);
uint64_t gci_paf_hits_count(const gci_paf_hits* r, int file);
int gci_paf_hits_export(const gci_paf_hits* r, int file, uint8_t* d_hits);
int gci_paf_hits_free(gci_paf_hits* r);
int gci_route_hits(gci_ctx* ctx, const uint8_t* d_hits, uint32_t n, const uint8_t* d_name_base, uint32_t n_parts, uint32_t cap,
                   uint8_t* d_out_hits, uint8_t* d_out_names, uint32_t name_slot, uint64_t* d_status);
int gci_paf_score_device(gci_ctx* ctx, const uint8_t* d_names, uint8_t* d_hits, const uint32_t* h_hits_upto, int n_files,
                         const char* const* targets, int n_targets, gci_paf_dev** out);
/* The PAF filter's pooled scratch (kept in the context between calls, up to 16 GB) given back to the driver: for a host whose other
 * stages allocate through an allocator of their own.  Synchronises when there is something to free. */
int gci_paf_pool_release(gci_ctx* ctx);

/* ---- the way of a memory-mapped input file to the device (staging.cpp) ---------------------------------------------------------
 * A ring of n_slots pinned host buffers of slot_bytes each, filled by `threads` host threads (parallel memcpy out of the page
 * cache) and emptied by DMA on the caller's stream.  gci_stage_send: h_src[0, n) -> d_dst[0, n) on `stream` (a hipStream_t);
 * returns when the last piece is enqueued -- its bytes are in a pinned slot by then, the caller may unmap the file.  forget != 0
 * drops the pages of h_src from the process's page table as they are read (madvise DONTNEED; the page cache keeps the data);
 * urgent == 0 lets urgent calls of other threads go first, piece by piece.  Calls from several threads take slots in turns. */
typedef struct gci_stage gci_stage;
int gci_stage_create(gci_ctx* ctx, uint64_t slot_bytes, int n_slots, int threads, gci_stage** out);
int gci_stage_send(gci_ctx* ctx, gci_stage* stage, const uint8_t* h_src, uint64_t n, uint8_t* d_dst, void* stream, int forget, int urgent);
/* ... from a file descriptor: bytes [offset, offset + n) by pread() straight into the pinned slots (no mapping, no page faults) */
int gci_stage_send_fd(gci_ctx* ctx, gci_stage* stage, int fd, uint64_t offset, uint64_t n, uint8_t* d_dst, void* stream, int urgent);
int gci_stage_free(gci_stage* stage);

/* ---- N1 on the GPU: BGZF inflate and the BAM record walk (k_inflate.hip; replaces pysam / htslib at GCI.py:150-151) ---------
 * gci_bgzf_inflate_device: d_raw = the bytes of a BGZF file (or of a run of its members) on the device; d_member_pos[m] =
 * offset of member m in d_raw, n_members + 1 entries (the last = end of the run; gci_bgzf_blocks makes the table on the
 * host); d_out_off[m] = where member m's output goes in d_out (exclusive scan of the members' ISIZE, n_members + 1 entries).
 * A WAVE per member (k_inflate_wave.hip: the block's body cut into 64 pieces decoded side by side from guessed bit offsets and
 * stitched where neighbouring decoders fall into step, the symbols as a stream in scratch memory of the context, the copies
 * resolved by pointer jumping over the member's output in LDS); the members that do not stitch -- a few in a thousand -- and,
 * with GCI_INFLATE=lane in the environment, all of them by one LANE per member (k_inflate.hip: decode tables in LDS, the output
 * buffer is the window).  Every member's length and -- check_crc != 0, by a further kernel, one wave per member -- CRC-32 are
 * verified.  d_raw must be readable for 8 bytes past its last member (the decoders fetch whole aligned words).
 * *d_status: min over failing members of (member << 8 | -status), UINT64_MAX if none (decode with gci_decode_status).
 * gci_bgzf_inflate_streams(ctx, 2): every other batch of members runs on a second stream of the context's own.  For a host
 * whose HIP runtime has hardware queues to spare -- GPU_MAX_HW_QUEUES >= 8 in the environment BEFORE the runtime started; the
 * default four are shared by all streams of a process, and a copy stream that shares one with the inflate waits for it (uploads
 * at 26 instead of 58 GB/s) -- and only the host knows whether the variable was there in time: the default is 1.
 * GCI_INFLATE_STREAMS=1|2 in the environment overrides (measurements).
 * gci_bgzf_inflate_last_stats (synchronises): how the members of the context's LAST gci_bgzf_inflate_device call fared with the
 * wave decoder -- h_counts[0] decoded, [1] header not taken, [2] no meeting point, [3] end-of-block codes on wrong paths,
 * [4] undecodable / copies, [5] length, [6] lanes, [7] not tried (GCI_INFLATE=lane); [1 .. 6] went to the lane decoder.
 *
 * gci_bam_record_offsets_device: the byte offset of every record of an inflated BAM stream on the device, without the serial
 * block_size chain: candidate record starts by a strict format test on every byte position, successor lookup, reachability
 * from `first_record` (= end of the BAM header) by pointer doubling.  d_result (device, 3 x uint64): [0] records found (may
 * exceed cap: nothing is written beyond it), [1] bytes consumed (n_bytes, or the offset of a partial last record when the
 * stream is a chunk), [2] != 0: the chain broke at offset [1] (a record the strict test rejects: take the host walk).
 * The call synchronises. */
int gci_bgzf_inflate_device(gci_ctx* ctx, const uint8_t* d_raw, const uint64_t* d_member_pos, const uint64_t* d_out_off,
                            uint32_t n_members, uint8_t* d_out, uint64_t out_cap, int check_crc, uint64_t* d_status);
int gci_bgzf_inflate_streams(gci_ctx* ctx, int n);
int gci_bgzf_inflate_last_stats(gci_ctx* ctx, uint32_t h_counts[32]);
/* Members the device decodes at a time (a launch takes a whole number of such rounds: size runs of a large file accordingly); 0 = unknown. */
uint32_t gci_bgzf_inflate_round(gci_ctx* ctx);
int gci_bam_record_offsets_device(gci_ctx* ctx, const uint8_t* d_stream, uint64_t n_bytes, uint64_t first_record, int32_t n_ref,
                                  uint64_t* d_offs, uint64_t cap, uint64_t* d_result);

#ifdef __cplusplus
}
#endif
#endif /* GCI_HIP_H */
