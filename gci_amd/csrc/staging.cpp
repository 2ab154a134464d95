// staging.cpp -- the bytes of a memory-mapped input file on their way to the device: a ring of pinned host buffers filled by
// host threads (parallel memcpy out of the page cache: the page faults are theirs, not the copy engine's), every slot leaving by
// DMA on the caller's stream and taken up again once that copy's event has passed.
//
// Round 3 / 4 ran this loop in Python (gci_amd/device.py: _Staging): per 64 MB slot eight futures submitted and awaited, a lock, a
// torch copy, an event -- a third of a millisecond of interpreter per slot next to the 1.1 ms the slot spends on the bus, and the
// uploads of a whole-genome BAM arrived at 28 - 35 GB/s where the link gives 57.  Now that the device inflates at 90 GB/s of output
// the uploads are what the command line waits for: the loop is native.
#include "gci_ctx.hpp"

#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>

struct gci_stage {
    size_t slot_bytes = 0;
    int n_slots = 0, threads = 1;
    std::vector<void*> slot;
    std::vector<hipEvent_t> ev;
    std::vector<char> used;
    std::mutex ring;                        // a send (or a stretch of a non-urgent one) has the ring to itself
    std::atomic<int> urgent{0};
    // the copying threads and the send they work for: tasks = (piece, part), taken from a counter -- a thread that is done with its
    // part of one piece goes on to the next piece (into the next slot) without waiting for the others
    std::vector<std::thread> workers;
    std::mutex m;
    std::condition_variable go, filled, room;
    const uint8_t* src = nullptr;          // a mapping ... or
    int fd = -1;                            // ... a descriptor read with pread (no page faults on a mapping: the kernel copies out of the page cache itself)
    uint64_t fd_off = 0, n = 0;
    size_t part_bytes = 0;
    uint64_t n_pieces = 0, parts_per_piece = 1;
    std::atomic<uint64_t> next_task{0};
    uint64_t n_tasks = 0;
    std::vector<int> remaining;             // per piece: parts not yet copied (under m)
    uint64_t fill_allowed = 0;              // pieces [0, fill_allowed) may be written into their slots (under m)
    int busy = 0;
    uint64_t gen = 0;
    bool stop = false;
    std::atomic<int> io_error{0};
    int forget = 0;                         // the parts' pages are dropped from the page table as they have been read
    long page = 4096;
};

namespace {

void worker(gci_stage* s)
{
    uint64_t seen = 0;
    for (;;) {
        {
            std::unique_lock<std::mutex> lk(s->m);
            s->go.wait(lk, [&] { return s->stop || s->gen != seen; });
            if (s->stop) return;
            seen = s->gen;
        }
        for (;;) {
            const uint64_t task = s->next_task.fetch_add(1);
            if (task >= s->n_tasks) break;
            const uint64_t piece = task / s->parts_per_piece, part = task % s->parts_per_piece;
            {
                std::unique_lock<std::mutex> lk(s->m);                     // its slot must be free (the copy of the piece that had it: over)
                s->room.wait(lk, [&] { return piece < s->fill_allowed; });
            }
            const uint64_t p0 = piece * (uint64_t)s->slot_bytes, p1 = std::min<uint64_t>(s->n, p0 + s->slot_bytes);
            const uint64_t a = p0 + part * (uint64_t)s->part_bytes, b = std::min<uint64_t>(p1, a + s->part_bytes);
            if (a < b) {
                uint8_t* dst = (uint8_t*)s->slot[(size_t)(piece % (uint64_t)s->n_slots)] + (a - p0);
                if (s->fd < 0) {
                    memcpy(dst, s->src + a, (size_t)(b - a));
                    if (s->forget) {
                        // the part's pages out of this process's page table, by the thread that read them (in the sending thread this
                        // stood between a piece and its DMA: 1.5 ms per 64 MB, i.e. 43 GB/s at most)
                        const uintptr_t pg = (uintptr_t)s->page, lo = ((uintptr_t)(s->src + a) + pg - 1) / pg * pg, hi = (uintptr_t)(s->src + b) / pg * pg;
                        if (hi > lo) (void)madvise((void*)lo, hi - lo, MADV_DONTNEED);
                    }
                } else {
                    for (uint64_t got = 0; got < b - a;) {
                        const ssize_t r = pread(s->fd, dst + got, (size_t)(b - a - got), (off_t)(s->fd_off + a + got));
                        if (r <= 0) { s->io_error = 1; break; }
                        got += (uint64_t)r;
                    }
                }
            }
            std::lock_guard<std::mutex> lk(s->m);
            if (--s->remaining[(size_t)piece] == 0) s->filled.notify_all();
        }
        std::lock_guard<std::mutex> lk(s->m);
        if (--s->busy == 0) s->filled.notify_all();
    }
}

}  // namespace

extern "C" int gci_stage_create(gci_ctx* ctx, uint64_t slot_bytes, int n_slots, int threads, gci_stage** out)
{
    if (!ctx || !out || slot_bytes < 4096 || n_slots < 2 || n_slots > 64 || threads < 1 || threads > 256) return GCI_E_INVALID;
    gci_stage* s = new (std::nothrow) gci_stage;
    if (!s) return GCI_E_NOMEM;
    s->slot_bytes = (size_t)slot_bytes; s->n_slots = n_slots; s->threads = threads;
    s->slot.assign((size_t)n_slots, nullptr);
    s->ev.assign((size_t)n_slots, nullptr);
    s->used.assign((size_t)n_slots, 0);
    for (int k = 0; k < n_slots; k++) {
        // (page-locking 64 MB takes ~20 ms: the slots are locked when they are first needed, not all of them in front of the first byte)
        if (hipEventCreateWithFlags(&s->ev[(size_t)k], hipEventDisableTiming) != hipSuccess) { delete s; return GCI_E_HIP; }
    }
    for (int t = 0; t < threads; t++) s->workers.emplace_back(worker, s);
    *out = s;
    return GCI_OK;
}

static int stage_stretch(gci_ctx* ctx, gci_stage* s, const uint8_t* h_src, int fd, uint64_t fd_off, uint64_t n, uint8_t* d_dst, hipStream_t stream, int forget)
{
    const uint64_t S = (uint64_t)s->n_slots;
    for (int k = 0; k < s->n_slots; k++) {
        if (!s->slot[(size_t)k] && hipHostMalloc(&s->slot[(size_t)k], s->slot_bytes, hipHostMallocDefault) != hipSuccess)
            return gci_fail(ctx, hipGetLastError(), "hipHostMalloc (staging slot)");
        if (s->used[(size_t)k]) {                                            // (a slot an earlier send left on the bus)
            if (hipEventSynchronize(s->ev[(size_t)k]) != hipSuccess) return gci_fail(ctx, hipGetLastError(), "hipEventSynchronize (staging slot)");
            s->used[(size_t)k] = 0;
        }
    }
    const long page = sysconf(_SC_PAGESIZE);
    const uint64_t pieces = (n + s->slot_bytes - 1) / s->slot_bytes;
    size_t part = (s->slot_bytes + 7) / 8;                                   // eight parts per piece (8 MB of a 64 MB slot)
    part = (part + 4095) / 4096 * 4096;
    {
        std::lock_guard<std::mutex> lk(s->m);
        s->src = h_src; s->fd = fd; s->fd_off = fd_off; s->n = n; s->part_bytes = part;
        s->forget = forget && h_src ? 1 : 0; s->page = page;
        s->n_pieces = pieces; s->parts_per_piece = (s->slot_bytes + part - 1) / part;
        s->n_tasks = pieces * s->parts_per_piece;
        s->remaining.assign((size_t)pieces, (int)s->parts_per_piece);
        s->fill_allowed = std::min<uint64_t>(pieces, S);
        s->next_task = 0;
        s->busy = (int)s->workers.size();
        s->io_error = 0;
        s->gen++;
    }
    s->go.notify_all();
    int rc = GCI_OK;
    for (uint64_t p = 0; p < pieces; p++) {
        {
            std::unique_lock<std::mutex> lk(s->m);
            s->filled.wait(lk, [&] { return s->remaining[(size_t)p] == 0; });
        }
        const uint64_t a = p * (uint64_t)s->slot_bytes;
        const size_t len = (size_t)std::min<uint64_t>(s->slot_bytes, n - a);
        const size_t k = (size_t)(p % S);
        if (rc == GCI_OK && fd >= 0 && s->io_error.load()) rc = GCI_E_INVALID;
        if (rc == GCI_OK && (hipMemcpyAsync(d_dst + a, s->slot[k], len, hipMemcpyHostToDevice, stream) != hipSuccess || hipEventRecord(s->ev[k], stream) != hipSuccess))
            rc = gci_fail(ctx, hipGetLastError(), "hipMemcpyAsync (staging slot)");
        if (rc == GCI_OK) s->used[k] = 1;
        // the slot of piece p - (S - 2) is wanted by piece p + 2: wait for the copy that reads it (two copies stay in flight behind it)
        if (p + 2 >= S && p + 2 < pieces + 0 + 0 && p + 2 - S < pieces) {
            const uint64_t q = p + 2 - S;
            if (rc == GCI_OK && s->used[(size_t)(q % S)] && hipEventSynchronize(s->ev[(size_t)(q % S)]) != hipSuccess)
                rc = gci_fail(ctx, hipGetLastError(), "hipEventSynchronize (staging slot)");
            std::lock_guard<std::mutex> lk(s->m);
            s->fill_allowed = std::max<uint64_t>(s->fill_allowed, std::min<uint64_t>(pieces, q + S + 1));
            s->room.notify_all();
        } else if (rc != GCI_OK) {
            std::lock_guard<std::mutex> lk(s->m);                            // (an error: let the threads run out)
            s->fill_allowed = pieces;
            s->room.notify_all();
        }
    }
    {
        std::unique_lock<std::mutex> lk(s->m);
        s->fill_allowed = pieces;
        s->room.notify_all();
        s->filled.wait(lk, [&] { return s->busy == 0; });
    }
    return rc;
}

static int stage_send(gci_ctx* ctx, gci_stage* s, const uint8_t* h_src, int fd, uint64_t fd_off, uint64_t n, uint8_t* d_dst, void* stream, int forget,
                      int urgent)
{
    if (urgent) s->urgent++;
    int rc = GCI_OK;
    // an urgent send takes the ring for all of its bytes; one that is not goes in stretches of four slots and lets urgent ones in between
    const uint64_t stretch = urgent ? n : 4ull * (uint64_t)s->slot_bytes;
    for (uint64_t a = 0; a < n && rc == GCI_OK; a += stretch) {
        while (!urgent && s->urgent.load() > 0) usleep(300);
        std::lock_guard<std::mutex> lk(s->ring);
        const uint64_t len = std::min<uint64_t>(stretch, n - a);
        rc = stage_stretch(ctx, s, h_src ? h_src + a : nullptr, fd, fd_off + a, len, d_dst + a, (hipStream_t)stream, forget);
    }
    if (urgent) s->urgent--;
    return rc;
}

// h_src[0, n) -> d_dst[0, n), enqueued on `stream` (a hipStream_t); returns when the last piece is ENQUEUED (its bytes are in a
// pinned slot by then: the caller may unmap the file).  forget != 0: the pages of h_src are dropped from the process's page table
// as they have been read (madvise DONTNEED: the page cache keeps the data; see device.py _forget_pages for why).  urgent == 0: the
// call lets urgent ones go first, piece by piece (the assembly, whose N runs nobody waits for, next to the runs of a BAM file).
extern "C" int gci_stage_send(gci_ctx* ctx, gci_stage* s, const uint8_t* h_src, uint64_t n, uint8_t* d_dst, void* stream, int forget, int urgent)
{
    if (!ctx || !s || (n && (!h_src || !d_dst))) return GCI_E_INVALID;
    return stage_send(ctx, s, h_src, -1, 0, n, d_dst, stream, forget, urgent);
}

// the same from a file descriptor: bytes [offset, offset + n) of `fd` by pread() into the slots -- no mapping, no page faults
extern "C" int gci_stage_send_fd(gci_ctx* ctx, gci_stage* s, int fd, uint64_t offset, uint64_t n, uint8_t* d_dst, void* stream, int urgent)
{
    if (!ctx || !s || fd < 0 || (n && !d_dst)) return GCI_E_INVALID;
    return stage_send(ctx, s, nullptr, fd, offset, n, d_dst, stream, 0, urgent);
}

extern "C" int gci_stage_free(gci_stage* s)
{
    if (!s) return GCI_OK;
    {
        std::lock_guard<std::mutex> lk(s->m);
        s->stop = true;
    }
    s->go.notify_all();
    for (auto& t : s->workers) t.join();
    for (int k = 0; k < s->n_slots; k++) {
        if (s->used[(size_t)k]) (void)hipEventSynchronize(s->ev[(size_t)k]);
        if (s->ev[(size_t)k]) (void)hipEventDestroy(s->ev[(size_t)k]);
        if (s->slot[(size_t)k]) (void)hipHostFree(s->slot[(size_t)k]);
    }
    delete s;
    return GCI_OK;
}
