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
    int next = 0;
    std::mutex piece;                       // a piece (one slot's worth) at a time: two senders take slots in turns
    std::atomic<int> urgent{0};
    // the copying threads
    std::vector<std::thread> workers;
    std::mutex m;
    std::condition_variable go, done;
    const uint8_t* src = nullptr;          // a mapping ... or
    int fd = -1;                            // ... a descriptor read with pread (no page faults on a mapping: the kernel copies out of the page cache itself)
    uint64_t fd_off = 0;
    uint8_t* dst = nullptr;
    size_t n = 0, step = 0;
    std::atomic<int> io_error{0};
    std::atomic<size_t> part{0};
    int busy = 0;
    uint64_t gen = 0;
    bool stop = false;
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
            const size_t at = s->part.fetch_add(1) * s->step;
            if (at >= s->n) break;
            const size_t len = std::min(s->step, s->n - at);
            if (s->fd < 0) memcpy(s->dst + at, s->src + at, len);
            else {
                for (size_t got = 0; got < len;) {
                    const ssize_t r = pread(s->fd, s->dst + at + got, len - got, (off_t)(s->fd_off + at + got));
                    if (r <= 0) { s->io_error = 1; break; }
                    got += (size_t)r;
                }
            }
        }
        std::lock_guard<std::mutex> lk(s->m);
        if (--s->busy == 0) s->done.notify_one();
    }
}

// src[0, n) -> dst by all the threads; returns when the last byte is there
void copy_parallel(gci_stage* s, uint8_t* dst, const uint8_t* src, int fd, uint64_t fd_off, size_t n)
{
    const size_t T = (size_t)s->threads;
    size_t step = (n + T - 1) / T;
    step = (step + 4095) / 4096 * 4096;
    {
        std::lock_guard<std::mutex> lk(s->m);
        s->src = src; s->fd = fd; s->fd_off = fd_off; s->dst = dst; s->n = n; s->step = step;
        s->part = 0;
        s->busy = (int)s->workers.size();
        s->gen++;
    }
    s->go.notify_all();
    std::unique_lock<std::mutex> lk(s->m);
    s->done.wait(lk, [&] { return s->busy == 0; });
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

// h_src[0, n) -> d_dst[0, n), enqueued on `stream` (a hipStream_t); returns when the last piece is ENQUEUED (its bytes are in a
// pinned slot by then: the caller may unmap the file).  forget != 0: the pages of h_src are dropped from the process's page table
// as they have been read (madvise DONTNEED: the page cache keeps the data; see device.py _forget_pages for why).  urgent == 0: the
// call lets urgent ones go first, piece by piece (the assembly, whose N runs nobody waits for, next to the runs of a BAM file).
static int stage_send(gci_ctx* ctx, gci_stage* s, const uint8_t* h_src, int fd, uint64_t fd_off, uint64_t n, uint8_t* d_dst, void* stream, int forget,
                      int urgent)
{
    if (urgent) s->urgent++;
    int rc = GCI_OK;
    const long page = sysconf(_SC_PAGESIZE);
    for (uint64_t a = 0; a < n && rc == GCI_OK; a += s->slot_bytes) {
        const size_t len = (size_t)std::min<uint64_t>(s->slot_bytes, n - a);
        while (!urgent && s->urgent.load() > 0) usleep(300);
        std::lock_guard<std::mutex> lk(s->piece);
        const int k = s->next;
        s->next = (k + 1) % s->n_slots;
        if (!s->slot[(size_t)k]) {
            if (hipHostMalloc(&s->slot[(size_t)k], s->slot_bytes, hipHostMallocDefault) != hipSuccess) { rc = gci_fail(ctx, hipGetLastError(), "hipHostMalloc (staging slot)"); break; }
        }
        if (s->used[(size_t)k] && hipEventSynchronize(s->ev[(size_t)k]) != hipSuccess) { rc = gci_fail(ctx, hipGetLastError(), "hipEventSynchronize (staging slot)"); break; }
        copy_parallel(s, (uint8_t*)s->slot[(size_t)k], h_src ? h_src + a : nullptr, fd, fd_off + a, len);
        if (fd >= 0 && s->io_error.load()) { rc = GCI_E_INVALID; break; }
        if (forget && h_src) {
            const uintptr_t lo = ((uintptr_t)(h_src + a) + (uintptr_t)page - 1) / (uintptr_t)page * (uintptr_t)page, hi = (uintptr_t)(h_src + a + len) / (uintptr_t)page * (uintptr_t)page;
            if (hi > lo) (void)madvise((void*)lo, hi - lo, MADV_DONTNEED);
        }
        if (hipMemcpyAsync(d_dst + a, s->slot[(size_t)k], len, hipMemcpyHostToDevice, (hipStream_t)stream) != hipSuccess ||
            hipEventRecord(s->ev[(size_t)k], (hipStream_t)stream) != hipSuccess) { rc = gci_fail(ctx, hipGetLastError(), "hipMemcpyAsync (staging slot)"); break; }
        s->used[(size_t)k] = 1;
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
    s->io_error = 0;
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
