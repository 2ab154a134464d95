// k_hbm.hip -- device memory, streams and events for a host that brings no tensor library (include/gci_hip.h, "gci_dev_*"):
// what the single-GPU command line holds its HBM buffers with instead of `import torch` (gci_amd/hbm.py).  Plumbing only: thin
// status-returning wrappers over the HIP runtime, plus three element-wise helpers the host side needs on buffers it cannot touch.
#include "gci_ctx.hpp"

#include <map>
#include <mutex>
#include <unordered_map>

// ---- the arena: the many small and medium device allocations of this library (its contexts' scratch tables, status words, record
// buffers; gci_malloc, gci_dev_malloc) are cut from slabs, so that a run makes tens of driver calls instead of hundreds.  What a
// driver allocation costs on this chip is not the call but the MEMORY: the kernel driver hands out VRAM it knows to be clean, and
// clears what it does not -- measured 35 - 45 ms per GB (tools/hwtests/reserve_timing.py: 16 GB 0.56 s, 64 GB 2.8 s; ~0 when the
// blocks were wiped when a previous process released them).  So nothing is reserved ahead of need (a 64 GB reservation at the start
// of the genome-size command line cost it 2.8 s: profiles/r06c), slabs stay small (256 MiB growing to GCI_ARENA_SLAB_MB, 1024: at
// most ~40 ms under the arena's lock), a request of ARENA_DIRECT (64 MiB) or more that no free block fits becomes a slab of exactly
// its size, made OUTSIDE the lock, and -- the point of the arena at genome size -- memory that one phase of a run gives back is
// what the next phase is cut from: blocks coalesce, the ingestion's 8 GB buffers become the depth track without the driver (and its
// clearing) being asked again.  A slab is never returned while the process lives.  Blocks are 4 KiB granular, best fit.
// GCI_ARENA=0: straight hipMalloc / hipFree.
namespace {
struct Arena {
    std::mutex mu;
    struct Slab { uintptr_t base; size_t size; };
    std::vector<Slab> slabs;
    std::map<uintptr_t, size_t> free_by_addr;
    std::multimap<size_t, uintptr_t> free_by_size;
    std::unordered_map<uintptr_t, size_t> live;
    size_t reserved = 0, in_use = 0;
};
Arena g_arena[16];
const bool g_arena_on = [] { const char* e = getenv("GCI_ARENA"); return !(e && e[0] == '0'); }();
const size_t g_slab_bytes = [] { const char* e = getenv("GCI_ARENA_SLAB_MB"); const long v = e ? atol(e) : 1024; return (size_t)(v < 256 ? 256 : v) << 20; }();
constexpr size_t ARENA_GRAIN = 4096;
constexpr size_t ARENA_DIRECT = (size_t)64 << 20;

void arena_erase_free(Arena& a, uintptr_t addr, size_t size)
{
    a.free_by_addr.erase(addr);
    auto r = a.free_by_size.equal_range(size);
    for (auto it = r.first; it != r.second; ++it) if (it->second == addr) { a.free_by_size.erase(it); break; }
}
void arena_insert_free(Arena& a, uintptr_t addr, size_t size)
{
    a.free_by_addr[addr] = size;
    a.free_by_size.emplace(size, addr);
}
const Arena::Slab* arena_slab_of(const Arena& a, uintptr_t addr)
{
    for (const auto& s : a.slabs) if (addr >= s.base && addr < s.base + s.size) return &s;
    return nullptr;
}
hipError_t arena_add_slab(Arena& a, int device, size_t at_least)
{
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return e;
    // slabs for the small and medium blocks: 256 MiB, then GCI_ARENA_SLAB_MB (1 GiB)
    size_t step = a.slabs.empty() ? ((size_t)256 << 20) : a.slabs.back().size * 4;
    if (step > g_slab_bytes) step = g_slab_bytes;
    if (step < ((size_t)256 << 20)) step = (size_t)256 << 20;
    size_t want = at_least > step ? (at_least + ((size_t)256 << 20) - 1) / ((size_t)256 << 20) * ((size_t)256 << 20) : step;
    void* p = nullptr;
    e = hipMalloc(&p, want);
    if (e != hipSuccess && want > at_least) {              // the device is filling up: exactly what is asked for
        (void)hipGetLastError();
        want = (at_least + (2u << 20) - 1) / (2u << 20) * (2u << 20);
        e = hipMalloc(&p, want);
    }
    if (e != hipSuccess) { (void)hipGetLastError(); return e; }
    a.slabs.push_back({(uintptr_t)p, want});
    a.reserved += want;
    arena_insert_free(a, (uintptr_t)p, want);
    return hipSuccess;
}
}  // namespace

hipError_t gci_dmalloc(int device, void** out, size_t bytes)
{
    if (!out || device < 0 || device >= 16) return hipErrorInvalidValue;
    if (!g_arena_on) { hipError_t e = hipSetDevice(device); return e != hipSuccess ? e : hipMalloc(out, bytes ? bytes : 16); }
    const size_t need = (bytes + ARENA_GRAIN - 1) / ARENA_GRAIN * ARENA_GRAIN + (bytes ? 0 : ARENA_GRAIN);
    Arena& a = g_arena[device];
    std::unique_lock<std::mutex> lock(a.mu);
    auto it = a.free_by_size.lower_bound(need);
    if (it == a.free_by_size.end() && need >= ARENA_DIRECT) {
        // a large block nothing free can hold: a slab of exactly its size, made without the lock (the driver may take 40 ms per GB)
        lock.unlock();
        hipError_t e = hipSetDevice(device);
        if (e != hipSuccess) return e;
        void* p = nullptr;
        e = hipMalloc(&p, need);
        if (e != hipSuccess) { (void)hipGetLastError(); return e; }
        lock.lock();
        a.slabs.push_back({(uintptr_t)p, need});
        a.reserved += need;
        a.live[(uintptr_t)p] = need;
        a.in_use += need;
        *out = p;
        return hipSuccess;
    }
    if (it == a.free_by_size.end()) {
        hipError_t e = arena_add_slab(a, device, need);
        if (e != hipSuccess) return e;
        it = a.free_by_size.lower_bound(need);
        if (it == a.free_by_size.end()) return hipErrorOutOfMemory;
    }
    const size_t have = it->first;
    const uintptr_t addr = it->second;
    a.free_by_size.erase(it);
    a.free_by_addr.erase(addr);
    if (have > need) arena_insert_free(a, addr + need, have - need);
    a.live[addr] = need;
    a.in_use += need;
    *out = (void*)addr;
    return hipSuccess;
}

hipError_t gci_dfree(void* p)
{
    if (!p) return hipSuccess;
    if (!g_arena_on) return hipFree(p);
    const uintptr_t addr = (uintptr_t)p;
    for (Arena& a : g_arena) {
        std::lock_guard<std::mutex> lock(a.mu);
        auto lv = a.live.find(addr);
        if (lv == a.live.end()) continue;
        size_t size = lv->second;
        a.live.erase(lv);
        a.in_use -= size;
        const Arena::Slab* slab = arena_slab_of(a, addr);
        uintptr_t lo = addr;
        auto nx = a.free_by_addr.find(addr + size);                       // the free neighbour behind, inside the same slab
        if (nx != a.free_by_addr.end() && slab && nx->first < slab->base + slab->size) { const size_t ns = nx->second; arena_erase_free(a, addr + size, ns); size += ns; }
        auto pv = a.free_by_addr.lower_bound(addr);                       // ... and the one in front
        if (pv != a.free_by_addr.begin()) {
            --pv;
            if (pv->first + pv->second == addr && slab && pv->first >= slab->base) { lo = pv->first; const size_t ps = pv->second; arena_erase_free(a, lo, ps); size += ps; }
        }
        arena_insert_free(a, lo, size);
        return hipSuccess;
    }
    return hipFree(p);                                     // (not the arena's: a pointer from before GCI_ARENA, or a caller's mistake the runtime reports)
}

namespace {
thread_local std::string g_dev_err;
int dev_fail(hipError_t e, const char* what) { g_dev_err = std::string(what) + ": " + hipGetErrorString(e); return GCI_E_HIP; }
#define DEVCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return dev_fail(e_, #call); } while (0)

__global__ __launch_bounds__(256) void k_i64_add(const int64_t* __restrict__ in, uint64_t n, int64_t delta, int64_t* __restrict__ out)
{
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) out[i] = in[i] + delta;
}

__global__ __launch_bounds__(256) void k_rec_flags_and(gci_rec* __restrict__ recs, uint64_t n, uint32_t mask)
{
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) recs[i].flags &= (uint8_t)mask;
}

// out[0] = 0, out[i + 1] = in[0] + ... + in[i] (uint32 -> uint64), one workgroup: the member sizes of a .depth.gz (<= a few 10^4)
__global__ __launch_bounds__(1024) void k_u32_scan_u64(const uint32_t* __restrict__ in, uint32_t n, uint64_t* __restrict__ out)
{
    __shared__ uint64_t part[1024];
    const uint32_t t = threadIdx.x, per = (n + 1023u) / 1024u;
    const uint32_t lo = t * per < n ? t * per : n, hi = lo + per < n ? lo + per : n;
    uint64_t s = 0;
    for (uint32_t i = lo; i < hi; i++) s += in[i];
    part[t] = s;
    __syncthreads();
    for (uint32_t d = 1; d < 1024u; d <<= 1) {                   // Hillis-Steele over the 1024 partial sums
        const uint64_t v = t >= d ? part[t - d] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    uint64_t run = t ? part[t - 1] : 0;
    if (t == 0) out[0] = 0;
    for (uint32_t i = lo; i < hi; i++) { run += in[i]; out[i + 1] = run; }
}
}  // namespace

extern "C" {

const char* gci_dev_last_error(void) { return g_dev_err.c_str(); }

int gci_dev_count(int* n_out)
{
    if (!n_out) return GCI_E_INVALID;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
    *n_out = n;
    return GCI_OK;
}

int gci_dev_malloc(int device, size_t bytes, void** d_out)
{
    if (!d_out) return GCI_E_INVALID;
    hipError_t e = gci_dmalloc(device, d_out, bytes);
    if (e == hipErrorOutOfMemory) { (void)hipGetLastError(); g_dev_err = "device memory: out of memory"; return GCI_E_NOMEM; }
    if (e != hipSuccess) return dev_fail(e, "gci_dmalloc");
    return GCI_OK;
}
int gci_dev_free(int device, void* d_ptr)
{
    (void)device;
    if (d_ptr) DEVCHK(gci_dfree(d_ptr));
    return GCI_OK;
}
// Slabs for at least `bytes` in all, made now (one driver allocation for what is missing -- at the driver's 35 - 45 ms per GB of
// memory it has to clear).  For a host that wants the cost in a place of its choosing; the command line does not use it.
// *reserved_out: the arena's slabs after it.
int gci_dev_reserve(int device, uint64_t bytes, uint64_t* reserved_out)
{
    if (device < 0 || device >= 16) return GCI_E_INVALID;
    Arena& a = g_arena[device];
    if (g_arena_on) {
        std::lock_guard<std::mutex> lock(a.mu);
        while (a.reserved < bytes) {
            const size_t before = a.reserved;
            hipError_t e = arena_add_slab(a, device, (size_t)(bytes - a.reserved));
            if (e == hipErrorOutOfMemory) { g_dev_err = "gci_dev_reserve: out of memory"; break; }
            if (e != hipSuccess) return dev_fail(e, "gci_dev_reserve");
            if (a.reserved == before) break;
        }
    }
    if (reserved_out) *reserved_out = a.reserved;
    return GCI_OK;
}
int gci_dev_arena_info(int device, uint64_t* reserved, uint64_t* in_use, uint32_t* n_slabs)
{
    if (device < 0 || device >= 16) return GCI_E_INVALID;
    Arena& a = g_arena[device];
    std::lock_guard<std::mutex> lock(a.mu);
    if (reserved) *reserved = a.reserved;
    if (in_use) *in_use = a.in_use;
    if (n_slabs) *n_slabs = (uint32_t)a.slabs.size();
    return GCI_OK;
}
int gci_dev_mem_info(int device, uint64_t* free_bytes, uint64_t* total_bytes)
{
    DEVCHK(hipSetDevice(device));
    size_t f = 0, t = 0;
    DEVCHK(hipMemGetInfo(&f, &t));
    if (free_bytes) *free_bytes = f;
    if (total_bytes) *total_bytes = t;
    return GCI_OK;
}
int gci_dev_sync(int device)
{
    DEVCHK(hipSetDevice(device));
    DEVCHK(hipDeviceSynchronize());
    return GCI_OK;
}
int gci_dev_host_alloc(int device, size_t bytes, void** h_out)
{
    if (!h_out) return GCI_E_INVALID;
    DEVCHK(hipSetDevice(device));
    DEVCHK(hipHostMalloc(h_out, bytes ? bytes : 16, hipHostMallocDefault));
    return GCI_OK;
}
int gci_dev_host_free(int device, void* h_ptr)
{
    DEVCHK(hipSetDevice(device));
    if (h_ptr) DEVCHK(hipHostFree(h_ptr));
    return GCI_OK;
}

int gci_dev_stream_create(int device, void** out)
{
    if (!out) return GCI_E_INVALID;
    DEVCHK(hipSetDevice(device));
    hipStream_t s = nullptr;
    DEVCHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *out = (void*)s;
    return GCI_OK;
}
int gci_dev_stream_destroy(int device, void* stream)
{
    DEVCHK(hipSetDevice(device));
    if (stream) DEVCHK(hipStreamDestroy((hipStream_t)stream));
    return GCI_OK;
}
int gci_dev_stream_sync(int device, void* stream)
{
    DEVCHK(hipSetDevice(device));
    DEVCHK(hipStreamSynchronize((hipStream_t)stream));
    return GCI_OK;
}
int gci_dev_event_create(int device, int timing, void** out)
{
    if (!out) return GCI_E_INVALID;
    DEVCHK(hipSetDevice(device));
    hipEvent_t e = nullptr;
    DEVCHK(hipEventCreateWithFlags(&e, timing ? hipEventDefault : hipEventDisableTiming));
    *out = (void*)e;
    return GCI_OK;
}
int gci_dev_event_destroy(int device, void* event)
{
    DEVCHK(hipSetDevice(device));
    if (event) DEVCHK(hipEventDestroy((hipEvent_t)event));
    return GCI_OK;
}
int gci_dev_event_record(int device, void* event, void* stream)
{
    DEVCHK(hipSetDevice(device));
    DEVCHK(hipEventRecord((hipEvent_t)event, (hipStream_t)stream));
    return GCI_OK;
}
int gci_dev_event_sync(int device, void* event)
{
    DEVCHK(hipSetDevice(device));
    DEVCHK(hipEventSynchronize((hipEvent_t)event));
    return GCI_OK;
}
int gci_dev_event_elapsed_ms(int device, void* a, void* b, double* ms)
{
    if (!ms) return GCI_E_INVALID;
    DEVCHK(hipSetDevice(device));
    float f = 0.f;
    DEVCHK(hipEventElapsedTime(&f, (hipEvent_t)a, (hipEvent_t)b));
    *ms = f;
    return GCI_OK;
}
int gci_dev_stream_wait_event(int device, void* stream, void* event)
{
    DEVCHK(hipSetDevice(device));
    DEVCHK(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)event, 0));
    return GCI_OK;
}

// kind: 1 = host -> device, 2 = device -> host, 3 = device -> device.  Asynchronous on `stream` (pinned host memory; a pageable
// host buffer makes the runtime stage the bytes, and the caller synchronises before it lets go of the buffer).
int gci_dev_memcpy_async(int device, void* dst, const void* src, size_t bytes, int kind, void* stream)
{
    if (kind < 1 || kind > 3) return GCI_E_INVALID;
    DEVCHK(hipSetDevice(device));
    if (!bytes) return GCI_OK;
    const hipMemcpyKind k = kind == 1 ? hipMemcpyHostToDevice : kind == 2 ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
    DEVCHK(hipMemcpyAsync(dst, src, bytes, k, (hipStream_t)stream));
    return GCI_OK;
}
int gci_dev_memset_async(int device, void* d_dst, int byte, size_t bytes, void* stream)
{
    DEVCHK(hipSetDevice(device));
    if (bytes) DEVCHK(hipMemsetAsync(d_dst, byte, bytes, (hipStream_t)stream));
    return GCI_OK;
}

int gci_dev_i64_add(int device, const int64_t* d_in, uint64_t n, int64_t delta, int64_t* d_out, void* stream)
{
    DEVCHK(hipSetDevice(device));
    if (!n) return GCI_OK;
    if (!d_in || !d_out) return GCI_E_INVALID;
    const uint64_t wg = (n + 255) / 256;
    hipLaunchKernelGGL(k_i64_add, dim3((uint32_t)(wg < 4096 ? wg : 4096)), dim3(256), 0, (hipStream_t)stream, d_in, n, delta, d_out);
    DEVCHK(hipGetLastError());
    return GCI_OK;
}
int gci_dev_rec_flags_and(int device, gci_rec* d_recs, uint64_t n, uint32_t mask, void* stream)
{
    DEVCHK(hipSetDevice(device));
    if (!n) return GCI_OK;
    if (!d_recs) return GCI_E_INVALID;
    const uint64_t wg = (n + 255) / 256;
    hipLaunchKernelGGL(k_rec_flags_and, dim3((uint32_t)(wg < 4096 ? wg : 4096)), dim3(256), 0, (hipStream_t)stream, d_recs, n, mask);
    DEVCHK(hipGetLastError());
    return GCI_OK;
}
int gci_dev_u32_scan_u64(int device, const uint32_t* d_in, uint32_t n, uint64_t* d_out, void* stream)
{
    DEVCHK(hipSetDevice(device));
    if (!d_out || (n && !d_in)) return GCI_E_INVALID;
    hipLaunchKernelGGL(k_u32_scan_u64, dim3(1), dim3(1024), 0, (hipStream_t)stream, d_in, n, d_out);
    DEVCHK(hipGetLastError());
    return GCI_OK;
}

}  // extern "C"
