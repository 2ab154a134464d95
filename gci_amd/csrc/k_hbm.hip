// k_hbm.hip -- device memory, streams and events for a host that brings no tensor library (include/gci_hip.h, "gci_dev_*"):
// what the single-GPU command line holds its HBM buffers with instead of `import torch` (gci_amd/hbm.py).  Plumbing only: thin
// status-returning wrappers over the HIP runtime, plus three element-wise helpers the host side needs on buffers it cannot touch.
#include "gci_ctx.hpp"

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
    DEVCHK(hipSetDevice(device));
    hipError_t e = hipMalloc(d_out, bytes ? bytes : 16);
    if (e == hipErrorOutOfMemory) { (void)hipGetLastError(); g_dev_err = "hipMalloc: out of memory"; return GCI_E_NOMEM; }
    if (e != hipSuccess) return dev_fail(e, "hipMalloc");
    return GCI_OK;
}
int gci_dev_free(int device, void* d_ptr)
{
    DEVCHK(hipSetDevice(device));
    if (d_ptr) DEVCHK(hipFree(d_ptr));
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
