// unaligned8.hip -- what the inflate kernel's output path costs on this chip (MI355X): 8 lanes of a wave, each walking its own 64 KB
// region the way a DEFLATE decoder does -- store 8 bytes, load 8 bytes back from a little further behind (the match copy reads
// the member's own output), dependent -- with the addresses 8-byte ALIGNED or at an odd byte offset.
//   A  aligned   store + dependent aligned load
//   U  unaligned store + dependent unaligned load   (what k_bgzf_inflate does: ld8 / st8 at any address)
//   S  unaligned stores only      L  unaligned loads only (of earlier data)
// Build: hipcc --offload-arch=gfx950 -O3 -o unaligned8 unaligned8.hip ; run: ./unaligned8
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ unsigned long long ld8(const uint8_t* p) { unsigned long long v; __builtin_memcpy(&v, p, 8); return v; }
__device__ __forceinline__ void st8(uint8_t* p, unsigned long long v) { __builtin_memcpy(p, &v, 8); }

template <int MODE>
__global__ __launch_bounds__(64) void k_walk(uint8_t* __restrict__ buf, uint32_t region, uint32_t iters, uint32_t skew, unsigned long long* __restrict__ sink)
{
    const int lane = threadIdx.x;
    if (lane >= 8) return;
    uint8_t* base = buf + ((size_t)blockIdx.x * 8 + lane) * region + skew;     // skew = 0: every access 8-byte aligned
    unsigned long long v = blockIdx.x * 8 + lane;
    uint32_t op = 64;
    for (uint32_t i = 0; i < iters; i++) {
        if (MODE == 0 || MODE == 1 || MODE == 2) st8(base + op, v);
        if (MODE == 0 || MODE == 1 || MODE == 3) v += ld8(base + op - 24 - 8 * (i & 3));      // behind: written a few iterations ago
        else v += i;
        op += 8;
        if (op + 16 > region) op = 64;
    }
    if (v == 0x123456789ull) sink[0] = v;
}

template <int MODE>
static void run(const char* what, uint8_t* buf, uint32_t n_blocks, uint32_t region, uint32_t iters, uint32_t skew, unsigned long long* sink)
{
    hipEvent_t a, b;
    CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    float best = 1e30f;
    for (int r = 0; r < 3; r++) {
        CHK(hipEventRecord(a));
        hipLaunchKernelGGL(k_walk<MODE>, dim3(n_blocks), dim3(64), 0, 0, buf, region, iters, skew, sink);
        CHK(hipEventRecord(b));
        CHK(hipEventSynchronize(b));
        float ms; CHK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    const double acc = (double)n_blocks * 8 * iters;
    printf("%-52s %8.2f ms   %7.2f G lane-iterations/s   %6.1f ns per iteration of a lane\n", what, best, acc / best / 1e6,
           best * 1e6 / iters);
}

int main()
{
    const uint32_t region = 65536, n_blocks = 256 * 14, iters = 8000;      // 112 members per CU, like the product kernel
    uint8_t* buf; unsigned long long* sink;
    CHK(hipMalloc(&buf, (size_t)n_blocks * 8 * region + 4096)); CHK(hipMalloc(&sink, 8));
    CHK(hipMemset(buf, 1, (size_t)n_blocks * 8 * region + 4096));
    run<0>("A  aligned store + dependent aligned load", buf, n_blocks, region, iters, 0, sink);
    run<1>("U  unaligned (+3) store + dependent unaligned load", buf, n_blocks, region, iters, 3, sink);
    run<1>("U4 offset +4 (dword aligned) store + load", buf, n_blocks, region, iters, 4, sink);
    run<2>("S  unaligned (+3) stores only", buf, n_blocks, region, iters, 3, sink);
    run<2>("S0 aligned stores only", buf, n_blocks, region, iters, 0, sink);
    run<3>("L  unaligned (+3) loads only", buf, n_blocks, region, iters, 3, sink);
    run<3>("L0 aligned loads only", buf, n_blocks, region, iters, 0, sink);
    return 0;
}
