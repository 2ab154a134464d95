// random_read.hip -- what a random read of HBM costs on this chip (MI355X), to size the join's name comparisons:
//   A  one 16-byte load per thread at a random 128-byte line
//   B  two 16-byte loads per thread, both halves (+0, +64) of ONE random 128-byte line
//   C  one 32-byte read (two 16-byte loads) at a random 32-byte slot
//   D  two independent random lines per thread (memory-level parallelism)
// If B costs what A costs, the memory side moves 128-byte lines; if it costs ~2x, 64-byte sectors.
// Build: hipcc --offload-arch=gfx950 -O3 -o random_read random_read.hip ; run: ./random_read [buffer MiB]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x)
{
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull; x ^= x >> 27; x *= 0x94d049bb133111ebull; x ^= x >> 31;
    return x;
}

template <int MODE>
__global__ __launch_bounds__(256) void k_rand(const uint8_t* __restrict__ buf, uint64_t n_lines, uint64_t n_threads, uint32_t* __restrict__ sink, uint64_t salt)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_threads) return;
    const uint64_t r = mix(i + salt);
    const uint8_t* p = buf + (r % n_lines) * 128;
    uint4 a = make_uint4(0, 0, 0, 0), b = a;
    if (MODE == 0) a = *reinterpret_cast<const uint4*>(p);
    if (MODE == 1) { a = *reinterpret_cast<const uint4*>(p); b = *reinterpret_cast<const uint4*>(p + 64); }
    if (MODE == 2) { const uint8_t* q = p + 32 * ((r >> 40) & 3); a = *reinterpret_cast<const uint4*>(q); b = *reinterpret_cast<const uint4*>(q + 16); }
    if (MODE == 3) { a = *reinterpret_cast<const uint4*>(p); b = *reinterpret_cast<const uint4*>(buf + (mix(r) % n_lines) * 128); }
    const uint32_t v = a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w;
    if (v == 0x12345678u) sink[0] = v;
}

template <int MODE>
static void run(const char* what, const uint8_t* buf, uint64_t n_lines, uint64_t n_threads, uint32_t* sink, int lines_per_thread)
{
    hipEvent_t a, b;
    CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    const uint32_t grid = (uint32_t)((n_threads + 255) / 256);
    hipLaunchKernelGGL((k_rand<MODE>), dim3(grid), dim3(256), 0, 0, buf, n_lines, n_threads, sink, 1ull);
    CHK(hipDeviceSynchronize());
    float best = 1e9f;
    for (int rep = 0; rep < 5; rep++) {
        CHK(hipEventRecord(a, 0));
        hipLaunchKernelGGL((k_rand<MODE>), dim3(grid), dim3(256), 0, 0, buf, n_lines, n_threads, sink, 1000ull * (rep + 2));
        CHK(hipEventRecord(b, 0));
        CHK(hipEventSynchronize(b));
        float ms; CHK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    const double lines = (double)n_threads * lines_per_thread;
    printf("%-58s %8.3f ms  %6.2f G lines/s  = %5.2f TB/s at 64 B, %5.2f TB/s at 128 B per line\n", what, best,
           lines / best / 1e6, lines * 64 / best / 1e9, lines * 128 / best / 1e9);
}

int main(int argc, char** argv)
{
    const uint64_t mib = argc > 1 ? strtoull(argv[1], nullptr, 0) : 4096;
    const uint64_t bytes = mib << 20, n_lines = bytes / 128;
    uint8_t* buf; uint32_t* sink;
    CHK(hipMalloc(&buf, bytes)); CHK(hipMalloc(&sink, 64));
    CHK(hipMemset(buf, 1, bytes));
    const uint64_t n = 32ull << 20;
    printf("buffer %llu MiB, %llu threads\n", (unsigned long long)mib, (unsigned long long)n);
    run<0>("A one 16-byte load at a random 128-byte line", buf, n_lines, n, sink, 1);
    run<1>("B both 64-byte halves of one random line", buf, n_lines, n, sink, 1);
    run<2>("C 32 bytes at a random 32-byte slot", buf, n_lines, n, sink, 1);
    run<3>("D two independent random lines per thread", buf, n_lines, n, sink, 2);
    return 0;
}
