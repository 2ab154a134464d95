// Does gfx950 serve unaligned ds_read_b32 / ds_write_b32 / ds_read_b64 / global dwordx4 correctly?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
__global__ void k(const uint8_t* g, uint32_t* out32, unsigned long long* out64, uint32_t* outw, uint4* outg)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[256];
    for (int i = threadIdx.x; i < 256; i += blockDim.x) lds[i] = (uint8_t)i;
    __syncthreads();
    int t = threadIdx.x;            // 0..15: byte offset t
    uint32_t v; __builtin_memcpy(&v, lds + t, 4); out32[t] = v;
    unsigned long long w; __builtin_memcpy(&w, lds + t, 8); out64[t] = w;
    __syncthreads();
    uint32_t x = 0xAABBCCDDu; __builtin_memcpy(lds + 64 + t * 8 + (t & 3), &x, 4);
    __syncthreads();
    uint32_t y; __builtin_memcpy(&y, lds + 64 + t * 8 + (t & 3), 4); outw[t] = y;
    uint4 q; __builtin_memcpy(&q, g + t, 16); outg[t] = q;
}
int main()
{
    uint8_t h[64]; for (int i = 0; i < 64; i++) h[i] = (uint8_t)(100 + i);
    uint8_t* g; uint32_t *o32, *ow; unsigned long long* o64; uint4* og;
    hipMalloc(&g, 64); hipMalloc(&o32, 64); hipMalloc(&o64, 128); hipMalloc(&ow, 64); hipMalloc(&og, 256);
    hipMemcpy(g, h, 64, hipMemcpyHostToDevice);
    k<<<1, 16>>>(g, o32, o64, ow, og);
    uint32_t r32[16], rw[16]; unsigned long long r64[16]; uint4 rg[16];
    hipMemcpy(r32, o32, 64, hipMemcpyDeviceToHost); hipMemcpy(r64, o64, 128, hipMemcpyDeviceToHost);
    hipMemcpy(rw, ow, 64, hipMemcpyDeviceToHost); hipMemcpy(rg, og, 256, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < 16; t++) {
        uint32_t e = t | ((t + 1) << 8) | ((t + 2) << 16) | ((uint32_t)(t + 3) << 24);
        if (r32[t] != e) { bad++; printf("ds_read_b32 off %d: got %08x want %08x\n", t, r32[t], e); }
        unsigned long long e64 = 0; for (int b = 0; b < 8; b++) e64 |= (unsigned long long)(t + b) << (8 * b);
        if (r64[t] != e64) { bad++; printf("ds_read_b64 off %d mismatch\n", t); }
        if (rw[t] != 0xAABBCCDDu) { bad++; printf("ds_write_b32 off %d: got %08x\n", t, rw[t]); }
        uint8_t eg[16]; for (int b = 0; b < 16; b++) eg[b] = (uint8_t)(100 + t + b);
        if (memcmp(&rg[t], eg, 16)) { bad++; printf("global dwordx4 off %d mismatch\n", t); }
    }
    printf("unaligned access test: %s (%d bad)\n", bad ? "FAIL" : "PASS", bad);
    return bad != 0;
}
