// gci_common.h -- helpers shared by the host and device halves of libgci_hip.so.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define GCI_HD __host__ __device__ __forceinline__
#else
#define GCI_HD inline
#endif

// splitmix64 finaliser (bijective on 64 bits)
GCI_HD uint64_t gci_mix64(uint64_t x)
{
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull;
    x ^= x >> 27; x *= 0x94d049bb133111ebull;
    x ^= x >> 31;
    return x;
}

// Contribution of the k-th little-endian 8-byte word of a name (zero padded past the end): multilinear,
// (w ^ key_k) * m_k with an odd multiplier per position -- ONE 64-bit multiplication per word (round 2 ran the whole
// splitmix finaliser, two multiplications and three shifts, on every word: a sixth of the record filter's time).  The sum of
// the contributions is order sensitive (key and multiplier depend on the position), lanes hash words independently and add,
// and the finaliser below avalanches the sum.  Names are always confirmed on their bytes by the join: the hash only has to
// spread them.
GCI_HD uint64_t gci_hash_word(uint64_t w, uint32_t k)
{
    const uint64_t key = 0x9E3779B97F4A7C15ull * (uint64_t)(k + 1);
    return (w ^ key) * (((key >> 1) ^ 0xbf58476d1ce4e5b9ull) | 1ull);
}

GCI_HD uint64_t gci_hash_finish(uint64_t acc, uint32_t len)
{
    return gci_mix64(acc ^ ((uint64_t)len * 0xD6E8FEB86659FD93ull));
}

// Python / NumPy slice-bound normalisation for a sequence of length L
// (depths[t][a:b] at GCI.py:306 and :328).
GCI_HD int64_t gci_slice_bound(int64_t v, int64_t L)
{
    if (v < 0) { v += L; if (v < 0) v = 0; }
    else if (v > L) v = L;
    return v;
}
