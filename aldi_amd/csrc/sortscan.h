// Block-level building blocks for the index-producing stages (top-k, NMS order, sampling
// lists).  Everything here is deterministic: ties always resolve to the lower index, which is
// the order the oracle defines (oracle/d2_rcnn.py: stable descending sort).
#pragma once
#include "common.h"

// order-preserving float -> uint32 (ascending); -0.0 is canonicalised to +0.0
__device__ __forceinline__ uint32_t float_key_asc(float f) {
    f = f + 0.0f;
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_asc_to_float(uint32_t k) {
    uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(u);
}

// exclusive prefix of `flag` over the block in thread order; returns rank, sets *total.
// blockDim.x multiple of 64, <= 1024.  smem: >= 17 ints.
__device__ __forceinline__ int block_rank(bool flag, int* smem, int* total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    unsigned long long b = __ballot(flag);
    int r = __popcll(b & ((1ull << lane) - 1ull));
    __syncthreads();
    if (lane == 0) smem[w] = __popcll(b);
    __syncthreads();
    int nw = blockDim.x >> 6, base = 0, tot = 0;
    for (int i = 0; i < nw; ++i) {
        int c = smem[i];
        if (i < w) base += c;
        tot += c;
    }
    *total = tot;
    return base + r;
}

// in-LDS bitonic sort, ascending, n a power of two, all threads of the block participate
__device__ __forceinline__ void bitonic_sort_u64(unsigned long long* keys, int n) {
    for (int k = 2; k <= n; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            __syncthreads();
            for (int i = threadIdx.x; i < n; i += blockDim.x) {
                int ixj = i ^ j;
                if (ixj > i) {
                    unsigned long long a = keys[i], b = keys[ixj];
                    bool up = (i & k) == 0;
                    if ((a > b) == up) { keys[i] = b; keys[ixj] = a; }
                }
            }
        }
    }
    __syncthreads();
}
