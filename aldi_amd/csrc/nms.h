// Batched NMS on score-sorted boxes (shared by the RPN and the ROI-head inference path).
#pragma once
#include "common.h"

// ---------------------------------------------------------------------------------------
// generic batched NMS on score-sorted boxes (shared with the ROI-head inference path)
// ---------------------------------------------------------------------------------------
static __device__ __forceinline__ bool nms_over(const float4 a, const float4 b, float thr) {
    // torchvision nms: inter / (area_a + area_b - inter) > thr
    float w = fminf(a.z, b.z) - fmaxf(a.x, b.x);
    float h = fminf(a.w, b.w) - fmaxf(a.y, b.y);
    w = w > 0.f ? w : 0.f;
    h = h > 0.f ? h : 0.f;
    float inter = w * h;
    float aa = (a.z - a.x) * (a.w - a.y), ab = (b.z - b.x) * (b.w - b.y);
    return inter / (aa + ab - inter) > thr;
}

// mask[b][row][cw] bit j: box (cw*64+j) is suppressed by `row` (j > row, same category, both valid)
static __global__ __launch_bounds__(64) void nms_mask_kernel(const float4* __restrict__ boxes, const int* __restrict__ valid, const int* __restrict__ cat,
                                                      const int* __restrict__ count, int cap, float thr, unsigned long long* __restrict__ mask) {
    const int b = blockIdx.z, rb = blockIdx.y, cb = blockIdx.x;
    const int n = count ? count[b] : cap;
    if (rb * 64 >= n || cb * 64 >= n || cb < rb) return;
    __shared__ float4 cbox[64];
    __shared__ int ccat[64];
    const int j = cb * 64 + threadIdx.x;
    bool jv = j < n && valid[(long)b * cap + j];
    cbox[threadIdx.x] = jv ? boxes[(long)b * cap + j] : make_float4(0, 0, 0, 0);
    ccat[threadIdx.x] = jv ? (cat ? cat[(long)b * cap + j] : 0) : -1;
    __syncthreads();
    const int i = rb * 64 + threadIdx.x;
    unsigned long long bits = 0;
    if (i < n && valid[(long)b * cap + i]) {
        const float4 bi = boxes[(long)b * cap + i];
        const int ci = cat ? cat[(long)b * cap + i] : 0;
        for (int t = 0; t < 64; ++t) {
            int jj = cb * 64 + t;
            if (jj > i && ccat[t] == ci && nms_over(bi, cbox[t], thr)) bits |= 1ull << t;
        }
    }
    if (i < n) mask[((long)b * cap + i) * (cap / 64) + cb] = bits;
}

// one wave per batch item: sequential resolution in 64-box chunks
static __global__ __launch_bounds__(64) void nms_scan_kernel(const unsigned long long* __restrict__ mask, const int* __restrict__ valid,
                                                      const int* __restrict__ count, int cap, int max_keep,
                                                      int* __restrict__ keep /*[B][cap]*/, int* __restrict__ keep_count) {
    extern __shared__ unsigned long long removed[];   // cap/64 words
    const int b = blockIdx.x, lane = threadIdx.x;
    const int n = count ? count[b] : cap;
    const int words = cap / 64;
    for (int w = lane; w < words; w += 64) removed[w] = 0;
    __syncthreads();
    int nk = 0;
    const int chunks = (n + 63) / 64;
    for (int c = 0; c < chunks && nk < max_keep; ++c) {
        const int i = c * 64 + lane;
        const bool v = i < n && valid[(long)b * cap + i];
        const unsigned long long diag = (i < n) ? ((c * 64 <= i) ? mask[((long)b * cap + i) * words + c] : 0ull) : 0ull;
        unsigned long long rem = removed[c];
        unsigned long long inval = ~__ballot(v);
        rem |= inval;
        // resolve inside the chunk: box t survives iff not removed when reached
        unsigned long long kept = 0;
        for (int t = 0; t < 64; ++t) {
            unsigned long long dt = __shfl(diag, t, 64);
            if (!((rem >> t) & 1ull)) { kept |= 1ull << t; rem |= dt; }
        }
        // limit to max_keep
        int kc = __popcll(kept);
        if (nk + kc > max_keep) {
            int allow = max_keep - nk;
            unsigned long long kk = 0;
            for (int t = 0; t < 64 && allow > 0; ++t)
                if ((kept >> t) & 1ull) { kk |= 1ull << t; --allow; }
            kept = kk;
            kc = __popcll(kept);
        }
        if ((kept >> lane) & 1ull) keep[(long)b * cap + nk + __popcll(kept & ((1ull << lane) - 1ull))] = i;
        nk += kc;
        // OR the kept rows into the later words
        __syncthreads();
        for (int w = c + 1 + lane; w < words; w += 64) {
            unsigned long long acc = removed[w];
            unsigned long long kk = kept;
            while (kk) {
                int t = __ffsll((long long)kk) - 1;
                kk &= kk - 1;
                acc |= mask[((long)b * cap + c * 64 + t) * words + w];
            }
            removed[w] = acc;
        }
        __syncthreads();
    }
    if (lane == 0) keep_count[b] = nk;
}

