// ConvNeXt-specific kernels (reference aldi/backbone.py:189-225, ConvNextBlock): depthwise 7x7 convolution (forward,
// data gradient = the same kernel with the taps flipped, weight gradient) and the layer-scale residual
// out = x + s * gamma (.) y with its backward.  NHWC, 8 channels (16 B of bf16) per lane; the pointwise layers, LayerNorm and GELU
// of the block run on the igemm / LayerNorm / GELU kernels.
//
// No LDS tile: every output vector re-reads its input vectors through L1/L2 (XCD-aware block order keeps them in one L2).  Both
// kernels are instruction- and latency-bound, not HBM-bound: operands are prefetched one step ahead as raw 16-B vectors into
// ping-pong register buffers and unpacked a channel pair at a time (v_pk_fma_f32 on the pair).
#include "common.h"

namespace {

__device__ __forceinline__ void load8(const bf16_t* p, float v[8]) {
    const uint4 t = *reinterpret_cast<const uint4*>(p);
    const uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) { v[2 * k] = __uint_as_float(w[k] << 16); v[2 * k + 1] = __uint_as_float(w[k] & 0xffff0000u); }
}
__device__ __forceinline__ void load8(const float* p, float v[8]) { load4(p, v); load4(p + 4, v + 4); }
__device__ __forceinline__ void store8(bf16_t* p, const float v[8]) {
    uint4 t;
    t.x = pack2_bf16(v[0], v[1]); t.y = pack2_bf16(v[2], v[3]); t.z = pack2_bf16(v[4], v[5]); t.w = pack2_bf16(v[6], v[7]);
    *reinterpret_cast<uint4*>(p) = t;
}
__device__ __forceinline__ void store8(float* p, const float v[8]) { store4(p, v); store4(p + 4, v + 4); }

// 8 channels as they sit in memory (bf16: one 16-B vector), unpacked a channel pair at a time where they are used
template <typename T> struct Raw8;
template <> struct Raw8<float> { float v[8]; };
template <> struct Raw8<bf16_t> { uint32_t w[4]; };
__device__ __forceinline__ void ldraw8(const float* p, Raw8<float>& r) { load4(p, r.v); load4(p + 4, r.v + 4); }
__device__ __forceinline__ void ldraw8(const bf16_t* p, Raw8<bf16_t>& r) {
    const uint4 t = *reinterpret_cast<const uint4*>(p);
    r.w[0] = t.x; r.w[1] = t.y; r.w[2] = t.z; r.w[3] = t.w;
}
__device__ __forceinline__ void zero8(Raw8<float>& r) {
#pragma unroll
    for (int k = 0; k < 8; ++k) r.v[k] = 0.f;
}
__device__ __forceinline__ void zero8(Raw8<bf16_t>& r) { r.w[0] = r.w[1] = r.w[2] = r.w[3] = 0u; }
__device__ __forceinline__ void pair_of(const Raw8<float>& r, int p2, float o[2]) { o[0] = r.v[2 * p2]; o[1] = r.v[2 * p2 + 1]; }
__device__ __forceinline__ void pair_of(const Raw8<bf16_t>& r, int p2, float o[2]) {
    o[0] = __uint_as_float(r.w[p2] << 16); o[1] = __uint_as_float(r.w[p2] & 0xffff0000u);
}

// y[n][h][w][c] = bias[c] + sum_{kh,kw} x[n][h+kh-3][w+kw-3][c] * wt[kh][kw][c]     (flip: taps mirrored, no bias: data gradient)
// A thread owns WB = 4 consecutive output pixels of a row for 8 channels: per kernel row it loads the 10 input vectors the four
// windows share (instead of 4 x 7) and the 7 tap vectors once -- 2.8x fewer loads per output than one pixel per thread.
constexpr int WB = 4;
template <typename T>
__global__ __launch_bounds__(256, 2) void dwconv7_kernel(const T* __restrict__ x, const T* __restrict__ wt, const float* __restrict__ bias,
                                                       T* __restrict__ y, int N, int H, int W, int C, int flip) {
    const int c8 = C >> 3, wblocks = (W + WB - 1) / WB;
    const long total = (long)N * H * wblocks * c8;
    // workgroup b runs on XCD b % 8, each with its own L2: give every XCD a contiguous band of rows, so that the 7 input rows an
    // output row needs are re-read from THAT L2 and not fetched from HBM by eight different ones (gridDim.x is a multiple of 8)
    const long lb = (long)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const long stride = (long)gridDim.x * blockDim.x;
    // One kernel row of one output quad = 10 input + 7 tap vectors feeding 224 FMAs.  The vectors of the NEXT kernel row (of the
    // next quad after row 6) are requested, raw, before the current row is multiplied; the current row is unpacked a channel pair
    // at a time, so that two rows of operands and the 32 accumulators fit the registers of two waves per SIMD.
    struct Pos { int cc, w0, h0, n; bool valid; };
    struct Row {
        Raw8<T> x[WB + 6];
        const T* taps;
        bool ok;
    };
    auto decode = [&](long i) {
        Pos p;
        p.valid = i < total;
        const long ii = p.valid ? i : 0;
        p.cc = (int)(ii % c8) * 8;
        long r = ii / c8;
        p.w0 = (int)(r % wblocks) * WB; r /= wblocks;
        p.h0 = (int)(r % H); p.n = (int)(r / H);
        return p;
    };
    auto fetch = [&](const Pos& p, int kh, Row& R) {
        const int h = p.h0 + kh - 3;
        R.ok = p.valid && h >= 0 && h < H;
        if (!R.ok) return;
        const T* xrow = x + (((long)p.n * H + h) * W) * C + p.cc;
        if (p.w0 >= 3 && p.w0 + WB + 3 <= W) {               // interior of the row: no per-vector bounds selects
#pragma unroll
            for (int t = 0; t < WB + 6; ++t) ldraw8(xrow + (long)(p.w0 + t - 3) * C, R.x[t]);
        } else {
#pragma unroll
            for (int t = 0; t < WB + 6; ++t) {
                const int w = p.w0 + t - 3;
                if (w >= 0 && w < W) ldraw8(xrow + (long)w * C, R.x[t]);
                else zero8(R.x[t]);
            }
        }
        R.taps = wt + (long)(flip ? (6 - kh) * 7 + 6 : kh * 7) * C + p.cc;
    };
    const long wstep = flip ? -(long)C : (long)C;
    long i = lb * blockDim.x + threadIdx.x;
    Pos pc = decode(i), pn;
    Row A, B;                                                 // ping-pong: even kernel rows in A, odd ones in B (no register copies)
    fetch(pc, 0, A);
    for (; i < total; i += stride) {
        pn = decode(i + stride);
        float acc[WB][8], b8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (bias && !flip) { load4(bias + pc.cc, b8); load4(bias + pc.cc + 4, b8 + 4); }
#pragma unroll
        for (int j = 0; j < WB; ++j)
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[j][k] = b8[k];
        auto multiply = [&](const Row& cur) {
            if (!cur.ok) return;
            Raw8<T> tw[7];                                   // the 7 tap vectors of this kernel row: L1-resident, 7 loads in flight
#pragma unroll
            for (int kw = 0; kw < 7; ++kw) ldraw8(cur.taps + kw * wstep, tw[kw]);
#pragma unroll
            for (int p2 = 0; p2 < 4; ++p2) {
                float xv[WB + 6][2], wv[7][2];
#pragma unroll
                for (int t = 0; t < WB + 6; ++t) pair_of(cur.x[t], p2, xv[t]);
#pragma unroll
                for (int kw = 0; kw < 7; ++kw) pair_of(tw[kw], p2, wv[kw]);
#pragma unroll
                for (int kw = 0; kw < 7; ++kw)
#pragma unroll
                    for (int j = 0; j < WB; ++j) {           // explicit FMAs on a channel pair: v_pk_fma_f32
                        acc[j][2 * p2] = fmaf(xv[j + kw][0], wv[kw][0], acc[j][2 * p2]);
                        acc[j][2 * p2 + 1] = fmaf(xv[j + kw][1], wv[kw][1], acc[j][2 * p2 + 1]);
                    }
            }
        };
#pragma unroll 1
        for (int t = 0; t < 4; ++t) {                        // rows 2t (A) and 2t+1 (B); "row 7" is empty
            if (t < 3) fetch(pc, 2 * t + 1, B);
            else B.ok = false;
            multiply(A);
            if (t < 3) fetch(pc, 2 * t + 2, A);
            else fetch(pn, 0, A);
            multiply(B);
        }
#pragma unroll
        for (int j = 0; j < WB; ++j)
            if (pc.w0 + j < W) store8(y + (((long)pc.n * H + pc.h0) * W + pc.w0 + j) * C + pc.cc, acc[j]);
        pc = pn;
    }
}

// dw[kh][kw][c] += sum_{n,h,w} x[n][h+kh-3][w+kw-3][c] * g[n][h][w][c].  One block = (chunk of pixel quads, up to 256 channels, kernel row
// kh): lanes run along the channels (CG channel groups of 8: one coalesced 16-B load per lane), 256 / CG pixel lanes walk the chunk in
// quads of WB = 4 consecutive pixels of a row, so the four windows share their 10 input vectors; each thread keeps the 7 taps of its
// kernel row for 8 channels, the pixel lanes meet in LDS, one atomic per (tap, channel) per block.
template <typename T>
__global__ __launch_bounds__(256) void dwconv7_wgrad_kernel(const T* __restrict__ x, const T* __restrict__ g, float* __restrict__ dw, int N, int H,
                                                             int W, int C, int quads_per_block, int CG) {
    __shared__ float red[256 * 8];
    const int c8 = C >> 3, ngrp = (c8 + CG - 1) / CG, wblocks = (W + WB - 1) / WB;
    // 1-D grid of 8 * ceil(chunks / 8) * (7 * ngrp) blocks: XCD b % 8 owns the chunks congruent to it and walks the 7 * ngrp (kernel row,
    // channel group) blocks of one chunk back to back, so g and the 7 input rows of that chunk are served by ONE L2 instead of
    // being fetched from HBM once per kernel row
    const int per = 7 * ngrp, xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
    const int chunk = (seq / per) * 8 + xcd, by = seq % per;
    const int kh = by / ngrp, cg = (by - kh * ngrp) * CG + threadIdx.x % CG, pl = threadIdx.x / CG, npl = 256 / CG;
    const bool live = cg < c8;
    const int cc = cg * 8;
    const long nquads = (long)N * H * wblocks, q0 = (long)chunk * quads_per_block, q1 = min(nquads, q0 + quads_per_block);
    float acc[7][8];
#pragma unroll
    for (int kw = 0; kw < 7; ++kw)
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[kw][k] = 0.f;
    // A quad is a load -> 224-FMA chain and the 56 accumulators leave room for two waves per SIMD: without help the kernel runs at
    // the latency of that chain.  The 14 vectors of the NEXT quad are requested (kept raw: 16 B per vector) before the current quad
    // is multiplied, and the current one is unpacked a channel pair at a time (28 live floats instead of 112).
    struct Quad {
        Raw8<T> g[WB], x[WB + 6];
        bool ok;
    };
    auto fetch = [&](long q, Quad& Q) {
        const int w0 = (int)(q % wblocks) * WB, h0 = (int)((q / wblocks) % H), n = (int)(q / ((long)wblocks * H));
        const int h = h0 + kh - 3;
        Q.ok = h >= 0 && h < H;
        if (!Q.ok) return;
        const T* grow = g + (((long)n * H + h0) * W) * C + cc;
        const T* xrow = x + (((long)n * H + h) * W) * C + cc;
        if (w0 >= 3 && w0 + WB + 3 <= W) {                   // interior of the row: no per-vector bounds selects
#pragma unroll
            for (int j = 0; j < WB; ++j) ldraw8(grow + (long)(w0 + j) * C, Q.g[j]);
#pragma unroll
            for (int t = 0; t < WB + 6; ++t) ldraw8(xrow + (long)(w0 + t - 3) * C, Q.x[t]);
        } else {
#pragma unroll
            for (int j = 0; j < WB; ++j) {
                if (w0 + j < W) ldraw8(grow + (long)(w0 + j) * C, Q.g[j]);
                else zero8(Q.g[j]);
            }
#pragma unroll
            for (int t = 0; t < WB + 6; ++t) {
                const int w = w0 + t - 3;
                if (w >= 0 && w < W) ldraw8(xrow + (long)w * C, Q.x[t]);
                else zero8(Q.x[t]);
            }
        }
    };
    auto multiply = [&](const Quad& cur) {
        if (!cur.ok) return;
#pragma unroll
        for (int p2 = 0; p2 < 4; ++p2) {          // channel pair p2 of the lane's 8
            float gv[WB][2], xv[WB + 6][2];
#pragma unroll
            for (int j = 0; j < WB; ++j) pair_of(cur.g[j], p2, gv[j]);
#pragma unroll
            for (int t = 0; t < WB + 6; ++t) pair_of(cur.x[t], p2, xv[t]);
#pragma unroll
            for (int kw = 0; kw < 7; ++kw)
#pragma unroll
                for (int j = 0; j < WB; ++j) {
                    acc[kw][2 * p2] = fmaf(xv[j + kw][0], gv[j][0], acc[kw][2 * p2]);
                    acc[kw][2 * p2 + 1] = fmaf(xv[j + kw][1], gv[j][1], acc[kw][2 * p2 + 1]);
                }
        }
    };
    if (live) {
        Quad A, B;                                // ping-pong buffers: no register copies
        long q = q0 + pl;
        A.ok = false;
        if (q < q1) fetch(q, A);
#pragma unroll 1
        for (; q < q1; q += 2 * npl) {
            B.ok = false;
            if (q + npl < q1) fetch(q + npl, B);
            multiply(A);
            A.ok = false;
            if (q + 2 * npl < q1) fetch(q + 2 * npl, A);
            multiply(B);
        }
    }
    for (int kw = 0; kw < 7; ++kw) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 8; ++k) red[threadIdx.x * 8 + k] = acc[kw][k];
        __syncthreads();
        if (threadIdx.x < CG * 8) {                    // thread t sums channel (t / 8 group, t % 8) over the pixel lanes
            const int grp = threadIdx.x >> 3, k = threadIdx.x & 7;
            float s = 0.f;
            for (int q = 0; q < npl; ++q) s += red[(q * CG + grp) * 8 + k];
            const int cgo = (by - kh * ngrp) * CG + grp;
            if (cgo < c8) atomicAdd(dw + (long)(kh * 7 + kw) * C + cgo * 8 + k, s);
        }
    }
}

// out[r][c] = x[r][c] + s(r) * gamma[c] * y[r][c]
template <typename T>
__global__ void scale_add_kernel(const T* __restrict__ x, const T* __restrict__ y, const float* __restrict__ gamma, const float* __restrict__ scale,
                                 T* __restrict__ out, long rows, int C, int rows_per_sample) {
    const int c8 = C >> 3;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < rows * c8; i += (long)gridDim.x * blockDim.x) {
        const long r = i / c8;
        const int cc = (int)(i - r * c8) * 8;
        float xv[8], yv[8], o[8];
        load8(x + i * 8, xv);
        load8(y + i * 8, yv);
        const float s = scale ? scale[r / rows_per_sample] : 1.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = xv[k] + s * gamma[cc + k] * yv[k];
        store8(out + i * 8, o);
    }
}

// dy[r][c] = s(r) * gamma[c] * g[r][c];  dgamma[c] += sum_r s(r) * g[r][c] * y[r][c].  Lanes run along the channel groups (CG of them,
// coalesced), 256 / CG row lanes walk the block's rows; the row lanes meet in LDS (ds_add_f32), one global atomic per channel.
template <typename T>
__global__ __launch_bounds__(256) void scale_add_bwd_kernel(const T* __restrict__ g, const T* __restrict__ y, const float* __restrict__ gamma,
                                                             const float* __restrict__ scale, T* __restrict__ dy, float* __restrict__ dgamma, long rows,
                                                             int C, int rows_per_sample, int rows_per_block, int CG) {
    __shared__ float dg_s[256 * 8];
    const int c8 = C >> 3, ngrp = (c8 + CG - 1) / CG;
    const int cg = (blockIdx.y % ngrp) * CG + threadIdx.x % CG, rl = threadIdx.x / CG, nrl = 256 / CG;
    const bool live = cg < c8;
    const long r0 = (long)blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, gm[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (live) {
#pragma unroll
        for (int k = 0; k < 8; ++k) gm[k] = gamma[cg * 8 + k];
        auto finish = [&](long r, const Raw8<T>& gr, const Raw8<T>& yr) {
            const float s = scale ? scale[r / rows_per_sample] : 1.f;
            float o[8];
#pragma unroll
            for (int p2 = 0; p2 < 4; ++p2) {
                float gv[2], yv[2];
                pair_of(gr, p2, gv);
                pair_of(yr, p2, yv);
#pragma unroll
                for (int e = 0; e < 2; ++e) { o[2 * p2 + e] = s * gm[2 * p2 + e] * gv[e]; acc[2 * p2 + e] += s * gv[e] * yv[e]; }
            }
            store8(dy + (r * c8 + cg) * 8, o);
        };
        // a streaming read-modify-write with one row per lane and step runs at the latency of its loads: four rows (8 vectors) in flight
        long r = r0 + rl;
        for (; r + 3 * nrl < r1; r += 4 * nrl) {
            Raw8<T> gr[4], yr[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                ldraw8(g + ((r + u * nrl) * c8 + cg) * 8, gr[u]);
                ldraw8(y + ((r + u * nrl) * c8 + cg) * 8, yr[u]);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) finish(r + u * nrl, gr[u], yr[u]);
        }
        for (; r < r1; r += nrl) {
            Raw8<T> gr, yr;
            ldraw8(g + (r * c8 + cg) * 8, gr);
            ldraw8(y + (r * c8 + cg) * 8, yr);
            finish(r, gr, yr);
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) dg_s[threadIdx.x * 8 + k] = acc[k];
    __syncthreads();
    if (threadIdx.x < CG * 8) {
        const int grp = threadIdx.x >> 3, k = threadIdx.x & 7;
        float s = 0.f;
        for (int q = 0; q < nrl; ++q) s += dg_s[(q * CG + grp) * 8 + k];
        const int cgo = (blockIdx.y % ngrp) * CG + grp;
        if (cgo < c8) atomicAdd(dgamma + cgo * 8 + k, s);
    }
}

inline int grid_for(long work, int block = 256) {
    long b = (work + block - 1) / block;
    return (int)(b < 1 ? 1 : (b > 65535 * 8 ? 65535 * 8 : b));
}

}  // namespace

#define CNX_DISPATCH(dtype, f32, bf16)                                              \
    do {                                                                            \
        if ((dtype) == ALDI_F32) { f32; }                                           \
        else if ((dtype) == ALDI_BF16) { bf16; }                                    \
        else return aldi_set_error_msg(ALDI_ERR_ARG, "convnext: bad dtype");       \
    } while (0)

extern "C" int aldi_dwconv7(const void* x, const void* wt, const float* bias, void* y, int N, int H, int W, int C, int flip, int dtype,
                            aldi_stream_t stream) {
    if (!x || !wt || !y || C % 8 || N <= 0 || H <= 0 || W <= 0) return aldi_set_error_msg(ALDI_ERR_ARG, "dwconv7: bad args (C % 8 == 0)");
    hipStream_t st = (hipStream_t)stream;
    const long work = (long)N * H * ((W + 3) / 4) * (C / 8);
    const int nblk = (grid_for(work) + 7) / 8 * 8;
    CNX_DISPATCH(dtype,
        hipLaunchKernelGGL(dwconv7_kernel<float>, dim3(nblk), dim3(256), 0, st, (const float*)x, (const float*)wt, bias, (float*)y, N, H, W, C, flip),
        hipLaunchKernelGGL(dwconv7_kernel<bf16_t>, dim3(nblk), dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)wt, bias, (bf16_t*)y, N, H, W, C, flip));
    ALDI_CHECK_LAUNCH();
    return ALDI_OK;
}

extern "C" int aldi_dwconv7_wgrad(const void* x, const void* g, float* dw, int N, int H, int W, int C, int dtype, aldi_stream_t stream) {
    if (!x || !g || !dw || C % 8 || N <= 0 || H <= 0 || W <= 0) return aldi_set_error_msg(ALDI_ERR_ARG, "dwconv7_wgrad: bad args (C % 8 == 0)");
    hipStream_t st = (hipStream_t)stream;
    const int c8 = C / 8, CG = c8 >= 32 ? 32 : (c8 >= 16 ? 16 : (c8 >= 8 ? 8 : (c8 >= 4 ? 4 : (c8 >= 2 ? 2 : 1))));   // channel groups per block
    const int ngrp = (c8 + CG - 1) / CG;
    // ~1024 blocks over the chip, at least 64 quads (256 pixels) each: the per-block LDS reduction and atomics stay a small fraction
    const long nquads = (long)N * H * ((W + 3) / 4);
    long chunks = 1024 / (7 * ngrp) + 1;
    if (chunks > nquads / 64 + 1) chunks = nquads / 64 + 1;
    const int ppb = (int)((nquads + chunks - 1) / chunks);
    const int nchunks8 = (cdiv(nquads, ppb) + 7) / 8;
    dim3 grid(8 * nchunks8 * (ngrp * 7));
    CNX_DISPATCH(dtype,
        hipLaunchKernelGGL(dwconv7_wgrad_kernel<float>, grid, dim3(256), 0, st, (const float*)x, (const float*)g, dw, N, H, W, C, ppb, CG),
        hipLaunchKernelGGL(dwconv7_wgrad_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)g, dw, N, H, W, C, ppb, CG));
    ALDI_CHECK_LAUNCH();
    return ALDI_OK;
}

extern "C" int aldi_scale_add(const void* x, const void* y, const float* gamma, const float* scale, void* out, long rows, int C, int rows_per_sample,
                              int dtype, aldi_stream_t stream) {
    if (!x || !y || !gamma || !out || C % 8 || rows <= 0 || rows_per_sample <= 0) return aldi_set_error_msg(ALDI_ERR_ARG, "scale_add: bad args");
    hipStream_t st = (hipStream_t)stream;
    CNX_DISPATCH(dtype,
        hipLaunchKernelGGL(scale_add_kernel<float>, dim3(grid_for(rows * (C / 8))), dim3(256), 0, st, (const float*)x, (const float*)y, gamma, scale, (float*)out, rows, C, rows_per_sample),
        hipLaunchKernelGGL(scale_add_kernel<bf16_t>, dim3(grid_for(rows * (C / 8))), dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)y, gamma, scale, (bf16_t*)out, rows, C, rows_per_sample));
    ALDI_CHECK_LAUNCH();
    return ALDI_OK;
}

extern "C" int aldi_scale_add_backward(const void* g, const void* y, const float* gamma, const float* scale, void* dy, float* dgamma, long rows, int C,
                                       int rows_per_sample, int dtype, aldi_stream_t stream) {
    if (!g || !y || !gamma || !dy || !dgamma || C % 8 || rows <= 0 || rows_per_sample <= 0) return aldi_set_error_msg(ALDI_ERR_ARG, "scale_add_backward: bad args");
    hipStream_t st = (hipStream_t)stream;
    const int c8 = C / 8, CG = c8 >= 32 ? 32 : (c8 >= 16 ? 16 : (c8 >= 8 ? 8 : (c8 >= 4 ? 4 : (c8 >= 2 ? 2 : 1))));
    const int ngrp = (c8 + CG - 1) / CG;
    // every workgroup ends with CG * 8 atomics onto the same C addresses, which serialise in L2 (as in the bias-gradient sum): few,
    // long workgroups with four rows in flight per lane instead of ~2000 short ones
    const int target_blocks = aldi_tuning().sab_blocks;
    long chunks = target_blocks / ngrp + 1;
    if (chunks > rows / 64 + 1) chunks = rows / 64 + 1;
    const int rpb = (int)((rows + chunks - 1) / chunks);
    dim3 grid(cdiv(rows, rpb), ngrp);
    CNX_DISPATCH(dtype,
        hipLaunchKernelGGL(scale_add_bwd_kernel<float>, grid, dim3(256), 0, st, (const float*)g, (const float*)y, gamma, scale, (float*)dy, dgamma, rows, C, rows_per_sample, rpb, CG),
        hipLaunchKernelGGL(scale_add_bwd_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)g, (const bf16_t*)y, gamma, scale, (bf16_t*)dy, dgamma, rows, C, rows_per_sample, rpb, CG));
    ALDI_CHECK_LAUNCH();
    return ALDI_OK;
}
