// Token-wise kernels of the ViTDet trunk (gfx950): LayerNorm fwd/bwd with window (un)partition folded into the row map,
// exact GELU fwd/bwd, row gather-add (residuals, drop-path scale, window un-partition), patch extraction, table / grid
// resampling (relative and absolute position embeddings) and AdamW.  All of them are HBM-bound: 16-B accesses, one pass.
//
// Reference semantics: detectron2 `modeling/backbone/vit.py` (Block / ViT), `backbone/utils.py` (window_partition,
// get_rel_pos, get_abs_pos) as driven by aldi/backbone.py:21-43; optimizer aldi/backbone.py:66-84 (AdamW, D2 common/optim.py).
#include "common.h"

namespace {

// VW (4 or 8) consecutive elements <-> fp32: bf16 rows move as one 8-B or 16-B vector per lane
template <int VW> __device__ __forceinline__ void loadv(const float* p, float* v) {
#pragma unroll
    for (int i = 0; i < VW; i += 4) load4(p + i, v + i);
}
template <int VW> __device__ __forceinline__ void loadv(const bf16_t* p, float* v) {
    if constexpr (VW == 8) {
        const uint4 t = *reinterpret_cast<const uint4*>(p);
        const uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) { v[2 * k] = __uint_as_float(w[k] << 16); v[2 * k + 1] = __uint_as_float(w[k] & 0xffff0000u); }
    } else {
        load4(p, v);
    }
}
template <int VW> __device__ __forceinline__ void storev(float* p, const float* v) {
#pragma unroll
    for (int i = 0; i < VW; i += 4) store4(p + i, v + i);
}
template <int VW> __device__ __forceinline__ void storev(bf16_t* p, const float* v) {
    if constexpr (VW == 8) {
        uint4 t;
        t.x = pack2_bf16(v[0], v[1]); t.y = pack2_bf16(v[2], v[3]); t.z = pack2_bf16(v[4], v[5]); t.w = pack2_bf16(v[6], v[7]);
        *reinterpret_cast<uint4*>(p) = t;
    } else {
        store4(p, v);
    }
}

// ------------------------------------------------------------------------------------------------ LayerNorm
// One wave per output row.  `map` (nullable) gathers: output row r normalises source row map[r]; map[r] < 0 is a padding
// row of a partitioned window and is written as zeros (detectron2 pads AFTER norm1, so the padded tokens are exact zeros).
// C <= 2048 (ConvNeXt-L stage 3 is 1536 wide), C % 4 == 0; instantiated for 4 chunks (C <= 1024: half the registers) and 8; lane chunk j (4 channels at (lane + 64 j) * 4) is live iff inside C

template <typename T, int LN_MAXCH, int VW>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* __restrict__ x, const int* __restrict__ map, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, T* __restrict__ y, float* __restrict__ mean,
                                                      float* __restrict__ rstd, int rows, int C, float eps, int relu) {
    const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const long src = map ? map[r] : r;
    T* yr = y + (long)r * C;
    if (src < 0) {
        float z[VW];
#pragma unroll
        for (int i = 0; i < VW; ++i) z[i] = 0.f;
        for (int j = 0; j < LN_MAXCH; ++j) if ((lane + 64 * j) * VW < C) storev<VW>(yr + (lane + 64 * j) * VW, z);
        if (lane == 0) { mean[r] = 0.f; rstd[r] = 0.f; }
        return;
    }
    const T* xr = x + src * C;
    float v[LN_MAXCH][VW];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < LN_MAXCH; ++j)
        if ((lane + 64 * j) * VW < C) {
            loadv<VW>(xr + (lane + 64 * j) * VW, v[j]);
            {
#pragma unroll
            for (int i = 0; i < VW; i += 4) s += (v[j][i] + v[j][i + 1]) + (v[j][i + 2] + v[j][i + 3]);
        }
        }
    const float mu = warp_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < LN_MAXCH; ++j)
        if ((lane + 64 * j) * VW < C) {
#pragma unroll
            for (int i = 0; i < VW; ++i) { v[j][i] -= mu; q += v[j][i] * v[j][i]; }
        }
    const float rs = rsqrtf(warp_sum(q) / (float)C + eps);
#pragma unroll
    for (int j = 0; j < LN_MAXCH; ++j)
        if ((lane + 64 * j) * VW < C) {
            float gm[VW], bt[VW], o[VW];
            loadv<VW>(gamma + (lane + 64 * j) * VW, gm);
            loadv<VW>(beta + (lane + 64 * j) * VW, bt);
#pragma unroll
            for (int i = 0; i < VW; ++i) {
                o[i] = v[j][i] * rs * gm[i] + bt[i];
                if (relu) o[i] = fmaxf(o[i], 0.f);
            }
            storev<VW>(yr + (lane + 64 * j) * VW, o);
        }
    if (lane == 0) { mean[r] = mu; rstd[r] = rs; }
}

// a VW-element run (4 or 8) as it sits in memory (bf16: 32-bit words holding two elements), converted to fp32 only when it is used
template <typename T, int VW> struct RawV;
template <int VW> struct RawV<float, VW> { float v[VW]; };
template <int VW> struct RawV<bf16_t, VW> { uint32_t w[VW / 2]; };
template <int VW> __device__ __forceinline__ void ldrawv(const float* p, RawV<float, VW>& r) {
#pragma unroll
    for (int i = 0; i < VW; i += 4) load4(p + i, r.v + i);
}
__device__ __forceinline__ void ldrawv(const bf16_t* p, RawV<bf16_t, 4>& r) {
    const uint2 t = *reinterpret_cast<const uint2*>(p);
    r.w[0] = t.x; r.w[1] = t.y;
}
__device__ __forceinline__ void ldrawv(const bf16_t* p, RawV<bf16_t, 8>& r) {
    const uint4 t = *reinterpret_cast<const uint4*>(p);
    r.w[0] = t.x; r.w[1] = t.y; r.w[2] = t.z; r.w[3] = t.w;
}
template <int VW> __device__ __forceinline__ void cvtv(const RawV<float, VW>& r, float* o) {
#pragma unroll
    for (int i = 0; i < VW; ++i) o[i] = r.v[i];
}
template <int VW> __device__ __forceinline__ void cvtv(const RawV<bf16_t, VW>& r, float* o) {
#pragma unroll
    for (int i = 0; i < VW / 2; ++i) { o[2 * i] = __uint_as_float(r.w[i] << 16); o[2 * i + 1] = __uint_as_float(r.w[i] & 0xffff0000u); }
}

// dx[src] = rstd * (g*gamma - mean(g*gamma) - xhat * mean(g*gamma*xhat)) (+ res[src]);  dgamma += g*xhat, dbeta += g
template <typename T, int LN_MAXCH, int VW>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const T* __restrict__ g, const T* __restrict__ x, const int* __restrict__ map,
                                                      const float* __restrict__ gamma, const float* __restrict__ mean,
                                                      const float* __restrict__ rstd, const T* __restrict__ res, const T* __restrict__ mask,
                                                      T* __restrict__ dx, float* __restrict__ dgamma, float* __restrict__ dbeta, int rows, int C, int rows_per_block) {
    __shared__ float red[4 * LN_MAXCH * 64 * VW];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float dg[LN_MAXCH][VW], db[LN_MAXCH][VW], gm[LN_MAXCH][VW];
#pragma unroll
    for (int j = 0; j < LN_MAXCH; ++j) {
#pragma unroll
        for (int i = 0; i < VW; ++i) { dg[j][i] = 0.f; db[j][i] = 0.f; gm[j][i] = 0.f; }
        if ((lane + 64 * j) * VW < C) loadv<VW>(gamma + (lane + 64 * j) * VW, gm[j]);
    }
    const int r_end = min(rows, (int)(blockIdx.x + 1) * rows_per_block);
    // A wave walks its rows one at a time, and a row is a load -> reduce -> store chain: with nothing else in flight the kernel
    // runs at the latency of that chain, not at HBM rate.  The operands of the NEXT row (kept as raw 16-bit pairs: half the
    // registers) are requested before the current row is reduced; the residual comes with them instead of after the reduction.
    constexpr bool kPrefetch = LN_MAXCH * VW <= 16 && sizeof(T) == 2;
    struct RowIn {
        RawV<T, VW> g[LN_MAXCH], x[LN_MAXCH], m[LN_MAXCH], rv[LN_MAXCH];
        float mu, rs;
        long src;
    };
    auto fetch = [&](int r, RowIn& in) {
        in.src = map ? map[r] : r;
        if (in.src < 0) return;
        in.mu = mean[r];
        in.rs = rstd[r];
#pragma unroll
        for (int j = 0; j < LN_MAXCH; ++j)
            if ((lane + 64 * j) * VW < C) {
                const int off = (lane + 64 * j) * VW;
                ldrawv(g + (long)r * C + off, in.g[j]);
                if (mask) ldrawv(mask + (long)r * C + off, in.m[j]);
                ldrawv(x + in.src * C + off, in.x[j]);
                if (res) ldrawv(res + in.src * C + off, in.rv[j]);
            }
    };
    RowIn cur, nxt;
    int r = blockIdx.x * rows_per_block + wave;
    if (kPrefetch && r < r_end) fetch(r, cur);
    for (; r < r_end; r += 4) {
        if constexpr (kPrefetch) {
            nxt.src = -1;
            if (r + 4 < r_end) fetch(r + 4, nxt);
        } else {
            fetch(r, cur);
        }
        if (cur.src >= 0) {
            const long src = cur.src;
            const float mu = cur.mu, rs = cur.rs;
            float gv[LN_MAXCH][VW], xh[LN_MAXCH][VW];
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int j = 0; j < LN_MAXCH; ++j)
                if ((lane + 64 * j) * VW < C) {
                    cvtv(cur.g[j], gv[j]);
                    if (mask) {      // ReLU after the norm: the gradient passes where the activation was positive
                        float mv[VW];
                        cvtv(cur.m[j], mv);
#pragma unroll
                        for (int i = 0; i < VW; ++i) gv[j][i] = mv[i] > 0.f ? gv[j][i] : 0.f;
                    }
                    cvtv(cur.x[j], xh[j]);
#pragma unroll
                    for (int i = 0; i < VW; ++i) {
                        xh[j][i] = (xh[j][i] - mu) * rs;
                        dg[j][i] += gv[j][i] * xh[j][i];
                        db[j][i] += gv[j][i];
                        gv[j][i] *= gm[j][i];
                        s1 += gv[j][i];
                        s2 += gv[j][i] * xh[j][i];
                    }
                }
            s1 = warp_sum(s1) / (float)C;
            s2 = warp_sum(s2) / (float)C;
#pragma unroll
            for (int j = 0; j < LN_MAXCH; ++j)
                if ((lane + 64 * j) * VW < C) {
                    float o[VW], rv[VW];
#pragma unroll
                    for (int i = 0; i < VW; ++i) rv[i] = 0.f;
                    if (res) cvtv(cur.rv[j], rv);
#pragma unroll
                    for (int i = 0; i < VW; ++i) o[i] = rs * (gv[j][i] - s1 - xh[j][i] * s2) + rv[i];
                    storev<VW>(dx + src * C + (lane + 64 * j) * VW, o);
                }
        }
        if constexpr (kPrefetch) cur = nxt;
    }
    // block reduction of the parameter gradients, then one atomic per column
    for (int pass = 0; pass < 2; ++pass) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < LN_MAXCH; ++j)
            if ((lane + 64 * j) * VW < C) {
#pragma unroll
                for (int i = 0; i < VW; ++i) red[wave * (LN_MAXCH * 64 * VW) + (lane + 64 * j) * VW + i] = pass ? db[j][i] : dg[j][i];
            }
        __syncthreads();
        for (int cidx = threadIdx.x; cidx < C; cidx += 256) {
            constexpr int RS = LN_MAXCH * 64 * VW;
            const float t = (red[cidx] + red[RS + cidx]) + (red[2 * RS + cidx] + red[3 * RS + cidx]);
            atomicAdd((pass ? dbeta : dgamma) + cidx, t);
        }
    }
}

// ------------------------------------------------------------------------------------------------ GELU (erf form)
// Phi(x) and exp(-x^2/2) from ONE exponential: 0.5 erfc(|z|) = 0.5 poly(t) exp(-z^2), z = |x|/sqrt2, t = 1/(1 + p z) (Abramowitz-Stegun
// 7.1.26, |error| <= 1.5e-7 absolute in erf -- below fp32 test tolerance, far below bf16), no cancellation in the negative tail.
// libm's erff costs ~3x the instructions and made this kernel VALU-bound (52 us where HBM allows 26).
__device__ __forceinline__ void gelu_terms(float x, float& cdf, float& e) {
    const float z = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.f));
    e = __expf(-z * z);
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float half_erfc = 0.5f * p * t * e;
    cdf = x >= 0.f ? 1.f - half_erfc : half_erfc;
}

template <typename T>
__global__ void gelu_kernel(const T* __restrict__ x, const T* __restrict__ g, T* __restrict__ out, long n4) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        float v[4], gv[4], o[4];
        load4(x + i * 4, v);
        if (g) load4(g + i * 4, gv);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float cdf, e;
            gelu_terms(v[k], cdf, e);
            o[k] = g ? gv[k] * fmaf(v[k] * 0.3989422804014327f, e, cdf) : v[k] * cdf;
        }
        store4(out + i * 4, o);
    }
}

// bf16, n % 8 == 0, 16-B aligned: 8 elements (one 16-B vector) per lane and step
__global__ void gelu8_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ g, bf16_t* __restrict__ out, long n8) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
        const uint4 xv = *reinterpret_cast<const uint4*>(x + i * 8);
        uint4 gq = make_uint4(0u, 0u, 0u, 0u);
        if (g) gq = *reinterpret_cast<const uint4*>(g + i * 8);
        const uint32_t xw[4] = {xv.x, xv.y, xv.z, xv.w}, gw[4] = {gq.x, gq.y, gq.z, gq.w};
        uint32_t ow[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float v0 = __uint_as_float(xw[k] << 16), v1 = __uint_as_float(xw[k] & 0xffff0000u);
            float c0, e0, c1, e1, o0, o1;
            gelu_terms(v0, c0, e0);
            gelu_terms(v1, c1, e1);
            if (g) {
                o0 = __uint_as_float(gw[k] << 16) * fmaf(v0 * 0.3989422804014327f, e0, c0);
                o1 = __uint_as_float(gw[k] & 0xffff0000u) * fmaf(v1 * 0.3989422804014327f, e1, c1);
            } else {
                o0 = v0 * c0; o1 = v1 * c1;
            }
            ow[k] = pack2_bf16(o0, o1);
        }
        *reinterpret_cast<uint4*>(out + i * 8) = make_uint4(ow[0], ow[1], ow[2], ow[3]);
    }
}

// ------------------------------------------------------------------------------------------------ row gather-add
// out[r] = (a ? a[r] : 0) + s(r) * (idx >= 0 ? b[idx] : 0), idx = map ? map[r] : r, s(r) = scale ? scale[r / rows_per_sample] : 1
template <typename T>
__global__ void rows_add_kernel(const T* __restrict__ a, const T* __restrict__ b, const int* __restrict__ map, const float* __restrict__ scale,
                                T* __restrict__ out, int rows, int C, int rows_per_sample) {
    const int c4 = C >> 2;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < (long)rows * c4; i += (long)gridDim.x * blockDim.x) {
        const int r = (int)(i / c4), cc = (int)(i - (long)r * c4) * 4;
        const long idx = map ? map[r] : r;
        float av[4] = {0.f, 0.f, 0.f, 0.f}, bv[4] = {0.f, 0.f, 0.f, 0.f}, o[4];
        if (a) load4(a + (long)r * C + cc, av);
        if (idx >= 0) load4(b + idx * C + cc, bv);
        const float s = scale ? scale[r / rows_per_sample] : 1.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = av[k] + s * bv[k];
        store4(out + (long)r * C + cc, o);
    }
}

// bf16, C % 8 == 0, 16-B aligned rows: one 16-B vector per lane and step
__global__ void rows_add8_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b, const int* __restrict__ map, const float* __restrict__ scale,
                                 bf16_t* __restrict__ out, int rows, int C, int rows_per_sample) {
    const int c8 = C >> 3;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < (long)rows * c8; i += (long)gridDim.x * blockDim.x) {
        const int r = (int)(i / c8), cc = (int)(i - (long)r * c8) * 8;
        const long idx = map ? map[r] : r;
        uint4 av = make_uint4(0u, 0u, 0u, 0u), bv = make_uint4(0u, 0u, 0u, 0u);
        if (a) av = *reinterpret_cast<const uint4*>(a + (long)r * C + cc);
        if (idx >= 0) bv = *reinterpret_cast<const uint4*>(b + idx * C + cc);
        const float s = scale ? scale[r / rows_per_sample] : 1.f;
        const uint32_t aw[4] = {av.x, av.y, av.z, av.w}, bw[4] = {bv.x, bv.y, bv.z, bv.w};
        uint32_t ow[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            ow[k] = pack2_bf16(__uint_as_float(aw[k] << 16) + s * __uint_as_float(bw[k] << 16),
                               __uint_as_float(aw[k] & 0xffff0000u) + s * __uint_as_float(bw[k] & 0xffff0000u));
        *reinterpret_cast<uint4*>(out + (long)r * C + cc) = make_uint4(ow[0], ow[1], ow[2], ow[3]);
    }
}

// ------------------------------------------------------------------------------------------------ patch extraction
// uint8 CHW staging -> normalised bf16/fp32 rows [N*gh*gw][3*P*P] in the (c, ph, pw) order of a conv weight [Cout][3][P][P];
// pixels outside image n (the zero padding of ImageList) contribute (0 - mean)/std * 0 = 0, as in the reference where the
// batch is padded AFTER normalisation.
template <typename T>
__global__ void patchify_kernel(const uint8_t* __restrict__ img, T* __restrict__ out, int N, int Hs, int Ws, int P, int gh, int gw,
                                const int* __restrict__ hw, float m0, float m1, float m2, float is0, float is1, float is2) {
    const int K = 3 * P * P;
    const long total = (long)N * gh * gw * (K >> 2);
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int k = (int)(i % (K >> 2)) * 4;
        const long tok = i / (K >> 2);
        const int n = (int)(tok / (gh * gw)), t = (int)(tok - (long)n * gh * gw), ty = t / gw, tx = t - ty * gw;
        const int c = k / (P * P), rem = k - c * P * P, py = rem / P, px = rem - py * P;
        const int y = ty * P + py, x0 = tx * P + px;
        const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2), is = c == 0 ? is0 : (c == 1 ? is1 : is2);
        const int h = hw[2 * n], w = hw[2 * n + 1];
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int x = x0 + j;
            o[j] = (y < h && x < w) ? ((float)img[(((long)n * 3 + c) * Hs + y) * Ws + x] - mean) * is : 0.f;
        }
        store4(out + tok * K + k, o);
    }
}

// ------------------------------------------------------------------------------------------------ resampling
// 1-D linear resize of a table [L0][C] -> [L1][C] (F.interpolate mode="linear", align_corners=False), fwd and bwd (fp32)
__device__ __forceinline__ void lin_taps(int i, int L0, int L1, int& i0, int& i1, float& lam) {
    const float scale = (float)L0 / (float)L1;
    float src = ((float)i + 0.5f) * scale - 0.5f;
    src = src < 0.f ? 0.f : src;
    i0 = (int)src;
    i1 = i0 + (i0 < L0 - 1 ? 1 : 0);
    lam = src - (float)i0;
}
__global__ void linear_resize_kernel(const float* __restrict__ in, float* __restrict__ out, int L0, int L1, int C, int backward) {
    const int i = blockIdx.x;
    int i0, i1; float lam;
    lin_taps(i, L0, L1, i0, i1, lam);
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        if (!backward) out[i * C + c] = (1.f - lam) * in[i0 * C + c] + lam * in[i1 * C + c];
        else {   // in = grad wrt resized [L1][C], out = grad wrt table [L0][C] (pre-zeroed or accumulating)
            const float gv = in[i * C + c];
            atomicAdd(out + i0 * C + c, (1.f - lam) * gv);
            atomicAdd(out + i1 * C + c, lam * gv);
        }
    }
}

// bicubic (A = -0.75, align_corners=False, border indices clamped) resize of a grid [S0][S0][C] -> [gh][gw][C]
__device__ __forceinline__ void cubic_w(float t, float w[4]) {
    const float A = -0.75f;
    const float x0 = t + 1.f, x1 = t, x2 = 1.f - t, x3 = 2.f - t;
    w[0] = ((A * x0 - 5.f * A) * x0 + 8.f * A) * x0 - 4.f * A;
    w[1] = ((A + 2.f) * x1 - (A + 3.f)) * x1 * x1 + 1.f;
    w[2] = ((A + 2.f) * x2 - (A + 3.f)) * x2 * x2 + 1.f;
    w[3] = ((A * x3 - 5.f * A) * x3 + 8.f * A) * x3 - 4.f * A;
}
__global__ void bicubic_resize_kernel(const float* __restrict__ in, float* __restrict__ out, int S0h, int S0w, int gh, int gw, int C,
                                      int backward) {
    const int oy = blockIdx.x / gw, ox = blockIdx.x - oy * gw;
    const float sy = ((float)oy + 0.5f) * ((float)S0h / (float)gh) - 0.5f, sx = ((float)ox + 0.5f) * ((float)S0w / (float)gw) - 0.5f;
    const float fy = floorf(sy), fx = floorf(sx);
    float wy[4], wx[4];
    cubic_w(sy - fy, wy);
    cubic_w(sx - fx, wx);
    int iy[4], ix[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        iy[k] = min(max((int)fy - 1 + k, 0), S0h - 1);
        ix[k] = min(max((int)fx - 1 + k, 0), S0w - 1);
    }
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        if (!backward) {
            float acc = 0.f;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                float row = 0.f;
#pragma unroll
                for (int b = 0; b < 4; ++b) row += wx[b] * in[((long)iy[a] * S0w + ix[b]) * C + c];
                acc += wy[a] * row;
            }
            out[(long)blockIdx.x * C + c] = acc;
        } else {
            const float gv = in[(long)blockIdx.x * C + c];
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) atomicAdd(out + ((long)iy[a] * S0w + ix[b]) * C + c, wy[a] * wx[b] * gv);
        }
    }
}

// y[n][t][:] = x[n][t][:] + pos[t][:]  (pos fp32 [T][C]); backward of pos = sum over n of g
template <typename T>
__global__ void add_pos_kernel(const T* __restrict__ x, const float* __restrict__ pos, T* __restrict__ y, int N, long TC4) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < (long)N * TC4; i += (long)gridDim.x * blockDim.x) {
        float v[4], p[4];
        load4(x + i * 4, v);
        load4(pos + (i % TC4) * 4, p);
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] += p[k];
        store4(y + i * 4, v);
    }
}
template <typename T>
__global__ void sum_batch_kernel(const T* __restrict__ g, float* __restrict__ out, int N, long TC4) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < TC4; i += (long)gridDim.x * blockDim.x) {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int n = 0; n < N; ++n) {
            float v[4];
            load4(g + ((long)n * TC4 + i) * 4, v);
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] += v[k];
        }
        store4(out + i * 4, acc);
    }
}


// ------------------------------------------------------------------------------------------------ 2x2/2 max pool (NHWC)
// forward writes the pooled value and the winning tap (first maximum in row-major window order, torch's tie rule);
// backward routes the gradient to that tap and zeroes the rest.
template <typename T>
__global__ void maxpool2_kernel(const T* __restrict__ x, T* __restrict__ y, uint8_t* __restrict__ idx, int N, int H, int W, int C) {
    const int Ho = H >> 1, Wo = W >> 1, c4 = C >> 2;
    const long total = (long)N * Ho * Wo * c4;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int cc = (int)(i % c4) * 4;
        long pix = i / c4;
        const int wo = (int)(pix % Wo); pix /= Wo;
        const int ho = (int)(pix % Ho), n = (int)(pix / Ho);
        float best[4];
        uint8_t bi[4] = {0, 0, 0, 0};
        load4(x + (((long)n * H + 2 * ho) * W + 2 * wo) * C + cc, best);
#pragma unroll
        for (int t = 1; t < 4; ++t) {
            float v[4];
            load4(x + (((long)n * H + 2 * ho + (t >> 1)) * W + 2 * wo + (t & 1)) * C + cc, v);
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (v[k] > best[k] || v[k] != v[k]) { best[k] = v[k]; bi[k] = (uint8_t)t; }
        }
        store4(y + i * 4, best);
        *reinterpret_cast<uint32_t*>(idx + i * 4) = bi[0] | (bi[1] << 8) | (bi[2] << 16) | ((uint32_t)bi[3] << 24);
    }
}
template <typename T>
__global__ void maxpool2_bwd_kernel(const T* __restrict__ g, const uint8_t* __restrict__ idx, T* __restrict__ dx, int N, int H, int W, int C) {
    const int Ho = H >> 1, Wo = W >> 1, c4 = C >> 2;
    const long total = (long)N * Ho * Wo * c4;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int cc = (int)(i % c4) * 4;
        long pix = i / c4;
        const int wo = (int)(pix % Wo); pix /= Wo;
        const int ho = (int)(pix % Ho), n = (int)(pix / Ho);
        float gv[4];
        load4(g + i * 4, gv);
        const uint32_t b = *reinterpret_cast<const uint32_t*>(idx + i * 4);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float o[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = ((b >> (8 * k)) & 0xffu) == (uint32_t)t ? gv[k] : 0.f;
            store4(dx + (((long)n * H + 2 * ho + (t >> 1)) * W + 2 * wo + (t & 1)) * C + cc, o);
        }
    }
}

// ------------------------------------------------------------------------------------------------ AdamW
// torch.optim.AdamW (decoupled decay, bias-corrected): p *= 1 - lr*wd; m, v moments; p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps)
// lr_scale / wd_mask are per-element-segment factors resolved by the host into per-launch scalars.
template <typename T>
__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                             T* __restrict__ p_compute, long n, float lr, float beta1, float beta2, float eps, float wd, float bc1,
                             float bc2_sqrt, float grad_scale) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float gr = g[i] * grad_scale;
        float pv = p[i] * (1.f - lr * wd);
        const float mv = beta1 * m[i] + (1.f - beta1) * gr;
        const float vv = beta2 * v[i] + (1.f - beta2) * gr * gr;
        m[i] = mv; v[i] = vv;
        const float denom = sqrtf(vv) / bc2_sqrt + eps;
        pv -= (lr / bc1) * (mv / denom);
        p[i] = pv;
        if (p_compute) Elem<T>::st(p_compute + i, pv);
    }
}

inline int grid_for(long work, int block = 256) {
    long b = (work + block - 1) / block;
    return (int)(b < 1 ? 1 : (b > 65535 * 4 ? 65535 * 4 : b));
}

}  // namespace

#define VIT_DISPATCH(dtype, expr_f32, expr_bf16)                                   \
    do {                                                                           \
        if ((dtype) == ALDI_F32) { expr_f32; }                                     \
        else if ((dtype) == ALDI_BF16) { expr_bf16; }                              \
        else return aldi_set_error_msg(ALDI_ERR_ARG, "vit: bad dtype");            \
    } while (0)

extern "C" int aldi_layernorm_forward(const void* x, const int* map, const float* gamma, const float* beta, void* y, float* mean,
                                      float* rstd, int rows, int C, float eps, int relu, int dtype, aldi_stream_t stream) {
    if (C % 4 || C > 2048 || rows <= 0) return aldi_set_error_msg(ALDI_ERR_ARG, "layernorm: C must be a multiple of 4, <= 2048");
    hipStream_t st = (hipStream_t)stream;
    // bf16 rows of whole 16-B vectors: 8 channels per lane chunk (one 16-B access), half the chunks
    const bool v8 = dtype == ALDI_BF16 && C % 8 == 0 && (((uintptr_t)x | (uintptr_t)y) & 15) == 0;
#define LN_FWD(T_, MC, VW_) hipLaunchKernelGGL((ln_fwd_kernel<T_, MC, VW_>), dim3(cdiv(rows, 4)), dim3(256), 0, st, (const T_*)x, map, gamma, beta, (T_*)y, mean, rstd, rows, C, eps, relu)
    if (v8) {
        if (C <= 512) LN_FWD(bf16_t, 1, 8);          // one chunk per lane: half the registers of the 2-chunk form
        else if (C <= 1024) LN_FWD(bf16_t, 2, 8);
        else LN_FWD(bf16_t, 4, 8);
    } else if (C <= 1024) {
        VIT_DISPATCH(dtype, LN_FWD(float, 4, 4), LN_FWD(bf16_t, 4, 4));
    } else {
        VIT_DISPATCH(dtype, LN_FWD(float, 8, 4), LN_FWD(bf16_t, 8, 4));
    }
#undef LN_FWD
    ALDI_CHECK_LAUNCH();
    return ALDI_OK;
}

extern "C" int aldi_layernorm_backward(const void* g, const void* x, const int* map, const float* gamma, const float* mean, const float* rstd,
                                       const void* res, const void* mask, void* dx, float* dgamma, float* dbeta, int rows, int C, int dtype,
                                       aldi_stream_t stream) {
    if (C % 4 || C > 2048 || rows <= 0) return aldi_set_error_msg(ALDI_ERR_ARG, "layernorm: C must be a multiple of 4, <= 2048");
    hipStream_t st = (hipStream_t)stream;
    // one round of workgroups (2 per CU at this register count: 513 would run as two rounds), each ending with 2*C atomics
    // (the single-chunk form for C <= 512 needs half the registers: four workgroups per CU)
    const int target_wide = aldi_tuning().ln_bwd_blocks;
    const int target_narrow = aldi_tuning().ln_bwd_blocks_narrow;
    const bool v8 = dtype == ALDI_BF16 && C % 8 == 0 && (((uintptr_t)g | (uintptr_t)x | (uintptr_t)res | (uintptr_t)mask | (uintptr_t)dx) & 15) == 0;
    const int target_blocks = v8 && C <= 512 ? target_narrow : target_wide;
    const int rpb = cdiv(rows, target_blocks) < 32 ? 32 : cdiv(rows, target_blocks);
#define LN_BWD(T_, MC, VW_) hipLaunchKernelGGL((ln_bwd_kernel<T_, MC, VW_>), dim3(cdiv(rows, rpb)), dim3(256), 0, st, (const T_*)g, (const T_*)x, map, gamma, mean, rstd, \
                                               (const T_*)res, (const T_*)mask, (T_*)dx, dgamma, dbeta, rows, C, rpb)
    if (v8) {
        if (C <= 512) LN_BWD(bf16_t, 1, 8);
        else if (C <= 1024) LN_BWD(bf16_t, 2, 8);
        else LN_BWD(bf16_t, 4, 8);
    } else if (C <= 1024) {
        VIT_DISPATCH(dtype, LN_BWD(float, 4, 4), LN_BWD(bf16_t, 4, 4));
    } else {
        VIT_DISPATCH(dtype, LN_BWD(float, 8, 4), LN_BWD(bf16_t, 8, 4));
    }
#undef LN_BWD
    ALDI_CHECK_LAUNCH();
    return ALDI_OK;
}

extern "C" int aldi_gelu(const void* x, const void* g, void* out, long n, int dtype, aldi_stream_t stream) {
    if (n % 4) return aldi_set_error_msg(ALDI_ERR_ARG, "gelu: n % 4 != 0");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == ALDI_BF16 && n % 8 == 0 && (((uintptr_t)x | (uintptr_t)g | (uintptr_t)out) & 15) == 0) {
        hipLaunchKernelGGL(gelu8_kernel, dim3(grid_for(n / 8)), dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)g, (bf16_t*)out, n / 8);
        ALDI_CHECK_LAUNCH();
        return ALDI_OK;
    }
    VIT_DISPATCH(dtype,
        hipLaunchKernelGGL(gelu_kernel<float>, dim3(grid_for(n / 4)), dim3(256), 0, st, (const float*)x, (const float*)g, (float*)out, n / 4),
        hipLaunchKernelGGL(gelu_kernel<bf16_t>, dim3(grid_for(n / 4)), dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)g, (bf16_t*)out, n / 4));
    ALDI_CHECK_LAUNCH();
    return ALDI_OK;
}

extern "C" int aldi_rows_add(const void* a, const void* b, const int* map, const float* scale, void* out, int rows, int C,
                             int rows_per_sample, int dtype, aldi_stream_t stream) {
    if (C % 4 || rows <= 0 || rows_per_sample <= 0) return aldi_set_error_msg(ALDI_ERR_ARG, "rows_add: bad sizes");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == ALDI_BF16 && C % 8 == 0 && (((uintptr_t)a | (uintptr_t)b | (uintptr_t)out) & 15) == 0) {
        hipLaunchKernelGGL(rows_add8_kernel, dim3(grid_for((long)rows * (C / 8))), dim3(256), 0, st, (const bf16_t*)a, (const bf16_t*)b, map, scale, (bf16_t*)out, rows, C,
                           rows_per_sample);
        ALDI_CHECK_LAUNCH();
        return ALDI_OK;
    }
    const long work = (long)rows * (C / 4);
    VIT_DISPATCH(dtype,
        hipLaunchKernelGGL(rows_add_kernel<float>, dim3(grid_for(work)), dim3(256), 0, st, (const float*)a, (const float*)b, map, scale, (float*)out, rows, C, rows_per_sample),
        hipLaunchKernelGGL(rows_add_kernel<bf16_t>, dim3(grid_for(work)), dim3(256), 0, st, (const bf16_t*)a, (const bf16_t*)b, map, scale, (bf16_t*)out, rows, C, rows_per_sample));
    ALDI_CHECK_LAUNCH();
    return ALDI_OK;
}

extern "C" int aldi_patchify(const uint8_t* img, void* out, int N, int Hs, int Ws, int P, const int* hw, const float* mean, const float* std,
                             int dtype, aldi_stream_t stream) {
    if (P % 4 || Hs % P || Ws % P || N <= 0) return aldi_set_error_msg(ALDI_ERR_ARG, "patchify: bad sizes");
    hipStream_t st = (hipStream_t)stream;
    const int gh = Hs / P, gw = Ws / P;
    const long work = (long)N * gh * gw * (3 * P * P / 4);
    VIT_DISPATCH(dtype,
        hipLaunchKernelGGL(patchify_kernel<float>, dim3(grid_for(work)), dim3(256), 0, st, img, (float*)out, N, Hs, Ws, P, gh, gw, hw, mean[0], mean[1], mean[2], 1.f / std[0], 1.f / std[1], 1.f / std[2]),
        hipLaunchKernelGGL(patchify_kernel<bf16_t>, dim3(grid_for(work)), dim3(256), 0, st, img, (bf16_t*)out, N, Hs, Ws, P, gh, gw, hw, mean[0], mean[1], mean[2], 1.f / std[0], 1.f / std[1], 1.f / std[2]));
    ALDI_CHECK_LAUNCH();
    return ALDI_OK;
}

extern "C" int aldi_linear_resize(const float* in, float* out, int L0, int L1, int C, int backward, aldi_stream_t stream) {
    if (L0 <= 0 || L1 <= 0 || C <= 0) return aldi_set_error_msg(ALDI_ERR_ARG, "linear_resize: bad sizes");
    hipLaunchKernelGGL(linear_resize_kernel, dim3(L1), dim3(64), 0, (hipStream_t)stream, in, out, L0, L1, C, backward);
    ALDI_CHECK_LAUNCH();
    return ALDI_OK;
}

extern "C" int aldi_bicubic_resize(const float* in, float* out, int S0h, int S0w, int gh, int gw, int C, int backward, aldi_stream_t stream) {
    if (S0h <= 0 || S0w <= 0 || gh <= 0 || gw <= 0 || C <= 0) return aldi_set_error_msg(ALDI_ERR_ARG, "bicubic_resize: bad sizes");
    hipLaunchKernelGGL(bicubic_resize_kernel, dim3(gh * gw), dim3(256), 0, (hipStream_t)stream, in, out, S0h, S0w, gh, gw, C, backward);
    ALDI_CHECK_LAUNCH();
    return ALDI_OK;
}

extern "C" int aldi_add_pos(const void* x, const float* pos, void* y, int N, long TC, int dtype, aldi_stream_t stream) {
    if (TC % 4 || N <= 0) return aldi_set_error_msg(ALDI_ERR_ARG, "add_pos: bad sizes");
    hipStream_t st = (hipStream_t)stream;
    VIT_DISPATCH(dtype,
        hipLaunchKernelGGL(add_pos_kernel<float>, dim3(grid_for(N * TC / 4)), dim3(256), 0, st, (const float*)x, pos, (float*)y, N, TC / 4),
        hipLaunchKernelGGL(add_pos_kernel<bf16_t>, dim3(grid_for(N * TC / 4)), dim3(256), 0, st, (const bf16_t*)x, pos, (bf16_t*)y, N, TC / 4));
    ALDI_CHECK_LAUNCH();
    return ALDI_OK;
}

extern "C" int aldi_sum_batch(const void* g, float* out, int N, long TC, int dtype, aldi_stream_t stream) {
    if (TC % 4 || N <= 0) return aldi_set_error_msg(ALDI_ERR_ARG, "sum_batch: bad sizes");
    hipStream_t st = (hipStream_t)stream;
    VIT_DISPATCH(dtype,
        hipLaunchKernelGGL(sum_batch_kernel<float>, dim3(grid_for(TC / 4)), dim3(256), 0, st, (const float*)g, out, N, TC / 4),
        hipLaunchKernelGGL(sum_batch_kernel<bf16_t>, dim3(grid_for(TC / 4)), dim3(256), 0, st, (const bf16_t*)g, out, N, TC / 4));
    ALDI_CHECK_LAUNCH();
    return ALDI_OK;
}

extern "C" int aldi_adamw_step(float* p, const float* g, float* m, float* v, void* p_compute, long n, float lr, float beta1, float beta2,
                               float eps, float weight_decay, int step, float grad_scale, int dtype, aldi_stream_t stream) {
    if (n <= 0 || step < 1) return aldi_set_error_msg(ALDI_ERR_ARG, "adamw: bad sizes");
    hipStream_t st = (hipStream_t)stream;
    const float bc1 = 1.f - powf(beta1, (float)step), bc2s = sqrtf(1.f - powf(beta2, (float)step));
    VIT_DISPATCH(dtype,
        hipLaunchKernelGGL(adamw_kernel<float>, dim3(grid_for(n)), dim3(256), 0, st, p, g, m, v, (float*)p_compute, n, lr, beta1, beta2, eps, weight_decay, bc1, bc2s, grad_scale),
        hipLaunchKernelGGL(adamw_kernel<bf16_t>, dim3(grid_for(n)), dim3(256), 0, st, p, g, m, v, (bf16_t*)p_compute, n, lr, beta1, beta2, eps, weight_decay, bc1, bc2s, grad_scale));
    ALDI_CHECK_LAUNCH();
    return ALDI_OK;
}

extern "C" int aldi_maxpool2(const void* x, void* y, unsigned char* idx, int N, int H, int W, int C, int backward, int dtype,
                             aldi_stream_t stream) {
    if (H % 2 || W % 2 || C % 4 || N <= 0) return aldi_set_error_msg(ALDI_ERR_ARG, "maxpool2: H, W must be even and C % 4 == 0");
    hipStream_t st = (hipStream_t)stream;
    const long work = (long)N * (H / 2) * (W / 2) * (C / 4);
    if (!backward) {
        VIT_DISPATCH(dtype,
            hipLaunchKernelGGL(maxpool2_kernel<float>, dim3(grid_for(work)), dim3(256), 0, st, (const float*)x, (float*)y, idx, N, H, W, C),
            hipLaunchKernelGGL(maxpool2_kernel<bf16_t>, dim3(grid_for(work)), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y, idx, N, H, W, C));
    } else {   // x = gradient wrt the pooled map, y = gradient wrt the input map (fully written)
        VIT_DISPATCH(dtype,
            hipLaunchKernelGGL(maxpool2_bwd_kernel<float>, dim3(grid_for(work)), dim3(256), 0, st, (const float*)x, idx, (float*)y, N, H, W, C),
            hipLaunchKernelGGL(maxpool2_bwd_kernel<bf16_t>, dim3(grid_for(work)), dim3(256), 0, st, (const bf16_t*)x, idx, (bf16_t*)y, N, H, W, C));
    }
    ALDI_CHECK_LAUNCH();
    return ALDI_OK;
}
