// Deformable-DETR pieces around the multi-scale deformable attention op (msda.hip), fp32 (the reference runs this detector with AMP
// off: configs/Base-DETR.yaml:56-58).  The detector's own source is an absent submodule of the reference (.gitmodules:4-6); the
// arithmetic follows oracle/deformable_detr.py, which is pinned against transformers' implementation.
//
//   * GroupNorm(32) of the input projections over NHWC maps: per (image, group) statistics in two deterministic stages (partial sums per
//     pixel chunk, chunks added in order), then one normalising pass;
//   * the glue between a deformable-attention layer's linear outputs and the sampling op: softmax over the (level, point) logits of every
//     head and sampling locations = reference point + offset / (W_l, H_l);
//   * multi-head self attention of the decoder's queries (a few hundred queries, 32-wide heads: one thread per query, keys and values of
//     the head streamed through the LDS, online softmax);
//   * the box head's last step: sigmoid(t + (logit(reference), 0, 0)).
#include "common.h"

namespace {

// ---------------------------------------------------------------------------------------------------- dropout
// Stateless: element i of site `seed` is kept when a 64-bit mix of (seed, i) lands at or above p -- the backward pass recomputes the same
// decision from the same (seed, i) instead of reading a stored mask.  (Not torch's generator: the reference's dropout draws from the CUDA
// Philox stream, which no other implementation reproduces either; the statistics are the same.)
__device__ __forceinline__ float drop_scale(unsigned long long seed, unsigned long long i, float p, float inv_keep) {
    unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (i + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    const float u = (float)(z >> 40) * (1.0f / 16777216.0f);         // 24 bits -> [0, 1)
    return u >= p ? inv_keep : 0.f;
}
// out = (res ? res : 0) + x * keep / (1 - p)
__global__ __launch_bounds__(256) void dropout_add_kernel(const float* __restrict__ x, const float* __restrict__ res, float* __restrict__ out, long n, float p,
                                                          unsigned long long seed) {
    const float inv_keep = 1.f / (1.f - p);
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        out[i] = (res ? res[i] : 0.f) + x[i] * drop_scale(seed, (unsigned long long)i, p, inv_keep);
}

// ---------------------------------------------------------------------------------------------------- GroupNorm
// x [N][HW][C]; part [N][chunks][G][2] = (sum, sum of squares) of the chunk's pixels over the group's C / G channels
__global__ __launch_bounds__(256) void gn_partial_kernel(const float* __restrict__ x, float* __restrict__ part, int HW, int C, int G, int chunk_px) {
    extern __shared__ float sm[];                     // [C][2]
    const int n = blockIdx.y, ch = blockIdx.x, chunks = gridDim.x;
    const int p0 = ch * chunk_px, p1 = min(HW, p0 + chunk_px);
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float s = 0.f, q = 0.f;
        const float* col = x + ((long)n * HW + p0) * C + c;
        for (int p = p0; p < p1; ++p, col += C) { const float v = *col; s += v; q += v * v; }
        sm[2 * c] = s; sm[2 * c + 1] = q;
    }
    __syncthreads();
    const int cpg = C / G;
    for (int g = threadIdx.x; g < G; g += blockDim.x) {
        float s = 0.f, q = 0.f;
        for (int k = 0; k < cpg; ++k) { s += sm[2 * (g * cpg + k)]; q += sm[2 * (g * cpg + k) + 1]; }
        float* o = part + (((long)n * chunks + ch) * G + g) * 2;
        o[0] = s; o[1] = q;
    }
}
__global__ void gn_stats_kernel(const float* __restrict__ part, float* __restrict__ mean, float* __restrict__ rstd, int chunks, int G, float inv_count, float eps) {
    const int n = blockIdx.x;
    for (int g = threadIdx.x; g < G; g += blockDim.x) {
        double s = 0.0, q = 0.0;                      // (a handful of chunks: the order is fixed, the width costs nothing)
        for (int c = 0; c < chunks; ++c) {
            const float* o = part + (((long)n * chunks + c) * G + g) * 2;
            s += o[0]; q += o[1];
        }
        const double m = s * inv_count;
        double var = q * inv_count - m * m;
        if (var < 0.0) var = 0.0;
        mean[n * G + g] = (float)m;
        rstd[n * G + g] = (float)(1.0 / sqrt(var + (double)eps));
    }
}
__global__ __launch_bounds__(256) void gn_apply_kernel(const float* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ y, long total, int HW,
                                                       int C, int G) {
    const int cpg = C / G;
    for (long i = (blockIdx.x * (long)blockDim.x + threadIdx.x) * 4; i < total; i += (long)gridDim.x * blockDim.x * 4) {
        const int c = (int)(i % C);
        const int n = (int)(i / ((long)HW * C));
        const float4 v = *reinterpret_cast<const float4*>(x + i);
        const float in[4] = {v.x, v.y, v.z, v.w};
        float out[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int g = (c + k) / cpg;
            out[k] = (in[k] - mean[n * G + g]) * rstd[n * G + g] * gamma[c + k] + beta[c + k];
        }
        *reinterpret_cast<float4*>(y + i) = make_float4(out[0], out[1], out[2], out[3]);
    }
}

// ---------------------------------------------------------------------------------------------------- deformable attention glue
// raw [T][M*L*P*3]: the sampling-offset linear's outputs ([M][L][P][2]) followed by the attention-weight linear's ([M][L*P]);
// ref [T][L][2] reference points in [0, 1] of every level; -> loc [T][M][L][P][2], aw [T][M][L][P]
__global__ __launch_bounds__(256) void msda_prepare_kernel(const float* __restrict__ raw, const float* __restrict__ ref, const int* __restrict__ shapes,
                                                           float* __restrict__ loc, float* __restrict__ aw, long T, int M, int L, int P) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;          // (token, head)
    if (i >= T * M) return;
    const long t = i / M;
    const int m = (int)(i % M);
    const int LP = L * P;
    const float* off = raw + t * (long)(M * LP * 3) + (long)m * LP * 2;
    const float* lg = raw + t * (long)(M * LP * 3) + (long)M * LP * 2 + (long)m * LP;
    float mx = -3.4e38f;
    for (int k = 0; k < LP; ++k) mx = fmaxf(mx, lg[k]);
    float sum = 0.f;
    for (int k = 0; k < LP; ++k) sum += expf(lg[k] - mx);
    const float inv = 1.f / sum;
    float* o_aw = aw + i * LP;
    float* o_loc = loc + i * LP * 2;
    for (int l = 0; l < L; ++l) {
        const float rx = ref[(t * L + l) * 2], ry = ref[(t * L + l) * 2 + 1];
        const float iw = 1.f / (float)shapes[2 * l + 1], ih = 1.f / (float)shapes[2 * l];
        for (int p = 0; p < P; ++p) {
            const int k = l * P + p;
            o_aw[k] = expf(lg[k] - mx) * inv;
            o_loc[2 * k] = rx + off[2 * k] * iw;
            o_loc[2 * k + 1] = ry + off[2 * k + 1] * ih;
        }
    }
}

// ---------------------------------------------------------------------------------------------------- decoder self attention
// q, k, v [B][Q][H*D] (row strides ldq / ldk / ldv floats); out [B][Q][H*D]; lse [B][H][Q] (for a backward pass).  One thread per query.
template <int D>
__global__ __launch_bounds__(64) void mha_small_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v, float* __restrict__ out,
                                                       float* __restrict__ lse, int Q, int H, int ldq, int ldk, int ldv, float scale, float drop_p,
                                                       unsigned long long seed) {
    __shared__ float ks[64][D], vs[64][D];
    const int bh = blockIdx.x, b = bh / H, h = bh % H;
    const int qi = blockIdx.y * 64 + threadIdx.x;
    const float inv_keep = 1.f / (1.f - drop_p);
    float qr[D], acc[D];
    const bool live = qi < Q;
#pragma unroll
    for (int d = 0; d < D; ++d) { qr[d] = live ? q[((long)b * Q + qi) * ldq + h * D + d] * scale : 0.f; acc[d] = 0.f; }
    float mx = -3.4e38f, l = 0.f;
    for (int j0 = 0; j0 < Q; j0 += 64) {
        __syncthreads();
        const int j = j0 + threadIdx.x;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            ks[threadIdx.x][d] = j < Q ? k[((long)b * Q + j) * ldk + h * D + d] : 0.f;
            vs[threadIdx.x][d] = j < Q ? v[((long)b * Q + j) * ldv + h * D + d] : 0.f;
        }
        __syncthreads();
        const int nj = min(64, Q - j0);
        for (int jj = 0; jj < nj; ++jj) {
            float s = 0.f;
#pragma unroll
            for (int d = 0; d < D; ++d) s += qr[d] * ks[jj][d];
            const float nm = fmaxf(mx, s);
            const float c = expf(mx - nm), p = expf(s - nm);
            l = l * c + p;                  // (the softmax normalises over ALL keys; dropout acts on the normalised probabilities)
            const float pk = drop_p > 0.f ? p * drop_scale(seed, ((unsigned long long)bh * Q + qi) * Q + (j0 + jj), drop_p, inv_keep) : p;
#pragma unroll
            for (int d = 0; d < D; ++d) acc[d] = acc[d] * c + pk * vs[jj][d];
            mx = nm;
        }
    }
    if (!live) return;
    const float inv = 1.f / l;
#pragma unroll
    for (int d = 0; d < D; ++d) out[((long)b * Q + qi) * (H * D) + h * D + d] = acc[d] * inv;
    if (lse) lse[((long)b * H + h) * Q + qi] = mx + logf(l);
}

// boxes [R][4] = sigmoid(t[R][4] + (logit(ref[r % refs][0..1]), 0, 0)), logit with the reference's clamps (eps 1e-5)
__global__ void box_finish_kernel(const float* __restrict__ t, const float* __restrict__ ref, float* __restrict__ boxes, long R, long refs) {
    const long r = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (r >= R) return;
    const float* rr = ref + (r % refs) * 2;
    float o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float x = t[r * 4 + k];
        if (k < 2) {
            const float u = fminf(fmaxf(rr[k], 0.f), 1.f);
            x += logf(fmaxf(u, 1e-5f) / fmaxf(1.f - u, 1e-5f));
        }
        o[k] = 1.f / (1.f + expf(-x));
    }
    *reinterpret_cast<float4*>(boxes + r * 4) = make_float4(o[0], o[1], o[2], o[3]);
}

}  // namespace

extern "C" size_t aldi_group_norm_workspace(int N, int HW, int G) {
    const int chunk = 256;
    return (size_t)N * cdiv(HW, chunk) * G * 2 * sizeof(float);
}
extern "C" int aldi_group_norm_forward(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd, void* workspace, int N, int HW,
                                       int C, int G, float eps, aldi_stream_t stream) {
    if (!x || !gamma || !beta || !y || !mean || !rstd || !workspace || N <= 0 || HW <= 0 || G <= 0 || C % G || C % 4 || C > 4096)
        return aldi_set_error_msg(ALDI_ERR_ARG, "group_norm_forward: bad args (C a multiple of the group count and of 4, at most 4096)");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int chunk = 256, chunks = cdiv(HW, chunk);
    float* part = static_cast<float*>(workspace);
    hipLaunchKernelGGL(gn_partial_kernel, dim3(chunks, N), dim3(256), (size_t)C * 2 * sizeof(float), st, x, part, HW, C, G, chunk);
    hipLaunchKernelGGL(gn_stats_kernel, dim3(N), dim3(64), 0, st, part, mean, rstd, chunks, G, 1.f / ((float)HW * (float)(C / G)), eps);
    const long total = (long)N * HW * C;
    hipLaunchKernelGGL(gn_apply_kernel, dim3((unsigned)((total / 4 + 255) / 256 < 4096 ? (total / 4 + 255) / 256 : 4096)), dim3(256), 0, st, x, mean, rstd, gamma, beta, y, total, HW, C, G);
    ALDI_CHECK_LAUNCH();
    return ALDI_OK;
}

extern "C" int aldi_msda_prepare(const float* raw, const float* ref, const int* spatial_shapes, float* sampling_loc, float* attn_weight, long T, int M, int L, int P,
                                 aldi_stream_t stream) {
    if (!raw || !ref || !spatial_shapes || !sampling_loc || !attn_weight || T <= 0 || M <= 0 || L <= 0 || P <= 0)
        return aldi_set_error_msg(ALDI_ERR_ARG, "msda_prepare: bad args");
    hipLaunchKernelGGL(msda_prepare_kernel, dim3((unsigned)((T * M + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), raw, ref, spatial_shapes, sampling_loc,
                       attn_weight, T, M, L, P);
    ALDI_CHECK_LAUNCH();
    return ALDI_OK;
}

extern "C" int aldi_dropout_add(const float* x, const float* res, float* out, long n, float p, unsigned long long seed, aldi_stream_t stream) {
    if (!x || !out || n <= 0 || !(p >= 0.f && p < 1.f)) return aldi_set_error_msg(ALDI_ERR_ARG, "dropout_add: bad args (0 <= p < 1)");
    const long blocks = (n + 255) / 256;
    hipLaunchKernelGGL(dropout_add_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, static_cast<hipStream_t>(stream), x, res, out, n, p, seed);
    ALDI_CHECK_LAUNCH();
    return ALDI_OK;
}

extern "C" int aldi_mha_small_forward(const float* q, const float* k, const float* v, float* out, float* lse, int B, int Q, int H, int D, int ldq, int ldk, int ldv,
                                      float scale, float drop_p, unsigned long long seed, aldi_stream_t stream) {
    if (!q || !k || !v || !out || B <= 0 || Q <= 0 || H <= 0 || !(drop_p >= 0.f && drop_p < 1.f)) return aldi_set_error_msg(ALDI_ERR_ARG, "mha_small_forward: bad args");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 grid(B * H, cdiv(Q, 64));
    if (D == 32) hipLaunchKernelGGL(mha_small_kernel<32>, grid, dim3(64), 0, st, q, k, v, out, lse, Q, H, ldq, ldk, ldv, scale, drop_p, seed);
    else if (D == 16) hipLaunchKernelGGL(mha_small_kernel<16>, grid, dim3(64), 0, st, q, k, v, out, lse, Q, H, ldq, ldk, ldv, scale, drop_p, seed);
    else if (D == 64) hipLaunchKernelGGL(mha_small_kernel<64>, grid, dim3(64), 0, st, q, k, v, out, lse, Q, H, ldq, ldk, ldv, scale, drop_p, seed);
    else return aldi_set_error_msg(ALDI_ERR_ARG, "mha_small_forward: head width 16, 32 or 64");
    ALDI_CHECK_LAUNCH();
    return ALDI_OK;
}

extern "C" int aldi_detr_box_finish(const float* t, const float* ref, float* boxes, long R, long refs, aldi_stream_t stream) {
    if (!t || !ref || !boxes || R <= 0 || refs <= 0) return aldi_set_error_msg(ALDI_ERR_ARG, "detr_box_finish: bad args");
    hipLaunchKernelGGL(box_finish_kernel, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), t, ref, boxes, R, refs);
    ALDI_CHECK_LAUNCH();
    return ALDI_OK;
}

namespace {
// rows of x [T][C] whose keep flag is 0 become zero (the value maps of padded pixels)
__global__ __launch_bounds__(256) void mask_rows_kernel(float* __restrict__ x, const unsigned char* __restrict__ keep, long T, int C) {
    for (long i = (blockIdx.x * (long)blockDim.x + threadIdx.x) * 4; i < T * C; i += (long)gridDim.x * blockDim.x * 4)
        if (!keep[i / C]) *reinterpret_cast<float4*>(x + i) = make_float4(0.f, 0.f, 0.f, 0.f);
}
}  // namespace

extern "C" int aldi_mask_rows(float* x, const unsigned char* keep, long T, int C, aldi_stream_t stream) {
    if (!x || !keep || T <= 0 || C <= 0 || C % 4) return aldi_set_error_msg(ALDI_ERR_ARG, "mask_rows: bad args");
    const long blocks = (T * C / 4 + 255) / 256;
    hipLaunchKernelGGL(mask_rows_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, static_cast<hipStream_t>(stream), x, keep, T, C);
    ALDI_CHECK_LAUNCH();
    return ALDI_OK;
}

// ==================================================================================================== backward passes
namespace {

// msda_prepare backward: g_raw [T][M*L*P*3] (fully written) and g_ref [T][L][2] (fully written: the sum over heads and points of the
// location gradients) from g_loc [T][M][L][P][2], g_aw [T][M][L][P] and the forward's attention weights aw.
__global__ __launch_bounds__(256) void msda_prepare_bwd_kernel(const float* __restrict__ g_loc, const float* __restrict__ g_aw, const float* __restrict__ aw,
                                                               const int* __restrict__ shapes, float* __restrict__ g_raw, long T, int M, int L, int P) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;          // (token, head)
    if (i >= T * M) return;
    const long t = i / M;
    const int m = (int)(i % M);
    const int LP = L * P;
    float* o_off = g_raw + t * (long)(M * LP * 3) + (long)m * LP * 2;
    float* o_lg = g_raw + t * (long)(M * LP * 3) + (long)M * LP * 2 + (long)m * LP;
    const float* a = aw + i * LP;
    const float* ga = g_aw + i * LP;
    const float* gl = g_loc + i * LP * 2;
    float dot = 0.f;
    for (int k = 0; k < LP; ++k) dot += a[k] * ga[k];
    for (int l = 0; l < L; ++l) {
        const float iw = 1.f / (float)shapes[2 * l + 1], ih = 1.f / (float)shapes[2 * l];
        for (int p = 0; p < P; ++p) {
            const int k = l * P + p;
            o_lg[k] = a[k] * (ga[k] - dot);
            o_off[2 * k] = gl[2 * k] * iw;
            o_off[2 * k + 1] = gl[2 * k + 1] * ih;
        }
    }
}
// g_ref [T][L][2] = sum over heads and points of g_loc (the reference point enters every location with weight 1)
__global__ __launch_bounds__(256) void msda_ref_bwd_kernel(const float* __restrict__ g_loc, float* __restrict__ g_ref, long T, int M, int L, int P) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;          // (token, level)
    if (i >= T * L) return;
    const long t = i / L;
    const int l = (int)(i % L);
    float sx = 0.f, sy = 0.f;
    for (int m = 0; m < M; ++m)
        for (int p = 0; p < P; ++p) {
            const float* g = g_loc + (((t * M + m) * L + l) * P + p) * 2;
            sx += g[0]; sy += g[1];
        }
    g_ref[i * 2] = sx; g_ref[i * 2 + 1] = sy;
}

// small attention backward.  delta[b][h][i] = sum_d dO * O.  dq: one thread per query; dk, dv: one thread per key.
template <int D>
__global__ __launch_bounds__(64) void mha_small_bwd_q_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                                             const float* __restrict__ o, const float* __restrict__ d_o, const float* __restrict__ lse,
                                                             float* __restrict__ dq, float* __restrict__ delta, int Q, int H, int ldq, int ldk, int ldv, int lddq,
                                                             float scale, float drop_p, unsigned long long seed) {
    __shared__ float ks[64][D], vs[64][D];
    const int bh = blockIdx.x, b = bh / H, h = bh % H;
    const int qi = blockIdx.y * 64 + threadIdx.x;
    const bool live = qi < Q;
    const float inv_keep = 1.f / (1.f - drop_p);
    float qr[D], gr[D], acc[D];
    float dl = 0.f;
#pragma unroll
    for (int d = 0; d < D; ++d) {
        qr[d] = live ? q[((long)b * Q + qi) * ldq + h * D + d] * scale : 0.f;
        gr[d] = live ? d_o[((long)b * Q + qi) * (H * D) + h * D + d] : 0.f;
        dl += live ? gr[d] * o[((long)b * Q + qi) * (H * D) + h * D + d] : 0.f;
        acc[d] = 0.f;
    }
    const float ls = live ? lse[((long)b * H + h) * Q + qi] : 0.f;
    for (int j0 = 0; j0 < Q; j0 += 64) {
        __syncthreads();
        const int j = j0 + threadIdx.x;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            ks[threadIdx.x][d] = j < Q ? k[((long)b * Q + j) * ldk + h * D + d] : 0.f;
            vs[threadIdx.x][d] = j < Q ? v[((long)b * Q + j) * ldv + h * D + d] : 0.f;
        }
        __syncthreads();
        const int nj = min(64, Q - j0);
        for (int jj = 0; jj < nj; ++jj) {
            float s = 0.f, dp = 0.f;
#pragma unroll
            for (int d = 0; d < D; ++d) { s += qr[d] * ks[jj][d]; dp += gr[d] * vs[jj][d]; }
            const float p = expf(s - ls);
            if (drop_p > 0.f) dp *= drop_scale(seed, ((unsigned long long)bh * Q + qi) * Q + (j0 + jj), drop_p, inv_keep);
            const float ds = p * (dp - dl);
#pragma unroll
            for (int d = 0; d < D; ++d) acc[d] += ds * ks[jj][d];
        }
    }
    if (!live) return;
#pragma unroll
    for (int d = 0; d < D; ++d) dq[((long)b * Q + qi) * lddq + h * D + d] = acc[d] * scale;
    delta[((long)b * H + h) * Q + qi] = dl;
}
template <int D>
__global__ __launch_bounds__(64) void mha_small_bwd_kv_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                                              const float* __restrict__ d_o, const float* __restrict__ lse, const float* __restrict__ delta,
                                                              float* __restrict__ dk, float* __restrict__ dv, int Q, int H, int ldq, int ldk, int ldv, int lddk,
                                                              int lddv, float scale, float drop_p, unsigned long long seed) {
    __shared__ float qs[64][D], gs[64][D], ls_[64], dl_[64];
    const int bh = blockIdx.x, b = bh / H, h = bh % H;
    const int kj = blockIdx.y * 64 + threadIdx.x;
    const bool live = kj < Q;
    const float inv_keep = 1.f / (1.f - drop_p);
    float kr[D], vr[D], ak[D], av[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        kr[d] = live ? k[((long)b * Q + kj) * ldk + h * D + d] : 0.f;
        vr[d] = live ? v[((long)b * Q + kj) * ldv + h * D + d] : 0.f;
        ak[d] = av[d] = 0.f;
    }
    for (int i0 = 0; i0 < Q; i0 += 64) {
        __syncthreads();
        const int i = i0 + threadIdx.x;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            qs[threadIdx.x][d] = i < Q ? q[((long)b * Q + i) * ldq + h * D + d] * scale : 0.f;
            gs[threadIdx.x][d] = i < Q ? d_o[((long)b * Q + i) * (H * D) + h * D + d] : 0.f;
        }
        ls_[threadIdx.x] = i < Q ? lse[((long)b * H + h) * Q + i] : 0.f;
        dl_[threadIdx.x] = i < Q ? delta[((long)b * H + h) * Q + i] : 0.f;
        __syncthreads();
        const int ni = min(64, Q - i0);
        for (int ii = 0; ii < ni; ++ii) {
            float s = 0.f, dp = 0.f;
#pragma unroll
            for (int d = 0; d < D; ++d) { s += qs[ii][d] * kr[d]; dp += gs[ii][d] * vr[d]; }
            const float p = expf(s - ls_[ii]);
            const float m = drop_p > 0.f ? drop_scale(seed, ((unsigned long long)bh * Q + (i0 + ii)) * Q + kj, drop_p, inv_keep) : 1.f;
            const float ds = p * (dp * m - dl_[ii]);
#pragma unroll
            for (int d = 0; d < D; ++d) { av[d] += p * m * gs[ii][d]; ak[d] += ds * qs[ii][d]; }
        }
    }
    if (!live) return;
#pragma unroll
    for (int d = 0; d < D; ++d) {
        dk[((long)b * Q + kj) * lddk + h * D + d] = ak[d];           // (qs already carries the scale)
        dv[((long)b * Q + kj) * lddv + h * D + d] = av[d];
    }
}

// GroupNorm backward.  Stage 1 per (image, pixel chunk): per channel sum(g) and sum(g * xhat) -> part [N][chunks][C][2]; stage 2 per image:
// chunks added in order -> per-group sums ds, db (for dx) and per-channel sums accumulated into dgamma / dbeta over images in order.
__global__ __launch_bounds__(256) void gn_bwd_partial_kernel(const float* __restrict__ g, const float* __restrict__ x, const float* __restrict__ mean,
                                                             const float* __restrict__ rstd, float* __restrict__ part, int HW, int C, int G, int chunk_px) {
    const int n = blockIdx.y, ch = blockIdx.x, chunks = gridDim.x;
    const int p0 = ch * chunk_px, p1 = min(HW, p0 + chunk_px);
    const int cpg = C / G;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const float m = mean[n * G + c / cpg], r = rstd[n * G + c / cpg];
        float s = 0.f, q = 0.f;
        const long base = ((long)n * HW + p0) * C + c;
        for (int p = 0; p < p1 - p0; ++p) {
            const float gv = g[base + (long)p * C];
            s += gv;
            q += gv * (x[base + (long)p * C] - m) * r;
        }
        float* o = part + (((long)n * chunks + ch) * C + c) * 2;
        o[0] = s; o[1] = q;
    }
}
// one workgroup per image: sums [C][2] over chunks (LDS), group sums, then this image's contribution to dgamma / dbeta
__global__ __launch_bounds__(256) void gn_bwd_reduce_kernel(const float* __restrict__ part, const float* __restrict__ gamma, float* __restrict__ gsum /*[N][G][2]*/,
                                                            float* __restrict__ chan /*[N][C][2]*/, int chunks, int C, int G) {
    extern __shared__ float sm[];                     // [C][2]
    const int n = blockIdx.x, cpg = C / G;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float s = 0.f, q = 0.f;
        for (int k = 0; k < chunks; ++k) {
            const float* o = part + (((long)n * chunks + k) * C + c) * 2;
            s += o[0]; q += o[1];
        }
        sm[2 * c] = s; sm[2 * c + 1] = q;
        chan[((long)n * C + c) * 2] = s; chan[((long)n * C + c) * 2 + 1] = q;
    }
    __syncthreads();
    for (int gi = threadIdx.x; gi < G; gi += blockDim.x) {
        float a = 0.f, b = 0.f;                       // sum(g * gamma), sum(g * gamma * xhat) over the group
        for (int k = 0; k < cpg; ++k) {
            const int c = gi * cpg + k;
            a += sm[2 * c] * gamma[c];
            b += sm[2 * c + 1] * gamma[c];
        }
        gsum[((long)n * G + gi) * 2] = a; gsum[((long)n * G + gi) * 2 + 1] = b;
    }
}
__global__ void gn_bwd_params_kernel(const float* __restrict__ chan, float* __restrict__ dgamma, float* __restrict__ dbeta, int N, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float s = 0.f, q = 0.f;
    for (int n = 0; n < N; ++n) { s += chan[((long)n * C + c) * 2]; q += chan[((long)n * C + c) * 2 + 1]; }
    dbeta[c] += s; dgamma[c] += q;
}
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const float* __restrict__ g, const float* __restrict__ x, const float* __restrict__ mean,
                                                           const float* __restrict__ rstd, const float* __restrict__ gamma, const float* __restrict__ gsum,
                                                           float* __restrict__ dx, long total, int HW, int C, int G) {
    const int cpg = C / G;
    const float inv_cnt = 1.f / ((float)HW * (float)cpg);
    for (long i = (blockIdx.x * (long)blockDim.x + threadIdx.x) * 4; i < total; i += (long)gridDim.x * blockDim.x * 4) {
        const int c = (int)(i % C);
        const int n = (int)(i / ((long)HW * C));
        const float4 gv = *reinterpret_cast<const float4*>(g + i), xv = *reinterpret_cast<const float4*>(x + i);
        const float gg[4] = {gv.x, gv.y, gv.z, gv.w}, xx[4] = {xv.x, xv.y, xv.z, xv.w};
        float o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int gi = (c + k) / cpg;
            const float m = mean[n * G + gi], r = rstd[n * G + gi];
            const float a = gsum[((long)n * G + gi) * 2] * inv_cnt, b = gsum[((long)n * G + gi) * 2 + 1] * inv_cnt;
            const float xh = (xx[k] - m) * r;
            o[k] = r * (gg[k] * gamma[c + k] - a - xh * b);
        }
        *reinterpret_cast<float4*>(dx + i) = make_float4(o[0], o[1], o[2], o[3]);
    }
}

// box head's last step backward: g_t [R][4] (fully written), g_ref [refs][2] accumulated over the rows that share a reference point
__global__ void box_finish_bwd_kernel(const float* __restrict__ g_boxes, const float* __restrict__ boxes, const float* __restrict__ ref, float* __restrict__ g_t,
                                      long R) {
    const long r = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (r >= R) return;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float s = boxes[r * 4 + k];
        g_t[r * 4 + k] = g_boxes[r * 4 + k] * s * (1.f - s);
    }
}
// g_ref [refs][2] = sum over rows r with r % refs == j (in row order) of g_t[r][0..1] * d logit / d ref
__global__ void box_ref_bwd_kernel(const float* __restrict__ g_t, const float* __restrict__ ref, float* __restrict__ g_ref, long R, long refs) {
    const long j = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (j >= refs) return;
    float sx = 0.f, sy = 0.f;
    for (long r = j; r < R; r += refs) { sx += g_t[r * 4]; sy += g_t[r * 4 + 1]; }
    const float s[2] = {sx, sy};
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const float u = ref[j * 2 + k];
        // logit(u) = log(max(u, eps)) - log(max(1 - u, eps)): each clamp switches its term's derivative off
        float d = 0.f;
        if (u >= 0.f && u <= 1.f) d = (u > 1e-5f ? 1.f / u : 0.f) + (1.f - u > 1e-5f ? 1.f / (1.f - u) : 0.f);
        g_ref[j * 2 + k] += s[k] * d;
    }
}

}  // namespace

extern "C" int aldi_msda_prepare_backward(const float* g_loc, const float* g_aw, const float* attn_weight, const int* spatial_shapes, float* g_raw, float* g_ref, long T,
                                          int M, int L, int P, aldi_stream_t stream) {
    if (!g_loc || !g_aw || !attn_weight || !spatial_shapes || !g_raw || T <= 0) return aldi_set_error_msg(ALDI_ERR_ARG, "msda_prepare_backward: bad args");
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(msda_prepare_bwd_kernel, dim3((unsigned)((T * M + 255) / 256)), dim3(256), 0, st, g_loc, g_aw, attn_weight, spatial_shapes, g_raw, T, M, L, P);
    if (g_ref) hipLaunchKernelGGL(msda_ref_bwd_kernel, dim3((unsigned)((T * L + 255) / 256)), dim3(256), 0, st, g_loc, g_ref, T, M, L, P);
    ALDI_CHECK_LAUNCH();
    return ALDI_OK;
}

extern "C" int aldi_mha_small_backward(const float* q, const float* k, const float* v, const float* out, const float* d_out, const float* lse, float* dq, float* dk,
                                       float* dv, float* delta, int B, int Q, int H, int D, int ldq, int ldk, int ldv, int lddq, int lddk, int lddv, float scale,
                                       float drop_p, unsigned long long seed, aldi_stream_t stream) {
    if (!q || !k || !v || !out || !d_out || !lse || !dq || !dk || !dv || !delta) return aldi_set_error_msg(ALDI_ERR_ARG, "mha_small_backward: null pointer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 grid(B * H, cdiv(Q, 64));
#define ALDI_MHA_BWD(DD)                                                                                                                                         \
    hipLaunchKernelGGL(mha_small_bwd_q_kernel<DD>, grid, dim3(64), 0, st, q, k, v, out, d_out, lse, dq, delta, Q, H, ldq, ldk, ldv, lddq, scale, drop_p, seed); \
    hipLaunchKernelGGL(mha_small_bwd_kv_kernel<DD>, grid, dim3(64), 0, st, q, k, v, d_out, lse, delta, dk, dv, Q, H, ldq, ldk, ldv, lddk, lddv, scale, drop_p, seed)
    if (D == 32) { ALDI_MHA_BWD(32); }
    else if (D == 16) { ALDI_MHA_BWD(16); }
    else if (D == 64) { ALDI_MHA_BWD(64); }
    else return aldi_set_error_msg(ALDI_ERR_ARG, "mha_small_backward: head width 16, 32 or 64");
#undef ALDI_MHA_BWD
    ALDI_CHECK_LAUNCH();
    return ALDI_OK;
}

extern "C" size_t aldi_group_norm_backward_workspace(int N, int HW, int C, int G) {
    return ((size_t)N * cdiv(HW, 256) * C * 2 + (size_t)N * G * 2 + (size_t)N * C * 2) * sizeof(float);
}
extern "C" int aldi_group_norm_backward(const float* g, const float* x, const float* gamma, const float* mean, const float* rstd, float* dx, float* dgamma,
                                        float* dbeta, void* workspace, int N, int HW, int C, int G, aldi_stream_t stream) {
    if (!g || !x || !gamma || !mean || !rstd || !dx || !dgamma || !dbeta || !workspace || C % G || C % 4 || C > 4096)
        return aldi_set_error_msg(ALDI_ERR_ARG, "group_norm_backward: bad args");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int chunk = 256, chunks = cdiv(HW, chunk);
    float* part = static_cast<float*>(workspace);
    float* gsum = part + (size_t)N * chunks * C * 2;
    float* chan = gsum + (size_t)N * G * 2;
    hipLaunchKernelGGL(gn_bwd_partial_kernel, dim3(chunks, N), dim3(256), 0, st, g, x, mean, rstd, part, HW, C, G, chunk);
    hipLaunchKernelGGL(gn_bwd_reduce_kernel, dim3(N), dim3(256), (size_t)C * 2 * sizeof(float), st, part, gamma, gsum, chan, chunks, C, G);
    hipLaunchKernelGGL(gn_bwd_params_kernel, dim3(cdiv(C, 256)), dim3(256), 0, st, chan, dgamma, dbeta, N, C);
    const long total = (long)N * HW * C;
    hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3((unsigned)((total / 4 + 255) / 256 < 4096 ? (total / 4 + 255) / 256 : 4096)), dim3(256), 0, st, g, x, mean, rstd, gamma,
                       gsum, dx, total, HW, C, G);
    ALDI_CHECK_LAUNCH();
    return ALDI_OK;
}

extern "C" int aldi_detr_box_finish_backward(const float* g_boxes, const float* boxes, const float* ref, float* g_t, float* g_ref, long R, long refs,
                                             aldi_stream_t stream) {
    if (!g_boxes || !boxes || !ref || !g_t || R <= 0 || refs <= 0) return aldi_set_error_msg(ALDI_ERR_ARG, "detr_box_finish_backward: bad args");
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(box_finish_bwd_kernel, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, st, g_boxes, boxes, ref, g_t, R);
    if (g_ref) hipLaunchKernelGGL(box_ref_bwd_kernel, dim3((unsigned)((refs + 255) / 256)), dim3(256), 0, st, g_t, ref, g_ref, R, refs);
    ALDI_CHECK_LAUNCH();
    return ALDI_OK;
}

// ==================================================================================================== Hungarian cost and set loss
namespace {

__device__ __forceinline__ void cxcywh_to_xyxy(const float* b, float* o) {
    o[0] = b[0] - 0.5f * b[2]; o[1] = b[1] - 0.5f * b[3]; o[2] = b[0] + 0.5f * b[2]; o[3] = b[1] + 0.5f * b[3];
}
// generalized IoU of two xyxy boxes; optionally its gradient with respect to box a's corners
__device__ __forceinline__ float giou_xyxy(const float* a, const float* b, float* da) {
    const float aw = a[2] - a[0], ah = a[3] - a[1];
    const float area_a = aw * ah, area_b = (b[2] - b[0]) * (b[3] - b[1]);
    const float ix0 = fmaxf(a[0], b[0]), iy0 = fmaxf(a[1], b[1]), ix1 = fminf(a[2], b[2]), iy1 = fminf(a[3], b[3]);
    const float iw = fmaxf(ix1 - ix0, 0.f), ih = fmaxf(iy1 - iy0, 0.f);
    const float inter = iw * ih, uni = area_a + area_b - inter;
    const float hx0 = fminf(a[0], b[0]), hy0 = fminf(a[1], b[1]), hx1 = fmaxf(a[2], b[2]), hy1 = fmaxf(a[3], b[3]);
    const float hw = fmaxf(hx1 - hx0, 0.f), hh = fmaxf(hy1 - hy0, 0.f);
    const float hull = hw * hh;
    const float g = inter / uni - (hull - uni) / hull;
    if (da) {
        // corner k of a: derivative of every piece
        const float d_iw[4] = {(iw > 0.f && a[0] > b[0]) ? -1.f : 0.f, 0.f, (iw > 0.f && a[2] < b[2]) ? 1.f : 0.f, 0.f};
        const float d_ih[4] = {0.f, (ih > 0.f && a[1] > b[1]) ? -1.f : 0.f, 0.f, (ih > 0.f && a[3] < b[3]) ? 1.f : 0.f};
        const float d_aa[4] = {-ah, -aw, ah, aw};
        const float d_hw[4] = {(a[0] < b[0]) ? -1.f : 0.f, 0.f, (a[2] > b[2]) ? 1.f : 0.f, 0.f};
        const float d_hh[4] = {0.f, (a[1] < b[1]) ? -1.f : 0.f, 0.f, (a[3] > b[3]) ? 1.f : 0.f};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float d_inter = d_iw[k] * ih + iw * d_ih[k];
            const float d_uni = d_aa[k] - d_inter;
            const float d_hull = d_hw[k] * hh + hw * d_hh[k];
            da[k] = (d_inter * uni - inter * d_uni) / (uni * uni) + (d_uni * hull - uni * d_hull) / (hull * hull);
        }
    }
    return g;
}

// cost [LB][Nq][Gmax] of assigning query q to target g (focal-style class cost, L1, -GIoU); targets of image lb % B
__global__ __launch_bounds__(256) void detr_cost_kernel(const float* __restrict__ logits, const float* __restrict__ boxes, const int* __restrict__ t_labels,
                                                        const float* __restrict__ t_boxes, const int* __restrict__ t_count, float* __restrict__ cost, int LB, int B,
                                                        int Nq, int K, int Gmax, float w_class, float w_bbox, float w_giou, float alpha) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= (long)LB * Nq * Gmax) return;
    const int g = (int)(i % Gmax);
    const long r = i / Gmax;                          // (lb, q)
    const int b = (int)((r / Nq) % B);
    if (g >= t_count[b]) { cost[i] = 0.f; return; }
    const int lab = t_labels[b * Gmax + g];
    const float p = 1.f / (1.f + expf(-logits[r * K + lab]));
    const float neg = (1.f - alpha) * p * p * -logf(1.f - p + 1e-8f);
    const float pos = alpha * (1.f - p) * (1.f - p) * -logf(p + 1e-8f);
    const float* pb = boxes + r * 4;
    const float* tb = t_boxes + ((long)b * Gmax + g) * 4;
    const float l1 = fabsf(pb[0] - tb[0]) + fabsf(pb[1] - tb[1]) + fabsf(pb[2] - tb[2]) + fabsf(pb[3] - tb[3]);
    float pa[4], ta[4];
    cxcywh_to_xyxy(pb, pa); cxcywh_to_xyxy(tb, ta);
    cost[i] = w_bbox * l1 + w_class * (pos - neg) - w_giou * giou_xyxy(pa, ta, nullptr);
}

// One thread per (lb, q): the focal loss of its K logits (+ gradient), and, when the query is matched (match[lb][q] = target index or -1),
// the L1 and GIoU terms of its box (+ gradient).  rows [LB*Nq][3] = this row's (focal, L1, 1 - GIoU) sums; gradients already carry
// coefficient / num_boxes.
__global__ __launch_bounds__(256) void detr_loss_kernel(const float* __restrict__ logits, const float* __restrict__ boxes, const int* __restrict__ match,
                                                        const int* __restrict__ t_labels, const float* __restrict__ t_boxes, float* __restrict__ rows,
                                                        float* __restrict__ g_logits, float* __restrict__ g_boxes, int LB, int B, int Nq, int K, int Gmax,
                                                        float alpha, float c_ce, float c_bbox, float c_giou, float inv_num_boxes) {
    const long r = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (r >= (long)LB * Nq) return;
    const int b = (int)((r / Nq) % B);
    const int m = match[r];
    const int lab = m >= 0 ? t_labels[b * Gmax + m] : -1;
    float ce = 0.f;
    for (int k = 0; k < K; ++k) {
        const float x = logits[r * K + k];
        const float p = 1.f / (1.f + expf(-x));
        // binary cross entropy with logits, stable form; focal weight (1 - p_t)^2 and the alpha balance
        const float t = k == lab ? 1.f : 0.f;
        const float bce = fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x)));
        const float pt = p * t + (1.f - p) * (1.f - t);
        const float at = alpha * t + (1.f - alpha) * (1.f - t);
        ce += at * (1.f - pt) * (1.f - pt) * bce;
        float d;
        if (k == lab) d = alpha * (1.f - p) * (1.f - p) * (2.f * p * logf(fmaxf(p, 1e-38f)) - (1.f - p));
        else d = (1.f - alpha) * p * p * (p - 2.f * (1.f - p) * log1pf(-fminf(p, 1.f - 1e-7f)));
        g_logits[r * K + k] = d * c_ce * inv_num_boxes;
    }
    float l1 = 0.f, gl = 0.f;
    float gb[4] = {0.f, 0.f, 0.f, 0.f};
    if (m >= 0) {
        const float* pb = boxes + r * 4;
        const float* tb = t_boxes + ((long)b * Gmax + m) * 4;
        float pa[4], ta[4], da[4];
        cxcywh_to_xyxy(pb, pa); cxcywh_to_xyxy(tb, ta);
        const float g = giou_xyxy(pa, ta, da);
        gl = 1.f - g;
        // corners -> (cx, cy, w, h): x0 = cx - w/2, x1 = cx + w/2
        const float dg[4] = {da[0] + da[2], da[1] + da[3], 0.5f * (da[2] - da[0]), 0.5f * (da[3] - da[1])};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float df = pb[k] - tb[k];
            l1 += fabsf(df);
            gb[k] = (c_bbox * (df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f)) - c_giou * dg[k]) * inv_num_boxes;
        }
    }
    *reinterpret_cast<float4*>(g_boxes + r * 4) = make_float4(gb[0], gb[1], gb[2], gb[3]);
    rows[r * 3] = ce; rows[r * 3 + 1] = l1; rows[r * 3 + 2] = gl;
}
// losses [LB / B][3]: per decoder layer, the rows of its B images added in row order, / num_boxes
__global__ void detr_loss_reduce_kernel(const float* __restrict__ rows, float* __restrict__ losses, int rows_per_layer, float inv_num_boxes) {
    const int l = blockIdx.x, k = threadIdx.x;
    if (k >= 3) return;
    float s = 0.f;
    for (int r = 0; r < rows_per_layer; ++r) s += rows[((long)l * rows_per_layer + r) * 3 + k];
    losses[l * 3 + k] = s * inv_num_boxes;
}

}  // namespace

extern "C" int aldi_detr_match_cost(const float* logits, const float* boxes, const int* t_labels, const float* t_boxes, const int* t_count, float* cost, int LB, int B,
                                    int Nq, int K, int Gmax, float w_class, float w_bbox, float w_giou, float alpha, aldi_stream_t stream) {
    if (!logits || !boxes || !t_labels || !t_boxes || !t_count || !cost || LB <= 0 || B <= 0 || LB % B || Gmax <= 0)
        return aldi_set_error_msg(ALDI_ERR_ARG, "detr_match_cost: bad args");
    const long n = (long)LB * Nq * Gmax;
    hipLaunchKernelGGL(detr_cost_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), logits, boxes, t_labels, t_boxes, t_count, cost,
                       LB, B, Nq, K, Gmax, w_class, w_bbox, w_giou, alpha);
    ALDI_CHECK_LAUNCH();
    return ALDI_OK;
}

extern "C" int aldi_detr_set_loss(const float* logits, const float* boxes, const int* match, const int* t_labels, const float* t_boxes, float* rows, float* losses,
                                  float* g_logits, float* g_boxes, int LB, int B, int Nq, int K, int Gmax, float alpha, float c_ce, float c_bbox, float c_giou,
                                  float num_boxes, aldi_stream_t stream) {
    if (!logits || !boxes || !match || !t_labels || !t_boxes || !rows || !losses || !g_logits || !g_boxes || LB <= 0 || B <= 0 || LB % B || num_boxes <= 0.f)
        return aldi_set_error_msg(ALDI_ERR_ARG, "detr_set_loss: bad args");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const long n = (long)LB * Nq;
    hipLaunchKernelGGL(detr_loss_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, logits, boxes, match, t_labels, t_boxes, rows, g_logits, g_boxes, LB, B, Nq, K,
                       Gmax, alpha, c_ce, c_bbox, c_giou, 1.f / num_boxes);
    hipLaunchKernelGGL(detr_loss_reduce_kernel, dim3(LB / B), dim3(64), 0, st, rows, losses, B * Nq, 1.f / num_boxes);
    ALDI_CHECK_LAUNCH();
    return ALDI_OK;
}
