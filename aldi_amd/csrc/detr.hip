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
                                                       float* __restrict__ lse, int Q, int H, int ldq, int ldk, int ldv, float scale) {
    __shared__ float ks[64][D], vs[64][D];
    const int bh = blockIdx.x, b = bh / H, h = bh % H;
    const int qi = blockIdx.y * 64 + threadIdx.x;
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
            l = l * c + p;
#pragma unroll
            for (int d = 0; d < D; ++d) acc[d] = acc[d] * c + p * vs[jj][d];
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

extern "C" int aldi_mha_small_forward(const float* q, const float* k, const float* v, float* out, float* lse, int B, int Q, int H, int D, int ldq, int ldk, int ldv,
                                      float scale, aldi_stream_t stream) {
    if (!q || !k || !v || !out || B <= 0 || Q <= 0 || H <= 0) return aldi_set_error_msg(ALDI_ERR_ARG, "mha_small_forward: bad args");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 grid(B * H, cdiv(Q, 64));
    if (D == 32) hipLaunchKernelGGL(mha_small_kernel<32>, grid, dim3(64), 0, st, q, k, v, out, lse, Q, H, ldq, ldk, ldv, scale);
    else if (D == 16) hipLaunchKernelGGL(mha_small_kernel<16>, grid, dim3(64), 0, st, q, k, v, out, lse, Q, H, ldq, ldk, ldv, scale);
    else if (D == 64) hipLaunchKernelGGL(mha_small_kernel<64>, grid, dim3(64), 0, st, q, k, v, out, lse, Q, H, ldq, ldk, ldv, scale);
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
