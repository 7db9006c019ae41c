// ALDI-owned losses, forward + backward fused: RPN / ROI-head distillation
// (reference aldi/distill.py:193-278) and the domain-alignment BCE with constant label
// (reference aldi/align.py:76-90), plus the global-average-pool pair of the ConvDiscriminator
// (reference aldi/align.py:103-119).
#include "common.h"
#include "loss_rows.h"
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

namespace {

struct DGeom {
    int nl, A, C, sumA, N;
    int H[ALDI_MAX_LEVELS], W[ALDI_MAX_LEVELS], off[ALDI_MAX_LEVELS + 1];
    const float* s[ALDI_MAX_LEVELS];
    const float* t[ALDI_MAX_LEVELS];
    float* g[ALDI_MAX_LEVELS];
};

// Position q of the reference's cat([flatten(x) for x in per_level_raw_outputs]) where the raw
// level tensor is (N, CH, H, W)  ->  offset into our NHWC-with-C-channels level buffer.
// `chq` = CH per anchor-group (1 for logits: CH = A; 4 for deltas: CH = 4A), `cbase` = first
// channel of that group in our row (0 for logits, A for deltas).
__device__ __forceinline__ bool raw_flat_to_nhwc(const DGeom& g, long q, int chq, int cbase, int* lvl, long* off) {
    int l = 0;
#pragma unroll
    for (int k = 1; k < ALDI_MAX_LEVELS; ++k)
        if (k < g.nl && q >= (long)g.N * chq * g.off[k]) l = k;
    long rem = q - (long)g.N * chq * g.off[l];
    const int HW = g.H[l] * g.W[l];
    const int CH = g.A * chq;
    int n = (int)(rem / ((long)CH * HW));
    rem -= (long)n * CH * HW;
    int ch = (int)(rem / HW);
    int hw = (int)(rem - (long)ch * HW);
    *lvl = l;
    *off = ((long)n * HW + hw) * g.C + cbase + ch;
    return true;
}

__device__ __forceinline__ float bce_logits(float x, float y) {
    float m = fmaxf(-x, 0.f);
    return (1.f - y) * x + m + logf(expf(-m) + expf(-x - m));
}

// reference aldi/distill.py:203-227.  labels: [N][sumA] int in {-1,0,1} (flat index = mask position).
__global__ __launch_bounds__(256) void rpn_distill_kernel(DGeom g, const int* __restrict__ labels, float inv_T, float inv_valid, float inv_fg4,
                                                          const int* __restrict__ counts /* device {n_valid, n_fg} or null */,
                                                          int do_obj, int do_reg, float gscale, float* __restrict__ loss /*[2]*/) {
    __shared__ float red[16];
    if (counts) {
        const int nv = counts[0], nf = counts[1];
        inv_valid = nv > 0 ? 1.f / (float)nv : 0.f;
        inv_fg4 = nf > 0 ? 1.f / (float)(4 * nf) : 0.f;
    }
    const long q = blockIdx.x * (long)blockDim.x + threadIdx.x;
    float l_obj = 0.f, l_reg = 0.f;
    if (q < (long)g.N * g.sumA) {
        const int lab = labels[q];
        if (lab >= 0 && do_obj) {
            int l; long o;
            raw_flat_to_nhwc(g, q, 1, 0, &l, &o);
            const float xs = g.s[l][o];
            const float tp = 1.f / (1.f + expf(-(g.t[l][o] * inv_T)));
            l_obj = bce_logits(xs, tp);
            if (g.g[l] && gscale != 0.f) g.g[l][o] += (1.f / (1.f + expf(-xs)) - tp) * inv_valid * gscale;
        }
        if (lab == 1 && do_reg) {
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                int l; long o;
                raw_flat_to_nhwc(g, q * 4 + d, 4, g.A, &l, &o);
                const float df = g.s[l][o] - g.t[l][o];
                l_reg += fabsf(df);
                if (g.g[l] && gscale != 0.f) g.g[l][o] += (df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f)) * inv_fg4 * gscale;
            }
        }
    }
    float s0 = block_sum(l_obj, red);
    float s1 = block_sum(l_reg, red);
    if (threadIdx.x == 0) {
        if (s0 != 0.f) unsafeAtomicAdd(loss + 0, s0 * inv_valid);
        if (s1 != 0.f) unsafeAtomicAdd(loss + 1, s1 * inv_fg4);
    }
}

// reference aldi/distill.py:231-278.  pred rows: [0,K] logits, then 4K class-specific deltas.
__global__ __launch_bounds__(256) void roih_distill_kernel(const float* __restrict__ sp, const float* __restrict__ tp, int Cp, int K, int R,
                                                           float inv_T, int kl, int do_cls, int do_reg, float gscale,
                                                           float* __restrict__ grad, float* __restrict__ loss /*[2]*/) {
    __shared__ float red[16];
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    float l_cls = 0.f, l_reg = 0.f;
    const float invR = 1.f / (float)max(R, 1);
    if (r < R)
        roih_distill_row(sp + (long)r * Cp, tp + (long)r * Cp, K, inv_T, kl, do_cls, do_reg, invR, gscale, gscale, grad ? grad + (long)r * Cp : nullptr, l_cls, l_reg);
    float s0 = block_sum(l_cls, red);
    float s1 = block_sum(l_reg, red);
    if (threadIdx.x == 0) {
        if (s0 != 0.f) unsafeAtomicAdd(loss + 0, s0 * invR);
        if (s1 != 0.f) unsafeAtomicAdd(loss + 1, s1 * invR);
    }
}

// weight * mean_r BCEWithLogits(pred[r][0], label); grad[r][0] = weight*(sigmoid - label)/R * gscale (other columns 0)
template <typename T>
__global__ __launch_bounds__(256) void bce_const_kernel(const float* __restrict__ pred, int ld, int R, float label, float weight, float gscale,
                                                        T* __restrict__ grad, float* __restrict__ loss) {
    __shared__ float red[16];
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    float l = 0.f;
    const float invR = 1.f / (float)max(R, 1);
    if (r < R) {
        const float x = pred[(long)r * ld];
        l = bce_logits(x, label);
        if (grad) {
            for (int c = 1; c < ld; ++c) Elem<T>::st(grad + (long)r * ld + c, 0.f);
            Elem<T>::st(grad + (long)r * ld, weight * (1.f / (1.f + expf(-x)) - label) * invR * gscale);
        }
    }
    float s = block_sum(l, red);
    if (threadIdx.x == 0 && s != 0.f) unsafeAtomicAdd(loss, weight * s * invR);
}

// AdaptiveAvgPool2d(1): x [N][HW][C] -> y [N][C]   (grid (C/64, N), 256 threads = 4 row-lanes x 64 channels)
template <typename T>
__global__ __launch_bounds__(256) void avgpool_kernel(const T* __restrict__ x, T* __restrict__ y, int HW, int C) {
    __shared__ float red[4][64];
    const int n = blockIdx.y, c = blockIdx.x * 64 + (threadIdx.x & 63), w = threadIdx.x >> 6;
    float s = 0.f;
    if (c < C)
        for (int p = w; p < HW; p += 4) s += Elem<T>::ld(x + ((long)n * HW + p) * C + c);
    red[w][threadIdx.x & 63] = s;
    __syncthreads();
    if (w == 0 && c < C) Elem<T>::st(y + (long)n * C + c, (red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]) / (float)HW);
}

// The same in two launches over pixel ranges (the image discriminator pools a 196 x 332 map: eight workgroups walking 65 k pixels
// each took 4.4 ms): part[n][s][c] = sum over the s-th pixel range (fp32), then y = sum_s part / HW.  Deterministic.
constexpr int kPoolSplits = 128;
template <typename T>
__global__ __launch_bounds__(256) void avgpool_part_kernel(const T* __restrict__ x, float* __restrict__ part, int HW, int C) {
    __shared__ float red[4][64];
    const int n = blockIdx.z, sp = blockIdx.y, c = blockIdx.x * 64 + (threadIdx.x & 63), w = threadIdx.x >> 6;
    const int per = (HW + kPoolSplits - 1) / kPoolSplits, p0 = sp * per, p1 = min(HW, p0 + per);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (c < C) {
        const T* xp = x + (long)n * HW * C + c;
        int p = p0 + w;
        for (; p + 12 < p1; p += 16) {                 // four independent loads in flight per lane
            s0 += Elem<T>::ld(xp + (long)p * C); s1 += Elem<T>::ld(xp + (long)(p + 4) * C);
            s2 += Elem<T>::ld(xp + (long)(p + 8) * C); s3 += Elem<T>::ld(xp + (long)(p + 12) * C);
        }
        for (; p < p1; p += 4) s0 += Elem<T>::ld(xp + (long)p * C);
    }
    red[w][threadIdx.x & 63] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (w == 0 && c < C) part[((long)n * kPoolSplits + sp) * C + c] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
template <typename T>
__global__ void avgpool_finish_kernel(const float* __restrict__ part, T* __restrict__ y, int N, int HW, int C) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * C) return;
    const int n = i / C, c = i - n * C;
    float s = 0.f;
    for (int sp = 0; sp < kPoolSplits; ++sp) s += part[((long)n * kPoolSplits + sp) * C + c];
    Elem<T>::st(y + i, s / (float)HW);
}

// backward of ReLU -> avgpool: gx[n][p][c] = act[n][p][c] > 0 ? gy[n][c] / HW : 0
template <typename T>
__global__ void avgpool_bwd_kernel(const T* __restrict__ gy, const T* __restrict__ act, T* __restrict__ gx, int N, int HW, int C) {
    long total = (long)N * HW * (C / 4);
    const float inv = 1.f / (float)HW;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int c4 = (int)(i % (C / 4));
        int n = (int)(i / ((long)HW * (C / 4)));
        float g[4], a[4];
        load4(gy + (long)n * C + c4 * 4, g);
        load4(act + i * 4, a);
#pragma unroll
        for (int k = 0; k < 4; ++k) g[k] = a[k] > 0.f ? g[k] * inv : 0.f;
        store4(gx + i * 4, g);
    }
}

}  // namespace

extern "C" int aldi_rpn_distill_loss(const aldi_rpn_geom* gm, float* const* student_head, float* const* teacher_head, float* const* grad,
                                     const int* labels, int N, float obj_temperature, int n_valid, int n_fg, const int* n_valid_fg_dev,
                                     int do_obj, int do_reg, float grad_scale, float* loss2, aldi_stream_t stream) {
    if (!gm || !student_head || !teacher_head || !labels || !loss2) return aldi_set_error_msg(ALDI_ERR_ARG, "rpn_distill_loss: null pointer");
    DGeom g;
    g.nl = gm->num_levels; g.A = gm->A; g.C = gm->C; g.sumA = gm->off[gm->num_levels]; g.N = N;
    for (int l = 0; l < ALDI_MAX_LEVELS; ++l) {
        g.H[l] = gm->H[l]; g.W[l] = gm->W[l]; g.off[l] = gm->off[l];
        g.s[l] = student_head[l]; g.t[l] = teacher_head[l]; g.g[l] = grad ? grad[l] : nullptr;
    }
    g.off[ALDI_MAX_LEVELS] = gm->off[ALDI_MAX_LEVELS];
    long tot = (long)N * g.sumA;
    float inv_valid = n_valid > 0 ? 1.f / (float)n_valid : 0.f;
    float inv_fg4 = n_fg > 0 ? 1.f / (float)(4 * n_fg) : 0.f;
    hipLaunchKernelGGL(rpn_distill_kernel, dim3((int)((tot + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), g, labels,
                       1.f / obj_temperature, inv_valid, inv_fg4, n_valid_fg_dev, do_obj, do_reg, grad_scale, loss2);
    ALDI_CHECK_LAUNCH();
    return ALDI_OK;
}

extern "C" int aldi_roih_distill_loss(const float* student_pred, const float* teacher_pred, int Cp, int K, int R, float cls_temperature,
                                      int kl, int do_cls, int do_reg, float grad_scale, float* grad, float* loss2, aldi_stream_t stream) {
    if (!student_pred || !teacher_pred || !loss2) return aldi_set_error_msg(ALDI_ERR_ARG, "roih_distill_loss: null pointer");
    if (R <= 0) return ALDI_OK;
    hipLaunchKernelGGL(roih_distill_kernel, dim3(cdiv(R, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), student_pred, teacher_pred, Cp, K, R,
                       1.f / cls_temperature, kl, do_cls, do_reg, grad_scale, grad, loss2);
    ALDI_CHECK_LAUNCH();
    return ALDI_OK;
}

extern "C" int aldi_domain_bce(const float* pred, int ld, int R, float label, float weight, float grad_scale, void* grad, float* loss,
                               int dtype, aldi_stream_t stream) {
    if (!pred || !loss || R <= 0) return aldi_set_error_msg(ALDI_ERR_ARG, "domain_bce: bad args");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == ALDI_BF16) hipLaunchKernelGGL(bce_const_kernel<bf16_t>, dim3(cdiv(R, 256)), dim3(256), 0, st, pred, ld, R, label, weight, grad_scale, (bf16_t*)grad, loss);
    else hipLaunchKernelGGL(bce_const_kernel<float>, dim3(cdiv(R, 256)), dim3(256), 0, st, pred, ld, R, label, weight, grad_scale, (float*)grad, loss);
    ALDI_CHECK_LAUNCH();
    return ALDI_OK;
}

extern "C" size_t aldi_avgpool_workspace(int N, int C) { return (size_t)N * kPoolSplits * (size_t)C * sizeof(float); }

extern "C" int aldi_avgpool(const void* x, void* y, int N, int HW, int C, int dtype, void* workspace, aldi_stream_t stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (workspace && HW >= 4 * kPoolSplits) {            // large maps: pixel ranges in parallel, then the sum of the partial sums
        float* part = static_cast<float*>(workspace);
        dim3 g2(cdiv(C, 64), kPoolSplits, N);
        if (dtype == ALDI_BF16) {
            hipLaunchKernelGGL(avgpool_part_kernel<bf16_t>, g2, dim3(256), 0, st, (const bf16_t*)x, part, HW, C);
            hipLaunchKernelGGL(avgpool_finish_kernel<bf16_t>, dim3(cdiv(N * C, 256)), dim3(256), 0, st, part, (bf16_t*)y, N, HW, C);
        } else {
            hipLaunchKernelGGL(avgpool_part_kernel<float>, g2, dim3(256), 0, st, (const float*)x, part, HW, C);
            hipLaunchKernelGGL(avgpool_finish_kernel<float>, dim3(cdiv(N * C, 256)), dim3(256), 0, st, part, (float*)y, N, HW, C);
        }
        ALDI_CHECK_LAUNCH();
        return ALDI_OK;
    }
    dim3 grid(cdiv(C, 64), N);
    if (dtype == ALDI_BF16) hipLaunchKernelGGL(avgpool_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y, HW, C);
    else hipLaunchKernelGGL(avgpool_kernel<float>, grid, dim3(256), 0, st, (const float*)x, (float*)y, HW, C);
    ALDI_CHECK_LAUNCH();
    return ALDI_OK;
}

extern "C" int aldi_avgpool_bwd(const void* gy, const void* act, void* gx, int N, int HW, int C, int dtype, aldi_stream_t stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    long total = (long)N * HW * (C / 4);
    int blocks = (int)((total + 255) / 256 > 16384 ? 16384 : (total + 255) / 256);
    if (dtype == ALDI_BF16) hipLaunchKernelGGL(avgpool_bwd_kernel<bf16_t>, dim3(blocks), dim3(256), 0, st, (const bf16_t*)gy, (const bf16_t*)act, (bf16_t*)gx, N, HW, C);
    else hipLaunchKernelGGL(avgpool_bwd_kernel<float>, dim3(blocks), dim3(256), 0, st, (const float*)gy, (const float*)act, (float*)gx, N, HW, C);
    ALDI_CHECK_LAUNCH();
    return ALDI_OK;
}
