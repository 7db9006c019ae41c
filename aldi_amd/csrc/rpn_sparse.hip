// Sparse backward of the RPN head.
//
// The gradient of the RPN losses (and of the RPN distillation losses) with respect to the head outputs is non-zero only
// at the positions the SAMPLED anchors select: 256 per image for the RPN losses, and for the distillation losses the
// <= 256 + 4 * 128 positions the reference's masks pick (aldi/distill.py:200-227), i.e. at <= 1024 * N of the
// N * sum(H_l * W_l) = 358 k pixel positions of the five levels.  Everything downstream of it inside
// the head -- the 1x1 heads' data/weight gradients, the ReLU mask, the shared 3x3 conv's weight gradient and its data
// gradient -- is a sum over those pixels only.  Detectron2 / autograd (reached from aldi/trainer.py:79) run them as
// dense convolutions over all five levels; here the active pixels are listed, their rows gathered into small dense
// matrices ([S][16] head gradient, [S][256] hidden activation, [S][9][256] im2col of the level feature), the GEMMs run
// on S <= 1024 N rows through the ordinary igemm / wgrad kernels, and the data gradient is scattered back into the fp32
// level gradients.  Exact: the dropped terms are products with zeros.
#include "common.h"
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

namespace {

struct SGeom {
    int nl, N;
    int H[ALDI_MAX_LEVELS], W[ALDI_MAX_LEVELS];
    int row0[ALDI_MAX_LEVELS + 1];        // first global row (pixel position) of each level: N * sum_{k<l} H_k W_k
    const float* ghead[ALDI_MAX_LEVELS];  // [N][H][W][Ch] fp32
    const void* hidden[ALDI_MAX_LEVELS];  // [N][H][W][Cf]  (ReLU output of the shared 3x3 conv)
    const void* feat[ALDI_MAX_LEVELS];    // [N][H][W][Cf]  (level feature = input of the 3x3 conv)
    float* gfeat[ALDI_MAX_LEVELS];        // [N][H][W][Cf] fp32 gradient accumulators
};

__device__ __forceinline__ int level_of(const SGeom& g, int row) {
    int l = 0;
#pragma unroll
    for (int k = 1; k < ALDI_MAX_LEVELS; ++k)
        if (k < g.nl && row >= g.row0[k]) l = k;
    return l;
}

// one thread per pixel position: any of the Ch gradient channels non-zero -> append the global row index
__global__ __launch_bounds__(256) void active_rows_kernel(SGeom g, int Ch, int cap, int* __restrict__ idx, int* __restrict__ count, int* __restrict__ err) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    bool act = false;
    if (row < g.row0[g.nl]) {
        const int l = level_of(g, row);
        const float4* p = reinterpret_cast<const float4*>(g.ghead[l] + (long)(row - g.row0[l]) * Ch);
        for (int c = 0; c < Ch / 4; ++c) {
            const float4 v = p[c];
            act = act || v.x != 0.f || v.y != 0.f || v.z != 0.f || v.w != 0.f;
        }
    }
    const unsigned long long bal = __ballot(act);
    if (bal == 0ull) return;
    const int lane = threadIdx.x & 63;
    int base = 0;
    if (lane == 0) base = atomicAdd(count, __popcll(bal));
    base = __shfl(base, 0, 64);
    if (act) {
        const int slot = base + __popcll(bal & ((1ull << lane) - 1ull));
        if (slot < cap) idx[slot] = row;
        else atomicOr(err, 2);              // more active pixels than sampled anchors can produce: caller's bound is wrong
    }
}

// block s: row s of G (head gradient in T), of Tm (hidden activation) and of X9 ([9][Cf] im2col of the level feature); rows past
// the count are zero-filled so that the GEMMs can run over all `cap` rows
template <typename T>
__global__ __launch_bounds__(256) void sparse_gather_kernel(SGeom g, int Ch, int Cf, int cap, const int* __restrict__ idx, const int* __restrict__ count,
                                                            T* __restrict__ G, T* __restrict__ Tm, T* __restrict__ X9) {
    constexpr int EP = Elem<T>::kPer16B;
    const int s = blockIdx.x;
    const int n_act = min(*count, cap);
    const bool live = s < n_act;
    int l = 0, n = 0, h = 0, w = 0;
    long pix = 0;
    if (live) {
        const int row = idx[s];
        l = level_of(g, row);
        pix = row - g.row0[l];
        const int hw = g.H[l] * g.W[l];
        n = (int)(pix / hw);
        const int r = (int)(pix - (long)n * hw);
        h = r / g.W[l]; w = r - h * g.W[l];
    }
    const uint4 zero = make_uint4(0u, 0u, 0u, 0u);
    if (threadIdx.x < Ch) {
        const float v = live ? g.ghead[l][pix * Ch + threadIdx.x] : 0.f;
        Elem<T>::st(G + (long)s * Ch + threadIdx.x, v);
    }
    const int cpr = Cf / EP;                       // 16-B chunks per feature row
    for (int c = threadIdx.x; c < cpr; c += blockDim.x) {
        uint4 v = zero;
        if (live) v = reinterpret_cast<const uint4*>(static_cast<const T*>(g.hidden[l]) + pix * Cf)[c];
        reinterpret_cast<uint4*>(Tm + (long)s * Cf)[c] = v;
    }
    for (int c = threadIdx.x; c < 9 * cpr; c += blockDim.x) {
        const int tap = c / cpr, cc = c - tap * cpr;
        const int hh = h + tap / 3 - 1, ww = w + tap % 3 - 1;
        uint4 v = zero;
        if (live && (unsigned)hh < (unsigned)g.H[l] && (unsigned)ww < (unsigned)g.W[l])
            v = reinterpret_cast<const uint4*>(static_cast<const T*>(g.feat[l]) + (((long)n * g.H[l] + hh) * g.W[l] + ww) * Cf)[cc];
        reinterpret_cast<uint4*>(X9 + (long)s * 9 * Cf)[c] = v;
    }
}

// bf16 gradient maps (the compute dtype's own: what the reference's autocast sums, aldi/trainer.py:79): packed two-channel atomics
typedef short s16x2_t __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void sparse_scatter_bf16_kernel(SGeom g, int Cf, int cap, const int* __restrict__ idx, const int* __restrict__ count,
                                                                  const bf16_t* __restrict__ Y) {
    const int s = blockIdx.x;
    if (s >= min(*count, cap)) return;
    const int row = idx[s];
    const int l = level_of(g, row);
    const long pix = row - g.row0[l];
    const int hw = g.H[l] * g.W[l];
    const int n = (int)(pix / hw);
    const int r = (int)(pix - (long)n * hw);
    const int h = r / g.W[l], w = r - h * g.W[l];
    const int half = Cf / 2;
    for (int e = threadIdx.x; e < 9 * half; e += blockDim.x) {
        const int tap = e / half, c2 = e - tap * half;
        const int hh = h + tap / 3 - 1, ww = w + tap % 3 - 1;
        if ((unsigned)hh >= (unsigned)g.H[l] || (unsigned)ww >= (unsigned)g.W[l]) continue;
        const unsigned v = *reinterpret_cast<const unsigned*>(Y + (long)s * 9 * Cf + tap * Cf + c2 * 2);
        if ((v & 0x7fff7fffu) == 0u) continue;
        bf16_t* dst = reinterpret_cast<bf16_t*>(g.gfeat[l]) + (((long)n * g.H[l] + hh) * g.W[l] + ww) * Cf + c2 * 2;
        __builtin_amdgcn_global_atomic_fadd_v2bf16((__attribute__((address_space(1))) s16x2_t*)dst, __builtin_bit_cast(s16x2_t, v));
    }
}

// block s: gfeat[l][pixel + (tap - centre)][ci] += Y[s][tap][ci] for the nine taps (fp32 atomics: neighbouring active pixels overlap)
template <typename T>
__global__ __launch_bounds__(256) void sparse_scatter_kernel(SGeom g, int Cf, int cap, const int* __restrict__ idx, const int* __restrict__ count,
                                                             const T* __restrict__ Y) {
    const int s = blockIdx.x;
    if (s >= min(*count, cap)) return;
    const int row = idx[s];
    const int l = level_of(g, row);
    const long pix = row - g.row0[l];
    const int hw = g.H[l] * g.W[l];
    const int n = (int)(pix / hw);
    const int r = (int)(pix - (long)n * hw);
    const int h = r / g.W[l], w = r - h * g.W[l];
    for (int e = threadIdx.x; e < 9 * Cf; e += blockDim.x) {
        const int tap = e / Cf, ci = e - tap * Cf;
        const int hh = h + tap / 3 - 1, ww = w + tap % 3 - 1;
        if ((unsigned)hh >= (unsigned)g.H[l] || (unsigned)ww >= (unsigned)g.W[l]) continue;
        const float v = Elem<T>::ld(Y + (long)s * 9 * Cf + e);
        if (v != 0.f) unsafeAtomicAdd(g.gfeat[l] + (((long)n * g.H[l] + hh) * g.W[l] + ww) * Cf + ci, v);
    }
}

int fill_geom(SGeom& g, const aldi_rpn_geom* gm, int N) {
    if (!gm || gm->num_levels < 1 || gm->num_levels > ALDI_MAX_LEVELS || N < 1) return aldi_set_error_msg(ALDI_ERR_ARG, "rpn_sparse: bad geometry");
    g.nl = gm->num_levels; g.N = N;
    long r = 0;
    for (int l = 0; l < ALDI_MAX_LEVELS; ++l) {
        g.row0[l] = (int)r;
        g.H[l] = l < g.nl ? gm->H[l] : 0; g.W[l] = l < g.nl ? gm->W[l] : 0;
        if (l < g.nl) r += (long)N * gm->H[l] * gm->W[l];
        g.ghead[l] = nullptr; g.hidden[l] = nullptr; g.feat[l] = nullptr; g.gfeat[l] = nullptr;
    }
    if (r > 0x7fffffffL) return aldi_set_error_msg(ALDI_ERR_ARG, "rpn_sparse: too many pixel positions");
    for (int l = g.nl; l <= ALDI_MAX_LEVELS; ++l) g.row0[l] = (int)r;
    return ALDI_OK;
}

}  // namespace

extern "C" int aldi_rpn_active_pixels(const aldi_rpn_geom* gm, float* const* ghead, int N, int cap, int* idx, int* count, int* err_flag,
                                      aldi_stream_t stream) {
    if (!ghead || !idx || !count || !err_flag || cap < 1) return aldi_set_error_msg(ALDI_ERR_ARG, "rpn_active_pixels: bad args");
    SGeom g;
    if (int rc = fill_geom(g, gm, N)) return rc;
    if (gm->C % 4) return aldi_set_error_msg(ALDI_ERR_ARG, "rpn_active_pixels: head channels must be a multiple of 4");
    for (int l = 0; l < g.nl; ++l) g.ghead[l] = ghead[l];
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipError_t e = hipMemsetAsync(count, 0, sizeof(int), st);
    if (e != hipSuccess) return aldi_set_error(e, __FILE__, __LINE__);
    hipLaunchKernelGGL(active_rows_kernel, dim3(cdiv(g.row0[g.nl], 256)), dim3(256), 0, st, g, gm->C, cap, idx, count, err_flag);
    ALDI_CHECK_LAUNCH();
    return ALDI_OK;
}

extern "C" int aldi_rpn_sparse_gather(const aldi_rpn_geom* gm, float* const* ghead, const void* const* hidden, const void* const* feat, int N, int Cf,
                                      int cap, const int* idx, const int* count, void* G, void* Tm, void* X9, int dtype, aldi_stream_t stream) {
    if (!ghead || !hidden || !feat || !idx || !count || !G || !Tm || !X9) return aldi_set_error_msg(ALDI_ERR_ARG, "rpn_sparse_gather: null pointer");
    SGeom g;
    if (int rc = fill_geom(g, gm, N)) return rc;
    const int ep = dtype == ALDI_BF16 ? 8 : 4;
    if (Cf % ep || gm->C > 256) return aldi_set_error_msg(ALDI_ERR_ARG, "rpn_sparse_gather: Cf must be a multiple of a 16-B chunk, head channels <= 256");
    for (int l = 0; l < g.nl; ++l) { g.ghead[l] = ghead[l]; g.hidden[l] = hidden[l]; g.feat[l] = feat[l]; }
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == ALDI_BF16) hipLaunchKernelGGL(sparse_gather_kernel<bf16_t>, dim3(cap), dim3(256), 0, st, g, gm->C, Cf, cap, idx, count, (bf16_t*)G, (bf16_t*)Tm, (bf16_t*)X9);
    else if (dtype == ALDI_F32) hipLaunchKernelGGL(sparse_gather_kernel<float>, dim3(cap), dim3(256), 0, st, g, gm->C, Cf, cap, idx, count, (float*)G, (float*)Tm, (float*)X9);
    else return aldi_set_error_msg(ALDI_ERR_ARG, "rpn_sparse_gather: bad dtype");
    ALDI_CHECK_LAUNCH();
    return ALDI_OK;
}

extern "C" int aldi_rpn_sparse_scatter(const aldi_rpn_geom* gm, void* const* gfeat, const void* Y, int N, int Cf, int cap, const int* idx,
                                       const int* count, int dtype, int grad_dtype, aldi_stream_t stream) {
    if (!gfeat || !Y || !idx || !count) return aldi_set_error_msg(ALDI_ERR_ARG, "rpn_sparse_scatter: null pointer");
    SGeom g;
    if (int rc = fill_geom(g, gm, N)) return rc;
    for (int l = 0; l < g.nl; ++l) {
        if (!gfeat[l]) return aldi_set_error_msg(ALDI_ERR_ARG, "rpn_sparse_scatter: null level gradient");
        g.gfeat[l] = static_cast<float*>(gfeat[l]);
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (grad_dtype == ALDI_BF16) {
        if (dtype != ALDI_BF16 || Cf % 2) return aldi_set_error_msg(ALDI_ERR_ARG, "rpn_sparse_scatter: bf16 gradient maps take bf16 rows, even channel count");
        hipLaunchKernelGGL(sparse_scatter_bf16_kernel, dim3(cap), dim3(256), 0, st, g, Cf, cap, idx, count, (const bf16_t*)Y);
    } else if (grad_dtype != ALDI_F32) return aldi_set_error_msg(ALDI_ERR_ARG, "rpn_sparse_scatter: bad gradient dtype");
    else if (dtype == ALDI_BF16) hipLaunchKernelGGL(sparse_scatter_kernel<bf16_t>, dim3(cap), dim3(256), 0, st, g, Cf, cap, idx, count, (const bf16_t*)Y);
    else if (dtype == ALDI_F32) hipLaunchKernelGGL(sparse_scatter_kernel<float>, dim3(cap), dim3(256), 0, st, g, Cf, cap, idx, count, (const float*)Y);
    else return aldi_set_error_msg(ALDI_ERR_ARG, "rpn_sparse_scatter: bad dtype");
    ALDI_CHECK_LAUNCH();
    return ALDI_OK;
}
