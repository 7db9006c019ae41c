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
// on S <= 1024 N rows through the ordinary igemm / wgrad kernels, and the data gradient is added back into the level
// gradients.  Exact: the dropped terms are products with zeros.  Reproducible: the pixel list is sorted and the final add is a
// gather per target pixel (no atomics anywhere), so two runs of a step give bit-identical gradients.
#include "common.h"

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

// Active-pixel list, in ascending row order (deterministic: the list order fixes the row order of G / Tm / X9, i.e. the summation
// order of the two weight-gradient GEMMs).  Two launches over segments of 4096 pixel positions, as the label compaction of
// rpn.hip: (1) a wave ballots 16 rows of 64 positions (any of the Ch gradient channels non-zero), keeps the 64-bit masks and
// counts them per segment; (2) every segment adds up the counts of the segments before it and writes its part of the list.
constexpr int kSegPix = 4096, kSegK = 16;              // 4 waves x 16 rows of 64 positions
__global__ __launch_bounds__(256) void active_mark_kernel(SGeom g, int Ch, unsigned long long* __restrict__ bits, int* __restrict__ segcnt) {
    __shared__ int wsum[4];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int total = g.row0[g.nl];
    const int r0 = blockIdx.x * kSegPix + w * (64 * kSegK);
    int cnt = 0;
    for (int k = 0; k < kSegK; ++k) {
        const int row = r0 + k * 64 + lane;
        bool act = false;
        if (row < total) {
            const int l = level_of(g, row);
            const float4* p = reinterpret_cast<const float4*>(g.ghead[l] + (long)(row - g.row0[l]) * Ch);
            for (int c = 0; c < Ch / 4; ++c) {
                const float4 v = p[c];
                act = act || v.x != 0.f || v.y != 0.f || v.z != 0.f || v.w != 0.f;
            }
        }
        const unsigned long long bal = __ballot(act);
        if (lane == 0 && r0 + k * 64 < total) bits[(r0 + k * 64) >> 6] = bal;
        cnt += __popcll(bal);
    }
    if (lane == 0) wsum[w] = cnt;
    __syncthreads();
    if (tid == 0) segcnt[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

__global__ __launch_bounds__(256) void active_write_kernel(int total, int segs, const unsigned long long* __restrict__ bits, const int* __restrict__ segcnt,
                                                           int cap, int* __restrict__ idx, int* __restrict__ count, int* __restrict__ err) {
    __shared__ int wsum[4];
    __shared__ int sbase;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, seg = blockIdx.x;
    if (w == 0) {
        int a = 0;
        for (int s_ = lane; s_ < seg; s_ += 64) a += segcnt[s_];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
        if (lane == 0) sbase = a;
    }
    const int r0 = seg * kSegPix + w * (64 * kSegK);
    // lane k < 16 holds the mask of this wave's row k; an exclusive prefix over the 16 popcounts gives every row's offset
    unsigned long long mine = 0ull;
    if (lane < kSegK && r0 + lane * 64 < total) mine = bits[(r0 + lane * 64) >> 6];
    const int pc = __popcll(mine);
    int incl = pc;
#pragma unroll
    for (int o = 1; o < kSegK; o <<= 1) {
        const int t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    const int wave_total = __shfl(incl, kSegK - 1, 64);
    if (lane == 0) wsum[w] = wave_total;
    __syncthreads();
    int base = sbase;
    for (int i = 0; i < w; ++i) base += wsum[i];
    const unsigned long long lt = (1ull << lane) - 1ull;
    bool over = false;
    for (int k = 0; k < kSegK; ++k) {
        const unsigned long long m = __shfl(mine, k, 64);
        const int off = __shfl(incl - pc, k, 64);
        if ((m >> lane) & 1ull) {
            const int slot = base + off + __popcll(m & lt);
            if (slot < cap) idx[slot] = r0 + k * 64 + lane;
            else over = true;                  // more active pixels than the caller's bound: the excess is dropped, err says so
        }
    }
    if (__ballot(over) != 0ull && lane == 0) atomicOr(err, 2);
    if (seg == segs - 1 && tid == 0) *count = sbase + wsum[0] + wsum[1] + wsum[2] + wsum[3];
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

// gfeat[l][t][ci] += sum over the active pixels q = t - d(tap) of Y[slot(q)][tap][ci]   (d(tap) = (tap/3-1, tap%3-1))
//
// Deterministic and atomic-free: every TARGET pixel t of the nine-neighbourhoods is finished by exactly one workgroup -- the one
// of the contributing active pixel with the lowest tap index -- which sums all (<= 9) contributions in tap order in fp32 and
// updates the map once (bf16 maps: the ROIAlign share already in the map + this sum, rounded once).  Workgroup s handles active
// pixel s: 81 threads look up the slots of the pixels around its nine targets (binary search in the sorted list), then a group
// of Cf/8 threads per target adds the rows.
template <typename T> struct Row8;
template <> struct Row8<bf16_t> {
    __device__ static __forceinline__ void add(const bf16_t* p, float a[8]) {
        const uint4 v = *reinterpret_cast<const uint4*>(p);
        const unsigned u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) { a[2 * q] += __uint_as_float(u[q] << 16); a[2 * q + 1] += __uint_as_float(u[q] & 0xffff0000u); }
    }
    __device__ static __forceinline__ void rmw(bf16_t* p, const float a[8]) {
        const uint4 v = *reinterpret_cast<const uint4*>(p);
        const unsigned u[4] = {v.x, v.y, v.z, v.w};
        uint4 o;
        unsigned* ou = reinterpret_cast<unsigned*>(&o);
#pragma unroll
        for (int q = 0; q < 4; ++q) ou[q] = pack2_bf16(__uint_as_float(u[q] << 16) + a[2 * q], __uint_as_float(u[q] & 0xffff0000u) + a[2 * q + 1]);
        *reinterpret_cast<uint4*>(p) = o;
    }
};
template <> struct Row8<float> {
    __device__ static __forceinline__ void add(const float* p, float a[8]) {
        const float4 v0 = reinterpret_cast<const float4*>(p)[0], v1 = reinterpret_cast<const float4*>(p)[1];
        a[0] += v0.x; a[1] += v0.y; a[2] += v0.z; a[3] += v0.w; a[4] += v1.x; a[5] += v1.y; a[6] += v1.z; a[7] += v1.w;
    }
    __device__ static __forceinline__ void rmw(float* p, const float a[8]) {
        float4 v0 = reinterpret_cast<float4*>(p)[0], v1 = reinterpret_cast<float4*>(p)[1];
        v0.x += a[0]; v0.y += a[1]; v0.z += a[2]; v0.w += a[3]; v1.x += a[4]; v1.y += a[5]; v1.z += a[6]; v1.w += a[7];
        reinterpret_cast<float4*>(p)[0] = v0; reinterpret_cast<float4*>(p)[1] = v1;
    }
};

template <typename T, typename GT>
__global__ __launch_bounds__(256) void sparse_apply_kernel(SGeom g, int Cf, int cap, const int* __restrict__ idx, const int* __restrict__ count,
                                                           const T* __restrict__ Y) {
    __shared__ int slot[9][9];
    const int s = blockIdx.x;
    const int n_act = min(*count, cap);
    if (s >= n_act) return;
    const int row = idx[s];
    const int l = level_of(g, row);
    const long pix = row - g.row0[l];
    const int H = g.H[l], W = g.W[l], hw = H * W;
    const int n = (int)(pix / hw);
    const int r = (int)(pix - (long)n * hw);
    const int h = r / W, w = r - h * W;
    const int tid = threadIdx.x;
    if (tid < 81) {
        const int tap = tid / 9, tp = tid - tap * 9;
        const int th = h + tap / 3 - 1, tw = w + tap % 3 - 1;           // target pixel of this tap
        const int qh = th - (tp / 3 - 1), qw = tw - (tp % 3 - 1);       // the pixel that reaches it through tap tp
        int found = -1;
        if ((unsigned)th < (unsigned)H && (unsigned)tw < (unsigned)W && (unsigned)qh < (unsigned)H && (unsigned)qw < (unsigned)W) {
            const int qrow = g.row0[l] + (n * H + qh) * W + qw;
            int lo = 0, hi = n_act;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (idx[mid] < qrow) lo = mid + 1; else hi = mid;
            }
            if (lo < n_act && idx[lo] == qrow) found = lo;
        }
        slot[tap][tp] = found;
    }
    __syncthreads();
    const int tpg = Cf / 8;                       // threads per target (8 channels each)
    const int groups = blockDim.x / tpg;
    const int gi = tid / tpg, c8 = (tid - gi * tpg) * 8;
    if (gi >= groups) return;
    for (int tap = gi; tap < 9; tap += groups) {
        const int th = h + tap / 3 - 1, tw = w + tap % 3 - 1;
        if ((unsigned)th >= (unsigned)H || (unsigned)tw >= (unsigned)W) continue;
        bool owner = true;
        for (int tp = 0; tp < tap; ++tp) owner = owner && slot[tap][tp] < 0;
        if (!owner) continue;
        float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int tp = tap; tp < 9; ++tp) {
            const int sq = slot[tap][tp];
            if (sq >= 0) Row8<T>::add(Y + ((long)sq * 9 + tp) * Cf + c8, a);
        }
        Row8<GT>::rmw(reinterpret_cast<GT*>(g.gfeat[l]) + (((long)n * H + th) * W + tw) * Cf + c8, a);
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

extern "C" size_t aldi_rpn_active_pixels_workspace(const aldi_rpn_geom* gm, int N) {
    SGeom g;
    if (fill_geom(g, gm, N)) return 0;
    const size_t rows = (size_t)g.row0[g.nl];
    return ((rows + 63) / 64) * sizeof(unsigned long long) + (size_t)cdiv((long)rows, kSegPix) * sizeof(int) + 64;
}

extern "C" int aldi_rpn_active_pixels(const aldi_rpn_geom* gm, float* const* ghead, int N, int cap, int* idx, int* count, void* workspace,
                                      int* err_flag, aldi_stream_t stream) {
    if (!ghead || !idx || !count || !workspace || !err_flag || cap < 1) return aldi_set_error_msg(ALDI_ERR_ARG, "rpn_active_pixels: bad args");
    SGeom g;
    if (int rc = fill_geom(g, gm, N)) return rc;
    if (gm->C % 4) return aldi_set_error_msg(ALDI_ERR_ARG, "rpn_active_pixels: head channels must be a multiple of 4");
    for (int l = 0; l < g.nl; ++l) g.ghead[l] = ghead[l];
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int total = g.row0[g.nl], segs = cdiv(total, kSegPix);
    unsigned long long* bits = static_cast<unsigned long long*>(workspace);
    int* segcnt = reinterpret_cast<int*>(bits + (total + 63) / 64);
    hipLaunchKernelGGL(active_mark_kernel, dim3(segs), dim3(256), 0, st, g, gm->C, bits, segcnt);
    ALDI_CHECK_LAUNCH();
    hipLaunchKernelGGL(active_write_kernel, dim3(segs), dim3(256), 0, st, total, segs, bits, segcnt, cap, idx, count, err_flag);
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
    if (Cf % 8 || Cf > 2048) return aldi_set_error_msg(ALDI_ERR_ARG, "rpn_sparse_scatter: Cf must be a multiple of 8, <= 2048");
    if (grad_dtype == ALDI_BF16) {
        if (dtype != ALDI_BF16) return aldi_set_error_msg(ALDI_ERR_ARG, "rpn_sparse_scatter: bf16 gradient maps take bf16 rows");
        hipLaunchKernelGGL((sparse_apply_kernel<bf16_t, bf16_t>), dim3(cap), dim3(256), 0, st, g, Cf, cap, idx, count, (const bf16_t*)Y);
    } else if (grad_dtype != ALDI_F32) return aldi_set_error_msg(ALDI_ERR_ARG, "rpn_sparse_scatter: bad gradient dtype");
    else if (dtype == ALDI_BF16) hipLaunchKernelGGL((sparse_apply_kernel<bf16_t, float>), dim3(cap), dim3(256), 0, st, g, Cf, cap, idx, count, (const bf16_t*)Y);
    else if (dtype == ALDI_F32) hipLaunchKernelGGL((sparse_apply_kernel<float, float>), dim3(cap), dim3(256), 0, st, g, Cf, cap, idx, count, (const float*)Y);
    else return aldi_set_error_msg(ALDI_ERR_ARG, "rpn_sparse_scatter: bad dtype");
    ALDI_CHECK_LAUNCH();
    return ALDI_OK;
}
