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

// The same mask from a triangular grid: one WAVE per 64 x 64 block (rb <= cb) of a batch item, four blocks per workgroup, no LDS -- lane j holds
// column box j and every row thread reads it through v_readlane (scalar operands); the square grid above launches 2 x the workgroups (half
// return at once) of one wave each and stages the column boxes through the LDS behind a barrier.  Identical arithmetic, identical bits.
static __global__ __launch_bounds__(256) void nms_mask_tri_kernel(const float4* __restrict__ boxes, const int* __restrict__ valid, const int* __restrict__ cat,
                                                                  const int* __restrict__ count, int cap, float thr, unsigned long long* __restrict__ mask) {
    const int b = blockIdx.y, lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int nb = cap / 64;
    int pidx = (int)blockIdx.x * 4 + wave;               // index into the row-major list of blocks (rb, cb >= rb)
    if (pidx >= nb * (nb + 1) / 2) return;
    int rb = 0;
    while (pidx >= nb - rb) { pidx -= nb - rb; ++rb; }   // (scalar: at most nb steps)
    const int cb = rb + pidx;
    const int n = count ? count[b] : cap;
    if (rb * 64 >= n || cb * 64 >= n) return;
    const int j = cb * 64 + lane;
    const bool jv = j < n && valid[(long)b * cap + j];
    const float4 cbx = jv ? boxes[(long)b * cap + j] : make_float4(0, 0, 0, 0);
    const int ccat = jv ? (cat ? cat[(long)b * cap + j] : 0) : -1;
    const int i = rb * 64 + lane;
    const bool iv = i < n && valid[(long)b * cap + i];
    const float4 bi = iv ? boxes[(long)b * cap + i] : make_float4(0, 0, 0, 0);
    const int ci = cat ? (iv ? cat[(long)b * cap + i] : 0) : 0;
    unsigned long long bits = 0;
#pragma unroll
    for (int t = 0; t < 64; ++t) {
        float4 cbt;
        cbt.x = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cbx.x), t));
        cbt.y = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cbx.y), t));
        cbt.z = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cbx.z), t));
        cbt.w = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cbx.w), t));
        const int ct = __builtin_amdgcn_readlane(ccat, t);
        const int jj = cb * 64 + t;
        if (iv && jj > i && ct == ci && nms_over(bi, cbt, thr)) bits |= 1ull << t;
    }
    if (i < n) mask[((long)b * cap + i) * (cap / 64) + cb] = bits;
}
static inline void nms_mask_launch(hipStream_t st, int B, const float4* boxes, const int* valid, const int* cat, const int* count, int cap, float thr,
                                   unsigned long long* mask, int tri) {
    const int nb = cap / 64;
    if (tri) hipLaunchKernelGGL(nms_mask_tri_kernel, dim3((nb * (nb + 1) / 2 + 3) / 4, B), dim3(256), 0, st, boxes, valid, cat, count, cap, thr, mask);
    else hipLaunchKernelGGL(nms_mask_kernel, dim3(nb, nb, B), dim3(64), 0, st, boxes, valid, cat, count, cap, thr, mask);
}

// OR over the 64 lanes of a wave by DPP (row rotations, then the two row broadcasts): ~7 VALU ops instead of a six-deep chain of
// LDS permutes per 32 bits.  The total is read from lane 63 and returned wave-uniform.
static __device__ __forceinline__ unsigned wave_or_u32(unsigned v) {
    int x = (int)v;
    x |= __builtin_amdgcn_update_dpp(0, x, 0xb1, 0xf, 0xf, false);      // quad_perm [1,0,3,2]
    x |= __builtin_amdgcn_update_dpp(0, x, 0x4e, 0xf, 0xf, false);      // quad_perm [2,3,0,1]
    x |= __builtin_amdgcn_update_dpp(0, x, 0x124, 0xf, 0xf, false);     // row_ror:4
    x |= __builtin_amdgcn_update_dpp(0, x, 0x128, 0xf, 0xf, false);     // row_ror:8
    x |= __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);     // row_bcast:15 -> rows 1, 3
    x |= __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);     // row_bcast:31 -> rows 2, 3
    return (unsigned)__builtin_amdgcn_readlane(x, 63);
}
// max over the 64 lanes of unsigned values (bit patterns of non-negative floats order like the floats), same DPP ladder
static __device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
    int x = (int)v;
#define ALDI_DPP_MAX(ctrl, rows) { const unsigned o_ = (unsigned)__builtin_amdgcn_update_dpp(0, x, ctrl, rows, 0xf, false); x = (int)((unsigned)x > o_ ? (unsigned)x : o_); }
    ALDI_DPP_MAX(0xb1, 0xf) ALDI_DPP_MAX(0x4e, 0xf) ALDI_DPP_MAX(0x124, 0xf) ALDI_DPP_MAX(0x128, 0xf) ALDI_DPP_MAX(0x142, 0xa) ALDI_DPP_MAX(0x143, 0xc)
#undef ALDI_DPP_MAX
    return (unsigned)__builtin_amdgcn_readlane(x, 63);
}
static __device__ __forceinline__ unsigned long long wave_or_u64(unsigned long long v) {
    return ((unsigned long long)wave_or_u32((unsigned)(v >> 32)) << 32) | wave_or_u32((unsigned)v);
}

// One 1024-thread workgroup per batch item; greedy resolution in 64-box chunks.  The chain is serial by nature, so the
// kernel is built around its latency: wave 0 resolves a chunk by fixed-point rounds of one wave-wide OR each, while ALL 16
// waves hold that chunk's suppression rows for the later words in registers -- loaded one chunk ahead,
// unconditionally (64 rows x words x 8 B), so no global load sits on the chain -- and fold the kept rows into
// `removed` with a 6-step wave OR-reduction.
template <int KMAX, int NW>   // 64-bit row words per thread: ceil((cap/64 - 1) / NW); NW waves per workgroup
static __global__ __launch_bounds__(64 * NW) void nms_scan_kernel(const unsigned long long* __restrict__ mask, const int* __restrict__ valid,
                                                               const int* __restrict__ count, int cap, int max_keep,
                                                               int* __restrict__ keep /*[B][cap]*/, int* __restrict__ keep_count) {
    extern __shared__ unsigned long long removed[];   // cap/64 words, then 2 x [kept word, nk] (double-buffered by chunk parity)
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = count ? count[b] : cap;
    const int words = cap / 64;
    unsigned long long* kept_sh = removed + words;
    for (int w = tid; w < words + 4; w += 64 * NW) removed[w] = 0;
    const int chunks = (n + 63) / 64;
    const unsigned long long* mrow = mask + (long)b * cap * words;
    // rows of chunk c + 2 are requested at the top of iteration c: a full iteration (two barriers) between the request and the
    // first use, so the ~2 us global-load latency is not paid once per chunk on the serial chain
    unsigned long long cur[KMAX], nxt[KMAX], nx2[KMAX], diag_cur = 0, diag_nxt = 0, diag_nx2 = 0;
    int v_cur = 0, v_nxt = 0, v_nx2 = 0;
    auto load_rows = [&](int c, unsigned long long* r, unsigned long long& diag, int& v) {
        const int i = c * 64 + lane;
        const bool in = i < n;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const int w = c + 1 + wave + k * NW;
            r[k] = (in && w < words) ? mrow[(long)i * words + w] : 0ull;
        }
        if (wave == 0) {
            diag = in ? mrow[(long)i * words + c] : 0ull;
            v = in ? valid[(long)b * cap + i] : 0;
        }
    };
    if (chunks > 0) load_rows(0, cur, diag_cur, v_cur);
    if (chunks > 1) load_rows(1, nxt, diag_nxt, v_nxt);
    __syncthreads();
    int nk = 0;
    for (int c = 0; c < chunks; ++c) {
        if (c + 2 < chunks) load_rows(c + 2, nx2, diag_nx2, v_nx2);
        if (wave == 0) {
            // Greedy resolution of the chunk as a fixed point: K = alive & ~S(K), S(K) = OR of the kept rows' suppression bits
            // (lane t holds row t).  Starting from K = alive the iterates alternate around the unique solution and pin down at
            // least one more box per round (box t depends on boxes < t only); a chunk needs as many rounds as its longest
            // suppression chain (a handful), each one wave-wide OR -- instead of 64 dependent scalar steps.
            const unsigned long long alive = ~(removed[c] | ~__ballot(v_cur != 0));
            unsigned long long kept = alive;
            for (int it = 0; it < 65; ++it) {
                const unsigned long long sup = wave_or_u64(((kept >> lane) & 1ull) ? diag_cur : 0ull);
                const unsigned long long next = alive & ~sup;
                if (next == kept) break;
                kept = next;
            }
            int kc = __popcll(kept);
            if (nk + kc > max_keep) {           // keep only the first (max_keep - nk) survivors
                int allow = max_keep - nk;
                unsigned long long kk = 0;
                for (int t = 0; t < 64 && allow > 0; ++t)
                    if ((kept >> t) & 1ull) { kk |= 1ull << t; --allow; }
                kept = kk;
                kc = __popcll(kept);
            }
            if ((kept >> lane) & 1ull) keep[(long)b * cap + nk + __popcll(kept & ((1ull << lane) - 1ull))] = c * 64 + lane;
            if (lane == 0) { kept_sh[2 * (c & 1)] = kept; kept_sh[2 * (c & 1) + 1] = (unsigned long long)(nk + kc); }
        }
        // ONE barrier per chunk.  Wave 0 needs no second one: the word it reads next, removed[c + 1], is the word it folds itself
        // below (w = c + 1 + wave), and every older contribution to it was made before its owner reached this barrier; the other
        // waves read the kept word of THIS parity while wave 0 may already be writing the other one.
        __syncthreads();
        const unsigned long long kept = kept_sh[2 * (c & 1)];
        nk = (int)kept_sh[2 * (c & 1) + 1];
        const bool mine = (kept >> lane) & 1ull;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const int w = c + 1 + wave + k * NW;
            if (w < words) {                                       // wave uniform
                const unsigned long long x = wave_or_u64(mine ? cur[k] : 0ull);
                if (lane == 0) removed[w] |= x;                    // word w is owned by exactly one wave
            }
        }
        if (nk >= max_keep) break;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) { cur[k] = nxt[k]; nxt[k] = nx2[k]; }
        diag_cur = diag_nxt; v_cur = v_nxt;
        diag_nxt = diag_nx2; v_nxt = v_nx2;
    }
    if (tid == 0) keep_count[b] = nk;
}

// launch helper: picks the per-thread row-word count for `cap`
static inline bool nms_scan_launch(hipStream_t st, int B, const unsigned long long* mask, const int* valid, const int* count, int cap, int max_keep,
                                   int* keep, int* keep_count) {
    const int words = cap / 64;
    const size_t sh = (size_t)(words + 4) * 8;
    // (16 waves x 2 words measured fastest: 77 us against 81 / 90 for 8 x 4 / 4 x 8 on the 2000-box lists)
    if (words <= 33) hipLaunchKernelGGL((nms_scan_kernel<2, 16>), dim3(B), dim3(1024), sh, st, mask, valid, count, cap, max_keep, keep, keep_count);
    else if (words <= 129) hipLaunchKernelGGL((nms_scan_kernel<8, 16>), dim3(B), dim3(1024), sh, st, mask, valid, count, cap, max_keep, keep, keep_count);
    else return false;
    return true;
}
