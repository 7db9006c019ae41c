// ROI-head side: proposal labelling/sampling glue, ROIAlign forward/backward, box-head
// losses (fwd+bwd) and the inference post-processing that turns teacher outputs into
// pseudo-labels (softmax, score threshold, per-class NMS, top-k, pseudo-label threshold).
//
// Replaces detectron2 StandardROIHeads.label_and_sample_proposals, ROIPooler + torchvision
// roi_align (aligned=True, sampling_ratio=0), FastRCNNOutputLayers.losses / inference and
// the reference's own pseudo-label filter (aldi/pseudolabeler.py:51-67); call sites
// aldi/distill.py:157,162 and aldi/pseudolabeler.py:21.
#include "common.h"
#include "loss_rows.h"
#include "sortscan.h"
#include "nms.h"
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>


namespace {

// cand = proposals (count[n]) followed by the image's GT boxes (if any)  [add_ground_truth_to_proposals]
__global__ void roi_append_gt_kernel(const float4* __restrict__ props, const int* __restrict__ pcount, int P,
                                     const float4* __restrict__ gt, const int* __restrict__ gcount, int Gmax,
                                     float4* __restrict__ cand, int* __restrict__ ccount, int Pcap) {
    const int n = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int pc = pcount[n], gc = gcount[n];
    if (i == 0) ccount[n] = pc + gc;
    if (i >= Pcap) return;
    float4 v = make_float4(0, 0, 0, 0);
    if (i < pc) v = props[(long)n * P + i];
    else if (i < pc + gc) v = gt[(long)n * Gmax + (i - pc)];
    cand[(long)n * Pcap + i] = v;
}

// gt_classes per candidate: matched class for fg, K for bg, -2 padding
__global__ void roi_classes_kernel(const int* __restrict__ labels, const int* __restrict__ best_idx, const int* __restrict__ gt_classes,
                                   const int* __restrict__ gcount, int Gmax, int L, int K, int* __restrict__ cls) {
    const int n = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= L) return;
    const int lab = labels[(long)n * L + i];
    int c;
    if (lab == -2) c = -2;
    else if (gcount[n] == 0) c = K;
    else c = lab == 1 ? gt_classes[n * Gmax + best_idx[(long)n * L + i]] : K;
    cls[(long)n * L + i] = c;
}

// sampled_idxs = cat(fg_list[sel_fg], bg_list[sel_bg]); rows of image n start at row_off[n]
__global__ void roi_gather_kernel(const float4* __restrict__ cand, const int* __restrict__ cls, const int* __restrict__ best_idx, int L,
                                  const int* __restrict__ lists, const int* __restrict__ sel, const int* __restrict__ nsel, int S,
                                  const int* __restrict__ row_off, const float4* __restrict__ gt, const int* __restrict__ gcount, int Gmax,
                                  float* __restrict__ rois /*[R][5]*/, int* __restrict__ r_cls, float4* __restrict__ r_gt, int* __restrict__ r_idx) {
    const int n = blockIdx.y;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int nf = nsel[n * 2], nb = nsel[n * 2 + 1];
    if (j >= nf + nb) return;
    const int kind = j < nf ? 0 : 1;
    const int pos = sel[(n * 2 + kind) * S + (kind ? j - nf : j)];
    const int idx = lists[((long)n * 2 + kind) * L + pos];
    const int row = row_off[n] + j;
    const float4 b = cand[(long)n * L + idx];
    rois[row * 5 + 0] = (float)n;
    rois[row * 5 + 1] = b.x; rois[row * 5 + 2] = b.y; rois[row * 5 + 3] = b.z; rois[row * 5 + 4] = b.w;
    r_cls[row] = cls[(long)n * L + idx];
    r_gt[row] = gcount[n] > 0 ? gt[(long)n * Gmax + best_idx[(long)n * L + idx]] : make_float4(0, 0, 0, 0);
    r_idx[row] = idx;
}

// rois for inference: every proposal of every image, rows packed as [N][P] (padding rows get b = -1)
__global__ void roi_from_proposals_kernel(const float4* __restrict__ props, const int* __restrict__ pcount, int P, float* __restrict__ rois) {
    const int n = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const long row = (long)n * P + i;
    const bool ok = i < pcount[n];
    const float4 b = ok ? props[row] : make_float4(0, 0, 0, 0);
    rois[row * 5 + 0] = ok ? (float)n : -1.f;
    rois[row * 5 + 1] = b.x; rois[row * 5 + 2] = b.y; rois[row * 5 + 3] = b.z; rois[row * 5 + 4] = b.w;
}

struct Feats {
    const void* f[4];
    float* g[4];
    unsigned gbytes[4];
    int H[4], W[4];
    float scale[4];
    int C;
};

__device__ __forceinline__ int roi_level(float x1, float y1, float x2, float y2) {
    // floor(4 + log2(sqrt(area)/224 + 1e-8)) clamped to [2,5], minus 2
    float s = sqrtf((x2 - x1) * (y2 - y1));
    float lv = floorf(4.f + log2f(s / 224.f + 1e-8f));
    lv = fminf(fmaxf(lv, 2.f), 5.f);
    return (int)lv - 2;
}

struct Bilin { int lo, hi; float l, h; bool dead; };
__device__ __forceinline__ Bilin bilin_prep(float v, int size) {
    Bilin b;
    b.dead = v < -1.0f || v > (float)size;
    if (v <= 0.f) v = 0.f;
    int lo = (int)v, hi;
    if (lo >= size - 1) { hi = lo = size - 1; v = (float)lo; }
    else hi = lo + 1;
    b.lo = lo; b.hi = hi;
    b.l = v - (float)lo;
    b.h = 1.f - b.l;
    return b;
}

// grid (R, P): block = one output row (ph) of one ROI; thread = channel (C == blockDim.x)
template <typename T, bool BWD>
__global__ __launch_bounds__(256) void roialign_kernel(Feats ft, const float* __restrict__ rois, int P, T* __restrict__ pooled /*[R][P][P][C]*/) {
    const int r = blockIdx.x, ph = blockIdx.y, c = threadIdx.x;
    const float* rp = rois + (long)r * 5;
    const int b = (int)rp[0];
    if (b < 0) {
        if (!BWD) for (int pw = 0; pw < P; ++pw) Elem<T>::st(pooled + (((long)r * P + ph) * P + pw) * ft.C + c, 0.f);
        return;
    }
    const int l = roi_level(rp[1], rp[2], rp[3], rp[4]);
    const int H = ft.H[l], W = ft.W[l];
    const float sc = ft.scale[l];
    const float x1 = rp[1] * sc - 0.5f, y1 = rp[2] * sc - 0.5f, x2 = rp[3] * sc - 0.5f, y2 = rp[4] * sc - 0.5f;
    const float rw = x2 - x1, rh = y2 - y1;
    const float bw = rw / (float)P, bh = rh / (float)P;
    const int gh = (int)ceilf(rh / (float)P), gw = (int)ceilf(rw / (float)P);
    const float count = (float)max(gh * gw, 1);
    const T* F = static_cast<const T*>(ft.f[l]) + (long)b * H * W * ft.C + c;
    if constexpr (!BWD) {
        for (int pw = 0; pw < P; ++pw) {
            T* op = pooled + (((long)r * P + ph) * P + pw) * ft.C + c;
            float acc = 0.f;
            for (int iy = 0; iy < gh; ++iy) {
                const float y = y1 + (float)ph * bh + ((float)iy + 0.5f) * bh / (float)gh;
                const Bilin by = bilin_prep(y, H);
                for (int ix = 0; ix < gw; ++ix) {
                    const float x = x1 + (float)pw * bw + ((float)ix + 0.5f) * bw / (float)gw;
                    const Bilin bx = bilin_prep(x, W);
                    if (by.dead || bx.dead) continue;
                    const float w1 = by.h * bx.h, w2 = by.h * bx.l, w3 = by.l * bx.h, w4 = by.l * bx.l;
                    const long o1 = ((long)by.lo * W + bx.lo) * ft.C, o2 = ((long)by.lo * W + bx.hi) * ft.C;
                    const long o3 = ((long)by.hi * W + bx.lo) * ft.C, o4 = ((long)by.hi * W + bx.hi) * ft.C;
                    acc += w1 * Elem<T>::ld(F + o1) + w2 * Elem<T>::ld(F + o2) + w3 * Elem<T>::ld(F + o3) + w4 * Elem<T>::ld(F + o4);
                }
            }
            Elem<T>::st(op, acc / count);
        }
    } else {
        // Backward.  For one sample row (iy) the sample columns sweep left -> right over all P bins, touching the pixel
        // pairs (xlo, xlo+1) of the two feature rows (ylo, yhi): keep that 2x2 window in registers and flush a column with
        // ONE atomic per row when the sweep leaves it -- ~2 atomics per touched pixel instead of 4 per sample.
        // fire-and-forget buffer atomics on this level's gradient map (block-uniform descriptor); a zero contribution
        // or an out-of-map column becomes an out-of-range offset, which the hardware drops -- no branch, no wait
        const __amdgpu_buffer_rsrc_t rg = make_rsrc_uniform(ft.g[l], ft.gbytes[l]);
        const unsigned gbase = ((unsigned)b * (unsigned)(H * W) * (unsigned)ft.C + (unsigned)c) * 4u;
        float gbin[8];
#pragma unroll
        for (int pw = 0; pw < 8; ++pw)
            gbin[pw] = pw < P ? Elem<T>::ld(pooled + (((long)r * P + ph) * P + pw) * ft.C + c) / count : 0.f;
        for (int iy = 0; iy < gh; ++iy) {
            const float y = y1 + (float)ph * bh + ((float)iy + 0.5f) * bh / (float)gh;
            const Bilin by = bilin_prep(y, H);
            if (by.dead) continue;
            const unsigned Glo = gbase + (unsigned)(by.lo * W) * (unsigned)ft.C * 4u;
            const unsigned Ghi = gbase + (unsigned)(by.hi * W) * (unsigned)ft.C * 4u;
            const unsigned cstep = (unsigned)ft.C * 4u;
            int cur = -1;                       // window covers pixel columns cur, cur+1
            float a0l = 0.f, a0h = 0.f, a1l = 0.f, a1h = 0.f;   // [column 0/1][row lo/hi]
            auto flush0 = [&]() {
                buf_atomic_add_f32(rg, a0l != 0.f ? Glo + (unsigned)cur * cstep : kBufOOB, a0l);
                buf_atomic_add_f32(rg, a0h != 0.f ? Ghi + (unsigned)cur * cstep : kBufOOB, a0h);
            };
            auto flush1 = [&]() {
                buf_atomic_add_f32(rg, a1l != 0.f ? Glo + (unsigned)(cur + 1) * cstep : kBufOOB, a1l);
                buf_atomic_add_f32(rg, a1h != 0.f ? Ghi + (unsigned)(cur + 1) * cstep : kBufOOB, a1h);
            };
            for (int pw = 0; pw < P; ++pw) {
                const float gv = gbin[pw];
                for (int ix = 0; ix < gw; ++ix) {
                    const float x = x1 + (float)pw * bw + ((float)ix + 0.5f) * bw / (float)gw;
                    const Bilin bx = bilin_prep(x, W);
                    if (bx.dead) continue;
                    if (bx.lo != cur) {         // the sweep moved on (lo is non-decreasing along the row)
                        if (cur >= 0) {
                            flush0();
                            if (bx.lo == cur + 1) { a0l = a1l; a0h = a1h; }
                            else { flush1(); a0l = 0.f; a0h = 0.f; }
                        }
                        a1l = 0.f; a1h = 0.f;
                        cur = bx.lo;
                    }
                    const float gl = gv * by.h, gh_ = gv * by.l;
                    a0l += gl * bx.h; a0h += gh_ * bx.h;
                    if (bx.hi != bx.lo) { a1l += gl * bx.l; a1h += gh_ * bx.l; }
                    else { a0l += gl * bx.l; a0h += gh_ * bx.l; }     // clamped at the right border: both taps hit the same pixel
                }
            }
            if (cur >= 0) {
                flush0();
                if (cur + 1 < W) flush1();
            }
        }
    }
}

// Forward, vector form: a lane owns one 16-B group of channels (8 bf16 / 4 fp32) of one output bin, so the bilinear
// bookkeeping (identical for every channel) is paid once per 16 B of feature data instead of once per element, and a tap
// is fetched as 512 contiguous bytes per bin.  grid (R, P): block = one output row of one ROI, bins side by side.
template <typename T>
__global__ __launch_bounds__(256) void roialign_fwd_vec_kernel(Feats ft, const float* __restrict__ rois, int P, T* __restrict__ pooled /*[R][P][P][C]*/) {
    constexpr int EP = Elem<T>::kPer16B;
    const int r = blockIdx.x, ph = blockIdx.y;
    const int lpb = ft.C / EP;                       // lanes per bin (32 / 64)
    const int cg = threadIdx.x % lpb, bin0 = threadIdx.x / lpb, nbin = 256 / lpb;
    const float* rp = rois + (long)r * 5;
    const int b = (int)rp[0];
    if (b < 0) {
        for (int pw = bin0; pw < P; pw += nbin)
            *reinterpret_cast<uint4*>(pooled + (((long)r * P + ph) * P + pw) * ft.C + cg * EP) = make_uint4(0, 0, 0, 0);
        return;
    }
    const int l = roi_level(rp[1], rp[2], rp[3], rp[4]);
    const int H = ft.H[l], W = ft.W[l];
    const float sc = ft.scale[l];
    const float x1 = rp[1] * sc - 0.5f, y1 = rp[2] * sc - 0.5f, x2 = rp[3] * sc - 0.5f, y2 = rp[4] * sc - 0.5f;
    const float rw = x2 - x1, rh = y2 - y1;
    const float bw = rw / (float)P, bh = rh / (float)P;
    const int gh = (int)ceilf(rh / (float)P), gw = (int)ceilf(rw / (float)P);
    const float count = (float)max(gh * gw, 1);
    const T* F = static_cast<const T*>(ft.f[l]) + (long)b * H * W * ft.C + cg * EP;
    for (int pw = bin0; pw < P; pw += nbin) {
        float acc[EP];
#pragma unroll
        for (int e = 0; e < EP; ++e) acc[e] = 0.f;
        for (int iy = 0; iy < gh; ++iy) {
            const float y = y1 + (float)ph * bh + ((float)iy + 0.5f) * bh / (float)gh;
            const Bilin by = bilin_prep(y, H);
            for (int ix = 0; ix < gw; ++ix) {
                const float x = x1 + (float)pw * bw + ((float)ix + 0.5f) * bw / (float)gw;
                const Bilin bx = bilin_prep(x, W);
                if (by.dead || bx.dead) continue;
                const float w1 = by.h * bx.h, w2 = by.h * bx.l, w3 = by.l * bx.h, w4 = by.l * bx.l;
                const uint4 q1 = *reinterpret_cast<const uint4*>(F + ((long)by.lo * W + bx.lo) * ft.C);
                const uint4 q2 = *reinterpret_cast<const uint4*>(F + ((long)by.lo * W + bx.hi) * ft.C);
                const uint4 q3 = *reinterpret_cast<const uint4*>(F + ((long)by.hi * W + bx.lo) * ft.C);
                const uint4 q4 = *reinterpret_cast<const uint4*>(F + ((long)by.hi * W + bx.hi) * ft.C);
                const uint32_t* u1 = reinterpret_cast<const uint32_t*>(&q1);
                const uint32_t* u2 = reinterpret_cast<const uint32_t*>(&q2);
                const uint32_t* u3 = reinterpret_cast<const uint32_t*>(&q3);
                const uint32_t* u4 = reinterpret_cast<const uint32_t*>(&q4);
                if constexpr (EP == 8) {
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        // same association as the scalar form: ((w1*f1 + w2*f2) + w3*f3) + w4*f4, then += into acc
                        acc[2 * d] += w1 * __uint_as_float(u1[d] << 16) + w2 * __uint_as_float(u2[d] << 16) + w3 * __uint_as_float(u3[d] << 16) +
                                      w4 * __uint_as_float(u4[d] << 16);
                        acc[2 * d + 1] += w1 * __uint_as_float(u1[d] & 0xffff0000u) + w2 * __uint_as_float(u2[d] & 0xffff0000u) +
                                          w3 * __uint_as_float(u3[d] & 0xffff0000u) + w4 * __uint_as_float(u4[d] & 0xffff0000u);
                    }
                } else {
#pragma unroll
                    for (int d = 0; d < 4; ++d)
                        acc[d] += w1 * __uint_as_float(u1[d]) + w2 * __uint_as_float(u2[d]) + w3 * __uint_as_float(u3[d]) + w4 * __uint_as_float(u4[d]);
                }
            }
        }
        T* op = pooled + (((long)r * P + ph) * P + pw) * ft.C + cg * EP;
        if constexpr (EP == 8) {
            uint4 o;
            o.x = pack2_bf16(acc[0] / count, acc[1] / count); o.y = pack2_bf16(acc[2] / count, acc[3] / count);
            o.z = pack2_bf16(acc[4] / count, acc[5] / count); o.w = pack2_bf16(acc[6] / count, acc[7] / count);
            *reinterpret_cast<uint4*>(op) = o;
        } else {
            *reinterpret_cast<float4*>(op) = make_float4(acc[0] / count, acc[1] / count, acc[2] / count, acc[3] / count);
        }
    }
}

template <typename T> struct Raw8;                      // 8 consecutive channels as loaded; unpacked when consumed
template <> struct Raw8<bf16_t> {
    uint4 v;
    __device__ __forceinline__ void load(const bf16_t* p) { v = *reinterpret_cast<const uint4*>(p); }
    static __device__ __forceinline__ void store_zero(bf16_t* p) { *reinterpret_cast<uint4*>(p) = make_uint4(0, 0, 0, 0); }
    static __device__ __forceinline__ void store(bf16_t* p, const float* o) {
        *reinterpret_cast<uint4*>(p) = make_uint4(pack2_bf16(o[0], o[1]), pack2_bf16(o[2], o[3]), pack2_bf16(o[4], o[5]), pack2_bf16(o[6], o[7]));
    }
    __device__ __forceinline__ void unpack(float* o) const {
        o[0] = __uint_as_float(v.x << 16); o[1] = __uint_as_float(v.x & 0xffff0000u);
        o[2] = __uint_as_float(v.y << 16); o[3] = __uint_as_float(v.y & 0xffff0000u);
        o[4] = __uint_as_float(v.z << 16); o[5] = __uint_as_float(v.z & 0xffff0000u);
        o[6] = __uint_as_float(v.w << 16); o[7] = __uint_as_float(v.w & 0xffff0000u);
    }
};
template <> struct Raw8<float> {
    float4 a, b;
    __device__ __forceinline__ void load(const float* p) { a = *reinterpret_cast<const float4*>(p); b = *reinterpret_cast<const float4*>(p + 4); }
    static __device__ __forceinline__ void store_zero(float* p) {
        *reinterpret_cast<float4*>(p) = make_float4(0.f, 0.f, 0.f, 0.f); *reinterpret_cast<float4*>(p + 4) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    static __device__ __forceinline__ void store(float* p, const float* o) {
        *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]); *reinterpret_cast<float4*>(p + 4) = make_float4(o[4], o[5], o[6], o[7]);
    }
    __device__ __forceinline__ void unpack(float* o) const { o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w; }
};

// acc[0..7] += w * u[0..7] as four fused packed multiply-adds (v_pk_fma_f32).  The file is built with -ffp-contract=off, where `acc += w * u` is a
// v_pk_mul_f32 + v_pk_add_f32 pair; the RoIAlign kernels are bound by VALU issue (r06: forward 2048 ROIs x 4 waves x ~3 700 instructions = 58 of its
// 62 us).  One rounding instead of two per term.
__device__ __forceinline__ void fma8(float* acc, const float w, const float* u) {
#pragma unroll
    for (int i = 0; i < 8; i += 2) {
        const f32x2_t r2 = __builtin_elementwise_fma(f32x2_t{w, w}, f32x2_t{u[i], u[i + 1]}, f32x2_t{acc[i], acc[i + 1]});
        acc[i] = r2[0]; acc[i + 1] = r2[1];
    }
}

// Forward, separable form (what aldi_roialign runs): the adaptive sample grid of a bin is a product grid, so
//   out[ph][pw][c] = 1/count * sum_yy rowc[ph][yy] * ( sum_xx colc[pw][xx] * f[yy][xx][c] )
// with rowc / colc the summed bilinear weights the bin's sample rows / columns put on feature row yy / column xx (the same
// bilin_prep, clamps and dead samples as above: a sample is dead when either coordinate is).  One workgroup per ROI; a thread
// owns (bin column pw, 8 channels) and walks the footprint rows once: per row it reads the ~bw + 2 pixels under its bin column
// (16 B each) instead of 4 corners per sample -- a 19 x 10-pixel ROI reads ~215 KB instead of ~600 KB, and the bilinear
// bookkeeping is done once per ROI in LDS tables instead of once per (sample, 16 B).  Sums associate differently from the
// sample-by-sample form (same terms; fp32 differences of a few ulps).
template <int V> struct SepTag { static constexpr int value = V; };
constexpr int kSepMaxH = 208, kSepMaxW = 344;      // footprint bounds: the p2 map of a 1333 x 800 image is 200 x 336
template <typename T>
__global__ __launch_bounds__(256) void roialign_fwd_sep_kernel(Feats ft, const float* __restrict__ rois, int P, T* __restrict__ pooled /*[R][P][P][C]*/) {
    __shared__ float rowc[7][kSepMaxH];
    __shared__ float colc[7][kSepMaxW];
    __shared__ int xlo_s[8], xhi_s[8];
    const int r = blockIdx.x, tid = threadIdx.x;
    const int pw = tid >> 5, c8 = tid & 31;            // bin column (7 = idle), channels c8 * 8 ... + 7
    const int C = ft.C;
    const float* rp = rois + (long)r * 5;
    const int b = (int)rp[0];
    T* out_base = pooled + (long)r * P * P * C + c8 * 8;
    if (b < 0) {
        if (pw < P)
            for (int ph = 0; ph < P; ++ph) Raw8<T>::store_zero(out_base + (long)(ph * P + pw) * C);
        return;
    }
    const int l = roi_level(rp[1], rp[2], rp[3], rp[4]);
    const int H = ft.H[l], W = ft.W[l];
    const float sc = ft.scale[l];
    const float x1 = rp[1] * sc - 0.5f, y1 = rp[2] * sc - 0.5f, x2 = rp[3] * sc - 0.5f, y2 = rp[4] * sc - 0.5f;
    const float rw = x2 - x1, rh = y2 - y1;
    const float bw = rw / (float)P, bh = rh / (float)P;
    const int gh = (int)ceilf(rh / (float)P), gw = (int)ceilf(rw / (float)P);
    const float inv_count = 1.f / (float)max(gh * gw, 1);
    // footprint (every live sample's two taps lie inside it)
    const int fy0 = min(max((int)floorf(y1) - 1, 0), H - 1), fy1 = min(max((int)floorf(y2) + 2, 0), H - 1);
    const int fx0 = min(max((int)floorf(x1) - 1, 0), W - 1), fx1 = min(max((int)floorf(x2) + 2, 0), W - 1);
    const int nrow = fy1 - fy0 + 1, ncol = fx1 - fx0 + 1;
    for (int e = tid; e < 7 * nrow; e += 256) {
        const int ph = e / nrow, yy = fy0 + e - ph * nrow;
        float a = 0.f;
        if (ph < P)
            for (int iy = 0; iy < gh; ++iy) {
                const Bilin by = bilin_prep(y1 + (float)ph * bh + ((float)iy + 0.5f) * bh / (float)gh, H);
                if (by.dead) continue;
                if (by.lo == yy) a += by.h;
                if (by.hi == yy) a += by.l;
            }
        rowc[ph][yy - fy0] = a;
    }
    for (int e = tid; e < 7 * ncol; e += 256) {
        const int q = e / ncol, xx = fx0 + e - q * ncol;
        float a = 0.f;
        if (q < P)
            for (int ix = 0; ix < gw; ++ix) {
                const Bilin bx = bilin_prep(x1 + (float)q * bw + ((float)ix + 0.5f) * bw / (float)gw, W);
                if (bx.dead) continue;
                if (bx.lo == xx) a += bx.h;
                if (bx.hi == xx) a += bx.l;
            }
        colc[q][xx - fx0] = a;
    }
    __syncthreads();
    if (tid < 7) {                                     // pixels with weight under bin column `tid`: a contiguous run
        int lo = ncol, hi = -1;
        for (int x = 0; x < ncol; ++x)
            if (colc[tid][x] != 0.f) { lo = min(lo, x); hi = x; }
        xlo_s[tid] = lo; xhi_s[tid] = hi;
    }
    __syncthreads();
    if (pw >= P) return;
    const int xlo = xlo_s[pw], xhi = xhi_s[pw];
    float acc[7][8];
#pragma unroll
    for (int ph = 0; ph < 7; ++ph)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[ph][i] = 0.f;
    const T* F = static_cast<const T*>(ft.f[l]) + (long)b * H * W * C + c8 * 8;
    auto row_stage = [&](const float* wr, const float* t) {     // acc[ph] += rowc[ph][y] * t for the bin rows with weight on this feature row
#pragma unroll
        for (int ph = 0; ph < 7; ++ph)
            if (wr[ph] != 0.f) {
                if constexpr (sizeof(T) == 2) fma8(acc[ph], wr[ph], t);
                else {
#pragma unroll
                    for (int i = 0; i < 8; ++i) acc[ph][i] += wr[ph] * t[i];
                }
            }
    };
    // The kernel is bound by VALU issue (2048 ROIs x 4 waves x ~3 700 instructions = 58 of its 62 us; -ffp-contract=off made every multiply-add a
    // v_pk_mul + v_pk_add pair: fused for bf16 maps, r06).  A bin column's run of pixels with weight is 4-5 long; walked four at a time a 5-pixel run cost
    // 8 pixels of unpack + multiply-add per row, and every row re-read the column weights from LDS and rebuilt the pixel addresses.  Here the wave's
    // LONGEST run (a wave holds two bin columns) picks an exactly unrolled row loop (1 .. 6 pixels), the column weights and pixel offsets are
    // computed once per ROI; longer runs (large ROIs) take the generic loop.  Same terms in the same order: identical results.
    const int run = xhi - xlo + 1;
    int maxrun = run;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) maxrun = max(maxrun, __shfl_xor(maxrun, o, 64));
    maxrun = __builtin_amdgcn_readfirstlane(maxrun);
    auto rows_exact = [&](auto NPt) {
        constexpr int NP = decltype(NPt)::value;
        float wcr[NP];
        int po[NP];
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int xk = min(xlo + k, xhi);
            wcr[k] = k < run ? colc[pw][xk] : 0.f;
            po[k] = xk * C;
        }
        for (int y = 0; y < nrow; ++y) {
            float wr[7];
            bool any = false;
#pragma unroll
            for (int ph = 0; ph < 7; ++ph) { wr[ph] = rowc[ph][y]; any = any || wr[ph] != 0.f; }
            if (!any) continue;                            // (uniform: margin rows of the footprint)
            const T* Fr = F + ((long)(fy0 + y) * W + fx0) * C;
            Raw8<T> q[NP];
#pragma unroll
            for (int k = 0; k < NP; ++k) q[k].load(Fr + po[k]);
            float t[8], u[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) t[i] = 0.f;
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                q[k].unpack(u);
                if constexpr (sizeof(T) == 2) fma8(t, wcr[k], u);
                else {
#pragma unroll
                    for (int i = 0; i < 8; ++i) t[i] += wcr[k] * u[i];
                }
            }
            row_stage(wr, t);
        }
    };
    switch (maxrun) {
        case 1: rows_exact(SepTag<1>{}); break;
        case 2: rows_exact(SepTag<2>{}); break;
        case 3: rows_exact(SepTag<3>{}); break;
        case 4: rows_exact(SepTag<4>{}); break;
        case 5: rows_exact(SepTag<5>{}); break;
        case 6: rows_exact(SepTag<6>{}); break;
        default:
            for (int y = 0; y < nrow; ++y) {
                float wr[7];
                bool any = false;
#pragma unroll
                for (int ph = 0; ph < 7; ++ph) { wr[ph] = rowc[ph][y]; any = any || wr[ph] != 0.f; }
                if (!any) continue;
                float t[8], u[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) t[i] = 0.f;
                const T* Fr = F + ((long)(fy0 + y) * W + fx0) * C;
                constexpr int XB = 4;       // pixels whose loads are in flight together (past the run: the last pixel again, weight 0)
                for (int x = xlo; x <= xhi; x += XB) {
                    Raw8<T> q[XB];
                    float wc[XB];
#pragma unroll
                    for (int k = 0; k < XB; ++k) {
                        const int xk = min(x + k, xhi);
                        q[k].load(Fr + (long)xk * C);
                        wc[k] = x + k <= xhi ? colc[pw][xk] : 0.f;
                    }
#pragma unroll
                    for (int k = 0; k < XB; ++k) {
                        q[k].unpack(u);
                        if constexpr (sizeof(T) == 2) fma8(t, wc[k], u);
                        else {
#pragma unroll
                            for (int i = 0; i < 8; ++i) t[i] += wc[k] * u[i];
                        }
                    }
                }
                row_stage(wr, t);
            }
    }
#pragma unroll
    for (int ph = 0; ph < 7; ++ph)
        if (ph < P) {
            float o[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = acc[ph][i] * inv_count;
            Raw8<T>::store(out_base + (long)(ph * P + pw) * C, o);
        }
}

// Backward as a GATHER (no atomics, deterministic): one workgroup owns a 32-pixel segment of one feature-map row of one
// (level, image), finds the ROIs that can touch it, and for each of them builds the separable bilinear footprint
//   rowc[ph]      = sum over the bin's sample rows of the weight they put on THIS row        (7 values)
//   colc[px][pw]  = the same along x for every pixel of the segment                           (32 x 7)
// with the forward's own bilin_prep (same clamps, same dead samples), so that
//   dG[py][px][c] = sum_roi 1/count * sum_pw colc[px][pw] * ( sum_ph rowc[ph] * g_pooled[roi][ph][pw][c] ).
// Every gradient element is written exactly once (the scatter form issues ~880 MB of fp32 atomics per step and is bound by
// them).  The benchmark step's ROIs are small (512 per image, nearly all on P2, ~19 x 10 pixels there: ~5 candidates per
// segment, 48 k (segment, ROI) pairs), so the kernel is bound by instruction issue and L2 round trips per pair, not by bytes:
//   * two thread roles per pair.  Row reduction: thread = (bin column, 8 channels), three 16-byte loads (the run of bin rows
//     with weight on this row; 2-byte loads per channel cost 7x the load instructions) -> gsum[pw][c] in LDS.
//     Column spread: thread = (pixel, 4-channel groups), only the bins with weight on ITS pixel (2-3 of 7).
//   * the candidate's geometry (divisions, ceilings) is computed once by the scanning thread and parked in LDS;
//   * a software pipeline with ONE barrier per pair: iteration q reduces the rows it loaded an iteration ago, issues the
//     loads of q + 1, builds the tables of q + 2 and spreads q - 1.
constexpr int kSeg = 32;
struct GatherGeom { int blk_off[5]; int segs[4]; int N; };

template <typename T, typename GT>      // GT: type of the gradient maps (float, or bf16 = what the FPN backward consumes: no cast pass)
__global__ __launch_bounds__(256, 5) void roialign_bwd_gather_kernel(Feats ft, GatherGeom gg, const float* __restrict__ rois, int R, int P,
                                                                  const T* __restrict__ gp /*[R][P][P][C]*/, int sorted) {
    __shared__ int cand[256];
    __shared__ float cx1[256], cy1[256], cbw[256], cbh[256], cinv[256];
    __shared__ int cgw[256], cgh[256];
    __shared__ int range[2];
    __shared__ int sm[17];
    __shared__ __attribute__((aligned(16))) float rowc[4][8];
    __shared__ __attribute__((aligned(16))) float colc[4][kSeg][8];
    __shared__ __attribute__((aligned(16))) float gsum[2][7][256];
    const int tid = threadIdx.x;
    const int px = tid >> 3, cg = tid & 7;               // column spread: pixel of the segment, channels (j * 8 + cg) * 4 ... + 3
    const int qw = tid >> 5, c8 = tid & 31;              // row reduction: bin column, channels c8 * 8 ... + 7
    // coarse levels first: their segments see an image's large ROIs (many candidates each) and would otherwise start last,
    // alone on the chip, after the thousands of light P2 segments
    const int bid = (int)(gridDim.x - 1 - blockIdx.x);
    int l = 0;
    while (l < 3 && bid >= gg.blk_off[l + 1]) ++l;
    const int H = ft.H[l], W = ft.W[l], C = ft.C;
    int t = bid - gg.blk_off[l];
    const int seg = t % gg.segs[l]; t /= gg.segs[l];
    const int py = t % H, b = t / H;
    const int px0 = seg * kSeg;
    const float sc = ft.scale[l];
    float4 acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);

    // the separable footprint of candidate k on this row / these 32 pixels -> table `buf` (every thread fills one colc entry)
    auto tables = [&](int k, int buf) {
        const float x1 = cx1[k], y1 = cy1[k], bw = cbw[k], bh = cbh[k];
        const int gw = cgw[k], gh = cgh[k];
        if (tid < 8) {
            float a = 0.f;
            if (tid < P)
                for (int iy = 0; iy < gh; ++iy) {
                    const Bilin by = bilin_prep(y1 + (float)tid * bh + ((float)iy + 0.5f) * bh / (float)gh, H);
                    if (by.dead) continue;
                    if (by.lo == py) a += by.h;
                    if (by.hi == py) a += by.l;
                }
            rowc[buf][tid] = a;
        }
        {
            const int pw = tid & 7;
            float a = 0.f;
            if (pw < P) {
                const int pxa = px0 + px;
                for (int ix = 0; ix < gw; ++ix) {
                    const Bilin bx = bilin_prep(x1 + (float)pw * bw + ((float)ix + 0.5f) * bw / (float)gw, W);
                    if (bx.dead) continue;
                    if (bx.lo == pxa) a += bx.h;
                    if (bx.hi == pxa) a += bx.l;
                }
            }
            colc[buf][px][pw] = a;
        }
    };

    // ROI rows are grouped by image (the engine concatenates the per-image samples): only this image's rows can hit the segment
    // -- a binary search instead of scanning all R rows in every one of the ~12 k workgroups
    int r_lo = 0, r_hi = R;
    if (sorted) {
        if (tid < 2) {
            const float key = (float)(b + tid);          // first row with image index >= b, resp. >= b + 1
            int lo = 0, hi = R;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (rois[(long)mid * 5] < key) lo = mid + 1; else hi = mid;
            }
            range[tid] = lo;
        }
        __syncthreads();
        r_lo = range[0]; r_hi = range[1];
    }
    for (int base = r_lo; base < r_hi; base += 256) {
        // ---- candidates of this chunk: same image, same level, footprint bounding box meets the segment
        const int r = base + tid;
        bool hit = false;
        float x1 = 0.f, y1 = 0.f, rw = 0.f, rh = 0.f;
        if (r < r_hi) {
            const float* rp = rois + (long)r * 5;
            if ((int)rp[0] == b && roi_level(rp[1], rp[2], rp[3], rp[4]) == l) {
                x1 = rp[1] * sc - 0.5f; y1 = rp[2] * sc - 0.5f;
                const float x2 = rp[3] * sc - 0.5f, y2 = rp[4] * sc - 0.5f;
                rw = x2 - x1; rh = y2 - y1;
                const int r0 = min(max((int)floorf(y1) - 1, 0), H - 1), r1 = min(max((int)floorf(y2) + 2, 0), H - 1);
                const int c0 = min(max((int)floorf(x1) - 1, 0), W - 1), c1 = min(max((int)floorf(x2) + 2, 0), W - 1);
                hit = py >= r0 && py <= r1 && c1 >= px0 && c0 < px0 + kSeg;
            }
        }
        int ncand;
        const int rank = block_rank(hit, sm, &ncand);
        if (hit) {
            const int gh = (int)ceilf(rh / (float)P), gw = (int)ceilf(rw / (float)P);
            cand[rank] = r;
            cx1[rank] = x1; cy1[rank] = y1; cbw[rank] = rw / (float)P; cbh[rank] = rh / (float)P;
            cgw[rank] = gw; cgh[rank] = gh;
            cinv[rank] = 1.f / (float)max(gh * gw, 1);
        }
        __syncthreads();
        if (ncand > 0) tables(0, 0);
        if (ncand > 1) tables(1, 1);
        __syncthreads();

        // state of the candidate whose pooled-gradient rows are in flight
        Raw8<T> raw[3];
        float rc[3] = {0.f, 0.f, 0.f};
        int plo = 8, phi = -1;
        bool live_c = false, live_p = false;
        // bin rows with a non-zero weight on this feature row: a run of <= 3 unless the bins are thinner than a pixel or the
        // samples clamp at the border (the rest of the run is then fetched when the rows are reduced); the margin rows of the
        // candidate test have none at all
        auto issue = [&](int k, int buf) {
            plo = 8; phi = -1;
#pragma unroll
            for (int ph = 0; ph < 7; ++ph)
                if (ph < P && rowc[buf][ph] != 0.f) { plo = min(plo, ph); phi = ph; }
            live_c = phi >= 0;
            if (!live_c) return;
            const T* g0 = gp + (long)cand[k] * P * P * C + c8 * 8;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int pk = min(plo + j, P - 1);
                rc[j] = plo + j <= phi ? rowc[buf][pk] : 0.f;
                if (qw < P) raw[j].load(g0 + (long)(pk * P + qw) * C);
            }
        };
        if (ncand > 0) issue(0, 0);
        for (int q = 0; q <= ncand; ++q) {                  // (the last iteration only spreads the last candidate)
            bool live_q = false;
            if (q < ncand) {
                live_q = live_c;
                if (live_q && qw < P) {
                    // sum_ph rowc[ph] * g[ph][qw][c], ascending ph, then 1 / (samples per bin)
                    float g8[8], u[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) g8[i] = 0.f;
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        raw[j].unpack(u);
                        fma8(g8, rc[j], u);
                    }
                    if (phi - plo > 2) {
                        const T* g0 = gp + (long)cand[q] * P * P * C + c8 * 8;
                        for (int ph = plo + 3; ph <= phi; ++ph) {
                            const float w = rowc[q & 3][ph];
                            Raw8<T> x;
                            x.load(g0 + (long)(ph * P + qw) * C);
                            x.unpack(u);
                            fma8(g8, w, u);
                        }
                    }
                    const float inv = cinv[q];
                    float* gd = &gsum[q & 1][qw][c8 * 8];
                    *reinterpret_cast<float4*>(gd) = make_float4(g8[0] * inv, g8[1] * inv, g8[2] * inv, g8[3] * inv);
                    *reinterpret_cast<float4*>(gd + 4) = make_float4(g8[4] * inv, g8[5] * inv, g8[6] * inv, g8[7] * inv);
                }
                if (q + 1 < ncand) issue(q + 1, (q + 1) & 3);
                if (q + 2 < ncand) tables(q + 2, (q + 2) & 3);
            }
            if (q > 0 && live_p) {
                // this pixel's bins of candidate q - 1 (a run of 2-3 of the 7), four channels at a time from gsum
                const int tb = (q - 1) & 3;
                const float4 w0 = *reinterpret_cast<const float4*>(&colc[tb][px][0]);
                const float4 w1 = *reinterpret_cast<const float4*>(&colc[tb][px][4]);
                const float wv[7] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z};
                const float* gsb = &gsum[(q - 1) & 1][0][cg * 4];
#pragma unroll
                for (int pw = 0; pw < 7; ++pw) {
                    if (wv[pw] != 0.f) {
                        const float w = wv[pw];
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const float4 g = *reinterpret_cast<const float4*>(gsb + pw * 256 + j * 32);
                            acc[j].x = __builtin_fmaf(w, g.x, acc[j].x); acc[j].y = __builtin_fmaf(w, g.y, acc[j].y); acc[j].z = __builtin_fmaf(w, g.z, acc[j].z); acc[j].w = __builtin_fmaf(w, g.w, acc[j].w);
                        }
                    }
                }
            }
            live_p = live_q;
            __syncthreads();
        }
    }
    if (px0 + px < W) {
        const long o = (((long)b * H + py) * W + px0 + px) * C + cg * 4;
        if constexpr (sizeof(GT) == 2) {
            bf16_t* G = reinterpret_cast<bf16_t*>(ft.g[l]) + o;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                uint2 v;
                v.x = pack2_bf16(acc[j].x, acc[j].y); v.y = pack2_bf16(acc[j].z, acc[j].w);
                *reinterpret_cast<uint2*>(G + j * 32) = v;
            }
        } else {
            float* G = ft.g[l] + o;
#pragma unroll
            for (int j = 0; j < 8; ++j) *reinterpret_cast<float4*>(G + j * 32) = acc[j];
        }
    }
}

// The gather with TWO feature rows per workgroup (what aldi_roialign_backward runs on bf16 pooled gradients).  The one-row kernel above is
// bound by instruction issue and the per-pair barrier chain, not by bytes: its parts add up when they are removed one at a time
// (profiles/r04_roialign_bwd_ablation.txt, the real step's ROIs: 199 us; writing the maps alone 35, + candidate scan 65, + the bare pair loop
// 115, + row reduction 132, + loads 150, + column tables 174, + column spread 199).  A workgroup that owns rows py0, py0 + 1 of a 32-pixel
// segment shares between the two rows the scan, the column tables, the pooled rows it loads (a run of <= 4 bin rows instead of 2 x 3) and
// the barrier of every (segment, ROI) pair, and sees ~1.07x the candidates of one row: 213 -> 186 us on the step's ROIs.  Same terms in the
// same order as the one-row kernel (the zero weights of a shared run contribute +0): the results are identical.  Tried on top, slower:
// a spread with a thread owning 4 channels x 8 pixels and scalar skips of the zero weights (one branch per FMA quad: 2.5x), four
// workgroups per CU by spilling (228 us), per-row / per-load branches in the reduction instead of zero weights (+5 us).
template <typename GT>
__global__ __launch_bounds__(256, 3) void roialign_bwd_gather2_kernel(Feats ft, GatherGeom gg, const float* __restrict__ rois, int R, int P,
                                                                   const bf16_t* __restrict__ gp /*[R][P][P][C]*/, int sorted) {
    __shared__ int cand[256];
    __shared__ float cx1[256], cy1[256], cbw[256], cbh[256], cinv[256];
    __shared__ int cgwh[256];
    __shared__ int range[2];
    __shared__ int sm[17];
    __shared__ int rrun[4];                              // first / last bin row with weight on either row, which rows have any
    __shared__ __attribute__((aligned(16))) float rowc[4][2][8];
    __shared__ __attribute__((aligned(16))) float colc[4][kSeg][8];
    __shared__ __attribute__((aligned(16))) float gsum[2][2][7][256];
    const int tid = threadIdx.x;
    const int px = tid >> 3, cg = tid & 7;               // column spread: pixel of the segment, channels (j * 8 + cg) * 4 ... + 3
    const int qw = tid >> 5, c8 = tid & 31;              // row reduction: bin column, channels c8 * 8 ... + 7
    const int bid = (int)(gridDim.x - 1 - blockIdx.x);   // coarse levels first (see above)
    int l = 0;
    while (l < 3 && bid >= gg.blk_off[l + 1]) ++l;
    const int H = ft.H[l], W = ft.W[l], C = ft.C;
    const int HP = (H + 1) >> 1;
    int t = bid - gg.blk_off[l];
    const int seg = t % gg.segs[l]; t /= gg.segs[l];
    const int py0 = (t % HP) * 2, b = t / HP;
    const int px0 = seg * kSeg;
    const float sc = ft.scale[l];
    float4 acc[2][8];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[r][j] = make_float4(0.f, 0.f, 0.f, 0.f);

    auto tables = [&](int k, int buf) {
        const float x1 = cx1[k], y1 = cy1[k], bw = cbw[k], bh = cbh[k];
        const int gw = cgwh[k] & 0xffff, gh = cgwh[k] >> 16;
        if (tid < 64) {
            const int r = tid >> 3, ph = tid & 7, py = py0 + r;
            float a = 0.f;
            if (tid < 16 && ph < P && py < H)
                for (int iy = 0; iy < gh; ++iy) {
                    const Bilin by = bilin_prep(y1 + (float)ph * bh + ((float)iy + 0.5f) * bh / (float)gh, H);
                    if (by.dead) continue;
                    if (by.lo == py) a += by.h;
                    if (by.hi == py) a += by.l;
                }
            const unsigned long long m = __ballot(a != 0.f);
            if (tid < 16) rowc[buf][r][ph] = a;
            if (tid == 0) {
                const unsigned m0 = (unsigned)m & 0x7fu, m1 = (unsigned)(m >> 8) & 0x7fu, mm = m0 | m1;
                rrun[buf] = mm ? (__ffs((int)mm) - 1) | ((31 - __clz((int)mm)) << 4) | ((m0 != 0) << 8) | ((m1 != 0) << 9) : 0;
            }
        }
        {
            const int pw = tid & 7;
            float a = 0.f;
            if (pw < P) {
                const int pxa = px0 + px;
                for (int ix = 0; ix < gw; ++ix) {
                    const Bilin bx = bilin_prep(x1 + (float)pw * bw + ((float)ix + 0.5f) * bw / (float)gw, W);
                    if (bx.dead) continue;
                    if (bx.lo == pxa) a += bx.h;
                    if (bx.hi == pxa) a += bx.l;
                }
            }
            colc[buf][px][pw] = a;
        }
    };

    int r_lo = 0, r_hi = R;
    if (sorted) {
        if (tid < 2) {
            const float key = (float)(b + tid);
            int lo = 0, hi = R;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (rois[(long)mid * 5] < key) lo = mid + 1; else hi = mid;
            }
            range[tid] = lo;
        }
        __syncthreads();
        r_lo = range[0]; r_hi = range[1];
    }
    for (int base = r_lo; base < r_hi; base += 256) {
        const int r = base + tid;
        bool hit = false;
        float x1 = 0.f, y1 = 0.f, rw = 0.f, rh = 0.f;
        if (r < r_hi) {
            const float* rp = rois + (long)r * 5;
            if ((int)rp[0] == b && roi_level(rp[1], rp[2], rp[3], rp[4]) == l) {
                x1 = rp[1] * sc - 0.5f; y1 = rp[2] * sc - 0.5f;
                const float x2 = rp[3] * sc - 0.5f, y2 = rp[4] * sc - 0.5f;
                rw = x2 - x1; rh = y2 - y1;
                const int r0 = min(max((int)floorf(y1) - 1, 0), H - 1), r1 = min(max((int)floorf(y2) + 2, 0), H - 1);
                const int c0 = min(max((int)floorf(x1) - 1, 0), W - 1), c1 = min(max((int)floorf(x2) + 2, 0), W - 1);
                hit = py0 + 1 >= r0 && py0 <= r1 && c1 >= px0 && c0 < px0 + kSeg;
            }
        }
        int ncand;
        const int rank = block_rank(hit, sm, &ncand);
        if (hit) {
            const int gh = (int)ceilf(rh / (float)P), gw = (int)ceilf(rw / (float)P);
            cand[rank] = r;
            cx1[rank] = x1; cy1[rank] = y1; cbw[rank] = rw / (float)P; cbh[rank] = rh / (float)P;
            cgwh[rank] = max(gw, 0) | (max(gh, 0) << 16);
            cinv[rank] = 1.f / (float)max(gh * gw, 1);
        }
        __syncthreads();
        if (ncand > 0) tables(0, 0);
        if (ncand > 1) tables(1, 1);
        __syncthreads();

        // the candidate whose pooled rows are in flight: the run of bin rows [plo, phi], their weights on the two feature rows
        uint4 raw[4];
        float rc0[4] = {0.f, 0.f, 0.f, 0.f}, rc1[4] = {0.f, 0.f, 0.f, 0.f};
        int plo = 0, phi = -1, lv = 0;
        auto issue = [&](int k, int buf) {
            const int m = rrun[buf];
            lv = (m >> 8) & 3;
            plo = m & 15; phi = lv ? (m >> 4) & 15 : -1;
            if (!lv) return;
            const bf16_t* g0 = gp + (long)cand[k] * P * P * C + c8 * 8;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int ph = plo + j;
                raw[j] = make_uint4(0u, 0u, 0u, 0u);
                rc0[j] = 0.f; rc1[j] = 0.f;
                if (ph <= phi) {
                    rc0[j] = rowc[buf][0][ph]; rc1[j] = rowc[buf][1][ph];
                    if (qw < P) raw[j] = *reinterpret_cast<const uint4*>(g0 + (long)(ph * P + qw) * C);
                }
            }
        };
        if (ncand > 0) issue(0, 0);
        for (int q = 0; q <= ncand; ++q) {
            if (q < ncand) {
                if (lv && qw < P) {
                    float g0[8], g1[8], u[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) { g0[i] = 0.f; g1[i] = 0.f; }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        Raw8<bf16_t> x; x.v = raw[j];
                        x.unpack(u);
                        fma8(g0, rc0[j], u);
                        fma8(g1, rc1[j], u);
                    }
                    if (phi - plo > 3) {
                        const bf16_t* gq = gp + (long)cand[q] * P * P * C + c8 * 8;
                        for (int ph = plo + 4; ph <= phi; ++ph) {
                            const float w0 = rowc[q & 3][0][ph], w1 = rowc[q & 3][1][ph];
                            Raw8<bf16_t> x;
                            x.load(gq + (long)(ph * P + qw) * C);
                            x.unpack(u);
                            { fma8(g0, w0, u); fma8(g1, w1, u); }
                        }
                    }
                    const float inv = cinv[q];
                    if (lv & 1) {
                        float* gd = &gsum[q & 1][0][qw][c8 * 8];
                        *reinterpret_cast<float4*>(gd) = make_float4(g0[0] * inv, g0[1] * inv, g0[2] * inv, g0[3] * inv);
                        *reinterpret_cast<float4*>(gd + 4) = make_float4(g0[4] * inv, g0[5] * inv, g0[6] * inv, g0[7] * inv);
                    }
                    if (lv & 2) {
                        float* gd = &gsum[q & 1][1][qw][c8 * 8];
                        *reinterpret_cast<float4*>(gd) = make_float4(g1[0] * inv, g1[1] * inv, g1[2] * inv, g1[3] * inv);
                        *reinterpret_cast<float4*>(gd + 4) = make_float4(g1[4] * inv, g1[5] * inv, g1[6] * inv, g1[7] * inv);
                    }
                }
                if (q + 1 < ncand) issue(q + 1, (q + 1) & 3);
                if (q + 2 < ncand) tables(q + 2, (q + 2) & 3);
            }
            if (q > 0) {
                // this pixel's bins of candidate q - 1 (a run of 2-3 of the 7) on the rows that carry weight, four channels at a time
                const int tb = (q - 1) & 3;
                const int lvp = (rrun[tb] >> 8) & 3;
                if (lvp) {
                    const float4 w0 = *reinterpret_cast<const float4*>(&colc[tb][px][0]);
                    const float4 w1 = *reinterpret_cast<const float4*>(&colc[tb][px][4]);
                    const float wv[7] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z};
#pragma unroll
                    for (int r = 0; r < 2; ++r) {
                        if (!((lvp >> r) & 1)) continue;
                        const float* gsb = &gsum[(q - 1) & 1][r][0][cg * 4];
#pragma unroll
                        for (int pw = 0; pw < 7; ++pw) {
                            if (wv[pw] != 0.f) {
                                const float w = wv[pw];
#pragma unroll
                                for (int j = 0; j < 8; ++j) {
                                    const float4 g = *reinterpret_cast<const float4*>(gsb + pw * 256 + j * 32);
                                    float4& a = acc[r][j];
                                    a.x = __builtin_fmaf(w, g.x, a.x); a.y = __builtin_fmaf(w, g.y, a.y); a.z = __builtin_fmaf(w, g.z, a.z); a.w = __builtin_fmaf(w, g.w, a.w);
                                }
                            }
                        }
                    }
                }
            }
            __syncthreads();
        }
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int py = py0 + r;
        if (py >= H || px0 + px >= W) continue;
        const long o = (((long)b * H + py) * W + px0 + px) * C + cg * 4;
        if constexpr (sizeof(GT) == 2) {
            bf16_t* G = reinterpret_cast<bf16_t*>(ft.g[l]) + o;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                uint2 v;
                v.x = pack2_bf16(acc[r][j].x, acc[r][j].y); v.y = pack2_bf16(acc[r][j].z, acc[r][j].w);
                *reinterpret_cast<uint2*>(G + j * 32) = v;
            }
        } else {
            float* G = ft.g[l] + o;
#pragma unroll
            for (int j = 0; j < 8; ++j) *reinterpret_cast<float4*>(G + j * 32) = acc[r][j];
        }
    }
}

// FastRCNNOutputLayers.losses: CE(mean over R) + L1 on the gt-class deltas of fg rows / R
// pred row: [0,K] class logits, [K+1, K+1+4K) deltas (class*4+d).  grad += d(loss*gscale)/d(pred)
__global__ __launch_bounds__(256) void box_loss_kernel(const float* __restrict__ pred, int Cp, int K, int R,
                                                       const float* __restrict__ rois, const int* __restrict__ cls, const float4* __restrict__ gtb,
                                                       float wx, float wy, float ww, float wh, float gs_cls, float gs_box,
                                                       float* __restrict__ grad, float* __restrict__ loss /*[2]*/) {
    __shared__ float red[16];
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    float l_cls = 0.f, l_box = 0.f;
    const float invR = 1.f / (float)max(R, 1);
    if (r < R)
        box_loss_row(pred + (long)r * Cp, K, cls[r], rois + (long)r * 5, gtb[r], wx, wy, ww, wh, invR, gs_cls, gs_box, grad ? grad + (long)r * Cp : nullptr,
                     l_cls, l_box);
    float s0 = block_sum(l_cls, red);
    float s1 = block_sum(l_box, red);
    if (threadIdx.x == 0) {
        if (s0 != 0.f) unsafeAtomicAdd(loss + 0, s0 * invR);
        if (s1 != 0.f) unsafeAtomicAdd(loss + 1, s1 * invR);
    }
}

// The box head's losses of EVERY chunk of a fused step in one launch: FastRCNNOutputLayers.losses per chunk (its own row count, scales and
// loss slots), the RoI distillation of the chunks that have a teacher, and the compute-dtype copy of the finished gradient rows -- four
// launches (box_loss x chunks, roih_distill, cast) on the chain between the box head's forward and its backward, where a dependent small
// launch costs 10-20 us of a replayed graph.  A chunk's rows are whole workgroups (its block sums go to its slots).  grad: fp32 [R][Cp],
// ZEROED by the caller (the rows are accumulated into exactly as the separate kernels do: (0 + box) + distillation).
constexpr int kMaxLossChunks = 8;
struct BoxLossChunk {
    int r0, r1, blk0;
    float gs_cls, gs_box;
    float* loss_box;
    const float* tpred;                  // teacher rows of THIS chunk (row r0 first), or null
    float inv_T; int kl, do_cls, do_reg;
    float gs_dcls, gs_dreg;
    float* loss_d;
};
struct BoxLossArgs { int n; BoxLossChunk c[kMaxLossChunks]; };
__global__ __launch_bounds__(256) void box_losses_fused_kernel(const float* __restrict__ pred, int Cp, int K, const float* __restrict__ rois,
                                                               const int* __restrict__ cls, const float4* __restrict__ gtb, float wx, float wy, float ww, float wh,
                                                               float* grad, bf16_t* __restrict__ grad_lo, BoxLossArgs A) {
    __shared__ float red[16];
    int ci = 0;
    for (int k = 1; k < A.n; ++k)
        if ((int)blockIdx.x >= A.c[k].blk0) ci = k;
    const BoxLossChunk& ch = A.c[ci];
    const int r = ch.r0 + ((int)blockIdx.x - ch.blk0) * (int)blockDim.x + (int)threadIdx.x;
    const float invR = 1.f / (float)max(ch.r1 - ch.r0, 1);
    float l_cls = 0.f, l_box = 0.f, d_cls = 0.f, d_reg = 0.f;
    if (r < ch.r1) {
        float* grow = grad + (long)r * Cp;
        box_loss_row(pred + (long)r * Cp, K, cls[r], rois + (long)r * 5, gtb[r], wx, wy, ww, wh, invR, ch.gs_cls, ch.gs_box, grow, l_cls, l_box);
        if (ch.tpred)
            roih_distill_row(pred + (long)r * Cp, ch.tpred + (long)(r - ch.r0) * Cp, K, ch.inv_T, ch.kl, ch.do_cls, ch.do_reg, invR, ch.gs_dcls, ch.gs_dreg,
                             grow, d_cls, d_reg);
        if (grad_lo)
            for (int k = 0; k < Cp; k += 2) {
                const float a = grow[k], b = k + 1 < Cp ? grow[k + 1] : 0.f;
                if (k + 1 < Cp) *reinterpret_cast<uint32_t*>(grad_lo + (long)r * Cp + k) = pack2_bf16(a, b);
                else grad_lo[(long)r * Cp + k] = f32_to_bf16(a);
            }
    }
    float s0 = block_sum(l_cls, red);
    float s1 = block_sum(l_box, red);
    if (threadIdx.x == 0) {
        if (s0 != 0.f) unsafeAtomicAdd(ch.loss_box + 0, s0 * invR);
        if (s1 != 0.f) unsafeAtomicAdd(ch.loss_box + 1, s1 * invR);
    }
    if (ch.tpred) {
        float t0 = block_sum(d_cls, red);
        float t1 = block_sum(d_reg, red);
        if (threadIdx.x == 0) {
            if (t0 != 0.f) unsafeAtomicAdd(ch.loss_d + 0, t0 * invR);
            if (t1 != 0.f) unsafeAtomicAdd(ch.loss_d + 1, t1 * invR);
        }
    }
}

// ------------------------------------------------------------------------- inference post-processing
constexpr int kDetCap = 8192;

__device__ __forceinline__ float clampf(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }

// score = softmax(logits)[k] for k < K; candidates with score > thresh appended (unordered; the sort fixes the order)
__global__ void det_candidates_kernel(const float* __restrict__ pred, int Cp, int K, const int* __restrict__ pcount, int P, float score_thresh,
                                      unsigned long long* __restrict__ keys /*[N][cap]*/, int* __restrict__ cnt, int* __restrict__ err) {
    const int n = blockIdx.y;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int r = t / K, k = t - r * K;
    if (r >= pcount[n]) return;
    const float* p = pred + ((long)n * P + r) * Cp;
    float m = p[0];
    for (int j = 1; j <= K; ++j) m = fmaxf(m, p[j]);
    float s = 0.f;
    for (int j = 0; j <= K; ++j) s += expf(p[j] - m);
    const float prob = expf(p[k] - m) / s;
    if (!isfinite(prob)) { atomicOr(err, 2); return; }
    if (prob > score_thresh) {
        int slot = atomicAdd(cnt + n, 1);
        if (slot < kDetCap) keys[(long)n * kDetCap + slot] = ((unsigned long long)(~float_key_asc(prob)) << 32) | (unsigned)t;
        else atomicOr(err, 4);
    }
}

// sort candidates by (score desc, flat index asc); materialise boxes (decoded for the candidate's class, clipped)
__global__ __launch_bounds__(1024) void det_sort_kernel(const float* __restrict__ pred, int Cp, int K, const float4* __restrict__ props, int P,
                                                        const int* __restrict__ img_hw, float wx, float wy, float ww, float wh,
                                                        unsigned long long* __restrict__ keys, int* __restrict__ cnt,
                                                        float4* __restrict__ boxes, float* __restrict__ scores, int* __restrict__ cats, int* __restrict__ valid) {
    extern __shared__ unsigned long long sk[];
    const int n = blockIdx.x;
    const int c = min(cnt[n], kDetCap);
    int p2 = 1024;
    while (p2 < c) p2 <<= 1;
    for (int i = threadIdx.x; i < p2; i += blockDim.x) sk[i] = i < c ? keys[(long)n * kDetCap + i] : ~0ull;
    __syncthreads();
    bitonic_sort_u64(sk, p2);
    const float ih = (float)img_hw[n * 2], iw = (float)img_hw[n * 2 + 1];
    const float clampv = 4.135166556742356f;
    for (int i = threadIdx.x; i < c; i += blockDim.x) {
        unsigned long long key = sk[i];
        int t = (int)(key & 0xffffffffu);
        int r = t / K, k = t - r * K;
        const float* p = pred + ((long)n * P + r) * Cp;
        const float4 b = props[(long)n * P + r];
        float w = b.z - b.x, h = b.w - b.y;
        float cx = b.x + 0.5f * w, cy = b.y + 0.5f * h;
        float dx = p[K + 1 + k * 4 + 0] / wx, dy = p[K + 1 + k * 4 + 1] / wy;
        float dw = fminf(p[K + 1 + k * 4 + 2] / ww, clampv), dh = fminf(p[K + 1 + k * 4 + 3] / wh, clampv);
        float pcx = dx * w + cx, pcy = dy * h + cy, pw = expf(dw) * w, phh = expf(dh) * h;
        float4 o = make_float4(pcx - 0.5f * pw, pcy - 0.5f * phh, pcx + 0.5f * pw, pcy + 0.5f * phh);
        o.x = clampf(o.x, 0.f, iw); o.y = clampf(o.y, 0.f, ih); o.z = clampf(o.z, 0.f, iw); o.w = clampf(o.w, 0.f, ih);
        const long slot = (long)n * kDetCap + i;
        boxes[slot] = o;
        scores[slot] = key_asc_to_float(~(unsigned)(key >> 32));
        cats[slot] = k;
        valid[slot] = 1;
    }
    __syncthreads();
    if (threadIdx.x == 0) cnt[n] = c;
}

// first `keep_count` survivors -> detections; detections with score > thr -> pseudo-labels (order preserved)
__global__ void det_finish_kernel(const float4* __restrict__ boxes, const float* __restrict__ scores, const int* __restrict__ cats,
                                  const int* __restrict__ keep, const int* __restrict__ keep_count, int topk, int pl_rows, float pl_thresh,
                                  float4* __restrict__ det_boxes, float* __restrict__ det_scores, int* __restrict__ det_cls, int* __restrict__ det_count,
                                  float4* __restrict__ pl_boxes, int* __restrict__ pl_cls, float* __restrict__ pl_scores, int* __restrict__ pl_count) {
    // one wave per image, 64 detections per round (a single thread walking `topk` dependent loads was a 27 us chain)
    const int n = blockIdx.x, lane = threadIdx.x;
    const int kc = min(keep_count[n], topk);
    int np = 0;
    for (int j0 = 0; j0 < topk; j0 += 64) {
        const int j = j0 + lane;
        float4 b = make_float4(0, 0, 0, 0);
        float s = 0.f;
        int c = -1;
        if (j < kc) {
            const long slot = (long)n * kDetCap + keep[(long)n * kDetCap + j];
            b = boxes[slot]; s = scores[slot]; c = cats[slot];
        }
        const bool pl = j < kc && s > pl_thresh;                     // pseudo-label: order of the detections kept
        const unsigned long long m = __ballot(pl);
        if (pl) {
            const int q = np + __popcll(m & ((1ull << lane) - 1ull));
            pl_boxes[(long)n * pl_rows + q] = b; pl_cls[n * pl_rows + q] = c; pl_scores[n * pl_rows + q] = s;
        }
        np += __popcll(m);
        if (j < topk) { det_boxes[(long)n * topk + j] = b; det_scores[n * topk + j] = s; det_cls[n * topk + j] = c; }
    }
    // (rows of pl_rows >= topk entries -- the caller's ground-truth slots; the class of an unused slot is -1 up to topk, 0 beyond, as
    // a zero-filled buffer with a [topk] block copied in would read)
    for (int j = np + lane; j < pl_rows; j += 64) {
        pl_boxes[(long)n * pl_rows + j] = make_float4(0, 0, 0, 0); pl_cls[n * pl_rows + j] = j < topk ? -1 : 0; pl_scores[n * pl_rows + j] = 0.f;
    }
    if (lane == 0) { det_count[n] = kc; pl_count[n] = np; }
}

Feats make_feats(const aldi_roi_feats* f, bool bwd) {
    Feats ft;
    for (int l = 0; l < 4; ++l) {
        ft.f[l] = f->feat[l]; ft.g[l] = bwd ? f->grad[l] : nullptr;
        ft.gbytes[l] = 0x7fffffffu;   // the maps are far below 2 GiB; the bound only has to reject the "dropped" offsets (bit 31)
        ft.H[l] = f->H[l]; ft.W[l] = f->W[l]; ft.scale[l] = f->scale[l];
    }
    ft.C = f->C;
    return ft;
}

}  // namespace

extern "C" int aldi_roi_prepare(const float* props, const int* pcount, int P, const float* gt_boxes, const int* gt_classes, const int* gt_count,
                                int Gmax, int N, int K, float iou_thresh, float* cand, int* ccount, float* best_iou, int* best_idx,
                                unsigned* gt_best_scratch, int* labels, int* cls, aldi_stream_t stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int L = P + Gmax;
    hipLaunchKernelGGL(roi_append_gt_kernel, dim3(cdiv(L, 256), N), dim3(256), 0, st, (const float4*)props, pcount, P, (const float4*)gt_boxes, gt_count, Gmax,
                       (float4*)cand, ccount, L);
    ALDI_CHECK_LAUNCH();
    int rc = aldi_box_match(cand, L, ccount, L, gt_boxes, gt_count, Gmax, N, iou_thresh, iou_thresh, 0, best_iou, best_idx, gt_best_scratch, sizeof(unsigned) * (size_t)N * Gmax, labels, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(roi_classes_kernel, dim3(cdiv(L, 256), N), dim3(256), 0, st, labels, best_idx, gt_classes, gt_count, Gmax, L, K, cls);
    ALDI_CHECK_LAUNCH();
    return ALDI_OK;
}

extern "C" int aldi_roi_gather(const float* cand, const int* cls, const int* best_idx, int L, const int* lists, const int* sel, const int* nsel, int S,
                               const int* row_off, const float* gt_boxes, const int* gt_count, int Gmax, int N,
                               float* rois, int* r_cls, float* r_gt, int* r_idx, aldi_stream_t stream) {
    hipLaunchKernelGGL(roi_gather_kernel, dim3(cdiv(S * 2, 256), N), dim3(256), 0, static_cast<hipStream_t>(stream), (const float4*)cand, cls, best_idx, L, lists, sel, nsel, S,
                       row_off, (const float4*)gt_boxes, gt_count, Gmax, rois, r_cls, (float4*)r_gt, r_idx);
    ALDI_CHECK_LAUNCH();
    return ALDI_OK;
}

extern "C" int aldi_rois_from_proposals(const float* props, const int* pcount, int P, int N, float* rois, aldi_stream_t stream) {
    hipLaunchKernelGGL(roi_from_proposals_kernel, dim3(cdiv(P, 256), N), dim3(256), 0, static_cast<hipStream_t>(stream), (const float4*)props, pcount, P, rois);
    ALDI_CHECK_LAUNCH();
    return ALDI_OK;
}

extern "C" int aldi_roialign(const aldi_roi_feats* f, const float* rois, int R, int P, void* pooled, int backward, int dtype, aldi_stream_t stream) {
    if (!f || !rois || !pooled || f->C != 256) return aldi_set_error_msg(ALDI_ERR_ARG, "roialign: bad args (C must be 256)");
    if (R <= 0) return ALDI_OK;
    Feats ft = make_feats(f, backward != 0);
    hipStream_t st = static_cast<hipStream_t>(stream);
    dim3 grid(R, P);
    // forward: separable form when every level's map fits its footprint tables (roialign_sep knob: 0 = the sample-by-sample form)
    bool sep = aldi_tuning().roialign_sep != 0 && P <= 7;
    for (int l = 0; l < 4; ++l) sep = sep && ft.H[l] <= kSepMaxH && ft.W[l] <= kSepMaxW;
    if (dtype == ALDI_BF16) {
        if (backward) hipLaunchKernelGGL((roialign_kernel<bf16_t, true>), grid, dim3(256), 0, st, ft, rois, P, (bf16_t*)pooled);
        else if (sep) hipLaunchKernelGGL((roialign_fwd_sep_kernel<bf16_t>), dim3(R), dim3(256), 0, st, ft, rois, P, (bf16_t*)pooled);
        else hipLaunchKernelGGL((roialign_fwd_vec_kernel<bf16_t>), grid, dim3(256), 0, st, ft, rois, P, (bf16_t*)pooled);
    } else {
        if (backward) hipLaunchKernelGGL((roialign_kernel<float, true>), grid, dim3(256), 0, st, ft, rois, P, (float*)pooled);
        else if (sep) hipLaunchKernelGGL((roialign_fwd_sep_kernel<float>), dim3(R), dim3(256), 0, st, ft, rois, P, (float*)pooled);
        else hipLaunchKernelGGL((roialign_fwd_vec_kernel<float>), grid, dim3(256), 0, st, ft, rois, P, (float*)pooled);
    }
    ALDI_CHECK_LAUNCH();
    return ALDI_OK;
}

extern "C" int aldi_roialign_backward(const aldi_roi_feats* f, const float* rois, int R, int P, const void* g_pooled, int N, int rois_sorted,
                                      int dtype, int grad_dtype, aldi_stream_t stream) {
    if (!f || !rois || !g_pooled || f->C != 256 || P > 7 || N < 1) return aldi_set_error_msg(ALDI_ERR_ARG, "roialign_backward: bad args (C must be 256, P <= 7)");
    if (grad_dtype != ALDI_F32 && !(grad_dtype == ALDI_BF16 && dtype == ALDI_BF16))
        return aldi_set_error_msg(ALDI_ERR_ARG, "roialign_backward: gradient maps are fp32, or bf16 with bf16 pooled gradients");
    Feats ft = make_feats(f, true);
    GatherGeom gg;
    gg.N = N;
    // bf16 pooled gradients: two feature rows per workgroup (roialign_bwd_rows knob: 1 = the one-row kernel)
    const int rows = dtype == ALDI_BF16 && aldi_tuning().roialign_bwd_rows != 1 ? 2 : 1;
    int off = 0;
    for (int l = 0; l < 4; ++l) {
        if (!ft.g[l]) return aldi_set_error_msg(ALDI_ERR_ARG, "roialign_backward: missing gradient map");
        gg.blk_off[l] = off;
        gg.segs[l] = cdiv(ft.W[l], kSeg);
        off += N * cdiv(ft.H[l], rows) * gg.segs[l];
    }
    gg.blk_off[4] = off;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (rows == 2 && grad_dtype == ALDI_BF16)
        hipLaunchKernelGGL((roialign_bwd_gather2_kernel<bf16_t>), dim3(off), dim3(256), 0, st, ft, gg, rois, R, P, (const bf16_t*)g_pooled, rois_sorted);
    else if (rows == 2)
        hipLaunchKernelGGL((roialign_bwd_gather2_kernel<float>), dim3(off), dim3(256), 0, st, ft, gg, rois, R, P, (const bf16_t*)g_pooled, rois_sorted);
    else if (dtype == ALDI_BF16 && grad_dtype == ALDI_BF16)
        hipLaunchKernelGGL((roialign_bwd_gather_kernel<bf16_t, bf16_t>), dim3(off), dim3(256), 0, st, ft, gg, rois, R, P, (const bf16_t*)g_pooled, rois_sorted);
    else if (dtype == ALDI_BF16)
        hipLaunchKernelGGL((roialign_bwd_gather_kernel<bf16_t, float>), dim3(off), dim3(256), 0, st, ft, gg, rois, R, P, (const bf16_t*)g_pooled, rois_sorted);
    else if (dtype == ALDI_F32)
        hipLaunchKernelGGL((roialign_bwd_gather_kernel<float, float>), dim3(off), dim3(256), 0, st, ft, gg, rois, R, P, (const float*)g_pooled, rois_sorted);
    else return aldi_set_error_msg(ALDI_ERR_ARG, "roialign_backward: bad dtype");
    ALDI_CHECK_LAUNCH();
    return ALDI_OK;
}

extern "C" int aldi_box_loss(const float* pred, int Cp, int K, int R, const float* rois, const int* cls, const float* gt_boxes,
                             const float* weights4, float grad_scale_cls, float grad_scale_box, float* grad, float* loss2, aldi_stream_t stream) {
    if (!pred || !rois || !cls || !gt_boxes || !loss2 || !weights4) return aldi_set_error_msg(ALDI_ERR_ARG, "box_loss: null pointer");
    if (R <= 0) return ALDI_OK;
    hipLaunchKernelGGL(box_loss_kernel, dim3(cdiv(R, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), pred, Cp, K, R, rois, cls, (const float4*)gt_boxes,
                       weights4[0], weights4[1], weights4[2], weights4[3], grad_scale_cls, grad_scale_box, grad, loss2);
    ALDI_CHECK_LAUNCH();
    return ALDI_OK;
}

extern "C" int aldi_box_losses_fused(const float* pred, int Cp, int K, const float* rois, const int* cls, const float* gt_boxes, const float* weights4,
                                     const aldi_box_loss_chunk* chunks, int nchunks, float* grad, void* grad_lo, aldi_stream_t stream) {
    if (!pred || !rois || !cls || !gt_boxes || !weights4 || !chunks || !grad) return aldi_set_error_msg(ALDI_ERR_ARG, "box_losses_fused: null pointer");
    if (nchunks < 1 || nchunks > kMaxLossChunks || (Cp & 1)) return aldi_set_error_msg(ALDI_ERR_ARG, "box_losses_fused: 1 .. 8 chunks, even row length");
    BoxLossArgs A;
    A.n = 0;
    int blk = 0;
    for (int i = 0; i < nchunks; ++i) {
        const aldi_box_loss_chunk& q = chunks[i];
        if (q.r1 <= q.r0) continue;
        if (!q.loss_box || (q.teacher_pred && !q.loss_distill)) return aldi_set_error_msg(ALDI_ERR_ARG, "box_losses_fused: missing loss slot");
        BoxLossChunk& c = A.c[A.n++];
        c.r0 = q.r0; c.r1 = q.r1; c.blk0 = blk;
        c.gs_cls = q.grad_scale_cls; c.gs_box = q.grad_scale_box; c.loss_box = q.loss_box;
        c.tpred = q.teacher_pred; c.inv_T = q.teacher_pred ? 1.f / q.cls_temperature : 1.f; c.kl = q.kl; c.do_cls = q.do_cls; c.do_reg = q.do_reg;
        c.gs_dcls = q.grad_scale_distill_cls; c.gs_dreg = q.grad_scale_distill_reg; c.loss_d = q.loss_distill;
        blk += cdiv(q.r1 - q.r0, 256);
    }
    if (A.n == 0) return ALDI_OK;
    hipLaunchKernelGGL(box_losses_fused_kernel, dim3(blk), dim3(256), 0, static_cast<hipStream_t>(stream), pred, Cp, K, rois, cls, (const float4*)gt_boxes,
                       weights4[0], weights4[1], weights4[2], weights4[3], grad, (bf16_t*)grad_lo, A);
    ALDI_CHECK_LAUNCH();
    return ALDI_OK;
}

extern "C" size_t aldi_detections_workspace(int N) {
    size_t cap = kDetCap, s = 0;
    s += (size_t)N * cap * 8 + 256;                 // keys
    s += (size_t)N * 4 + 256;                       // cnt
    s += (size_t)N * cap * (16 + 4 + 4 + 4) + 1024; // boxes, scores, cats, valid
    s += (size_t)N * cap * (cap / 64) * 8 + 256;    // mask
    s += (size_t)N * cap * 4 + (size_t)N * 4 + 512; // keep, keep_count
    return s + 1024;
}

extern "C" int aldi_detections(const float* pred, int Cp, int K, const float* props, const int* pcount, int P, int N, const int* img_hw,
                               const float* weights4, float score_thresh, float nms_thresh, int topk, float pl_thresh, void* workspace,
                               float* det_boxes, float* det_scores, int* det_cls, int* det_count,
                               float* pl_boxes, int* pl_cls, float* pl_scores, int* pl_count, int pl_rows, int* err_flag, aldi_stream_t stream) {
    if (!pred || !props || !pcount || !img_hw || !workspace || !weights4) return aldi_set_error_msg(ALDI_ERR_ARG, "detections: null pointer");
    if (pl_rows < topk) return aldi_set_error_msg(ALDI_ERR_ARG, "detections: pl_rows < topk");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t cap = kDetCap;
    char* w = static_cast<char*>(workspace);
    auto take = [&](size_t bytes) { char* p = w; w += (bytes + 255) / 256 * 256; return p; };
    auto* keys = (unsigned long long*)take((size_t)N * cap * 8);
    auto* cnt = (int*)take((size_t)N * 4);
    auto* boxes = (float4*)take((size_t)N * cap * 16);
    auto* scores = (float*)take((size_t)N * cap * 4);
    auto* cats = (int*)take((size_t)N * cap * 4);
    auto* valid = (int*)take((size_t)N * cap * 4);
    auto* mask = (unsigned long long*)take((size_t)N * cap * (cap / 64) * 8);
    auto* keep = (int*)take((size_t)N * cap * 4);
    auto* keep_count = (int*)take((size_t)N * 4);
    hipError_t e = hipMemsetAsync(cnt, 0, (size_t)N * 4, st);
    if (e != hipSuccess) return aldi_set_error(e, __FILE__, __LINE__);
    hipLaunchKernelGGL(det_candidates_kernel, dim3(cdiv((long)P * K, 256), N), dim3(256), 0, st, pred, Cp, K, pcount, P, score_thresh, keys, cnt, err_flag);
    ALDI_CHECK_LAUNCH();
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(det_sort_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kDetCap * 8);
    hipLaunchKernelGGL(det_sort_kernel, dim3(N), dim3(1024), kDetCap * 8, st, pred, Cp, K, (const float4*)props, P, img_hw, weights4[0], weights4[1], weights4[2], weights4[3],
                       keys, cnt, boxes, scores, cats, valid);
    ALDI_CHECK_LAUNCH();
    nms_mask_launch(st, N, boxes, valid, cats, cnt, (int)cap, nms_thresh, mask, aldi_tuning().nms_mask_tri);
    ALDI_CHECK_LAUNCH();
    if (!nms_scan_launch(st, N, mask, valid, cnt, (int)cap, topk, keep, keep_count)) return aldi_set_error_msg(ALDI_ERR_ARG, "detections: NMS capacity too large");
    ALDI_CHECK_LAUNCH();
    hipLaunchKernelGGL(det_finish_kernel, dim3(N), dim3(64), 0, st, boxes, scores, cats, keep, keep_count, topk, pl_rows, pl_thresh,
                       (float4*)det_boxes, det_scores, det_cls, det_count, (float4*)pl_boxes, pl_cls, pl_scores, pl_count);
    ALDI_CHECK_LAUNCH();
    return ALDI_OK;
}
