// RPN-side integer/index work: IoU matching, sampling lists, RPN losses (fwd+bwd), and
// proposal generation (per-level top-k, box decode, batched NMS, post-NMS top-k).
//
// Everything that produces an INDEX is bit-exact against the oracle given identical fp32
// inputs: IoU / box arithmetic follows the oracle's fp32 operation order (compiled with
// -ffp-contract=off, IEEE division), ties resolve to the lower index, ordering is a total
// order on (score desc, index asc).
//
// Replaces detectron2 RPN.label_and_sample_anchors / losses / predict_proposals and
// torchvision batched_nms, reached from the reference at aldi/distill.py:157,162,200-202 and
// aldi/pseudolabeler.py:21.
#include "common.h"
#include "sortscan.h"
#include "nms.h"
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

namespace {

__device__ __forceinline__ float iou_d2(const float4 g, const float4 a) {
    // detectron2 pairwise_iou: inter > 0 ? inter / (area1 + area2 - inter) : 0
    float area1 = (g.z - g.x) * (g.w - g.y);
    float area2 = (a.z - a.x) * (a.w - a.y);
    float w = fminf(g.z, a.z) - fmaxf(g.x, a.x);
    float h = fminf(g.w, a.w) - fmaxf(g.y, a.y);
    w = w > 0.f ? w : 0.f;
    h = h > 0.f ? h : 0.f;
    float inter = w * h;
    return inter > 0.f ? inter / (area1 + area2 - inter) : 0.f;
}

// ---------------------------------------------------------------------------------------
// Matcher (detectron2 Matcher): per box max/argmax IoU over GT (first max wins), per-GT best
// ---------------------------------------------------------------------------------------
constexpr int kGtTile = 256;

// The boxes of one workgroup are 256 neighbouring anchors (one strip of a feature row): only the GT boxes that meet the strip's
// union box can overlap any of them.  Every thread tests one GT against the union, the hits are compacted IN GT ORDER into
// LDS, and the per-box loop runs over that short list (with 100 pseudo-label boxes per image: ~7 instead of 100 iterations,
// none of them a scalar-memory round trip).  Proposals are not spatially ordered: their union is the image, the list all GTs.
struct UnionBox { float x0, y0, x1, y1; };
__device__ __forceinline__ UnionBox block_union(const float4 b, bool active, float (*sbb)[4]) {
    float x0 = active ? b.x : INFINITY, y0 = active ? b.y : INFINITY, x1 = active ? b.z : -INFINITY, y1 = active ? b.w : -INFINITY;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        x0 = fminf(x0, __shfl_xor(x0, o, 64)); y0 = fminf(y0, __shfl_xor(y0, o, 64));
        x1 = fmaxf(x1, __shfl_xor(x1, o, 64)); y1 = fmaxf(y1, __shfl_xor(y1, o, 64));
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sbb[w][0] = x0; sbb[w][1] = y0; sbb[w][2] = x1; sbb[w][3] = y1; }
    __syncthreads();
    UnionBox u{sbb[0][0], sbb[0][1], sbb[0][2], sbb[0][3]};
    for (int k = 1; k < (int)(blockDim.x >> 6); ++k) {
        u.x0 = fminf(u.x0, sbb[k][0]); u.y0 = fminf(u.y0, sbb[k][1]); u.x1 = fmaxf(u.x1, sbb[k][2]); u.y1 = fmaxf(u.y1, sbb[k][3]);
    }
    return u;
}
__device__ __forceinline__ bool meets(const float4 g, const UnionBox u) {
    return fminf(g.z, u.x1) - fmaxf(g.x, u.x0) > 0.f && fminf(g.w, u.y1) - fmaxf(g.y, u.y0) > 0.f;
}

__global__ __launch_bounds__(1024) void match_iou_kernel(const float4* __restrict__ boxes, long box_stride_n, const int* __restrict__ box_count, int L,
                                                        const float4* __restrict__ gt, const int* __restrict__ gt_count, int Gmax,
                                                        float* __restrict__ best_iou, int* __restrict__ best_idx, unsigned* __restrict__ gt_best, int parts) {
    __shared__ float4 sgt[kGtTile];
    __shared__ int sidx[kGtTile];
    __shared__ unsigned smax[kGtTile];
    __shared__ float sbb[16][4];
    __shared__ int sm[17];
    __shared__ float pbest[4][64];
    __shared__ int pidx[4][64];
    // parts == 1: one box per thread.  parts == 4 (blockDim 256): 64 boxes, wave p walks the p-th quarter of the GT list -- the
    // proposals' lists are all GTs, and one wave per 64 boxes walking 100 of them is a 50 us dependent chain on a near-empty chip
    const int part = parts > 1 ? (int)(threadIdx.x >> 6) : 0;
    const int per_wg = (int)blockDim.x / parts;
    const int n = blockIdx.y;
    const int i = blockIdx.x * per_wg + (parts > 1 ? (int)(threadIdx.x & 63) : (int)threadIdx.x);
    const int cnt = box_count ? box_count[n] : L;
    const bool active = i < cnt;
    const int G = gt_count[n];
    const float4 b = active ? boxes[n * box_stride_n + i] : make_float4(0.f, 0.f, 0.f, 0.f);
    const UnionBox ub = block_union(b, active, sbb);
    // (the first GT always takes the running maximum to >= 0 with index 0; later ones must be strictly greater)
    float best = G > 0 ? 0.f : -1.f;
    int bi = 0;
    const int lane = threadIdx.x & 63;
    const int tile = min((int)blockDim.x, kGtTile);
    for (int g0 = 0; g0 < G; g0 += tile) {
        const int gq = g0 + (int)threadIdx.x;
        float4 gq_box = make_float4(0.f, 0.f, 0.f, 0.f);
        bool hit = false;
        if ((int)threadIdx.x < tile && gq < G) { gq_box = gt[n * Gmax + gq]; hit = meets(gq_box, ub); }
        int nl;
        const int rank = block_rank(hit, sm, &nl);          // (its barriers also fence the previous tile's readers)
        if (hit) { sgt[rank] = gq_box; sidx[rank] = gq; smax[rank] = 0u; }
        __syncthreads();
        const int per = (nl + parts - 1) / parts;
        for (int k = part * per; k < min(nl, (part + 1) * per); ++k) {
            const float4 gb = sgt[k];
            const bool ov = active && fminf(gb.z, b.z) - fmaxf(gb.x, b.x) > 0.f && fminf(gb.w, b.w) - fmaxf(gb.y, b.y) > 0.f;
            if (!__ballot(ov)) continue;
            const int g = sidx[k];
            float v = ov ? iou_d2(gb, b) : 0.f;
            if (v > best) { best = v; bi = g; }
            // per-GT best IoU: wave maximum -> workgroup maximum in LDS -> ONE global atomic per (workgroup, GT it overlaps).  The
            // few hundred gt_best words are all the atomics of the launch go to: one per wave was 2/3 of the kernel's time.
            if (__ballot(v > 0.f)) {
                const unsigned wm = wave_max_u32(__float_as_uint(v));       // (IoUs are >= 0: their bit patterns order like the values)
                if (lane == 0) atomicMax(&smax[k], wm);
            }
        }
        __syncthreads();
        if ((int)threadIdx.x < nl && smax[threadIdx.x] != 0u) {
            // same-address device-scope atomics from all 8 XCDs serialise at the memory side (~1 us each, ~70 workgroups per GT):
            // read first and only raise the maximum -- a stale read is lower than the truth, so skipping is always safe
            unsigned* dst = gt_best + n * Gmax + sidx[threadIdx.x];
            if (__hip_atomic_load(dst, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < smax[threadIdx.x]) atomicMax(dst, smax[threadIdx.x]);
        }
    }
    if (parts > 1) {
        // the quarters in GT order: a later one wins only if strictly greater (first maximum, as in one pass); part 0 starts
        // from the (0, index 0) the first GT always establishes, the others from "nothing seen"
        pbest[part][threadIdx.x & 63] = part == 0 || best > 0.f ? best : -2.f;
        pidx[part][threadIdx.x & 63] = bi;
        __syncthreads();
        if (part != 0) return;
        for (int p = 1; p < parts; ++p) {
            const float v = pbest[p][threadIdx.x];
            if (v > best) { best = v; bi = pidx[p][threadIdx.x]; }
        }
    }
    if (active) {
        best_iou[(long)n * L + i] = best;
        best_idx[(long)n * L + i] = bi;
    }
}

// r06: the same matcher with the GT list culled PER WAVE instead of per workgroup.  The kernels are bound by VALU issue (tools/match_bench.py: the time
// grows by ~1 us per GT box, 116 us at 100 boxes on four images; replacing the global atomics by plain stores changes nothing): a workgroup's 1024
// anchors are one feature row of p2, whose union box is as wide as the image and meets every GT that crosses the row (~16 of 100), and every wave walked
// that list.  A wave's 64 anchors are ~21 neighbouring pixels: their union meets 2-3 GTs.  Every wave builds its own list, IN GT ORDER (ballot
// compaction), from the GT tile in LDS; first maximum, per-GT best and the low-quality rule are unchanged -- same labels bit for bit.
struct WaveUnion { float x0, y0, x1, y1; };
__device__ __forceinline__ WaveUnion wave_union(const float4 b, bool active) {
    float x0 = active ? b.x : INFINITY, y0 = active ? b.y : INFINITY, x1 = active ? b.z : -INFINITY, y1 = active ? b.w : -INFINITY;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        x0 = fminf(x0, __shfl_xor(x0, o, 64)); y0 = fminf(y0, __shfl_xor(y0, o, 64));
        x1 = fmaxf(x1, __shfl_xor(x1, o, 64)); y1 = fmaxf(y1, __shfl_xor(y1, o, 64));
    }
    return WaveUnion{x0, y0, x1, y1};
}
// the GTs of the tile in LDS (count nt) that meet this wave's union, ascending, into wl[]; returns their number (wave uniform)
__device__ __forceinline__ int wave_hits(const float4* sgt, const unsigned* skip_zero, int nt, const WaveUnion u, unsigned char* wl) {
    const int lane = threadIdx.x & 63;
    int n = 0;
    for (int t = 0; t < nt; t += 64) {
        const int g = t + lane;
        bool hit = false;
        if (g < nt) {
            const float4 q = sgt[g];
            hit = fminf(q.z, u.x1) - fmaxf(q.x, u.x0) > 0.f && fminf(q.w, u.y1) - fmaxf(q.y, u.y0) > 0.f && (skip_zero == nullptr || skip_zero[g] != 0u);
        }
        const unsigned long long m = __ballot(hit);
        if (hit) wl[n + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned char)g;
        n += __popcll(m);
    }
    return n;
}

template <int NT>
__global__ __launch_bounds__(NT) void match_iou_wave_kernel(const float4* __restrict__ boxes, long box_stride_n, const int* __restrict__ box_count, int L,
                                                            const float4* __restrict__ gt, const int* __restrict__ gt_count, int Gmax,
                                                            float* __restrict__ best_iou, int* __restrict__ best_idx, unsigned* __restrict__ gt_best, int gstride) {
    constexpr int NW = NT / 64;
    static_assert(kGtTile == 256, "GT indices of a tile fit a byte");
    __shared__ float4 sgt[kGtTile];
    __shared__ unsigned smax[kGtTile];
    __shared__ unsigned char wlist[NW][kGtTile];
    const int n = blockIdx.y, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int i = blockIdx.x * NT + tid;
    const int cnt = box_count ? box_count[n] : L;
    const bool active = i < cnt;
    const int G = gt_count[n];
    const float4 b = active ? boxes[n * box_stride_n + i] : make_float4(0.f, 0.f, 0.f, 0.f);
    const WaveUnion u = wave_union(b, active);
    float best = G > 0 ? 0.f : -1.f;       // (the first GT always takes the running maximum to >= 0 with index 0; later ones must be strictly greater)
    int bi = 0;
    for (int g0 = 0; g0 < G; g0 += kGtTile) {
        const int nt = min(kGtTile, G - g0);
        __syncthreads();                   // (the previous tile's readers are done)
        if (tid < kGtTile) {
            smax[tid] = 0u;
            if (tid < nt) sgt[tid] = gt[n * Gmax + g0 + tid];
        }
        __syncthreads();
        const int nl = wave_hits(sgt, nullptr, nt, u, wlist[w]);
        for (int k = 0; k < nl; ++k) {
            const int g = wlist[w][k];
            const float4 gb = sgt[g];
            const bool ov = active && fminf(gb.z, b.z) - fmaxf(gb.x, b.x) > 0.f && fminf(gb.w, b.w) - fmaxf(gb.y, b.y) > 0.f;
            if (!__ballot(ov)) continue;
            const float v = ov ? iou_d2(gb, b) : 0.f;
            if (v > best) { best = v; bi = g0 + g; }
            if (__ballot(v > 0.f)) {
                const unsigned wm = wave_max_u32(__float_as_uint(v));       // (IoUs are >= 0: their bit patterns order like the values)
                if (lane == 0) atomicMax(&smax[g], wm);
            }
        }
        __syncthreads();
        if (tid < nt && smax[tid] != 0u) {
            // (read first and only raise the maximum: a stale read is lower than the truth, so skipping is always safe)
            // gstride 32: every GT's word in its OWN 128-byte line.  Packed (400 bytes for 100 GTs = 4 lines) the ~4 000 device-scope loads / atomics of
            // an image serialise on those lines at the memory side: the matcher's time grew by 0.76 us per GT box (tools/match_bench.py)
            unsigned* dst = gt_best + ((long)n * Gmax + g0 + tid) * gstride;
            if (__hip_atomic_load(dst, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < smax[tid]) atomicMax(dst, smax[tid]);
        }
    }
    if (active) {
        best_iou[(long)n * L + i] = best;
        best_idx[(long)n * L + i] = bi;
    }
}

__global__ __launch_bounds__(256) void match_label_wave_kernel(const float4* __restrict__ boxes, long box_stride_n, const int* __restrict__ box_count, int L,
                                                               const float4* __restrict__ gt, const int* __restrict__ gt_count, int Gmax,
                                                               const float* __restrict__ best_iou, const unsigned* __restrict__ gt_best, int gstride,
                                                               float lo, float hi, int allow_low_quality, int* __restrict__ labels) {
    __shared__ float4 sgt[kGtTile];
    __shared__ unsigned sbest[kGtTile];
    __shared__ unsigned char wlist[4][kGtTile];
    __shared__ int s_zero;
    const int n = blockIdx.y, tid = threadIdx.x, w = tid >> 6;
    const int i = blockIdx.x * blockDim.x + tid;
    const int cnt = box_count ? box_count[n] : L;
    const int G = gt_count[n];
    const bool active = i < cnt;
    int lab = -2;                                               // padding slot, never sampled
    if (active) {
        if (G == 0) lab = 0;
        else {
            const float v = best_iou[(long)n * L + i];
            lab = v < lo ? 0 : (v < hi ? -1 : 1);
        }
    }
    if (allow_low_quality && G > 0) {                           // (block uniform)
        // low-quality rule: IoU == the GT's best over all boxes.  A GT nothing overlaps has best 0 and claims EVERY box (the
        // reference's `match_quality_matrix == highest_quality_foreach_gt`); the others can only claim boxes they overlap.
        const float4 b = active ? boxes[n * box_stride_n + i] : make_float4(0.f, 0.f, 0.f, 0.f);
        const WaveUnion u = wave_union(b, active);
        if (tid == 0) s_zero = 0;
        bool done = false;
        for (int g0 = 0; g0 < G; g0 += kGtTile) {
            const int nt = min(kGtTile, G - g0);
            __syncthreads();
            if (tid < nt) {
                sgt[tid] = gt[n * Gmax + g0 + tid];
                const unsigned top = gt_best[((long)n * Gmax + g0 + tid) * gstride];
                sbest[tid] = top;
                if (top == 0u) s_zero = 1;
            }
            __syncthreads();
            const int nl = wave_hits(sgt, sbest, nt, u, wlist[w]);
            if (active && !done)
                for (int k = 0; k < nl; ++k) {
                    const int g = wlist[w][k];
                    const float4 gb = sgt[g];
                    if (fminf(gb.z, b.z) - fmaxf(gb.x, b.x) > 0.f && fminf(gb.w, b.w) - fmaxf(gb.y, b.y) > 0.f &&
                        __float_as_uint(iou_d2(gb, b)) == sbest[g]) { lab = 1; done = true; break; }
                }
        }
        __syncthreads();
        if (active && s_zero) lab = 1;
    }
    if (i < L) labels[(long)n * L + i] = lab;
}

__global__ __launch_bounds__(256) void match_label_kernel(const float4* __restrict__ boxes, long box_stride_n, const int* __restrict__ box_count, int L,
                                                          const float4* __restrict__ gt, const int* __restrict__ gt_count, int Gmax,
                                                          const float* __restrict__ best_iou, const unsigned* __restrict__ gt_best,
                                                          float lo, float hi, int allow_low_quality, int* __restrict__ labels) {
    __shared__ float4 sgt[kGtTile];
    __shared__ unsigned sbest[kGtTile];
    __shared__ float sbb[4][4];
    __shared__ int sm[17];
    __shared__ int s_zero;
    const int n = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int cnt = box_count ? box_count[n] : L;
    const int G = gt_count[n];
    const bool active = i < cnt;
    int lab = -2;                                               // padding slot, never sampled
    if (active) {
        if (G == 0) lab = 0;
        else {
            const float v = best_iou[(long)n * L + i];
            lab = v < lo ? 0 : (v < hi ? -1 : 1);
        }
    }
    if (allow_low_quality && G > 0) {                           // (block uniform)
        // low-quality rule: IoU == the GT's best over all boxes.  A GT nothing overlaps has best 0 and claims EVERY box (the
        // reference's `match_quality_matrix == highest_quality_foreach_gt`); the others can only claim boxes they overlap.
        const float4 b = active ? boxes[n * box_stride_n + i] : make_float4(0.f, 0.f, 0.f, 0.f);
        const UnionBox ub = block_union(b, active, sbb);
        if (threadIdx.x == 0) s_zero = 0;
        __syncthreads();
        bool done = false;
        for (int g0 = 0; g0 < G; g0 += kGtTile) {
            const int gq = g0 + (int)threadIdx.x;
            float4 gq_box = make_float4(0.f, 0.f, 0.f, 0.f);
            unsigned top = 1u;
            bool hit = false;
            if (gq < G) {
                gq_box = gt[n * Gmax + gq];
                top = gt_best[n * Gmax + gq];
                hit = top != 0u && meets(gq_box, ub);
                if (top == 0u) s_zero = 1;
            }
            int nl;
            const int rank = block_rank(hit, sm, &nl);
            if (hit) { sgt[rank] = gq_box; sbest[rank] = top; }
            __syncthreads();
            if (active && !done)
                for (int k = 0; k < nl; ++k) {
                    const float4 gb = sgt[k];
                    if (fminf(gb.z, b.z) - fmaxf(gb.x, b.x) > 0.f && fminf(gb.w, b.w) - fmaxf(gb.y, b.y) > 0.f &&
                        __float_as_uint(iou_d2(gb, b)) == sbest[k]) { lab = 1; done = true; break; }
                }
        }
        if (active && s_zero) lab = 1;
    }
    if (i < L) labels[(long)n * L + i] = lab;
}

// ---------------------------------------------------------------------------------------
// ordered lists for subsample_labels: pos = (v != -1 && v != bg && v != -2), neg = (v == bg)
// ---------------------------------------------------------------------------------------
// A WAVE owns 1024 consecutive labels as 16 coalesced rows of 64: a ballot per row gives the in-row rank, the 16 popcounts
// the wave total; lanes with the flag set write consecutive output slots (coalesced) straight to global memory.
// Two launches over (segment of 4096 labels, image) instead of one workgroup walking an image's 268 k labels in 16 serial
// passes (114 us on the critical path of the step, 8 workgroups on the chip): count both kinds per segment, then every
// segment adds up the counts before it and writes its part of both lists.
constexpr int kSegLabels = 4096, kSegRows = 16;        // 4 waves x 16 rows of 64
__device__ __forceinline__ void label_flags(int v, int bg, bool& pos, bool& neg) {
    pos = v != -1 && v != -2 && v != bg;
    neg = v == bg;
}
__global__ __launch_bounds__(256) void compact_count_kernel(const int* __restrict__ labels, int L, int bg, int segs, int* __restrict__ segcnt /*[N][2][segs]*/) {
    __shared__ int wsum[2][4];
    const int seg = blockIdx.x, n = blockIdx.y;
    const int* lab = labels + (long)n * L;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int i0 = seg * kSegLabels + w * (64 * kSegRows) + lane;
    int cp = 0, cn = 0;
#pragma unroll
    for (int k = 0; k < kSegRows; ++k) {
        const int i = i0 + k * 64;
        bool fp = false, fn = false;
        if (i < L) label_flags(lab[i], bg, fp, fn);
        cp += __popcll(__ballot(fp));
        cn += __popcll(__ballot(fn));
    }
    if (lane == 0) { wsum[0][w] = cp; wsum[1][w] = cn; }
    __syncthreads();
    if (tid < 2) segcnt[((long)n * 2 + tid) * segs + seg] = wsum[tid][0] + wsum[tid][1] + wsum[tid][2] + wsum[tid][3];
}
__global__ __launch_bounds__(256) void compact_write_kernel(const int* __restrict__ labels, int L, int bg, int segs, const int* __restrict__ segcnt,
                                                            int* __restrict__ lists /*[N][2][L]*/, int* __restrict__ counts /*[N][2]*/) {
    __shared__ int wsum[2][4];
    __shared__ int sbase[2];
    const int seg = blockIdx.x, n = blockIdx.y;
    const int* lab = labels + (long)n * L;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const unsigned long long lt = (1ull << lane) - 1ull;
    // list offsets of this segment: the counts of the segments before it (waves 0 / 1: kind 0 / 1)
    if (w < 2) {
        int a = 0;
        for (int s_ = lane; s_ < seg; s_ += 64) a += segcnt[((long)n * 2 + w) * segs + s_];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
        if (lane == 0) sbase[w] = a;
    }
    const int i0 = seg * kSegLabels + w * (64 * kSegRows) + lane;
    unsigned long long mp[kSegRows], mn[kSegRows];
    int cp = 0, cn = 0;
#pragma unroll
    for (int k = 0; k < kSegRows; ++k) {
        const int i = i0 + k * 64;
        bool fp = false, fn = false;
        if (i < L) label_flags(lab[i], bg, fp, fn);
        mp[k] = __ballot(fp); mn[k] = __ballot(fn);
        cp += __popcll(mp[k]); cn += __popcll(mn[k]);
    }
    if (lane == 0) { wsum[0][w] = cp; wsum[1][w] = cn; }
    __syncthreads();
    int pp = sbase[0], pn = sbase[1], tp = 0, tn = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (i < w) { pp += wsum[0][i]; pn += wsum[1][i]; }
        tp += wsum[0][i]; tn += wsum[1][i];
    }
    int* outp = lists + ((long)n * 2 + 0) * L;
    int* outn = lists + ((long)n * 2 + 1) * L;
#pragma unroll
    for (int k = 0; k < kSegRows; ++k) {
        if ((mp[k] >> lane) & 1ull) outp[pp + __popcll(mp[k] & lt)] = i0 + k * 64;
        if ((mn[k] >> lane) & 1ull) outn[pn + __popcll(mn[k] & lt)] = i0 + k * 64;
        pp += __popcll(mp[k]); pn += __popcll(mn[k]);
    }
    if (seg == segs - 1 && tid == 0) { counts[n * 2 + 0] = sbase[0] + tp; counts[n * 2 + 1] = sbase[1] + tn; }
}

// labels.fill_(-1); labels[pos_list[sel_pos]] = 1; labels[neg_list[sel_neg]] = 0
__global__ void fill_int_kernel(int* p, long n, int v) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = v;
}
__global__ void sample_scatter_kernel(int* __restrict__ labels, int L, const int* __restrict__ lists, const int* __restrict__ sel /*[N][2][S]*/,
                                      const int* __restrict__ nsel /*[N][2]*/, int S) {
    const int n = blockIdx.y, kind = blockIdx.x;
    const int cnt = nsel[n * 2 + kind];
    for (int j = threadIdx.x; j < cnt; j += blockDim.x) {
        int pos = sel[(n * 2 + kind) * S + j];
        int idx = lists[((long)n * 2 + kind) * L + pos];
        labels[(long)n * L + idx] = kind == 0 ? 1 : 0;
    }
}

// ---------------------------------------------------------------------------------------
// RPN losses, forward + backward into the head-output gradient
// ---------------------------------------------------------------------------------------
struct Geom {
    int nl, A, C, sumA;
    int H[ALDI_MAX_LEVELS], W[ALDI_MAX_LEVELS], off[ALDI_MAX_LEVELS + 1];
    float* head[ALDI_MAX_LEVELS];
    float* grad[ALDI_MAX_LEVELS];
};

__device__ __forceinline__ int find_level(const Geom& g, int i) {
    int l = 0;
#pragma unroll
    for (int k = 1; k < ALDI_MAX_LEVELS; ++k)
        if (k < g.nl && i >= g.off[k]) l = k;
    return l;
}

__device__ __forceinline__ float bce_logits(float x, float y) {
    // torch binary_cross_entropy_with_logits: (1-y)*x + max(-x,0) + log(exp(-m) + exp(-x-m)), m = max(-x,0)
    float m = fmaxf(-x, 0.f);
    return (1.f - y) * x + m + logf(expf(-m) + expf(-x - m));
}

__device__ __forceinline__ void box_deltas(const float4 src, const float4 tgt, float wx, float wy, float ww, float wh, float d[4]) {
    float sw = src.z - src.x, sh = src.w - src.y;
    float sx = src.x + 0.5f * sw, sy = src.y + 0.5f * sh;
    float tw = tgt.z - tgt.x, th = tgt.w - tgt.y;
    float tx = tgt.x + 0.5f * tw, ty = tgt.y + 0.5f * th;
    d[0] = wx * (tx - sx) / sw;
    d[1] = wy * (ty - sy) / sh;
    d[2] = ww * logf(tw / sw);
    d[3] = wh * logf(th / sh);
}

__global__ __launch_bounds__(256) void rpn_loss_kernel(Geom g, const float4* __restrict__ anchors, const int* __restrict__ labels,
                                                       const int* __restrict__ matched, const float4* __restrict__ gt, const int* __restrict__ gt_count,
                                                       int Gmax, int N, float inv_norm, float gs_cls, float gs_loc, float* __restrict__ loss /*[2]*/) {
    __shared__ float red[16];
    const long t = blockIdx.x * (long)blockDim.x + threadIdx.x;
    float l_cls = 0.f, l_loc = 0.f;
    if (t < (long)N * g.sumA) {
        const int n = (int)(t / g.sumA), i = (int)(t - (long)n * g.sumA);
        const int lab = labels[t];
        if (lab >= 0) {
            const int l = find_level(g, i);
            const int j = i - g.off[l];
            const int cell = j / g.A, a = j - cell * g.A;
            const long base = ((long)n * g.H[l] * g.W[l] + cell) * g.C;
            const float x = g.head[l][base + a];
            const float y = (float)lab;
            l_cls = bce_logits(x, y);
            if (g.grad[l] && gs_cls != 0.f) g.grad[l][base + a] += (1.f / (1.f + expf(-x)) - y) * inv_norm * gs_cls;
            if (lab == 1) {
                float4 tb = make_float4(0, 0, 0, 0);
                if (gt_count[n] > 0) tb = gt[n * Gmax + matched[t]];
                float d[4];
                box_deltas(anchors[i], tb, 1.f, 1.f, 1.f, 1.f, d);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float pr = g.head[l][base + g.A + a * 4 + k];
                    float df = pr - d[k];
                    l_loc += fabsf(df);
                    if (g.grad[l] && gs_loc != 0.f) g.grad[l][base + g.A + a * 4 + k] += (df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f)) * inv_norm * gs_loc;
                }
            }
        }
    }
    float s0 = block_sum(l_cls, red);
    float s1 = block_sum(l_loc, red);
    if (threadIdx.x == 0) {
        if (s0 != 0.f) unsafeAtomicAdd(loss + 0, s0 * inv_norm);
        if (s1 != 0.f) unsafeAtomicAdd(loss + 1, s1 * inv_norm);
    }
}

// ---------------------------------------------------------------------------------------
// proposals
// ---------------------------------------------------------------------------------------
constexpr int kTopkCap = 2048;     // >= PRE_NMS_TOPK (2000)
constexpr int kMergeCap = 16384;   // >= 5 * 2000

// order-preserving keys of every objectness logit, [N][sumA] (level-major, (h, w, a) inside a level)
// (it also clears the `zero_words` ints of the radix select's histograms / states behind it: a memset node of its own was one more
// dependent launch on the proposal chain)
__global__ void rpn_keys_kernel(Geom g, int N, unsigned* __restrict__ keys, int* __restrict__ zero_ptr, long zero_words) {
    const long t = blockIdx.x * (long)blockDim.x + threadIdx.x;
    for (long z = t; z < zero_words; z += (long)gridDim.x * blockDim.x) zero_ptr[z] = 0;
    if (t >= (long)N * g.sumA) return;
    const int n = (int)(t / g.sumA), i = (int)(t - (long)n * g.sumA);
    const int l = find_level(g, i);
    const int j = i - g.off[l];
    const int cell = j / g.A, a = j - cell * g.A;
    keys[t] = float_key_asc(g.head[l][((long)n * g.H[l] * g.W[l] + cell) * g.C + a]);
}

// Visit every key of a contiguous range with 16-B loads (scalar head/tail for alignment).  `f(key, index, active)` is
// called the same number of times by every thread of the block (it may ballot).
template <class F>
__device__ __forceinline__ void for_each_key(const unsigned* __restrict__ kp, int nel, F f) {
    const int tid = threadIdx.x, nt = blockDim.x;
    int head = (int)((4 - ((reinterpret_cast<uintptr_t>(kp) >> 2) & 3)) & 3);
    if (head > nel) head = nel;
    const int nvec = (nel - head) >> 2;
    for (int base = 0; base < nvec; base += nt) {
        const int v = base + tid;
        const bool ok = v < nvec;
        uint4 q = make_uint4(0, 0, 0, 0);
        if (ok) q = *reinterpret_cast<const uint4*>(kp + head + v * 4);
        const int i0 = head + v * 4;
        f(q.x, i0, ok); f(q.y, i0 + 1, ok); f(q.z, i0 + 2, ok); f(q.w, i0 + 3, ok);
    }
    const int tail0 = head + nvec * 4, rest = head + (nel - tail0);     // < 8 stragglers
    const bool ok = tid < rest;
    const int idx = tid < head ? tid : tail0 + (tid - head);
    f(ok ? kp[idx] : 0u, idx, ok);
}

// LDS histogram increment with wave aggregation: objectness logits cluster (at init nearly all share the leading key
// bits), and 64 lanes hitting one LDS counter serialise.  Up to three leader rounds fold equal buckets into one add.
__device__ __forceinline__ void hist_add(int* hist, unsigned bucket, bool active) {
    const int lane = threadIdx.x & 63;
    unsigned long long todo = __ballot(active);
#pragma unroll 1
    for (int r = 0; r < 3 && todo; ++r) {
        const int leader = __ffsll((long long)todo) - 1;
        const unsigned bsel = (unsigned)__builtin_amdgcn_readlane((int)bucket, leader);
        const unsigned long long same = __ballot(active && bucket == bsel) & todo;
        if (lane == leader) atomicAdd(&hist[bsel], __popcll(same));
        todo &= ~same;
    }
    if ((todo >> lane) & 1ull) atomicAdd(&hist[bucket], 1);
}

// ---------------------------------------------------------------------------------------
// Exact top-k per (level, image) by (logit desc, index asc): radix select (12+12+8 bits) of the k-th largest key, one
// streaming pass that collects the winners (unordered; a bitonic sort on (key desc, index asc) fixes the order), and an
// index-ordered pass only when MORE keys tie with the k-th than fit.  One workgroup per (level, image) would leave a
// 200K-key level on a single CU (the radix passes are instruction bound there), so a level is cut into 8192-key chunks,
// every pass is its own launch (the launch boundary is the grid-wide barrier), and the per-(level, image) histograms live in global
// memory:   hist(pass 0) -> [find bucket] hist(pass 1) -> [find] hist(pass 2) -> [find] collect -> sort.
// Each workgroup re-derives the selected bucket from the previous pass's histogram (4096 bins, trivial) instead of
// waiting for a separate "find" launch; chunk 0 records the chain state for the next launch.
// ---------------------------------------------------------------------------------------
constexpr int kTopkChunk = 8192;
struct TopkState { unsigned prefix, mask; int need, bucket_count; };

// highest bucket b with count(buckets > b) < need <= count(buckets >= b); 1024 threads, nb in {256, 4096}
__device__ __forceinline__ void find_bucket(const int* __restrict__ hist, int nb, int need, int* sm /*>=20*/, int* bucket, int* above, int* bcount) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int per = nb / 1024 > 0 ? nb / 1024 : 1;           // 4 bins per thread for 4096, 1 for 256 (threads >= nb idle)
    int c[4] = {0, 0, 0, 0}, sum = 0;
    const int hi = nb - 1 - tid * per;                       // thread 0 owns the highest bins
    if (hi >= 0)
        for (int k = 0; k < per; ++k) { c[k] = hist[hi - k]; sum += c[k]; }
    int incl = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    __syncthreads();
    if (lane == 63) sm[w] = incl;
    if (tid == 0) { sm[17] = 0; sm[18] = 0; sm[19] = 0; }
    __syncthreads();
    int wb = 0;
    for (int i = 0; i < w; ++i) wb += sm[i];
    const int excl = wb + incl - sum;
    if (excl < need && need <= excl + sum) {                 // exactly one thread when need <= total
        int acc = excl;
        for (int k = 0; k < per; ++k) {
            if (acc + c[k] >= need) { sm[17] = hi - k; sm[18] = acc; sm[19] = c[k]; break; }
            acc += c[k];
        }
    }
    __syncthreads();
    *bucket = sm[17]; *above = sm[18]; *bcount = sm[19];
    __syncthreads();
}

// pass p (0,1,2): advance the chain with the previous histogram, then histogram this chunk's matching keys
__global__ __launch_bounds__(1024) void topk_hist_kernel(Geom g, const unsigned* __restrict__ keys_all, int pre_nms_topk, int pass,
                                                         int* __restrict__ hists /*[3][B][4096]*/, TopkState* __restrict__ state /*[B]*/) {
    __shared__ int hist[4096];
    __shared__ int sm[20];
    const int c = blockIdx.x, l = blockIdx.y, n = blockIdx.z, B = gridDim.y * gridDim.z, bl = n * g.nl + l;
    const int nel = g.H[l] * g.W[l] * g.A;
    if (c * kTopkChunk >= nel) return;
    const int shifts[3] = {20, 8, 0}, widths[3] = {12, 12, 8};
    TopkState s;
    s.prefix = 0; s.mask = 0; s.need = min(pre_nms_topk, nel); s.bucket_count = 0;      // chain start (passes 0 and 1)
    if (pass > 0) {
        if (pass > 1) s = state[B + bl];                   // what pass 1 left (slot 1); this pass writes slot 0, which the collect reads
        int b, above, bc;
        find_bucket(hists + ((long)(pass - 1) * B + bl) * 4096, 1 << widths[pass - 1], s.need, sm, &b, &above, &bc);
        s.prefix |= (unsigned)b << shifts[pass - 1];
        s.mask |= (unsigned)((1 << widths[pass - 1]) - 1) << shifts[pass - 1];
        s.need -= above;
        s.bucket_count = bc;
    }
    const int shift = shifts[pass], nb = 1 << widths[pass];
    for (int i = threadIdx.x; i < nb; i += 1024) hist[i] = 0;
    __syncthreads();
    const unsigned* kp = keys_all + (long)n * g.sumA + g.off[l] + (long)c * kTopkChunk;
    const int cnt = min(kTopkChunk, nel - c * kTopkChunk);
    for_each_key(kp, cnt, [&](unsigned key, int, bool ok) {
        hist_add(hist, (key >> shift) & (nb - 1), ok && (key & s.mask) == s.prefix);
    });
    __syncthreads();
    int* out = hists + ((long)pass * B + bl) * 4096;
    for (int i = threadIdx.x; i < nb; i += 1024) {
        const int v = hist[i];
        if (v) atomicAdd(out + i, v);
    }
    // the next launch reads the state this launch started from (+ this launch's bucket).  The two slots alternate (pass 1 writes
    // slot 1, pass 2 reads slot 1 and writes slot 0), so no workgroup of a launch reads the slot another one writes
    if (c == 0 && threadIdx.x == 0 && pass > 0) state[(pass == 1 ? B : 0) + bl] = s;
}

// final: k-th key from the last histogram, collect the winners of this chunk (unordered append)
__global__ __launch_bounds__(1024) void topk_collect_kernel(Geom g, const unsigned* __restrict__ keys_all, int* __restrict__ hists, const TopkState* __restrict__ state,
                                                            unsigned long long* __restrict__ cand /*[B][kTopkCap]*/, int* __restrict__ fill /*[B]*/,
                                                            TopkState* __restrict__ final_state /*[B]*/) {
    __shared__ int sm[20];
    const int c = blockIdx.x, l = blockIdx.y, n = blockIdx.z, B = gridDim.y * gridDim.z, bl = n * g.nl + l;
    const int nel = g.H[l] * g.W[l] * g.A;
    if (c * kTopkChunk >= nel) return;
    TopkState s = state[bl];
    int b, above, bc;
    find_bucket(hists + ((long)2 * B + bl) * 4096, 256, s.need, sm, &b, &above, &bc);
    const unsigned kth = s.prefix | (unsigned)b;
    const int need = s.need - above;                 // copies of kth to take (lowest indices), of `bc`
    const bool take_all_eq = bc == need;
    if (c == 0 && threadIdx.x == 0) { TopkState f; f.prefix = kth; f.mask = ~0u; f.need = need; f.bucket_count = bc; final_state[bl] = f; }
    const unsigned* kp = keys_all + (long)n * g.sumA + g.off[l] + (long)c * kTopkChunk;
    const int cnt = min(kTopkChunk, nel - c * kTopkChunk);
    unsigned long long* out = cand + (long)bl * kTopkCap;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // ONE returning atomic per workgroup: count the chunk's winners first (wave ballots), reserve their slots together, then
    // write.  A returning atomic per wave and key row put ~1800 serialised round trips on each (level, image) counter -- 90 us
    // of the student's proposal chain.
    int mine = 0;
    for_each_key(kp, cnt, [&](unsigned key, int, bool ok) {
        mine += __popcll(__ballot(ok && (key > kth || (take_all_eq && key == kth))));
    });
    __syncthreads();                                  // (find_bucket's readers of sm are done)
    if (lane == 0) sm[wave] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        int tot = 0;
        for (int w = 0; w < 16; ++w) { const int v = sm[w]; sm[w] = tot; tot += v; }
        sm[16] = tot ? atomicAdd(fill + bl, tot) : 0;
    }
    __syncthreads();
    int pos = sm[16] + sm[wave];
    for_each_key(kp, cnt, [&](unsigned key, int i, bool ok) {
        const bool win = ok && (key > kth || (take_all_eq && key == kth));
        const unsigned long long m = __ballot(win);
        if (win) out[pos + __popcll(m & ((1ull << lane) - 1ull))] = ((unsigned long long)(~key) << 32) | (unsigned)(c * kTopkChunk + i);
        pos += __popcll(m);
    });
}

// per (level, image): bring the candidates into LDS, resolve ties beyond k (rare), sort by (key desc, index asc)
__device__ __forceinline__ float4 apply_deltas_d2(const float4 box, float dx, float dy, float dw, float dh, float wx, float wy, float ww, float wh) {
    const float clampv = 4.135166556742356f;   // log(1000/16)
    float w = box.z - box.x, h = box.w - box.y;
    float cx = box.x + 0.5f * w, cy = box.y + 0.5f * h;
    dx = dx / wx; dy = dy / wy; dw = dw / ww; dh = dh / wh;
    dw = fminf(dw, clampv); dh = fminf(dh, clampv);
    float pcx = dx * w + cx, pcy = dy * h + cy;
    float pw = expf(dw) * w, ph = expf(dh) * h;
    return make_float4(pcx - 0.5f * pw, pcy - 0.5f * ph, pcx + 0.5f * pw, pcy + 0.5f * ph);
}

__device__ __forceinline__ float clampf(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }

// decode + clip + validity of sorted candidate r of (level l, image n); k = the number of candidates of that list
__device__ __forceinline__ void decode_candidate(const Geom& g, const float4* __restrict__ anchors, unsigned long long key, int r, int k, int l, int n,
                                                 const int* __restrict__ img_hw /*[N][2]*/, float4* __restrict__ boxes /*[N][nl][cap]*/,
                                                 float* __restrict__ scores, int* __restrict__ valid, int* __restrict__ err) {
    const long slot = ((long)n * g.nl + l) * kTopkCap + r;
    if (r >= k) { valid[slot] = 0; return; }
    int i = (int)(key & 0xffffffffu);
    int cell = i / g.A, a = i - cell * g.A;
    const float* hp = g.head[l] + ((long)n * g.H[l] * g.W[l] + cell) * g.C;
    float sc = hp[a];
    float4 b = apply_deltas_d2(anchors[g.off[l] + i], hp[g.A + a * 4 + 0], hp[g.A + a * 4 + 1], hp[g.A + a * 4 + 2], hp[g.A + a * 4 + 3], 1.f, 1.f, 1.f, 1.f);
    bool fin = isfinite(b.x) && isfinite(b.y) && isfinite(b.z) && isfinite(b.w) && isfinite(sc);
    if (!fin) atomicOr(err, 1);
    float ih = (float)img_hw[n * 2], iw = (float)img_hw[n * 2 + 1];
    b.x = clampf(b.x, 0.f, iw); b.y = clampf(b.y, 0.f, ih); b.z = clampf(b.z, 0.f, iw); b.w = clampf(b.w, 0.f, ih);
    bool ok = fin && (b.z - b.x) > 0.f && (b.w - b.y) > 0.f;
    boxes[slot] = b;
    scores[slot] = sc;
    valid[slot] = ok ? 1 : 0;
}

__global__ __launch_bounds__(1024) void topk_sort_kernel(Geom g, const unsigned* __restrict__ keys_all, int pre_nms_topk, const TopkState* __restrict__ final_state,
                                                         const int* __restrict__ fill, unsigned long long* __restrict__ cand, int* __restrict__ cand_count,
                                                         const float4* __restrict__ anchors, const int* __restrict__ img_hw, float4* __restrict__ boxes,
                                                         float* __restrict__ scores, int* __restrict__ valid, int* __restrict__ err) {
    __shared__ unsigned long long keys[kTopkCap];
    __shared__ int sm[17];
    const int l = blockIdx.x, n = blockIdx.y, bl = n * g.nl + l;
    const int nel = g.H[l] * g.W[l] * g.A;
    const int k = min(pre_nms_topk, nel);
    const TopkState f = final_state[bl];
    const int have = fill[bl];
    unsigned long long* io = cand + (long)bl * kTopkCap;
    for (int i = threadIdx.x; i < kTopkCap; i += 1024) keys[i] = i < have ? io[i] : ~0ull;
    __syncthreads();
    if (f.bucket_count != f.need) {                  // more copies of the k-th key than fit: take the lowest indices
        const unsigned* kp = keys_all + (long)n * g.sumA + g.off[l];
        int base_eq = 0;
        for (int s0 = 0; s0 < nel && base_eq < f.need; s0 += 1024) {
            const int i = s0 + threadIdx.x;
            const bool eq = i < nel && kp[i] == f.prefix;
            int te;
            const int re = block_rank(eq, sm, &te);
            if (eq && base_eq + re < f.need) keys[(k - f.need) + base_eq + re] = ((unsigned long long)(~f.prefix) << 32) | (unsigned)i;
            base_eq += te;
        }
        __syncthreads();
    }
    bitonic_sort_u64(keys, kTopkCap);
    // ... and the sorted candidates are decoded here (boxes, scores, validity): a launch of its own was one more dependent step of the proposal chain
    for (int i = threadIdx.x; i < kTopkCap; i += 1024) {
        io[i] = keys[i];
        decode_candidate(g, anchors, keys[i], i, k, l, n, img_hw, boxes, scores, valid, err);
    }
    if (threadIdx.x == 0) cand_count[bl] = k;
}

// ---------------------------------------------------------------------------------------
// The whole exact top-k of a (level, image) list -- the three radix passes, the collect and the sort + decode -- as ONE launch.  The five
// launches above sit on the proposal chain, which ends phase A with the chip idle (~20 dependent launches, 0.37 ms): a
// dependent launch costs 8-20 us of that chain in a replayed graph whatever it computes.  Here the launch boundaries between the passes are
// barriers among the workgroups of ONE (level, image) group (1-25 of them: every workgroup owns an 8192-key chunk):
//   arrive = plain stores / device-scope atomics -> __syncthreads -> lane 0: agent-scope release fence, s_waitcnt, relaxed atomic add;
//   wait   = lane 0 polls the counter with relaxed agent-scope loads (s_sleep between polls), ONE agent-scope acquire fence, __syncthreads
// (cdna_hip_programming.md, guideline 16).  The groups are independent, all workgroups of a launch fit the chip several times over (<= 500 of
// 1024 threads: a spinning workgroup never keeps a peer from being scheduled), and the poll is BOUNDED: after ~2^22 polls a workgroup gives up,
// raises bit 3 of the error word and goes on (wrong proposals, flagged -- never a hang).  The LAST workgroup of a group to finish its collect
// (a ticket) sorts and decodes the group's candidates; the chain state stays in registers, the counters are cleared by rpn_keys_kernel.
// Same arithmetic as topk_hist / topk_collect / topk_sort: bit-identical candidates.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void topk_group_barrier(int* counter, int P, int* err) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < P) {
            __builtin_amdgcn_s_sleep(8);
            if (++spins > (1 << 22)) { atomicOr(err, 8); break; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

__global__ __launch_bounds__(1024) void topk_fused_kernel(Geom g, const unsigned* __restrict__ keys_all, int pre_nms_topk, int* __restrict__ hists /*[3][B][4096]*/,
                                                          int* __restrict__ sync /*[B][4]*/, int* __restrict__ fill /*[B]*/, unsigned long long* __restrict__ cand,
                                                          int* __restrict__ cand_count, const float4* __restrict__ anchors, const int* __restrict__ img_hw,
                                                          float4* __restrict__ boxes, float* __restrict__ scores, int* __restrict__ valid, int* __restrict__ err) {
    __shared__ unsigned long long buf[kTopkCap];     // the passes' 4096-bin histogram, then the sorter's keys (16 KB)
    __shared__ int sm[20];
    __shared__ int s_last;
    int* hist = reinterpret_cast<int*>(buf);
    const int c = blockIdx.x, l = blockIdx.y, n = blockIdx.z, B = gridDim.y * gridDim.z, bl = n * g.nl + l;
    const int nel = g.H[l] * g.W[l] * g.A;
    if (c * kTopkChunk >= nel) return;
    const int P = (nel + kTopkChunk - 1) / kTopkChunk;
    const int shifts[3] = {20, 8, 0}, widths[3] = {12, 12, 8};
    const unsigned* kp = keys_all + (long)n * g.sumA + g.off[l] + (long)c * kTopkChunk;
    const int cnt = min(kTopkChunk, nel - c * kTopkChunk);
    TopkState s;
    s.prefix = 0; s.mask = 0; s.need = min(pre_nms_topk, nel); s.bucket_count = 0;
    for (int pass = 0; pass < 3; ++pass) {
        if (pass > 0) {
            int b, above, bc;
            find_bucket(hists + ((long)(pass - 1) * B + bl) * 4096, 1 << widths[pass - 1], s.need, sm, &b, &above, &bc);
            s.prefix |= (unsigned)b << shifts[pass - 1];
            s.mask |= (unsigned)((1 << widths[pass - 1]) - 1) << shifts[pass - 1];
            s.need -= above;
            s.bucket_count = bc;
        }
        const int shift = shifts[pass], nb = 1 << widths[pass];
        for (int i = threadIdx.x; i < nb; i += 1024) hist[i] = 0;
        __syncthreads();
        for_each_key(kp, cnt, [&](unsigned key, int, bool ok) {
            hist_add(hist, (key >> shift) & (nb - 1), ok && (key & s.mask) == s.prefix);
        });
        __syncthreads();
        int* out = hists + ((long)pass * B + bl) * 4096;
        for (int i = threadIdx.x; i < nb; i += 1024) {
            const int v = hist[i];
            if (v) atomicAdd(out + i, v);
        }
        topk_group_barrier(sync + bl * 4 + pass, P, err);
    }
    // ---- the k-th key, this chunk's winners (unordered append)
    int b, above, bc;
    find_bucket(hists + ((long)2 * B + bl) * 4096, 256, s.need, sm, &b, &above, &bc);
    const unsigned kth = s.prefix | (unsigned)b;
    const int need = s.need - above;                 // copies of kth to take (lowest indices), of `bc`
    const bool take_all_eq = bc == need;
    unsigned long long* out = cand + (long)bl * kTopkCap;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int mine = 0;
    for_each_key(kp, cnt, [&](unsigned key, int, bool ok) {
        mine += __popcll(__ballot(ok && (key > kth || (take_all_eq && key == kth))));
    });
    __syncthreads();
    if (lane == 0) sm[wave] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        int tot = 0;
        for (int w = 0; w < 16; ++w) { const int v = sm[w]; sm[w] = tot; tot += v; }
        sm[16] = tot ? atomicAdd(fill + bl, tot) : 0;
    }
    __syncthreads();
    int pos = sm[16] + sm[wave];
    for_each_key(kp, cnt, [&](unsigned key, int i, bool ok) {
        const bool win = ok && (key > kth || (take_all_eq && key == kth));
        const unsigned long long m = __ballot(win);
        if (win) out[pos + __popcll(m & ((1ull << lane) - 1ull))] = ((unsigned long long)(~key) << 32) | (unsigned)(c * kTopkChunk + i);
        pos += __popcll(m);
    });
    // ---- ticket: the group's last workgroup sorts
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int t = __hip_atomic_fetch_add(sync + bl * 4 + 3, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = t == P - 1;
        if (t == P - 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    if (!s_last) return;
    const int k = min(pre_nms_topk, nel);
    const int have = __hip_atomic_load(fill + bl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned long long* keys = buf;
    for (int i = threadIdx.x; i < kTopkCap; i += 1024) keys[i] = i < have ? out[i] : ~0ull;
    __syncthreads();
    if (bc != need) {                                // more copies of the k-th key than fit: take the lowest indices
        const unsigned* kall = keys_all + (long)n * g.sumA + g.off[l];
        int base_eq = 0;
        for (int s0 = 0; s0 < nel && base_eq < need; s0 += 1024) {
            const int i = s0 + threadIdx.x;
            const bool eq = i < nel && kall[i] == kth;
            int te;
            const int re = block_rank(eq, sm, &te);
            if (eq && base_eq + re < need) keys[(k - need) + base_eq + re] = ((unsigned long long)(~kth) << 32) | (unsigned)i;
            base_eq += te;
        }
        __syncthreads();
    }
    bitonic_sort_u64(keys, kTopkCap);
    for (int i = threadIdx.x; i < kTopkCap; i += 1024) {
        out[i] = keys[i];
        decode_candidate(g, anchors, keys[i], i, k, l, n, img_hw, boxes, scores, valid, err);
    }
    if (threadIdx.x == 0) cand_count[bl] = k;
}

// merge the per-level survivors of one image by (score desc, level asc, rank asc); keep post_nms_topk.
// Every level's survivor list is already in that order (NMS keeps score order), so an element's final position is its
// own rank plus, for every other level, the number of that level's elements ordered before it: a binary search per
// level over keys staged in LDS -- O(n log n) with no barriers instead of a 16K-element bitonic network.
__global__ __launch_bounds__(1024) void rpn_merge_kernel(int nl, const float4* __restrict__ boxes, const float* __restrict__ scores,
                                                         const int* __restrict__ keep, const int* __restrict__ keep_count, int post_topk,
                                                         float4* __restrict__ out_boxes /*[N][post]*/, float* __restrict__ out_scores, int* __restrict__ out_count) {
    extern __shared__ unsigned mkeys[];          // kMergeCap keys, ascending key == descending score
    __shared__ int lbase[ALDI_MAX_LEVELS + 1];
    const int n = blockIdx.x;
    if (threadIdx.x == 0) {
        int acc = 0;
        for (int l = 0; l < nl; ++l) { lbase[l] = acc; acc += keep_count[n * nl + l]; }
        lbase[nl] = acc;
    }
    __syncthreads();
    const int total = lbase[nl];
    for (int l = 0; l < nl; ++l) {
        const int bl = n * nl + l, kc = lbase[l + 1] - lbase[l];
        for (int j = threadIdx.x; j < kc; j += blockDim.x)
            mkeys[lbase[l] + j] = ~float_key_asc(scores[(long)bl * kTopkCap + keep[(long)bl * kTopkCap + j]]);
    }
    __syncthreads();
    const int cnt = min(total, post_topk);
    for (int e = threadIdx.x; e < total; e += blockDim.x) {
        int l = 0;
        while (e >= lbase[l + 1]) ++l;
        const int j = e - lbase[l];
        const unsigned key = mkeys[e];
        int pos = j;
        for (int o = 0; o < nl; ++o) {
            if (o == l) continue;
            // elements of level o ordered before (key, l): key_o < key, or key_o == key when o < l
            const unsigned* ko = mkeys + lbase[o];
            int lo = 0, hi = lbase[o + 1] - lbase[o];
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                const unsigned km = ko[mid];
                const bool before = o < l ? km <= key : km < key;
                if (before) lo = mid + 1; else hi = mid;
            }
            pos += lo;
        }
        if (pos < post_topk) {
            const long slot = ((long)n * nl + l) * kTopkCap + keep[((long)n * nl + l) * kTopkCap + j];
            out_boxes[(long)n * post_topk + pos] = boxes[slot];
            out_scores[(long)n * post_topk + pos] = scores[slot];
        }
    }
    for (int j = cnt + threadIdx.x; j < post_topk; j += blockDim.x) {
        out_boxes[(long)n * post_topk + j] = make_float4(0, 0, 0, 0);
        out_scores[(long)n * post_topk + j] = 0.f;
    }
    if (threadIdx.x == 0) out_count[n] = cnt;
}

Geom make_geom(const aldi_rpn_geom* gm, float* const* head, float* const* grad) {
    Geom g;
    g.nl = gm->num_levels; g.A = gm->A; g.C = gm->C; g.sumA = gm->off[gm->num_levels];
    for (int l = 0; l < ALDI_MAX_LEVELS; ++l) {
        g.H[l] = gm->H[l]; g.W[l] = gm->W[l]; g.off[l] = gm->off[l];
        g.head[l] = head ? head[l] : nullptr;
        g.grad[l] = grad ? grad[l] : nullptr;
    }
    g.off[ALDI_MAX_LEVELS] = gm->off[ALDI_MAX_LEVELS];
    return g;
}

}  // namespace

extern "C" int aldi_box_match(const float* boxes, long box_stride_n, const int* box_count, int L,
                              const float* gt_boxes, const int* gt_count, int Gmax, int N,
                              float lo, float hi, int allow_low_quality,
                              float* best_iou, int* best_idx, unsigned* gt_best_scratch, size_t gt_best_bytes, int* labels, aldi_stream_t stream) {
    if (!boxes || !gt_boxes || !gt_count || !best_iou || !best_idx || !gt_best_scratch || !labels) return aldi_set_error_msg(ALDI_ERR_ARG, "box_match: null pointer");
    if (gt_best_bytes < sizeof(unsigned) * (size_t)N * Gmax) return aldi_set_error_msg(ALDI_ERR_ARG, "box_match: gt_best_scratch smaller than N * Gmax words");
    hipStream_t st = static_cast<hipStream_t>(stream);
    // one word per GT, or (a scratch of N * Gmax * 128 bytes) one 128-byte line per GT: see match_iou_wave_kernel
    const int gstride = gt_best_bytes >= (size_t)N * Gmax * 128 ? 32 : 1;
    hipError_t e = hipMemsetAsync(gt_best_scratch, 0, sizeof(unsigned) * (size_t)N * Gmax * gstride, st);
    if (e != hipSuccess) return aldi_set_error(e, __FILE__, __LINE__);
    dim3 grid(cdiv(L, 256), N);
    // anchors (spatially ordered, ~268 k per image): big workgroups = few global atomics per GT; proposals (a few thousand,
    // unordered: every GT is on every workgroup's list): one wave per workgroup so that they spread over the chip
    const bool big = (long)L * N >= (1 << 16);
    const int parts = !big && Gmax <= kGtTile ? 4 : 1;       // (the split keeps GT order only within one list tile)
    const bool wave_cull = (big && aldi_tuning().match_wave != 0) || gstride != 1;       // (the padded scratch is the wave kernels' layout)     // the anchors: per-wave GT lists (r06; match_wave 0 = the per-workgroup lists)
    if (wave_cull) {
        hipLaunchKernelGGL(match_iou_wave_kernel<1024>, dim3(cdiv(L, 1024), N), dim3(1024), 0, st, (const float4*)boxes, box_stride_n, box_count, L,
                           (const float4*)gt_boxes, gt_count, Gmax, best_iou, best_idx, gt_best_scratch, gstride);
        ALDI_CHECK_LAUNCH();
        hipLaunchKernelGGL(match_label_wave_kernel, grid, dim3(256), 0, st, (const float4*)boxes, box_stride_n, box_count, L, (const float4*)gt_boxes, gt_count, Gmax,
                           best_iou, gt_best_scratch, gstride, lo, hi, allow_low_quality, labels);
        ALDI_CHECK_LAUNCH();
        return ALDI_OK;
    }
    hipLaunchKernelGGL(match_iou_kernel, dim3(cdiv(L, big ? 1024 : 64), N), dim3(big ? 1024 : 64 * parts), 0, st, (const float4*)boxes, box_stride_n, box_count, L,
                       (const float4*)gt_boxes, gt_count, Gmax, best_iou, best_idx, gt_best_scratch, parts);
    ALDI_CHECK_LAUNCH();
    hipLaunchKernelGGL(match_label_kernel, grid, dim3(256), 0, st, (const float4*)boxes, box_stride_n, box_count, L, (const float4*)gt_boxes, gt_count, Gmax,
                       best_iou, gt_best_scratch, lo, hi, allow_low_quality, labels);
    ALDI_CHECK_LAUNCH();
    return ALDI_OK;
}

namespace {
// StandardROIHeads.label_and_sample_proposals up to the ordered lists in ONE launch: append the ground truth to the proposals, match
// (IoU >= thr, first maximum wins, no low-quality rule), classes, and the ordered foreground / background lists with their lengths.
// The same arithmetic as roi_append_gt + match_iou + match_label + roi_classes + compact_count + compact_write (aldi_roi_prepare,
// aldi_compact_labels): eight launches of a few microseconds each on the chain that ends in the list lengths the host waits for, with
// nothing left to run beside them at that point of the step -- a dependent small launch costs ~10 us of that chain in a replayed graph.
// kRoiPrepWgs workgroups per image match a slice of the candidates each (a candidate per thread, every ground-truth box from LDS); the
// LAST one of an image to finish (a ticket per image, reset by its taker) writes the two ordered lists from the classes.
constexpr int kRoiPrepWgs = 8;
__global__ __launch_bounds__(256) void roi_prepare_fused_kernel(const float4* __restrict__ props, const int* __restrict__ pcount, int P,
                                                                const float4* __restrict__ gt, const int* __restrict__ gt_classes, const int* __restrict__ gcount,
                                                                int Gmax, int K, float thr, int L, float4* __restrict__ cand, int* __restrict__ ccount,
                                                                float* __restrict__ best_iou, int* __restrict__ best_idx, int* __restrict__ labels,
                                                                int* __restrict__ cls, int* __restrict__ lists /*[N][2][L]*/, int* __restrict__ counts /*[N][2]*/,
                                                                unsigned* __restrict__ tickets /*[N], zero*/, const int* __restrict__ tail_a,
                                                                const int* __restrict__ tail_b /* words copied to counts[2N], counts[2N + 1] */) {
    __shared__ float4 sgt[kGtTile];
    __shared__ int sgc[kGtTile];
    __shared__ int wsum[2][4];
    __shared__ int s_last;
    const int n = blockIdx.y, part = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int pc = pcount[n], gc = gcount[n], cnt = pc + gc;
    if (part == 0 && tid == 0) ccount[n] = cnt;
    for (int g = tid; g < gc; g += (int)blockDim.x) { sgt[g] = gt[(long)n * Gmax + g]; sgc[g] = gt_classes[n * Gmax + g]; }
    __syncthreads();
    const int per = (L + kRoiPrepWgs - 1) / kRoiPrepWgs;
    const int i_end = min(L, (part + 1) * per);
    for (int i = part * per + tid; i < i_end; i += (int)blockDim.x) {
        float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < pc) b = props[(long)n * P + i];
        else if (i < cnt) b = sgt[i - pc];
        const bool active = i < cnt;
        float best = gc > 0 ? 0.f : -1.f;
        int bi = 0;
        if (active)
            for (int g = 0; g < gc; ++g) {
                const float4 gb = sgt[g];
                const bool ov = fminf(gb.z, b.z) - fmaxf(gb.x, b.x) > 0.f && fminf(gb.w, b.w) - fmaxf(gb.y, b.y) > 0.f;
                const float v = ov ? iou_d2(gb, b) : 0.f;
                if (v > best) { best = v; bi = g; }
            }
        int lab = -2;
        if (active) lab = gc == 0 ? 0 : (best < thr ? 0 : 1);
        int c = -2;
        if (lab != -2) c = gc == 0 ? K : (lab == 1 ? sgc[bi] : K);
        const long o = (long)n * L + i;
        cand[o] = b;
        if (active) { best_iou[o] = best; best_idx[o] = bi; }
        labels[o] = lab;
        cls[o] = c;
    }
    // the image's last workgroup compacts
    __threadfence();
    __syncthreads();
    if (tid == 0) {
        const unsigned t = atomicAdd(&tickets[n], 1u);
        s_last = t == (unsigned)(kRoiPrepWgs - 1);
        if (s_last) tickets[n] = 0u;                     // (for the next launch: nobody else touches it any more)
    }
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    const volatile int* vcls = cls + (long)n * L;
    const unsigned long long lt = (1ull << lane) - 1ull;
    int base_p = 0, base_n = 0;
    for (int i0 = 0; i0 < L; i0 += (int)blockDim.x) {
        const int i = i0 + tid;
        bool fp = false, fn = false;
        if (i < L) label_flags(vcls[i], K, fp, fn);
        const unsigned long long mp = __ballot(fp), mn = __ballot(fn);
        if (lane == 0) { wsum[0][w] = __popcll(mp); wsum[1][w] = __popcll(mn); }
        __syncthreads();
        int pp = base_p, pn = base_n, tp = 0, tn = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (k < w) { pp += wsum[0][k]; pn += wsum[1][k]; }
            tp += wsum[0][k]; tn += wsum[1][k];
        }
        if (fp) lists[((long)n * 2 + 0) * L + pp + __popcll(mp & lt)] = i;
        if (fn) lists[((long)n * 2 + 1) * L + pn + __popcll(mn & lt)] = i;
        base_p += tp; base_n += tn;
        __syncthreads();
    }
    if (tid == 0) {
        counts[n * 2] = base_p; counts[n * 2 + 1] = base_n;
        if (n == 0 && tail_a) { counts[2 * (int)gridDim.y] = *tail_a; counts[2 * (int)gridDim.y + 1] = tail_b ? *tail_b : 0; }
    }
}
}  // namespace

extern "C" int aldi_roi_prepare_lists(const float* props, const int* pcount, int P, const float* gt_boxes, const int* gt_classes, const int* gt_count,
                                      int Gmax, int N, int K, float iou_thresh, float* cand, int* ccount, float* best_iou, int* best_idx, int* labels,
                                      int* cls, int* lists, int* counts, unsigned* tickets, const int* tail_a, const int* tail_b, aldi_stream_t stream) {
    if (!props || !pcount || !gt_boxes || !gt_classes || !gt_count || !cand || !ccount || !best_iou || !best_idx || !labels || !cls || !lists || !counts || !tickets)
        return aldi_set_error_msg(ALDI_ERR_ARG, "roi_prepare_lists: null pointer");
    if (Gmax > kGtTile || Gmax < 1 || N < 1 || P < 0) return aldi_set_error_msg(ALDI_ERR_ARG, "roi_prepare_lists: Gmax must be 1 .. 256");
    const int L = P + Gmax;
    hipLaunchKernelGGL(roi_prepare_fused_kernel, dim3(kRoiPrepWgs, N), dim3(256), 0, static_cast<hipStream_t>(stream), (const float4*)props, pcount, P,
                       (const float4*)gt_boxes, gt_classes, gt_count, Gmax, K, iou_thresh, L, (float4*)cand, ccount, best_iou, best_idx, labels, cls, lists, counts,
                       tickets, tail_a, tail_b);
    ALDI_CHECK_LAUNCH();
    return ALDI_OK;
}

extern "C" size_t aldi_compact_labels_workspace(int L, int N) { return (size_t)N * 2 * (size_t)cdiv(L > 0 ? L : 1, kSegLabels) * sizeof(int); }

extern "C" int aldi_compact_labels(const int* labels, int L, int N, int bg_label, int* lists, int* counts, void* workspace, aldi_stream_t stream) {
    if (!labels || !lists || !counts || !workspace || L <= 0 || N <= 0) return aldi_set_error_msg(ALDI_ERR_ARG, "compact_labels: null pointer / empty shape");
    const int segs = cdiv(L, kSegLabels);
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(compact_count_kernel, dim3(segs, N), dim3(256), 0, st, labels, L, bg_label, segs, (int*)workspace);
    ALDI_CHECK_LAUNCH();
    hipLaunchKernelGGL(compact_write_kernel, dim3(segs, N), dim3(256), 0, st, labels, L, bg_label, segs, (const int*)workspace, lists, counts);
    ALDI_CHECK_LAUNCH();
    return ALDI_OK;
}

extern "C" int aldi_rpn_apply_sample(int* labels, int L, int N, const int* lists, const int* sel, const int* nsel, int S, aldi_stream_t stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    long tot = (long)N * L;
    hipLaunchKernelGGL(fill_int_kernel, dim3((int)((tot + 255) / 256 > 4096 ? 4096 : (tot + 255) / 256)), dim3(256), 0, st, labels, tot, -1);
    ALDI_CHECK_LAUNCH();
    hipLaunchKernelGGL(sample_scatter_kernel, dim3(2, N), dim3(256), 0, st, labels, L, lists, sel, nsel, S);
    ALDI_CHECK_LAUNCH();
    return ALDI_OK;
}

extern "C" int aldi_rpn_loss(const aldi_rpn_geom* gm, float* const* head, float* const* grad, const float* anchors, const int* labels, const int* matched,
                             const float* gt_boxes, const int* gt_count, int Gmax, int N, float inv_norm, float grad_scale_cls, float grad_scale_loc, float* loss2,
                             aldi_stream_t stream) {
    if (!gm || !head || !anchors || !labels || !matched || !loss2) return aldi_set_error_msg(ALDI_ERR_ARG, "rpn_loss: null pointer");
    Geom g = make_geom(gm, head, grad);
    long tot = (long)N * g.sumA;
    hipLaunchKernelGGL(rpn_loss_kernel, dim3((int)((tot + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), g, (const float4*)anchors, labels, matched,
                       (const float4*)gt_boxes, gt_count, Gmax, N, inv_norm, grad_scale_cls, grad_scale_loc, loss2);
    ALDI_CHECK_LAUNCH();
    return ALDI_OK;
}

extern "C" size_t aldi_rpn_proposals_workspace(int N, int num_levels) {
    size_t B = (size_t)N * num_levels, cap = kTopkCap;
    size_t s = 0;
    s += B * cap * 8;            // cand keys
    s += B * 4 + 256;            // cand_count
    s += B * cap * 16;           // boxes
    s += B * cap * 4 * 2;        // scores, valid
    s += B * cap * (cap / 64) * 8;   // nms mask
    s += B * cap * 4 + B * 4 + 256;  // keep, keep_count
    s += (size_t)N * 512 * 1024 * 4; // objectness keys (sumA <= 512K per image)
    s += (size_t)3 * B * 4096 * 4 + 3 * B * 16 + B * 4 + B * 16 + 2048;   // radix-select histograms, chain state, fill counters, group counters
    return s + 1024;
}

extern "C" int aldi_rpn_proposals(const aldi_rpn_geom* gm, float* const* head, const float* anchors, const int* img_hw, int N,
                                  int pre_nms_topk, int post_nms_topk, float nms_thresh, void* workspace,
                                  float* out_boxes, float* out_scores, int* out_count, int* err_flag, aldi_stream_t stream) {
    if (!gm || !head || !anchors || !img_hw || !workspace || !out_boxes || !out_scores || !out_count || !err_flag)
        return aldi_set_error_msg(ALDI_ERR_ARG, "rpn_proposals: null pointer");
    if (pre_nms_topk > kTopkCap || gm->num_levels * kTopkCap > kMergeCap) return aldi_set_error_msg(ALDI_ERR_ARG, "rpn_proposals: pre_nms_topk too large");
    Geom g = make_geom(gm, head, nullptr);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t B = (size_t)N * g.nl, cap = kTopkCap;
    char* w = static_cast<char*>(workspace);
    auto take = [&](size_t bytes) { char* p = w; w += (bytes + 255) / 256 * 256; return p; };
    auto* cand = (unsigned long long*)take(B * cap * 8);
    auto* cand_count = (int*)take(B * 4);
    auto* boxes = (float4*)take(B * cap * 16);
    auto* scores = (float*)take(B * cap * 4);
    auto* valid = (int*)take(B * cap * 4);
    auto* mask = (unsigned long long*)take(B * cap * (cap / 64) * 8);
    auto* keep = (int*)take(B * cap * 4);
    auto* keep_count = (int*)take(B * 4);
    if (g.sumA > 512 * 1024) return aldi_set_error_msg(ALDI_ERR_ARG, "rpn_proposals: more than 512K anchors per image");
    auto* okeys = (unsigned*)take((size_t)N * g.sumA * 4);
    {
        // exact top-k per (level, image): multi-workgroup radix select (see topk_hist_kernel)
        auto* hists = (int*)take((size_t)3 * B * 4096 * 4);
        auto* tstate = (TopkState*)take((size_t)3 * B * sizeof(TopkState));     // [current | staging | final]
        auto* fill = (int*)take(B * 4);
        auto* gsync = (int*)take(B * 4 * 4);                   // group barrier / ticket counters of the fused top-k
        const long zero_words = (long)(((char*)gsync + B * 16 - (char*)hists + 3) / 4);
        hipLaunchKernelGGL(rpn_keys_kernel, dim3(cdiv((long)N * g.sumA, 256)), dim3(256), 0, st, g, N, okeys, hists, zero_words);
        ALDI_CHECK_LAUNCH();
        int max_nel = 0;
        for (int l = 0; l < g.nl; ++l) max_nel = g.H[l] * g.W[l] * g.A > max_nel ? g.H[l] * g.W[l] * g.A : max_nel;
        dim3 grid(cdiv(max_nel, kTopkChunk), g.nl, N);
        // the fused form's barriers are among the grid.x workgroups of ONE (level, image) group, consecutive in dispatch order; they are plain
        // launches, so co-residency of a group is assumed, not checked: keep a group far below what the chip holds of 1024-thread workgroups
        // (256 CUs x 1-2) and take the five-launch path beyond that (the spin is bounded and flags error bit 8 either way)
        if (aldi_tuning().rpn_topk_fused && grid.x <= 64) {
            hipLaunchKernelGGL(topk_fused_kernel, grid, dim3(1024), 0, st, g, okeys, pre_nms_topk, hists, gsync, fill, cand, cand_count, (const float4*)anchors, img_hw,
                               boxes, scores, valid, err_flag);
            ALDI_CHECK_LAUNCH();
        } else {
        for (int pass = 0; pass < 3; ++pass) {
            hipLaunchKernelGGL(topk_hist_kernel, grid, dim3(1024), 0, st, g, okeys, pre_nms_topk, pass, hists, tstate);
            ALDI_CHECK_LAUNCH();
        }
        hipLaunchKernelGGL(topk_collect_kernel, grid, dim3(1024), 0, st, g, okeys, hists, tstate, cand, fill, tstate + 2 * B);
        ALDI_CHECK_LAUNCH();
        hipLaunchKernelGGL(topk_sort_kernel, dim3(g.nl, N), dim3(1024), 0, st, g, okeys, pre_nms_topk, tstate + 2 * B, fill, cand, cand_count,
                           (const float4*)anchors, img_hw, boxes, scores, valid, err_flag);
        ALDI_CHECK_LAUNCH();
        }
    }
    nms_mask_launch(st, (int)B, boxes, valid, (const int*)nullptr, cand_count, (int)cap, nms_thresh, mask, aldi_tuning().nms_mask_tri);
    ALDI_CHECK_LAUNCH();
    // (a level can place at most post_nms_topk boxes in the image's merged list: its scan stops there)
    if (!nms_scan_launch(st, (int)B, mask, valid, cand_count, (int)cap, post_nms_topk < (int)cap ? post_nms_topk : (int)cap, keep, keep_count))
        return aldi_set_error_msg(ALDI_ERR_ARG, "rpn_proposals: NMS capacity too large");
    ALDI_CHECK_LAUNCH();
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(rpn_merge_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kMergeCap * 4);
    hipLaunchKernelGGL(rpn_merge_kernel, dim3(N), dim3(1024), kMergeCap * 4, st, g.nl, boxes, scores, keep, keep_count, post_nms_topk, (float4*)out_boxes, out_scores, out_count);
    ALDI_CHECK_LAUNCH();
    return ALDI_OK;
}
