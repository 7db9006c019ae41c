// Multi-scale deformable attention sampling (Deformable-DETR), forward and backward, fp32 -- BASELINE configs[4] /
// SURVEY.md 8(f) rank 2.  The reference reaches this op through its (absent) `aldi/detr/libs` submodule
// (.gitmodules:4-6; configs/Base-DETR.yaml:1-81: 4 levels, 8 heads x 4 points, d_model 256 => head dim 32, AMP off), whose
// CUDA extension `MSDeformAttnFunction.apply(value, spatial_shapes, level_start_index, sampling_locations, attention_weights,
// im2col_step)` this replaces.  Semantics = the published pure-PyTorch statement (`ms_deform_attn_core_pytorch`):
// per (image, query, head):  out = sum_l sum_p  w[l][p] * bilinear(value_l[head], loc[l][p])   with grid_sample's
// align_corners=False pixel mapping (x = loc_x * W - 0.5) and zero padding.
//
// HBM / gather bound: a query-head pair is D/4 consecutive lanes of 4 channels each (8 pairs per wave at D = 32), so each corner
// read is one 128-B segment of value[n][pixel][head][:] made of 16-B loads; locations and weights are broadcasts inside the
// pair.  Backward scatters the value gradient with fp32 atomics (corners of neighbouring samples collide by design) and
// reduces the location / weight gradients over the pair's lanes with 3 shuffle steps -- one plain store per (query, head,
// level, point).
#include "common.h"

namespace {

struct MsdaDev {
    const float *value, *loc, *attw, *gout;
    const int *shapes, *lstart;
    float *out, *gvalue, *gloc, *gattw;
    int N, S, M, Lq, L, P;
    int gmask;          // gather form: bit l = target level l is gathered (the others take the atomic scatter)
};

// A (query, head) pair is D/4 consecutive lanes, each owning 4 channels (one 16-B load per corner; the D/4 lanes of a pair read
// one 128-B (D = 32) segment).  8 pairs per wave at D = 32: the location / weight gradients need 3 shuffle steps per sum.
template <int LANES>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int o = LANES / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float dot4(const float4& a, const float4& b) { return (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w); }

template <int D, bool BWD>
__global__ __launch_bounds__(256) void msda_kernel(MsdaDev a) {
    constexpr int LANES = D / 4, PPW = 64 / LANES;
    const int lane = threadIdx.x & 63, d = (lane % LANES) * 4;
    const long pair = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * PPW + lane / LANES;
    const long npairs = (long)a.N * a.Lq * a.M;
    const bool live = pair < npairs;                      // dead pairs still take part in the shuffles
    const long pr = live ? pair : 0;
    const int m = (int)(pr % a.M), n = (int)(pr / ((long)a.M * a.Lq));
    const long rowstride = (long)a.M * D;
    const float* locp = a.loc + pr * a.L * a.P * 2;
    const float* wp = a.attw + pr * a.L * a.P;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 g = zero, acc = zero;
    if (BWD && live) g = *reinterpret_cast<const float4*>(a.gout + pr * D + d);
    // backward: the pair's L * P results are kept spread over its lanes (lane j holds samples j and j + LANES) and written as contiguous
    // runs at the end -- one lane storing three scattered words per sample was a third of this kernel
    const int lp = a.L * a.P, sub = lane % LANES;
    const bool spread = BWD && lp <= 2 * LANES;
    float rw0 = 0.f, rw1 = 0.f, rx0 = 0.f, rx1 = 0.f, ry0 = 0.f, ry1 = 0.f;
    for (int l = 0; l < a.L; ++l) {
        const int H = a.shapes[2 * l], W = a.shapes[2 * l + 1];
        const long base = ((long)n * a.S + a.lstart[l]) * rowstride + (long)m * D + d;
        for (int p = 0; p < a.P; ++p) {
            const float x = locp[(l * a.P + p) * 2] * W - 0.5f, y = locp[(l * a.P + p) * 2 + 1] * H - 0.5f, w = wp[l * a.P + p];
            float4 v00 = zero, v01 = zero, v10 = zero, v11 = zero;
            float lx = 0.f, ly = 0.f;
            int x0 = 0, y0 = 0;
            const bool inside = y > -1.f && x > -1.f && y < (float)H && x < (float)W;
            bool c00 = false, c01 = false, c10 = false, c11 = false;
            if (inside) {
                const float fx = floorf(x), fy = floorf(y);
                x0 = (int)fx; y0 = (int)fy; lx = x - fx; ly = y - fy;
                const bool xa = x0 >= 0, xb = x0 + 1 < W, ya = y0 >= 0, yb = y0 + 1 < H;
                c00 = ya && xa; c01 = ya && xb; c10 = yb && xa; c11 = yb && xb;
                const float* v = a.value + base;
                if (c00) v00 = *reinterpret_cast<const float4*>(v + ((long)y0 * W + x0) * rowstride);
                if (c01) v01 = *reinterpret_cast<const float4*>(v + ((long)y0 * W + x0 + 1) * rowstride);
                if (c10) v10 = *reinterpret_cast<const float4*>(v + ((long)(y0 + 1) * W + x0) * rowstride);
                if (c11) v11 = *reinterpret_cast<const float4*>(v + ((long)(y0 + 1) * W + x0 + 1) * rowstride);
            }
            const float hx = 1.f - lx, hy = 1.f - ly;
            const float w00 = hy * hx, w01 = hy * lx, w10 = ly * hx, w11 = ly * lx;
            if (!BWD) {
                acc.x += w * (w00 * v00.x + w01 * v01.x + w10 * v10.x + w11 * v11.x);
                acc.y += w * (w00 * v00.y + w01 * v01.y + w10 * v10.y + w11 * v11.y);
                acc.z += w * (w00 * v00.z + w01 * v01.z + w10 * v10.z + w11 * v11.z);
                acc.w += w * (w00 * v00.w + w01 * v01.w + w10 * v10.w + w11 * v11.w);
            } else {
                // per-channel-group dot products with the upstream gradient, then one reduction over the pair's lanes
                const float d00 = dot4(g, v00), d01 = dot4(g, v01), d10 = dot4(g, v10), d11 = dot4(g, v11);
                const float t_w = group_sum<LANES>(w00 * d00 + w01 * d01 + w10 * d10 + w11 * d11);
                const float t_x = group_sum<LANES>(w * (hy * (d01 - d00) + ly * (d11 - d10))) * (float)W;
                const float t_y = group_sum<LANES>(w * (hx * (d10 - d00) + lx * (d11 - d01))) * (float)H;
                const int idx = l * a.P + p;
                if (spread) {
                    if (idx == sub) { rw0 = t_w; rx0 = t_x; ry0 = t_y; }
                    if (idx == sub + LANES) { rw1 = t_w; rx1 = t_x; ry1 = t_y; }
                } else if (live && d == 0) {
                    a.gattw[pr * lp + idx] = t_w;
                    a.gloc[(pr * lp + idx) * 2] = t_x;
                    a.gloc[(pr * lp + idx) * 2 + 1] = t_y;
                }
            }
        }
    }
    if (!BWD && live) *reinterpret_cast<float4*>(a.out + pr * D + d) = acc;
    if (spread && live) {
        if (sub < lp) {
            a.gattw[pr * lp + sub] = rw0;
            *reinterpret_cast<float2*>(a.gloc + (pr * lp + sub) * 2) = make_float2(rx0, ry0);
        }
        if (sub + LANES < lp) {
            a.gattw[pr * lp + sub + LANES] = rw1;
            *reinterpret_cast<float2*>(a.gloc + (pr * lp + sub + LANES) * 2) = make_float2(rx1, ry1);
        }
    }
}

// value gradient: one lane per channel (a pair = D consecutive lanes), so every atomic instruction of a pair covers one
// contiguous 128-B (D = 32) segment -- the 4-channels-per-lane mapping above would issue four strided atomics instead
template <int D>
__global__ __launch_bounds__(256) void msda_bwd_value_kernel(MsdaDev a) {
    constexpr int PPW = 64 / D;
    const int lane = threadIdx.x & 63, d = lane % D;
    const long pair = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * PPW + lane / D;
    if (pair >= (long)a.N * a.Lq * a.M) return;
    const int m = (int)(pair % a.M), n = (int)(pair / ((long)a.M * a.Lq));
    const long rowstride = (long)a.M * D;
    const float* locp = a.loc + pair * a.L * a.P * 2;
    const float* wp = a.attw + pair * a.L * a.P;
    const float g = a.gout[pair * D + d];
    for (int l = 0; l < a.L; ++l) {
        const int H = a.shapes[2 * l], W = a.shapes[2 * l + 1];
        float* gv = a.gvalue + ((long)n * a.S + a.lstart[l]) * rowstride + (long)m * D + d;
        for (int p = 0; p < a.P; ++p) {
            const float x = locp[(l * a.P + p) * 2] * W - 0.5f, y = locp[(l * a.P + p) * 2 + 1] * H - 0.5f, gw = g * wp[l * a.P + p];
            if (!(y > -1.f && x > -1.f && y < (float)H && x < (float)W)) continue;
            const float fx = floorf(x), fy = floorf(y), lx = x - fx, ly = y - fy, hx = 1.f - lx, hy = 1.f - ly;
            const int x0 = (int)fx, y0 = (int)fy;
            const bool xa = x0 >= 0, xb = x0 + 1 < W, ya = y0 >= 0, yb = y0 + 1 < H;
            if (ya && xa) unsafeAtomicAdd(gv + ((long)y0 * W + x0) * rowstride, gw * hy * hx);
            if (ya && xb) unsafeAtomicAdd(gv + ((long)y0 * W + x0 + 1) * rowstride, gw * hy * lx);
            if (yb && xa) unsafeAtomicAdd(gv + ((long)(y0 + 1) * W + x0) * rowstride, gw * ly * hx);
            if (yb && xb) unsafeAtomicAdd(gv + ((long)(y0 + 1) * W + x0 + 1) * rowstride, gw * ly * lx);
        }
    }
}

// ---- value gradient as a GATHER when the queries are the pyramid's positions (the encoder's self attention) ---------------------
// The scatter above is bound by the float-add rate of the L2 atomic units (tools/probes/atomic_scope_probe.hip: 326 G adds/s at any
// scope; 2 x 22 K queries x 8 heads x 64 corners x 32 channels = 730 M adds per encoder layer: 1.9 ms).  Here a workgroup OWNS a tile
// of value pixels of one level and one head (8 x 8 on the fine levels, 4 x 4 / 2 x 2 on the coarse ones: every level receives the same
// number of samples, so a coarse pixel collects ~1300 of them on a 13 x 21 map and its list is split over 4 / 16 threads).  It walks
// the queries whose samples can reach the tile -- query (x, y) of level lq is expected at c = ((x + 0.5) W_l / W_lq - 0.5, ...) on the
// target level and a sample counts as "near" when it lies within kGR pixels of c --, keeps those that touch the tile in an LDS list
// (position, weight, query; slots reserved with one integer LDS atomic per wave), then every (pixel, 8-channel group) thread runs down
// the list, takes the samples within one pixel of its own position with the bilinear weight (1 - |dx|)(1 - |dy|), reads the output
// gradient's 32 bytes and writes its sum ONCE.  Samples that are not "near" (large learnt offsets, strongly padded images) go through a
// second launch of the atomic scatter restricted to them; the predicate is the same float expression in both kernels, so every sample
// is counted exactly once.  A tile whose list would not fit (kSCap) adds its own pixels with atomics instead.
struct GatherTiles { int tile_begin[9]; int ts_log[8]; int L; };     // per TARGET level: first tile, log2 of the tile edge (8, 4 or 2 pixels)
constexpr int kGR = 5, kSCap = 2048;

__device__ __forceinline__ float msda_centre(int q, int Wt, int Wq) { return ((float)q + 0.5f) * ((float)Wt / (float)Wq) - 0.5f; }
__device__ __forceinline__ bool msda_near(float x, float y, float cx, float cy) { return fabsf(x - cx) <= (float)kGR && fabsf(y - cy) <= (float)kGR; }

// DIRECT = false: list the tile's samples in LDS; true: add them to the tile's pixels with atomics (the list overflowed)
template <bool DIRECT>
__device__ __forceinline__ void gather_walk(const MsdaDev& a, int n, int m, int l, int H, int W, int ty0, int tx0, int TS, float4* smp, int* count, float* gvl) {
    const int tid = threadIdx.x, lane = tid & 63;
    const long rowstride = (long)a.M * 32;
    for (int lq = 0; lq < a.L; ++lq) {
        const int Hq = a.shapes[2 * lq], Wq = a.shapes[2 * lq + 1];
        // queries whose centre on level l lies within (tile - 1 - kGR, tile + TS + kGR): conservative integer bounds (+-1)
        const float sx = (float)Wq / (float)W, sy = (float)Hq / (float)H;
        int xa = (int)floorf(((float)(tx0 - 1 - kGR) + 0.5f) * sx - 0.5f) - 1, xb = (int)ceilf(((float)(tx0 + TS + kGR) + 0.5f) * sx - 0.5f) + 1;
        int ya = (int)floorf(((float)(ty0 - 1 - kGR) + 0.5f) * sy - 0.5f) - 1, yb = (int)ceilf(((float)(ty0 + TS + kGR) + 0.5f) * sy - 0.5f) + 1;
        xa = max(xa, 0); ya = max(ya, 0); xb = min(xb, Wq - 1); yb = min(yb, Hq - 1);
        if (xa > xb || ya > yb) continue;
        const int span = xb - xa + 1, cnt = span * (yb - ya + 1) * a.P;
        // a trip's operands (location + weight of candidate base + tid) are requested one trip ahead: the walk is a chain of L2 round trips
        const long lq0 = (long)n * a.Lq + a.lstart[lq];
        auto fetch = [&](int idx, long& pair_, int& xq_, int& yq_, float2& lp_, float& w_) {
            pair_ = -1;
            if (idx < cnt) {
                const int qq = idx / a.P, p_ = idx - qq * a.P;
                yq_ = ya + qq / span; xq_ = xa + qq % span;
                pair_ = (lq0 + (long)yq_ * Wq + xq_) * a.M + m;
                const long e = (pair_ * a.L + l) * a.P + p_;
                lp_ = *reinterpret_cast<const float2*>(a.loc + e * 2);
                w_ = a.attw[e];
            }
        };
        long pair_n; int xq_n = 0, yq_n = 0; float2 lp_n = make_float2(0.f, 0.f); float w_n = 0.f;
        fetch(tid, pair_n, xq_n, yq_n, lp_n, w_n);
        for (int base = 0; base < cnt; base += 256) {                    // (whole-workgroup trips: the slot reservation is a wave operation)
            const long pair = pair_n; const int xq = xq_n, yq = yq_n; const float2 lp = lp_n; const float wgt_c = w_n;
            if (base + 256 < cnt) fetch(base + 256 + tid, pair_n, xq_n, yq_n, lp_n, w_n);
            bool hit = false;
            float x = 0.f, y = 0.f;
            if (pair >= 0) {
                x = lp.x * W - 0.5f; y = lp.y * H - 0.5f;
                if (y > -1.f && x > -1.f && y < (float)H && x < (float)W && msda_near(x, y, msda_centre(xq, W, Wq), msda_centre(yq, H, Hq))) {
                    const int x0 = (int)floorf(x), y0 = (int)floorf(y);
                    hit = x0 + 1 >= tx0 && x0 < tx0 + TS && y0 + 1 >= ty0 && y0 < ty0 + TS;
                }
            }
            if (!DIRECT) {
                const unsigned long long mask = __ballot(hit);
                if (mask == 0) continue;
                int slot0 = 0;
                const int leader = __ffsll((long long)mask) - 1;
                if (lane == leader) slot0 = atomicAdd(count, __popcll(mask));
                slot0 = __shfl(slot0, leader, 64);
                if (hit) {
                    const int slot = slot0 + __popcll(mask & ((1ull << lane) - 1ull));
                    if (slot < kSCap) smp[slot] = make_float4(x, y, wgt_c, __int_as_float((int)pair));
                }
            } else if (hit) {
                const float wgt = wgt_c;
                const float fx = floorf(x), fy = floorf(y), lx = x - fx, ly = y - fy, hx = 1.f - lx, hy = 1.f - ly;
                const int x0 = (int)fx, y0 = (int)fy;
                const float* g = a.gout + pair * 32;
                for (int c = 0; c < 4; ++c) {
                    const int xx = x0 + (c & 1), yy = y0 + (c >> 1);
                    if (xx < tx0 || yy < ty0 || xx >= min(W, tx0 + TS) || yy >= min(H, ty0 + TS)) continue;
                    const float v = wgt * ((c >> 1) ? ly : hy) * ((c & 1) ? lx : hx);
                    float* dst = gvl + ((long)yy * W + xx) * rowstride;
                    for (int d = 0; d < 32; ++d) unsafeAtomicAdd(dst + d, v * g[d]);
                }
            }
        }
    }
}

__global__ __launch_bounds__(256) void msda_bwd_value_gather_kernel(MsdaDev a, GatherTiles T) {
    __shared__ float4 smp[kSCap];
    __shared__ int count;
    const int tid = threadIdx.x, m = blockIdx.y, n = blockIdx.z;
    int l = 0;
    while (l + 1 < T.L && (int)blockIdx.x >= T.tile_begin[l + 1]) ++l;
    if (!((a.gmask >> l) & 1)) return;
    const int ts_log = T.ts_log[l], TS = 1 << ts_log, NP = TS * TS, PARTS = 64 / NP;
    const int H = a.shapes[2 * l], W = a.shapes[2 * l + 1];
    const int tiles_x = (W + TS - 1) >> ts_log, tl = blockIdx.x - T.tile_begin[l];
    const int ty0 = (tl / tiles_x) << ts_log, tx0 = (tl % tiles_x) << ts_log;
    float* gvl = a.gvalue + ((long)n * a.S + a.lstart[l]) * (long)a.M * 32 + (long)m * 32;
    if (tid == 0) count = 0;
    __syncthreads();
    gather_walk<false>(a, n, m, l, H, W, ty0, tx0, TS, smp, &count, gvl);
    __syncthreads();
    const int ns = count;
    if (ns > kSCap) {                                  // (gvalue was zeroed by the caller; nobody else writes this tile's pixels in this launch)
        gather_walk<true>(a, n, m, l, H, W, ty0, tx0, TS, smp, &count, gvl);
        return;
    }
    // thread = (part of the list, pixel, 8-channel group)
    const int cq = tid & 3, px = (tid >> 2) & (NP - 1), part = (tid >> 2) >> (2 * ts_log);
    const int X = tx0 + (px & (TS - 1)), Y = ty0 + (px >> ts_log);
    const float Xf = (float)X, Yf = (float)Y;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int i = part; i < ns; i += PARTS) {
        const float4 sp = smp[i];
        const float wx = 1.f - fabsf(sp.x - Xf), wy = 1.f - fabsf(sp.y - Yf);
        if (wx > 0.f && wy > 0.f) {
            const float v = sp.z * wy * wx;
            const float4* g = reinterpret_cast<const float4*>(a.gout + (long)__float_as_int(sp.w) * 32 + cq * 8);
            const float4 g0 = g[0], g1 = g[1];
            acc[0] += v * g0.x; acc[1] += v * g0.y; acc[2] += v * g0.z; acc[3] += v * g0.w;
            acc[4] += v * g1.x; acc[5] += v * g1.y; acc[6] += v * g1.z; acc[7] += v * g1.w;
        }
    }
    if (PARTS > 1) {                                   // the parts of a (pixel, channel group) meet in LDS (the list is consumed)
        __syncthreads();
        float* red = reinterpret_cast<float*>(smp);
#pragma unroll
        for (int k = 0; k < 8; ++k) red[tid * 8 + k] = acc[k];
        __syncthreads();
        if (part) return;
        for (int q = 1; q < PARTS; ++q)
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] += red[(((q << (2 * ts_log)) + px) * 4 + cq) * 8 + k];
    }
    if (X >= W || Y >= H) return;
    float4* dst = reinterpret_cast<float4*>(gvl + ((long)Y * W + X) * (long)a.M * 32 + cq * 8);
    dst[0] = make_float4(acc[0], acc[1], acc[2], acc[3]);
    dst[1] = make_float4(acc[4], acc[5], acc[6], acc[7]);
}

// the samples the gather does not take: not "near" their query's centre on the target level -> the atomic scatter (one lane per channel)
__global__ __launch_bounds__(256) void msda_bwd_value_far_kernel(MsdaDev a) {
    constexpr int D = 32;
    const int lane = threadIdx.x & 63, d = lane % D;
    const long pair = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + lane / D;
    if (pair >= (long)a.N * a.Lq * a.M) return;
    const int m = (int)(pair % a.M), n = (int)(pair / ((long)a.M * a.Lq));
    const int q = (int)((pair / a.M) % a.Lq);
    int lq = 0;
    while (lq + 1 < a.L && q >= a.lstart[lq + 1]) ++lq;
    const int Hq = a.shapes[2 * lq], Wq = a.shapes[2 * lq + 1];
    const int yq = (q - a.lstart[lq]) / Wq, xq = (q - a.lstart[lq]) - yq * Wq;
    const long rowstride = (long)a.M * D;
    const float* locp = a.loc + pair * a.L * a.P * 2;
    const float* wp = a.attw + pair * a.L * a.P;
    float g = 0.f;
    bool have_g = false;
    for (int l = 0; l < a.L; ++l) {
        const int H = a.shapes[2 * l], W = a.shapes[2 * l + 1];
        const float cx = msda_centre(xq, W, Wq), cy = msda_centre(yq, H, Hq);
        float* gv = a.gvalue + ((long)n * a.S + a.lstart[l]) * rowstride + (long)m * D + d;
        for (int p = 0; p < a.P; ++p) {
            const float x = locp[(l * a.P + p) * 2] * W - 0.5f, y = locp[(l * a.P + p) * 2 + 1] * H - 0.5f;
            if (!(y > -1.f && x > -1.f && y < (float)H && x < (float)W)) continue;
            if (((a.gmask >> l) & 1) && msda_near(x, y, cx, cy)) continue;
            if (!have_g) { g = a.gout[pair * D + d]; have_g = true; }
            const float gw = g * wp[l * a.P + p];
            const float fx = floorf(x), fy = floorf(y), lx = x - fx, ly = y - fy, hx = 1.f - lx, hy = 1.f - ly;
            const int x0 = (int)fx, y0 = (int)fy;
            const bool xa = x0 >= 0, xb = x0 + 1 < W, ya = y0 >= 0, yb = y0 + 1 < H;
            if (ya && xa) unsafeAtomicAdd(gv + ((long)y0 * W + x0) * rowstride, gw * hy * hx);
            if (ya && xb) unsafeAtomicAdd(gv + ((long)y0 * W + x0 + 1) * rowstride, gw * hy * lx);
            if (yb && xa) unsafeAtomicAdd(gv + ((long)(y0 + 1) * W + x0) * rowstride, gw * ly * hx);
            if (yb && xb) unsafeAtomicAdd(gv + ((long)(y0 + 1) * W + x0 + 1) * rowstride, gw * ly * lx);
        }
    }
}

// ---- the same gather with the lists built in HBM by ONE pass over the samples ("binned" form; needs a workspace) -----------------
// The walk above finds a tile's samples by scanning every query that could reach it (5.6x the samples on an 8 x 8 tile, 42x on a 2 x 2
// one) and needs the "near" predicate plus a second launch for the rest.  Here every sample is visited once: a thread per (query,
// head, level, point) computes the 1..4 tiles its 2 x 2 footprint touches and appends (x, y, weight, query) to each tile's list in the
// workspace (one returning integer atomic per tile touched); the second kernel is a workgroup per tile that streams its list through
// LDS and sums as before.  No predicate, no second launch, any offsets; small tiles cost nothing to find, so every level is gathered
// with tiles sized for ~128+ entries.  A list that overflows its capacity sends the extra samples through the atomic scatter, tile by
// tile (the accumulate kernel ADDS to what those atomics left).
constexpr int kCntPad = 32;        // one list counter per 128-B line: counters of neighbouring tiles on one line serialise in the L2 atomic unit
struct BinTiles { int tile_begin[9]; int ts_log[8]; int cap[8]; long ent_begin[9]; int L; };      // per level: first tile, tile edge, list capacity, first entry (per (n, m) slice)

// grid (64-query chunk, head, image); a wave takes four of the L * P (level, point) pairs in turn, lane = query of the chunk: the 64
// neighbouring queries' samples of one (level, point) fall into a handful of tiles, so the lanes that hit the same tile reserve their
// slots with ONE returning atomic (743 us with one atomic per sample and tile, 558 with one per group but a round trip per group, 322 with all groups of a (level, point) at once)
__global__ __launch_bounds__(256) void msda_bin_kernel(MsdaDev a, BinTiles T, int* __restrict__ cnt, float4* __restrict__ ent) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = blockIdx.x * 64 + lane, m = blockIdx.y, n = blockIdx.z;
    const bool live = q < a.Lq;
    const long pair = ((long)n * a.Lq + (live ? q : 0)) * a.M + m;
    const int lp = a.L * a.P;
    const long slice = (long)n * a.M + m;
    const long ent_slice = T.ent_begin[T.L];
    int* cnt_s = cnt + slice * T.tile_begin[T.L] * kCntPad;
    for (int r = wave; r < lp; r += 4) {
        const int l = r / a.P;
        const int H = a.shapes[2 * l], W = a.shapes[2 * l + 1];
        const int ts = T.ts_log[l], tiles_x = (W + (1 << ts) - 1) >> ts, cap = T.cap[l];
        float x = 0.f, y = 0.f, wgt = 0.f;
        bool ok = false;
        if (live) {
            const float2 lc = *reinterpret_cast<const float2*>(a.loc + (pair * lp + r) * 2);
            x = lc.x * W - 0.5f; y = lc.y * H - 0.5f;
            ok = y > -1.f && x > -1.f && y < (float)H && x < (float)W;
            if (ok) wgt = a.attw[pair * lp + r];
        }
        const int x0 = (int)floorf(x), y0 = (int)floorf(y);
        const int txa = x0 >= 0 ? x0 >> ts : -1, txb = x0 + 1 < W ? (x0 + 1) >> ts : -1;
        const int tya = y0 >= 0 ? y0 >> ts : -1, tyb = y0 + 1 < H ? (y0 + 1) >> ts : -1;
        // per corner: the tile, and among the lanes that hit the same tile a leader, this lane's rank and the group's size (register work
        // only); then ALL leaders of all four corners reserve their groups' slots at once -- one atomic round trip per (level, point)
        bool hit4[4]; int t4[4], leader4[4], rank4[4], base4[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int tx = (c & 1) ? txb : txa, ty = (c >> 1) ? tyb : tya;
            bool hit = ok && tx >= 0 && ty >= 0;
            if ((c & 1) && txb == txa) hit = false;      // the same tile as the left / upper corner
            if ((c >> 1) && tyb == tya) hit = false;
            const int t = hit ? T.tile_begin[l] + ty * tiles_x + tx : -1;
            int leader = -1, rank = 0, count = 0;
            unsigned long long todo = __ballot(hit);
            while (todo) {
                const int ld = __ffsll((long long)todo) - 1;
                const int tl = __shfl(t, ld, 64);
                const unsigned long long same = __ballot(hit && t == tl);
                if (hit && t == tl) { leader = ld; rank = __popcll(same & ((1ull << lane) - 1ull)); count = __popcll(same); }
                todo &= ~same;
            }
            hit4[c] = hit; t4[c] = t; leader4[c] = leader; rank4[c] = rank;
            base4[c] = 0;
            if (hit && lane == leader) base4[c] = atomicAdd(cnt_s + (long)t * kCntPad, count);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int tx = (c & 1) ? txb : txa, ty = (c >> 1) ? tyb : tya;
            const bool hit = hit4[c];
            const int t = t4[c];
            const int slot = __shfl(base4[c], leader4[c] < 0 ? lane : leader4[c], 64) + rank4[c];
            if (!hit) continue;
            if (slot < cap) {
                ent[slice * ent_slice + T.ent_begin[l] + (long)(t - T.tile_begin[l]) * cap + slot] = make_float4(x, y, wgt, __int_as_float((int)pair));
            } else {                                     // list full: this sample's corners inside the tile, with atomics
                const float lx = x - (float)x0, ly = y - (float)y0, hx = 1.f - lx, hy = 1.f - ly;
                const float* g = a.gout + pair * 32;
                float* gvl = a.gvalue + ((long)n * a.S + a.lstart[l]) * (long)a.M * 32 + (long)m * 32;
                for (int k = 0; k < 4; ++k) {
                    const int xx = x0 + (k & 1), yy = y0 + (k >> 1);
                    if (xx < 0 || yy < 0 || xx >= W || yy >= H || (xx >> ts) != tx || (yy >> ts) != ty) continue;
                    const float v = wgt * ((k >> 1) ? ly : hy) * ((k & 1) ? lx : hx);
                    float* dst = gvl + ((long)yy * W + xx) * (long)a.M * 32;
                    for (int d = 0; d < 32; ++d) unsafeAtomicAdd(dst + d, v * g[d]);
                }
            }
        }
    }
}

__global__ __launch_bounds__(256) void msda_bin_accumulate_kernel(MsdaDev a, BinTiles T, const int* __restrict__ cnt, const float4* __restrict__ ent) {
    constexpr int CH = 1024;
    __shared__ float4 smp[CH];
    const int tid = threadIdx.x, m = blockIdx.y, n = blockIdx.z;
    int l = 0;
    while (l + 1 < T.L && (int)blockIdx.x >= T.tile_begin[l + 1]) ++l;
    const int ts_log = T.ts_log[l], TS = 1 << ts_log, NP = TS * TS, PARTS = 64 / NP;
    const int H = a.shapes[2 * l], W = a.shapes[2 * l + 1];
    const int tiles_x = (W + TS - 1) >> ts_log, tl = blockIdx.x - T.tile_begin[l];
    const int ty0 = (tl / tiles_x) << ts_log, tx0 = (tl % tiles_x) << ts_log;
    const long slice = (long)n * a.M + m;
    const int ns = min(cnt[(slice * T.tile_begin[T.L] + blockIdx.x) * kCntPad], T.cap[l]);
    const float4* list = ent + slice * T.ent_begin[T.L] + T.ent_begin[l] + (long)tl * T.cap[l];
    const int cq = tid & 3, px = (tid >> 2) & (NP - 1), part = (tid >> 2) >> (2 * ts_log);
    const int X = tx0 + (px & (TS - 1)), Y = ty0 + (px >> ts_log);
    const float Xf = (float)X, Yf = (float)Y;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int base = 0; base < ns; base += CH) {
        const int nc = min(CH, ns - base);
        if (base) __syncthreads();
        for (int i = tid; i < nc; i += 256) smp[i] = list[base + i];
        __syncthreads();
        // a thread's entries in windows of 32: first the window's hits as a bit mask (LDS reads and compares only), then the hits two at
        // a time -- four independent 16-byte loads of the output gradient in flight instead of one dependent pair per hit
        for (int w0 = part; w0 < nc; w0 += 32 * PARTS) {
            unsigned hits = 0;
#pragma unroll 8
            for (int k = 0; k < 32; ++k) {
                const int i = w0 + k * PARTS;
                if (i < nc) {
                    const float4 sp = smp[i];
                    if (1.f - fabsf(sp.x - Xf) > 0.f && 1.f - fabsf(sp.y - Yf) > 0.f) hits |= 1u << k;
                }
            }
            while (hits) {
                const int k0 = __ffs(hits) - 1;
                hits &= hits - 1;
                const int k1 = hits ? __ffs(hits) - 1 : k0;
                const bool two = hits != 0;
                hits &= hits - 1 + (two ? 0u : 1u);            // (clear the second bit only when there is one)
                const float4 s0 = smp[w0 + k0 * PARTS], s1 = smp[w0 + k1 * PARTS];
                const float4* g0p = reinterpret_cast<const float4*>(a.gout + (long)__float_as_int(s0.w) * 32 + cq * 8);
                const float4* g1p = reinterpret_cast<const float4*>(a.gout + (long)__float_as_int(s1.w) * 32 + cq * 8);
                const float4 a0 = g0p[0], a1 = g0p[1], b0 = g1p[0], b1 = g1p[1];
                const float v0 = s0.z * (1.f - fabsf(s0.y - Yf)) * (1.f - fabsf(s0.x - Xf));
                const float v1 = two ? s1.z * (1.f - fabsf(s1.y - Yf)) * (1.f - fabsf(s1.x - Xf)) : 0.f;
                acc[0] += v0 * a0.x; acc[1] += v0 * a0.y; acc[2] += v0 * a0.z; acc[3] += v0 * a0.w;
                acc[4] += v0 * a1.x; acc[5] += v0 * a1.y; acc[6] += v0 * a1.z; acc[7] += v0 * a1.w;
                acc[0] += v1 * b0.x; acc[1] += v1 * b0.y; acc[2] += v1 * b0.z; acc[3] += v1 * b0.w;
                acc[4] += v1 * b1.x; acc[5] += v1 * b1.y; acc[6] += v1 * b1.z; acc[7] += v1 * b1.w;
            }
        }
    }
    if (PARTS > 1) {
        __syncthreads();
        float* red = reinterpret_cast<float*>(smp);
#pragma unroll
        for (int k = 0; k < 8; ++k) red[tid * 8 + k] = acc[k];
        __syncthreads();
        if (part) return;
        for (int q = 1; q < PARTS; ++q)
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] += red[(((q << (2 * ts_log)) + px) * 4 + cq) * 8 + k];
    }
    if (X >= W || Y >= H) return;
    float4* dst = reinterpret_cast<float4*>(a.gvalue + (((long)n * a.S + a.lstart[l] + (long)Y * W + X) * a.M + m) * 32 + cq * 8);
    float4 d0 = dst[0], d1 = dst[1];                    // (zero, or what an overflowing list's atomics left)
    d0.x += acc[0]; d0.y += acc[1]; d0.z += acc[2]; d0.w += acc[3];
    d1.x += acc[4]; d1.y += acc[5]; d1.z += acc[6]; d1.w += acc[7];
    dst[0] = d0; dst[1] = d1;
}

// tile sizes / capacities of the binned form: the smallest tile with >= `want` expected entries, capacity 2x the expectation + 256
int bin_plan(const int* shapes_host, int S, int L, int P, BinTiles& T) {
    T = BinTiles{};
    T.L = L;
    const double want = aldi_tuning().msda_bin_list;
    for (int l = 0; l < L; ++l) {
        const int H = shapes_host[2 * l], W = shapes_host[2 * l + 1];
        const double spp = (double)S * P / ((double)H * W);
        int ts_log = 3;
        for (int t = 0; t <= 3; ++t)
            if (spp * ((1 << t) + 1) * ((1 << t) + 1) >= want) { ts_log = t; break; }
        T.ts_log[l] = ts_log;
        const double expect = spp * ((1 << ts_log) + 1) * ((1 << ts_log) + 1);
        T.cap[l] = ((int)(2.0 * expect) + 256 + 255) / 256 * 256;
        const int tiles = cdiv(H, 1 << ts_log) * cdiv(W, 1 << ts_log);
        T.tile_begin[l + 1] = T.tile_begin[l] + tiles;
        T.ent_begin[l + 1] = T.ent_begin[l] + (long)tiles * T.cap[l];
    }
    return ALDI_OK;
}

template <bool BWD>
int launch(const MsdaDev& a, int D, hipStream_t st) {
    const long npairs = (long)a.N * a.Lq * a.M;
    if (D == 32) hipLaunchKernelGGL((msda_kernel<32, BWD>), dim3(cdiv(npairs, 4 * 8)), dim3(256), 0, st, a);
    else if (D == 64) hipLaunchKernelGGL((msda_kernel<64, BWD>), dim3(cdiv(npairs, 4 * 4)), dim3(256), 0, st, a);
    else return aldi_set_error_msg(ALDI_ERR_ARG, "ms_deform_attn: head dim must be 32 or 64");
    ALDI_CHECK_LAUNCH();
    if (BWD && a.out) return ALDI_OK;                  // (the gather form finishes the value gradient itself; `out` is unused by the backward)
    if (BWD) {
        if (D == 32) hipLaunchKernelGGL(msda_bwd_value_kernel<32>, dim3(cdiv(npairs, 8)), dim3(256), 0, st, a);
        else hipLaunchKernelGGL(msda_bwd_value_kernel<64>, dim3(cdiv(npairs, 4)), dim3(256), 0, st, a);
        ALDI_CHECK_LAUNCH();
    }
    return ALDI_OK;
}

}  // namespace

extern "C" int aldi_ms_deform_attn_forward(const float* value, const int* spatial_shapes, const int* level_start_index, const float* sampling_loc,
                                           const float* attn_weight, float* out, int N, int S, int M, int D, int Lq, int L, int P,
                                           aldi_stream_t stream) {
    if (!value || !spatial_shapes || !level_start_index || !sampling_loc || !attn_weight || !out || N <= 0 || S <= 0 || M <= 0 || Lq <= 0 || L <= 0 || P <= 0)
        return aldi_set_error_msg(ALDI_ERR_ARG, "ms_deform_attn: bad args");
    MsdaDev a{};
    a.value = value; a.loc = sampling_loc; a.attw = attn_weight; a.shapes = spatial_shapes; a.lstart = level_start_index; a.out = out;
    a.N = N; a.S = S; a.M = M; a.Lq = Lq; a.L = L; a.P = P;
    return launch<false>(a, D, (hipStream_t)stream);
}

extern "C" int aldi_ms_deform_attn_backward(const float* value, const int* spatial_shapes, const int* level_start_index, const float* sampling_loc,
                                            const float* attn_weight, const float* grad_out, float* grad_value, float* grad_sampling_loc,
                                            float* grad_attn_weight, int N, int S, int M, int D, int Lq, int L, int P, aldi_stream_t stream) {
    if (!value || !spatial_shapes || !level_start_index || !sampling_loc || !attn_weight || !grad_out || !grad_value || !grad_sampling_loc ||
        !grad_attn_weight || N <= 0 || S <= 0 || M <= 0 || Lq <= 0 || L <= 0 || P <= 0)
        return aldi_set_error_msg(ALDI_ERR_ARG, "ms_deform_attn: bad args");
    hipError_t e = hipMemsetAsync(grad_value, 0, (size_t)N * S * M * D * sizeof(float), (hipStream_t)stream);
    if (e != hipSuccess) return aldi_set_error(e, __FILE__, __LINE__);
    MsdaDev a{};
    a.value = value; a.loc = sampling_loc; a.attw = attn_weight; a.gout = grad_out; a.shapes = spatial_shapes; a.lstart = level_start_index;
    a.gvalue = grad_value; a.gloc = grad_sampling_loc; a.gattw = grad_attn_weight;
    a.N = N; a.S = S; a.M = M; a.Lq = Lq; a.L = L; a.P = P;
    if (int rc = launch<true>(a, D, (hipStream_t)stream)) return rc;
    aldi_note_dispatch("msda_bwd_value_scatter");
    return ALDI_OK;
}

extern "C" size_t aldi_ms_deform_attn_backward_self_workspace(const int* spatial_shapes_host, int N, int S, int M, int L, int P) {
    if (!spatial_shapes_host || N <= 0 || S <= 0 || M <= 0 || L <= 0 || L > 8 || P <= 0) return 0;
    BinTiles T;
    bin_plan(spatial_shapes_host, S, L, P, T);
    const size_t slices = (size_t)N * M;
    return slices * T.tile_begin[L] * kCntPad * sizeof(int) + slices * (size_t)T.ent_begin[L] * sizeof(float4);
}

extern "C" int aldi_ms_deform_attn_backward_self(const float* value, const int* spatial_shapes, const int* level_start_index, const int* spatial_shapes_host,
                                                 const float* sampling_loc, const float* attn_weight, const float* grad_out, float* grad_value,
                                                 float* grad_sampling_loc, float* grad_attn_weight, void* workspace, size_t workspace_bytes,
                                                 int N, int S, int M, int D, int L, int P, aldi_stream_t stream) {
    if (!value || !spatial_shapes || !level_start_index || !spatial_shapes_host || !sampling_loc || !attn_weight || !grad_out || !grad_value ||
        !grad_sampling_loc || !grad_attn_weight || N <= 0 || S <= 0 || M <= 0 || L <= 0 || L > 8 || P <= 0)
        return aldi_set_error_msg(ALDI_ERR_ARG, "ms_deform_attn_backward_self: bad args (1 <= L <= 8)");
    long sum = 0;
    GatherTiles T{};
    T.L = L;
    for (int l = 0; l < L; ++l) {
        const int H = spatial_shapes_host[2 * l], W = spatial_shapes_host[2 * l + 1];
        if (H <= 0 || W <= 0) return aldi_set_error_msg(ALDI_ERR_ARG, "ms_deform_attn_backward_self: bad level shape");
        sum += (long)H * W;
    }
    if (sum != S) return aldi_set_error_msg(ALDI_ERR_ARG, "ms_deform_attn_backward_self: the level shapes do not add up to S");
    for (int l = 0; l < L; ++l) {
        const int H = spatial_shapes_host[2 * l], W = spatial_shapes_host[2 * l + 1];
        // every level receives S * P samples per head; a sample touches a T x T tile when its 2 x 2 footprint does: ~spp (T + 1)^2 list
        // entries per tile for spp samples per pixel -- at most ~1500 (the list holds 2048)
        const double spp = (double)S * P / ((double)H * W);
        const double lim = aldi_tuning().msda_gather_list;
        const int ts_log = spp * 81 <= lim ? 3 : (spp * 25 <= lim ? 2 : (spp * 9 <= lim ? 1 : 0));
        T.ts_log[l] = ts_log;
        T.tile_begin[l + 1] = T.tile_begin[l] + cdiv(H, 1 << ts_log) * cdiv(W, 1 << ts_log);
    }
    if (D != 32 || !aldi_tuning().msda_gather || M > 65535 || N > 65535 || (long)N * S * M > 0x7fffffffL)
        return aldi_ms_deform_attn_backward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_out, grad_value, grad_sampling_loc,
                                            grad_attn_weight, N, S, M, D, S, L, P, stream);
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(grad_value, 0, (size_t)N * S * M * D * sizeof(float), st);
    if (e != hipSuccess) return aldi_set_error(e, __FILE__, __LINE__);
    MsdaDev a{};
    a.value = value; a.loc = sampling_loc; a.attw = attn_weight; a.gout = grad_out; a.shapes = spatial_shapes; a.lstart = level_start_index;
    a.gvalue = grad_value; a.gloc = grad_sampling_loc; a.gattw = grad_attn_weight;
    a.N = N; a.S = S; a.M = M; a.Lq = S; a.L = L; a.P = P;
    a.out = grad_value;                                 // marks "value gradient handled here" for launch<true>
    a.gmask = aldi_tuning().msda_gather;
    if (int rc = launch<true>(a, D, st)) return rc;
    if (workspace && aldi_tuning().msda_bin) {           // the binned form: lists built in the workspace by one pass over the samples
        BinTiles B;
        bin_plan(spatial_shapes_host, S, L, P, B);
        if (workspace_bytes < aldi_ms_deform_attn_backward_self_workspace(spatial_shapes_host, N, S, M, L, P))
            return aldi_set_error_msg(ALDI_ERR_ARG, "ms_deform_attn_backward_self: workspace too small (aldi_ms_deform_attn_backward_self_workspace)");
        const size_t slices = (size_t)N * M, cnt_bytes = slices * B.tile_begin[L] * kCntPad * sizeof(int);
        int* cnt = static_cast<int*>(workspace);
        float4* ent = reinterpret_cast<float4*>(static_cast<char*>(workspace) + cnt_bytes);
        e = hipMemsetAsync(cnt, 0, cnt_bytes, st);
        if (e != hipSuccess) return aldi_set_error(e, __FILE__, __LINE__);
        hipLaunchKernelGGL(msda_bin_kernel, dim3(cdiv(S, 64), M, N), dim3(256), 0, st, a, B, cnt, ent);
        ALDI_CHECK_LAUNCH();
        hipLaunchKernelGGL(msda_bin_accumulate_kernel, dim3(B.tile_begin[L], M, N), dim3(256), 0, st, a, B, cnt, ent);
        ALDI_CHECK_LAUNCH();
        aldi_note_dispatch("msda_bwd_value_binned");
        return ALDI_OK;
    }
    hipLaunchKernelGGL(msda_bwd_value_gather_kernel, dim3(T.tile_begin[L], M, N), dim3(256), 0, st, a, T);
    ALDI_CHECK_LAUNCH();
    hipLaunchKernelGGL(msda_bwd_value_far_kernel, dim3(cdiv((long)N * S * M, 8)), dim3(256), 0, st, a);
    ALDI_CHECK_LAUNCH();
    aldi_note_dispatch("msda_bwd_value_gather");
    return ALDI_OK;
}
