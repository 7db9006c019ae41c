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
                if (live && d == 0) {
                    a.gattw[pr * a.L * a.P + l * a.P + p] = t_w;
                    a.gloc[(pr * a.L * a.P + l * a.P + p) * 2] = t_x;
                    a.gloc[(pr * a.L * a.P + l * a.P + p) * 2 + 1] = t_y;
                }
            }
        }
    }
    if (!BWD && live) *reinterpret_cast<float4*>(a.out + pr * D + d) = acc;
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

template <bool BWD>
int launch(const MsdaDev& a, int D, hipStream_t st) {
    const long npairs = (long)a.N * a.Lq * a.M;
    if (D == 32) hipLaunchKernelGGL((msda_kernel<32, BWD>), dim3(cdiv(npairs, 4 * 8)), dim3(256), 0, st, a);
    else if (D == 64) hipLaunchKernelGGL((msda_kernel<64, BWD>), dim3(cdiv(npairs, 4 * 4)), dim3(256), 0, st, a);
    else return aldi_set_error_msg(ALDI_ERR_ARG, "ms_deform_attn: head dim must be 32 or 64");
    ALDI_CHECK_LAUNCH();
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
    return launch<true>(a, D, (hipStream_t)stream);
}
