// Small bandwidth-bound kernels around the dense path: weight re-layouts, casts, stem,
// pooling, FPN glue, optimizer and EMA streams.
#include "common.h"
#include <stdlib.h>

namespace {

template <typename T>
__global__ void dgrad_weights_kernel(const float* __restrict__ w, const float* __restrict__ scale, T* __restrict__ wt,
                                     int Cout, int KH, int KW, int Cin) {
    // one thread per output element; output index (ci, kh, kw, co) with co fastest
    long total = (long)Cout * KH * KW * Cin;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int co = (int)(i % Cout);
        long r = i / Cout;
        int kw = (int)(r % KW); r /= KW;
        int kh = (int)(r % KH);
        int ci = (int)(r / KH);
        float v = w[(((long)co * KH + (KH - 1 - kh)) * KW + (KW - 1 - kw)) * Cin + ci];
        if (scale) v *= scale[co];
        Elem<T>::st(wt + i, v);
    }
}

// all layers in one launch: block = one 32 (co) x 32 (ci) tile of one tap of one layer, transposed through LDS so that
// both the fp32 reads (along ci) and the `dtype` writes (along co) are contiguous runs.
// tiles of a layer: tap-major, then co-tile, then ci-tile.
template <typename T>
__global__ __launch_bounds__(256) void dgrad_weights_batch_kernel(const aldi_dgw_item* __restrict__ items, int n_items) {
    __shared__ float tl[32][33];
    const int tile = blockIdx.x;
    int it = 0, hi = n_items - 1;                     // last item whose tile_begin <= tile (hundreds of layers: bisect, not scan)
    while (it < hi) {
        const int mid = (it + hi + 1) >> 1;
        if (items[mid].tile_begin <= tile) it = mid;
        else hi = mid - 1;
    }
    const aldi_dgw_item d = items[it];
    const int Cout = d.Cout, KH = d.KH, KW = d.KW, Cin = d.Cin, T_ = KH * KW;
    const int nci = (Cin + 31) >> 5, nco = (Cout + 31) >> 5;
    int t = tile - d.tile_begin;
    const int cit = t % nci; t /= nci;
    const int cot = t % nco;
    const int tap = t / nco;                         // output tap (kh, kw); the source tap is the rotated one
    const int kh = tap / KW, kw = tap - kh * KW;
    const int stap = (KH - 1 - kh) * KW + (KW - 1 - kw);
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int co = cot * 32 + ty + k * 8, ci = cit * 32 + tx;
        float v = 0.f;
        if (co < Cout && ci < Cin) {
            v = d.w_master[((long)co * T_ + stap) * Cin + ci];
            if (d.scale) v *= d.scale[co];
        }
        tl[ty + k * 8][tx] = v;
    }
    __syncthreads();
    T* __restrict__ wt = static_cast<T*>(d.wt);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int ci = cit * 32 + ty + k * 8, co = cot * 32 + tx;
        if (co < Cout && ci < Cin) Elem<T>::st(wt + ((long)ci * T_ + tap) * Cout + co, tl[tx][ty + k * 8]);
    }
}

// ------------------------------------------------------------------------------------------
// Stem: (uint8 BGR - mean)/std, zero pad, conv 7x7 s2 p3 (fp32 math), FrozenBN, ReLU.
// One block = one 16x16 tile of conv outputs of one image; a thread owns one pixel x 64 channels.
// ------------------------------------------------------------------------------------------
struct StemDev {
    const uint8_t* img;   // [N][3][Hs][Ws] staging (image n occupies the top-left h[n] x w[n])
    const float* w;       // [64][7][7][3] fp32
    const float* scale; const float* shift;
    void* y;              // [N][Hc][Wc][64]
    int N, Hs, Ws, Hc, Wc;
    int h[ALDI_MAX_IMAGES], wd[ALDI_MAX_IMAGES];
    float mean[3], inv_std[3];
};

template <typename T>
__global__ __launch_bounds__(256) void stem_kernel(StemDev p) {
    constexpr int TI = 16 * 2 + 5;            // input tile edge
    __shared__ float ws[147 * 64];            // [tap][co]
    __shared__ float tile[3 * TI * TI];       // [c][y][x]
    const int n = blockIdx.z;
    const int oy0 = blockIdx.y * 16, ox0 = blockIdx.x * 16;
    for (int i = threadIdx.x; i < 147 * 64; i += 256) {
        int co = i & 63, tap = i >> 6;        // tap = (kh*7+kw)*3 + c
        ws[i] = p.w[co * 147 + tap];
    }
    const int iy0 = oy0 * 2 - 3, ix0 = ox0 * 2 - 3;
    const int h = p.h[n], w = p.wd[n];
    for (int i = threadIdx.x; i < 3 * TI * TI; i += 256) {
        int c = i / (TI * TI), r = i - c * TI * TI;
        int yy = r / TI, xx = r - yy * TI;
        int iy = iy0 + yy, ix = ix0 + xx;
        float v = 0.f;
        if (iy >= 0 && iy < h && ix >= 0 && ix < w)
            v = ((float)p.img[(((long)n * 3 + c) * p.Hs + iy) * p.Ws + ix] - p.mean[c]) * p.inv_std[c];
        tile[i] = v;
    }
    __syncthreads();
    const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
    const int oy = oy0 + ty, ox = ox0 + tx;
    float acc[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) acc[i] = 0.f;
    for (int kh = 0; kh < 7; ++kh)
        for (int kw = 0; kw < 7; ++kw)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float xv = tile[(c * TI + ty * 2 + kh) * TI + tx * 2 + kw];
                const float4* wr = reinterpret_cast<const float4*>(&ws[((kh * 7 + kw) * 3 + c) * 64]);
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    float4 wv = wr[q];
                    acc[q * 4 + 0] = fmaf(xv, wv.x, acc[q * 4 + 0]);
                    acc[q * 4 + 1] = fmaf(xv, wv.y, acc[q * 4 + 1]);
                    acc[q * 4 + 2] = fmaf(xv, wv.z, acc[q * 4 + 2]);
                    acc[q * 4 + 3] = fmaf(xv, wv.w, acc[q * 4 + 3]);
                }
            }
    if (oy < p.Hc && ox < p.Wc) {
        T* out = static_cast<T*>(p.y) + (((long)n * p.Hc + oy) * p.Wc + ox) * 64;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float t = acc[q * 4 + r] * p.scale[q * 4 + r] + p.shift[q * 4 + r];
                v[r] = t > 0.f ? t : 0.f;
            }
            store4(out + q * 4, v);
        }
    }
}

// bf16 form on the matrix cores.  GEMM view per block: 256 pixels (16x16) x 64 channels x K, with the reduction laid
// out as k = (c*7 + kh)*8 + kw (kw = 7 is a zero-weight pad, 21 (c,kh) rows padded to 24 -> K = 192 = 6 MFMA k-steps), so
// that the 8 consecutive k of a fragment lane are 8 consecutive input columns of one (c, row) of the LDS image tile --
// the im2col gather is four aligned ds_read_b32.  The fp32 kernel above is VALU bound (147x64 FMAs per pixel).
__global__ __launch_bounds__(256) void stem_mfma_kernel(StemDev p) {
    constexpr int TH = 37, TW = 40, WK = 200;            // tile rows / padded row (elements); padded weight row
    __shared__ __attribute__((aligned(16))) bf16_t wl[64 * WK];
    __shared__ __attribute__((aligned(16))) bf16_t tile[3 * TH * TW];
    const int n = blockIdx.z;
    const int oy0 = blockIdx.y * 16, ox0 = blockIdx.x * 16;
    for (int i = threadIdx.x; i < 64 * 192; i += 256) {
        const int co = i / 192, k = i - co * 192;
        const int idx = k >> 3, kw = k & 7;
        float v = 0.f;
        if (idx < 21 && kw < 7) {
            const int c = idx / 7, kh = idx - c * 7;
            v = p.w[co * 147 + (kh * 7 + kw) * 3 + c];
        }
        wl[co * WK + k] = f32_to_bf16(v);
    }
    const int iy0 = oy0 * 2 - 3, ix0 = ox0 * 2 - 3;
    const int h = p.h[n], w = p.wd[n];
    for (int i = threadIdx.x; i < 3 * TH * TW; i += 256) {
        const int c = i / (TH * TW), r = i - c * TH * TW;
        const int yy = r / TW, xx = r - yy * TW;
        const int iy = iy0 + yy, ix = ix0 + xx;
        float v = 0.f;
        if (xx < TH && iy >= 0 && iy < h && ix >= 0 && ix < w)
            v = ((float)p.img[(((long)n * 3 + c) * p.Hs + iy) * p.Ws + ix] - p.mean[c]) * p.inv_std[c];
        tile[i] = f32_to_bf16(v);
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const uint32_t* tile32 = reinterpret_cast<const uint32_t*>(tile);
    const uint4* wl16 = reinterpret_cast<const uint4*>(wl);
#pragma unroll
    for (int s = 0; s < 6; ++s) {
        int idx = s * 4 + fq;
        idx = idx < 21 ? idx : 20;                       // padded rows carry zero weights; keep the address inside the tile
        const int c = idx / 7, kh = idx - c * 7;
        uint4 xf[4], wf[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ty = wave * 4 + i;
            const int e = ((c * TH + ty * 2 + kh) * TW + fr * 2) >> 1;       // 32-bit word index (fr*2 is even)
            xf[i] = make_uint4(tile32[e], tile32[e + 1], tile32[e + 2], tile32[e + 3]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) wf[j] = wl16[((j * 16 + fr) * WK + (s * 4 + fq) * 8) >> 3];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8_t*>(&wf[j]), *reinterpret_cast<bf16x8_t*>(&xf[i]),
                                                                     acc[i][j], 0, 0, 0);
    }
    // lane owns pixel (row wave*4+i, column fr), channels j*16 + fq*4 .. +3
    bf16_t* Y = static_cast<bf16_t*>(p.y);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int ch = j * 16 + fq * 4;
        const float4 sc = *reinterpret_cast<const float4*>(p.scale + ch), sh = *reinterpret_cast<const float4*>(p.shift + ch);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int oy = oy0 + wave * 4 + i, ox = ox0 + fr;
            if (oy < p.Hc && ox < p.Wc) {
                uint2 o;
                o.x = pack2_bf16(fmaxf(acc[i][j][0] * sc.x + sh.x, 0.f), fmaxf(acc[i][j][1] * sc.y + sh.y, 0.f));
                o.y = pack2_bf16(fmaxf(acc[i][j][2] * sc.z + sh.z, 0.f), fmaxf(acc[i][j][3] * sc.w + sh.w, 0.f));
                *reinterpret_cast<uint2*>(Y + (((long)n * p.Hc + oy) * p.Wc + ox) * 64 + ch) = o;
            }
        }
    }
}

// Stem + max-pool in one kernel (bf16).  Unfused, the 7x7/2 conv writes its [N][400][672][64] map (206 MB for the step's six
// images) only for the 3x3/2 max-pool to read it back; both kernels are then bound by that round trip and by the per-workgroup
// weight conversion.  Here a workgroup produces a 7x7 tile of POOLED pixels: it computes the 16x16 tile of conv outputs that
// starts one row / column before the first window (rows 2*py0-1 .. 2*py0+14: fifteen of them cover the seven windows; one conv row
// per tile edge is computed twice), keeps it in LDS after FrozenBN + ReLU (-inf where the conv pixel lies outside the map: the
// pool pads with -inf), and writes 16-B channel chunks of the 3x3 maxima.  Weights arrive pre-packed in the MFMA k order
// (aldi_stem_pack_weights: once per weight refresh instead of 12 K scalar loads + conversions per workgroup).  Same MFMA sequence
// per conv pixel as stem_mfma_kernel, so the result equals maxpool3s2(stem) bit for bit.
constexpr int kStemWK = 200;              // padded weight row (elements) of the packed [64][WK] bf16 image
__global__ void stem_pack_kernel(const float* __restrict__ w, bf16_t* __restrict__ wpk) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 64 * kStemWK) return;
    const int co = i / kStemWK, k = i - co * kStemWK;
    const int idx = k >> 3, kw = k & 7;
    float v = 0.f;
    if (k < 192 && idx < 21 && kw < 7) {
        const int c = idx / 7, kh = idx - c * 7;
        v = w[co * 147 + (kh * 7 + kw) * 3 + c];
    }
    wpk[i] = f32_to_bf16(v);
}

struct StemPoolDev {
    StemDev s;
    const bf16_t* wpk;
    bf16_t* yp;           // [N][Hp][Wp][64]
    int Hp, Wp;
};

__global__ __launch_bounds__(256) void stem_pool_mfma_kernel(StemPoolDev q) {
    const StemDev& p = q.s;
    constexpr int TH = 37, TW = 40, WK = kStemWK, PT = 7, CS = 144;   // CS: bytes per conv pixel in the LDS conv tile (128 + pad)
    constexpr int kWlBytes = 64 * WK * 2, kTileBytes = 3 * TH * TW * 2, kCtBytes = 256 * CS;
    __shared__ __attribute__((aligned(16))) unsigned char smem[(kWlBytes + kTileBytes) > kCtBytes ? (kWlBytes + kTileBytes) : kCtBytes];
    bf16_t* wl = reinterpret_cast<bf16_t*>(smem);
    bf16_t* tile = reinterpret_cast<bf16_t*>(smem + kWlBytes);
    const int n = blockIdx.z;
    const int py0 = blockIdx.y * PT, px0 = blockIdx.x * PT;
    const int cy0 = 2 * py0 - 1, cx0 = 2 * px0 - 1;         // first conv pixel of the 16x16 tile
    {   // packed weights: 1600 16-B chunks
        const uint4* src = reinterpret_cast<const uint4*>(q.wpk);
        uint4* dst = reinterpret_cast<uint4*>(wl);
        for (int i = threadIdx.x; i < 64 * WK / 8; i += 256) dst[i] = src[i];
    }
    const int iy0 = cy0 * 2 - 3, ix0 = cx0 * 2 - 3;
    const int h = p.h[n], w = p.wd[n];
    {   // input tile: thread t copies 20 consecutive columns of one (channel, row)
        const int r = threadIdx.x >> 1, half = threadIdx.x & 1;
        if (r < 3 * TH) {
            const int c = r / TH, yy = r - c * TH;
            const int iy = iy0 + yy;
            const bool rok = iy >= 0 && iy < h;
            const uint8_t* row = p.img + (((long)n * 3 + c) * p.Hs + (rok ? iy : 0)) * p.Ws;
            const float mean = p.mean[c], inv = p.inv_std[c];
#pragma unroll
            for (int k = 0; k < 20; ++k) {
                const int xx = half * 20 + k, ix = ix0 + xx;
                float v = 0.f;
                if (rok && xx < TH && ix >= 0 && ix < w) v = ((float)row[ix] - mean) * inv;
                tile[(c * TH + yy) * TW + xx] = f32_to_bf16(v);
            }
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    {
        const uint32_t* tile32 = reinterpret_cast<const uint32_t*>(tile);
        const uint4* wl16 = reinterpret_cast<const uint4*>(wl);
#pragma unroll
        for (int s = 0; s < 6; ++s) {
            int idx = s * 4 + fq;
            idx = idx < 21 ? idx : 20;                       // padded rows carry zero weights; keep the address inside the tile
            const int c = idx / 7, kh = idx - c * 7;
            uint4 xf[4], wf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ty = wave * 4 + i;
                const int e = ((c * TH + ty * 2 + kh) * TW + fr * 2) >> 1;
                xf[i] = make_uint4(tile32[e], tile32[e + 1], tile32[e + 2], tile32[e + 3]);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) wf[j] = wl16[((j * 16 + fr) * WK + (s * 4 + fq) * 8) >> 3];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8_t*>(&wf[j]), *reinterpret_cast<bf16x8_t*>(&xf[i]),
                                                                         acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();                                        // everyone is done with wl / tile: the conv tile takes their place
    // lane owns conv pixel (row wave*4+i, column fr), channels j*16 + fq*4 .. +3
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int ch = j * 16 + fq * 4;
        const float4 sc = *reinterpret_cast<const float4*>(p.scale + ch), sh = *reinterpret_cast<const float4*>(p.shift + ch);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ry = wave * 4 + i, cy = cy0 + ry, cx = cx0 + fr;
            uint2 o;
            if (cy >= 0 && cy < p.Hc && cx >= 0 && cx < p.Wc) {
                o.x = pack2_bf16(fmaxf(acc[i][j][0] * sc.x + sh.x, 0.f), fmaxf(acc[i][j][1] * sc.y + sh.y, 0.f));
                o.y = pack2_bf16(fmaxf(acc[i][j][2] * sc.z + sh.z, 0.f), fmaxf(acc[i][j][3] * sc.w + sh.w, 0.f));
            } else {
                o.x = o.y = 0xff80ff80u;                     // -inf, -inf
            }
            *reinterpret_cast<uint2*>(smem + (ry * 16 + fr) * CS + ch * 2) = o;
        }
    }
    __syncthreads();
    typedef short s16x8_t __attribute__((ext_vector_type(8)));
    for (int item = threadIdx.x; item < PT * PT * 8; item += 256) {
        const int chunk = item & 7, pp = item >> 3;
        const int pi = pp / PT, pj = pp - pi * PT;
        const int py = py0 + pi, px = px0 + pj;
        if (py >= q.Hp || px >= q.Wp) continue;
        // ReLU outputs are >= 0 and the padding is -inf: as signed 16-bit integers bf16 patterns order like the values
        s16x8_t m = *reinterpret_cast<const s16x8_t*>(smem + ((2 * pi) * 16 + 2 * pj) * CS + chunk * 16);
#pragma unroll
        for (int d = 1; d < 9; ++d) {
            const s16x8_t v = *reinterpret_cast<const s16x8_t*>(smem + ((2 * pi + d / 3) * 16 + 2 * pj + d % 3) * CS + chunk * 16);
            m = __builtin_elementwise_max(m, v);
        }
        *reinterpret_cast<s16x8_t*>(q.yp + (((long)n * q.Hp + py) * q.Wp + px) * 64 + chunk * 8) = m;
    }
}

// max_pool2d(kernel 3, stride 2, pad 1), NHWC, 4 channels per thread
template <typename T>
__global__ void maxpool3s2_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int H, int W, int C, int Ho, int Wo) {
    long total = (long)N * Ho * Wo * (C / 4);
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int c4 = (int)(i % (C / 4));
        long r = i / (C / 4);
        int wo = (int)(r % Wo); r /= Wo;
        int ho = (int)(r % Ho);
        int n = (int)(r / Ho);
        float m[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        for (int dy = 0; dy < 3; ++dy) {
            int hi = ho * 2 - 1 + dy;
            if (hi < 0 || hi >= H) continue;
            for (int dx = 0; dx < 3; ++dx) {
                int wi = wo * 2 - 1 + dx;
                if (wi < 0 || wi >= W) continue;
                float v[4];
                load4(x + (((long)n * H + hi) * W + wi) * C + c4 * 4, v);
#pragma unroll
                for (int k = 0; k < 4; ++k) m[k] = fmaxf(m[k], v[k]);
            }
        }
        store4(y + i * 4, m);
    }
}

// y[n,ho,wo,:] = x[n,2ho,2wo,:]   (LastLevelMaxPool: max_pool2d(kernel 1, stride 2))
template <typename T>
__global__ void subsample2_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int H, int W, int C, int Ho, int Wo) {
    long total = (long)N * Ho * Wo * (C / 4);
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int c4 = (int)(i % (C / 4));
        long r = i / (C / 4);
        int wo = (int)(r % Wo); r /= Wo;
        int ho = (int)(r % Ho);
        int n = (int)(r / Ho);
        float v[4];
        load4(x + (((long)n * H + ho * 2) * W + wo * 2) * C + c4 * 4, v);
        store4(y + i * 4, v);
    }
}

// gx[n,2ho,2wo,:] += gy[n,ho,wo,:]
template <typename T>
__global__ void subsample2_bwd_kernel(const T* __restrict__ gy, T* __restrict__ gx, int N, int H, int W, int C, int Ho, int Wo) {
    long total = (long)N * Ho * Wo * (C / 4);
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int c4 = (int)(i % (C / 4));
        long r = i / (C / 4);
        int wo = (int)(r % Wo); r /= Wo;
        int ho = (int)(r % Ho);
        int n = (int)(r / Ho);
        float a[4], b[4];
        load4(gy + i * 4, a);
        T* dst = gx + (((long)n * H + ho * 2) * W + wo * 2) * C + c4 * 4;
        load4(dst, b);
#pragma unroll
        for (int k = 0; k < 4; ++k) b[k] += a[k];
        store4(dst, b);
    }
}

// backward of nearest-upsample x2 + add: out[n,h,w,:] (= or +=) sum of the 2x2 block of g
template <typename T>
__global__ void upsample2_bwd_kernel(const T* __restrict__ g, T* __restrict__ out, int N, int Hc, int Wc, int C, int accumulate) {
    long total = (long)N * Hc * Wc * (C / 4);
    const int H = Hc * 2, W = Wc * 2;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int c4 = (int)(i % (C / 4));
        long r = i / (C / 4);
        int w = (int)(r % Wc); r /= Wc;
        int h = (int)(r % Hc);
        int n = (int)(r / Hc);
        float s[4] = {0, 0, 0, 0};
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                float v[4];
                load4(g + (((long)n * H + h * 2 + dy) * W + w * 2 + dx) * C + c4 * 4, v);
#pragma unroll
                for (int k = 0; k < 4; ++k) s[k] += v[k];
            }
        if (accumulate) {
            float o[4];
            load4(out + i * 4, o);
#pragma unroll
            for (int k = 0; k < 4; ++k) s[k] += o[k];
        }
        store4(out + i * 4, s);
    }
}

// The FPN's top-down backward as ONE launch: o3 += blocks(g2); o4 += blocks(o3); o5 += blocks(o4) -- three dependent calls of the kernel
// above, each a launch on the chain between the FPN's output convolutions' and the lateral convolutions' data gradients.  A thread owns 4
// channels of one pixel of the COARSEST map and walks its 4 / 16 / 64 descendants; every level's sums are formed in the order of the kernel
// above and pass through the storage type (store4 + what it would read back) before the next level uses them: the same bits.
template <typename T> __device__ __forceinline__ void through_storage(float v[4]);
template <> __device__ __forceinline__ void through_storage<float>(float v[4]) {}
template <> __device__ __forceinline__ void through_storage<bf16_t>(float v[4]) {
    bf16_t t[4];
    store4(t, v);
    load4(t, v);
}
template <typename T>
__global__ void upsample2_bwd_chain_kernel(const T* __restrict__ g2, T* __restrict__ o3, T* __restrict__ o4, T* __restrict__ o5, int N, int H5, int W5, int C) {
    const long total = (long)N * H5 * W5 * (C / 4);
    const int H4 = H5 * 2, W4 = W5 * 2, H3 = H5 * 4, W3 = W5 * 4, H2 = H5 * 8, W2 = W5 * 8;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % (C / 4));
        long r = i / (C / 4);
        const int w5 = (int)(r % W5); r /= W5;
        const int h5 = (int)(r % H5);
        const int n = (int)(r / H5);
        float s5[4] = {0, 0, 0, 0};
        for (int a = 0; a < 4; ++a) {                                  // the 2 x 2 block of the middle map, (dy, dx) order
            const int h4 = h5 * 2 + (a >> 1), w4 = w5 * 2 + (a & 1);
            float s4[4] = {0, 0, 0, 0};
            for (int b = 0; b < 4; ++b) {
                const int h3 = h4 * 2 + (b >> 1), w3 = w4 * 2 + (b & 1);
                float s3[4] = {0, 0, 0, 0};
#pragma unroll
                for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 2; ++dx) {
                        float v[4];
                        load4(g2 + (((long)n * H2 + h3 * 2 + dy) * W2 + w3 * 2 + dx) * C + c4 * 4, v);
#pragma unroll
                        for (int k = 0; k < 4; ++k) s3[k] += v[k];
                    }
                T* p3 = o3 + (((long)n * H3 + h3) * W3 + w3) * C + c4 * 4;
                float o[4];
                load4(p3, o);
#pragma unroll
                for (int k = 0; k < 4; ++k) s3[k] += o[k];
                store4(p3, s3);
                through_storage<T>(s3);
#pragma unroll
                for (int k = 0; k < 4; ++k) s4[k] += s3[k];
            }
            T* p4 = o4 + (((long)n * H4 + h4) * W4 + w4) * C + c4 * 4;
            float o[4];
            load4(p4, o);
#pragma unroll
            for (int k = 0; k < 4; ++k) s4[k] += o[k];
            store4(p4, s4);
            through_storage<T>(s4);
#pragma unroll
            for (int k = 0; k < 4; ++k) s5[k] += s4[k];
        }
        T* p5 = o5 + i * 4;
        float o[4];
        load4(p5, o);
#pragma unroll
        for (int k = 0; k < 4; ++k) s5[k] += o[k];
        store4(p5, s5);
    }
}

// out = (a ? a : 0) + (b ? b : 0) * bscale, optionally masked by relu_src > 0; a is T, b is fp32
template <typename T>
__global__ void add_f32_kernel(const T* __restrict__ a, const float* __restrict__ b, const T* __restrict__ relu_src, T* __restrict__ out, long n4) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        float v[4] = {0, 0, 0, 0};
        if (a) load4(a + i * 4, v);
        if (b) {
            float4 t = *reinterpret_cast<const float4*>(b + i * 4);
            v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w;
        }
        if (relu_src) {
            float m[4];
            load4(relu_src + i * 4, m);
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = m[k] > 0.f ? v[k] : 0.f;
        }
        store4(out + i * 4, v);
    }
}

template <typename T>
__global__ void cast_from_f32_kernel(const float* __restrict__ src, T* __restrict__ dst, long n) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) Elem<T>::st(dst + i, src[i]);
}

// torch.optim.SGD (momentum, dampening 0, nesterov off):  d = g + wd*p ; buf = mu*buf + d ; p -= lr*buf.
// Also refreshes the compute-dtype copy of the weights.  Flat buffers; p/g/buf fp32.
template <typename T>
__global__ void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf, T* __restrict__ pc,
                           long n, float lr, float mu, float wd, float gscale, int first) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float w = p[i];
        float d = g[i] * gscale + wd * w;
        float b = first ? d : mu * buf[i] + d;
        buf[i] = b;
        w = w - lr * b;
        p[i] = w;
        if (pc) Elem<T>::st(pc + i, w);
    }
}

// the same update with its scalars read from device memory (hyper = lr, momentum, weight decay, gradient scale): the launch can sit
// in a replayed hipGraph while the learning-rate schedule moves (the host refreshes the four floats before the replay)
template <typename T>
__global__ void sgd_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf, T* __restrict__ pc,
                               long n, const float* __restrict__ hyper) {
    const float lr = hyper[0], mu = hyper[1], wd = hyper[2], gscale = hyper[3];
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float w = p[i];
        const float d = g[i] * gscale + wd * w;
        const float b = mu * buf[i] + d;
        buf[i] = b;
        w = w - lr * b;
        p[i] = w;
        if (pc) Elem<T>::st(pc + i, w);
    }
}

// reference aldi/ema.py:43-46:  t = s*(1-alpha) + t*alpha   (copy_only: t = s, aldi/ema.py:29-30)
// tc (nullable): the compute-dtype copy of the first nc elements (the weights), written in the same pass
template <typename T>
__global__ void ema_kernel(float* __restrict__ t, const float* __restrict__ s, T* __restrict__ tc, long n, long nc, float one_minus_alpha, float alpha,
                           int copy_only) {
    const long n4 = ((reinterpret_cast<uintptr_t>(t) | reinterpret_cast<uintptr_t>(s)) & 15) == 0 ? n >> 2 : 0;      // 16-byte body
    for (long q = blockIdx.x * (long)blockDim.x + threadIdx.x; q < n4; q += (long)gridDim.x * blockDim.x) {
        const float4 sv = reinterpret_cast<const float4*>(s)[q];
        float4 v = sv;
        if (!copy_only) {
            const float4 tv = reinterpret_cast<const float4*>(t)[q];
            v.x = sv.x * one_minus_alpha + tv.x * alpha; v.y = sv.y * one_minus_alpha + tv.y * alpha;
            v.z = sv.z * one_minus_alpha + tv.z * alpha; v.w = sv.w * one_minus_alpha + tv.w * alpha;
        }
        reinterpret_cast<float4*>(t)[q] = v;
        if (tc) {
            const long i = q << 2;
            if (i < nc) Elem<T>::st(tc + i, v.x);
            if (i + 1 < nc) Elem<T>::st(tc + i + 1, v.y);
            if (i + 2 < nc) Elem<T>::st(tc + i + 2, v.z);
            if (i + 3 < nc) Elem<T>::st(tc + i + 3, v.w);
        }
    }
    for (long i = (n4 << 2) + blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float v = copy_only ? s[i] : s[i] * one_minus_alpha + t[i] * alpha;
        t[i] = v;
        if (tc && i < nc) Elem<T>::st(tc + i, v);
    }
}

// FrozenBN fold: scale = w * rsqrt(var + eps); shift = b - mean * scale
__global__ void bn_fold_kernel(const float* __restrict__ w, const float* __restrict__ b, const float* __restrict__ mean,
                               const float* __restrict__ var, float* __restrict__ scale, float* __restrict__ shift, int C, float eps) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < C) {
        float sc = w[i] * (1.0f / sqrtf(var[i] + eps));
        scale[i] = sc;
        shift[i] = b[i] - mean[i] * sc;
    }
}

static inline int nblocks(long n, int per = 256, int cap = 16384) {
    long b = (n + per - 1) / per;
    return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

}  // namespace

#define DISPATCH_T(dtype, KERNEL, grid, block, st, ...)                                              \
    do {                                                                                             \
        if ((dtype) == ALDI_BF16) hipLaunchKernelGGL(KERNEL<bf16_t>, grid, block, 0, st, __VA_ARGS__); \
        else if ((dtype) == ALDI_F32) hipLaunchKernelGGL(KERNEL<float>, grid, block, 0, st, __VA_ARGS__); \
        else return aldi_set_error_msg(ALDI_ERR_ARG, #KERNEL ": bad dtype");                         \
        ALDI_CHECK_LAUNCH();                                                                         \
    } while (0)

extern "C" int aldi_stem_forward(const aldi_stem_args* a, aldi_stream_t stream) {
    if (!a || !a->img || !a->w || !a->y || a->N < 1 || a->N > ALDI_MAX_IMAGES) return aldi_set_error_msg(ALDI_ERR_ARG, "stem_forward: bad args");
    StemDev d;
    d.img = a->img; d.w = a->w; d.scale = a->scale; d.shift = a->shift; d.y = a->y;
    d.N = a->N; d.Hs = a->Hs; d.Ws = a->Ws; d.Hc = a->Hc; d.Wc = a->Wc;
    for (int i = 0; i < a->N; ++i) { d.h[i] = a->h[i]; d.wd[i] = a->w_img[i]; }
    for (int c = 0; c < 3; ++c) { d.mean[c] = a->mean[c]; d.inv_std[c] = 1.0f / a->std[c]; }
    dim3 grid(cdiv(a->Wc, 16), cdiv(a->Hc, 16), a->N);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int mfma_env = aldi_tuning().stem_mfma;
    if (a->dtype == ALDI_BF16 && mfma_env && a->scale && a->shift) {
        hipLaunchKernelGGL(stem_mfma_kernel, grid, dim3(256), 0, st, d);
        ALDI_CHECK_LAUNCH();
        return ALDI_OK;
    }
    DISPATCH_T(a->dtype, stem_kernel, grid, dim3(256), st, d);
    return ALDI_OK;
}

extern "C" int aldi_stem_pack_weights(const float* w, void* w_packed, aldi_stream_t stream) {
    if (!w || !w_packed) return aldi_set_error_msg(ALDI_ERR_ARG, "stem_pack_weights: null pointer");
    hipLaunchKernelGGL(stem_pack_kernel, dim3(cdiv(64 * kStemWK, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), w, (bf16_t*)w_packed);
    ALDI_CHECK_LAUNCH();
    return ALDI_OK;
}

extern "C" int aldi_stem_pool_forward(const aldi_stem_args* a, const void* w_packed, void* y_pool, aldi_stream_t stream) {
    if (!a || !a->img || !w_packed || !y_pool || !a->scale || !a->shift || a->N < 1 || a->N > ALDI_MAX_IMAGES || a->dtype != ALDI_BF16)
        return aldi_set_error_msg(ALDI_ERR_ARG, "stem_pool_forward: bad args (bf16 only)");
    StemPoolDev q;
    StemDev& d = q.s;
    d.img = a->img; d.w = a->w; d.scale = a->scale; d.shift = a->shift; d.y = nullptr;
    d.N = a->N; d.Hs = a->Hs; d.Ws = a->Ws; d.Hc = a->Hc; d.Wc = a->Wc;
    for (int i = 0; i < a->N; ++i) { d.h[i] = a->h[i]; d.wd[i] = a->w_img[i]; }
    for (int c = 0; c < 3; ++c) { d.mean[c] = a->mean[c]; d.inv_std[c] = 1.0f / a->std[c]; }
    q.wpk = static_cast<const bf16_t*>(w_packed);
    q.yp = static_cast<bf16_t*>(y_pool);
    q.Hp = (a->Hc - 1) / 2 + 1; q.Wp = (a->Wc - 1) / 2 + 1;
    dim3 grid(cdiv(q.Wp, 7), cdiv(q.Hp, 7), a->N);
    hipLaunchKernelGGL(stem_pool_mfma_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream), q);
    ALDI_CHECK_LAUNCH();
    return ALDI_OK;
}

extern "C" int aldi_maxpool3s2(const void* x, void* y, int N, int H, int W, int C, int dtype, aldi_stream_t stream) {
    int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    long total = (long)N * Ho * Wo * (C / 4);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == ALDI_BF16) hipLaunchKernelGGL(maxpool3s2_kernel<bf16_t>, dim3(nblocks(total)), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y, N, H, W, C, Ho, Wo);
    else hipLaunchKernelGGL(maxpool3s2_kernel<float>, dim3(nblocks(total)), dim3(256), 0, st, (const float*)x, (float*)y, N, H, W, C, Ho, Wo);
    ALDI_CHECK_LAUNCH();
    return ALDI_OK;
}

// Input staging: up to 16 uint8 CHW images of their own sizes -> rows [n][c][0..h)[0..w) of the padded batch buffer, ONE launch
// (the reference's ImageList.from_tensors, detectron2 via aldi/trainer.py:28; per-image strided copies were 6 launches on the
// chain before the step's first graph).  The padding of the buffer is not touched.
constexpr int kMaxStage = 16;
struct StageDev { const unsigned char* src[kMaxStage]; int h[kMaxStage], w[kMaxStage]; int n, C, Hs, Ws; };
__global__ __launch_bounds__(256) void stage_images_kernel(StageDev p, unsigned char* __restrict__ dst) {
    const int i = blockIdx.z, c = blockIdx.y;
    const int h = p.h[i], w = p.w[i];
    const unsigned char* s = p.src[i] + (long)c * h * w;
    unsigned char* d = dst + ((long)i * p.C + c) * p.Hs * p.Ws;
    for (int row = blockIdx.x; row < h; row += gridDim.x) {
        const unsigned char* sr = s + (long)row * w;
        unsigned char* dr = d + (long)row * p.Ws;
        // 4-byte body when both rows start aligned, bytes otherwise / for the tail
        if ((((uintptr_t)sr | (uintptr_t)dr) & 3) == 0) {
            const int w4 = w >> 2;
            for (int x = threadIdx.x; x < w4; x += blockDim.x) reinterpret_cast<unsigned*>(dr)[x] = reinterpret_cast<const unsigned*>(sr)[x];
            for (int x = (w4 << 2) + threadIdx.x; x < w; x += blockDim.x) dr[x] = sr[x];
        } else {
            for (int x = threadIdx.x; x < w; x += blockDim.x) dr[x] = sr[x];
        }
    }
}

extern "C" int aldi_stage_images(const void* const* images, const int* heights, const int* widths, int n, int C, int Hs, int Ws, void* batch,
                                 aldi_stream_t stream) {
    if (!images || !heights || !widths || !batch || n < 1 || n > kMaxStage || C < 1) return aldi_set_error_msg(ALDI_ERR_ARG, "stage_images: bad args (1..16 images)");
    StageDev p;
    p.n = n; p.C = C; p.Hs = Hs; p.Ws = Ws;
    int hmax = 1;
    for (int i = 0; i < kMaxStage; ++i) {
        p.src[i] = i < n ? static_cast<const unsigned char*>(images[i]) : nullptr;
        p.h[i] = i < n ? heights[i] : 0; p.w[i] = i < n ? widths[i] : 0;
        if (i < n && (!images[i] || heights[i] < 0 || widths[i] < 0 || heights[i] > Hs || widths[i] > Ws)) return aldi_set_error_msg(ALDI_ERR_ARG, "stage_images: image larger than the batch");
        if (p.h[i] > hmax) hmax = p.h[i];
    }
    hipLaunchKernelGGL(stage_images_kernel, dim3(hmax < 256 ? hmax : 256, C, n), dim3(256), 0, static_cast<hipStream_t>(stream), p, static_cast<unsigned char*>(batch));
    ALDI_CHECK_LAUNCH();
    return ALDI_OK;
}

extern "C" int aldi_subsample2(const void* x, void* y, int N, int H, int W, int C, int backward, int dtype, aldi_stream_t stream) {
    int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    long total = (long)N * Ho * Wo * (C / 4);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (!backward) {
        if (dtype == ALDI_BF16) hipLaunchKernelGGL(subsample2_kernel<bf16_t>, dim3(nblocks(total)), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y, N, H, W, C, Ho, Wo);
        else hipLaunchKernelGGL(subsample2_kernel<float>, dim3(nblocks(total)), dim3(256), 0, st, (const float*)x, (float*)y, N, H, W, C, Ho, Wo);
    } else {   // x = grad wrt small map [N][Ho][Wo][C], y = grad wrt big map (accumulated)
        if (dtype == ALDI_BF16) hipLaunchKernelGGL(subsample2_bwd_kernel<bf16_t>, dim3(nblocks(total)), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y, N, H, W, C, Ho, Wo);
        else hipLaunchKernelGGL(subsample2_bwd_kernel<float>, dim3(nblocks(total)), dim3(256), 0, st, (const float*)x, (float*)y, N, H, W, C, Ho, Wo);
    }
    ALDI_CHECK_LAUNCH();
    return ALDI_OK;
}

extern "C" int aldi_upsample2_bwd(const void* g, void* out, int N, int Hc, int Wc, int C, int accumulate, int dtype, aldi_stream_t stream) {
    long total = (long)N * Hc * Wc * (C / 4);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == ALDI_BF16) hipLaunchKernelGGL(upsample2_bwd_kernel<bf16_t>, dim3(nblocks(total)), dim3(256), 0, st, (const bf16_t*)g, (bf16_t*)out, N, Hc, Wc, C, accumulate);
    else hipLaunchKernelGGL(upsample2_bwd_kernel<float>, dim3(nblocks(total)), dim3(256), 0, st, (const float*)g, (float*)out, N, Hc, Wc, C, accumulate);
    ALDI_CHECK_LAUNCH();
    return ALDI_OK;
}

extern "C" int aldi_upsample2_bwd_chain(const void* g2, void* o3, void* o4, void* o5, int N, int H5, int W5, int C, int dtype, aldi_stream_t stream) {
    if (!g2 || !o3 || !o4 || !o5 || N < 1 || H5 < 1 || W5 < 1 || (C & 3)) return aldi_set_error_msg(ALDI_ERR_ARG, "upsample2_bwd_chain: bad args");
    const long total = (long)N * H5 * W5 * (C / 4);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == ALDI_BF16) hipLaunchKernelGGL(upsample2_bwd_chain_kernel<bf16_t>, dim3(cdiv(total, 256)), dim3(256), 0, st, (const bf16_t*)g2, (bf16_t*)o3, (bf16_t*)o4, (bf16_t*)o5, N, H5, W5, C);
    else hipLaunchKernelGGL(upsample2_bwd_chain_kernel<float>, dim3(cdiv(total, 256)), dim3(256), 0, st, (const float*)g2, (float*)o3, (float*)o4, (float*)o5, N, H5, W5, C);
    ALDI_CHECK_LAUNCH();
    return ALDI_OK;
}

extern "C" int aldi_add_f32(const void* a, const float* b, const void* relu_src, void* out, long n, int dtype, aldi_stream_t stream) {
    if (n % 4) return aldi_set_error_msg(ALDI_ERR_ARG, "add_f32: n must be a multiple of 4");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == ALDI_BF16) hipLaunchKernelGGL(add_f32_kernel<bf16_t>, dim3(nblocks(n / 4)), dim3(256), 0, st, (const bf16_t*)a, b, (const bf16_t*)relu_src, (bf16_t*)out, n / 4);
    else hipLaunchKernelGGL(add_f32_kernel<float>, dim3(nblocks(n / 4)), dim3(256), 0, st, (const float*)a, b, (const float*)relu_src, (float*)out, n / 4);
    ALDI_CHECK_LAUNCH();
    return ALDI_OK;
}

extern "C" int aldi_cast_from_f32(const float* src, void* dst, long n, int dtype, aldi_stream_t stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == ALDI_BF16) hipLaunchKernelGGL(cast_from_f32_kernel<bf16_t>, dim3(nblocks(n)), dim3(256), 0, st, src, (bf16_t*)dst, n);
    else hipLaunchKernelGGL(cast_from_f32_kernel<float>, dim3(nblocks(n)), dim3(256), 0, st, src, (float*)dst, n);
    ALDI_CHECK_LAUNCH();
    return ALDI_OK;
}

extern "C" int aldi_sgd_step(float* p, const float* g, float* buf, void* p_compute, long n, float lr, float momentum, float weight_decay,
                             float grad_scale, int first_step, int dtype, aldi_stream_t stream) {
    if (!p || !g || !buf) return aldi_set_error_msg(ALDI_ERR_ARG, "sgd_step: null pointer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == ALDI_BF16) hipLaunchKernelGGL(sgd_kernel<bf16_t>, dim3(nblocks(n)), dim3(256), 0, st, p, g, buf, (bf16_t*)p_compute, n, lr, momentum, weight_decay, grad_scale, first_step);
    else hipLaunchKernelGGL(sgd_kernel<float>, dim3(nblocks(n)), dim3(256), 0, st, p, g, buf, (float*)nullptr, n, lr, momentum, weight_decay, grad_scale, first_step);
    ALDI_CHECK_LAUNCH();
    return ALDI_OK;
}

extern "C" int aldi_sgd_step_dev(float* p, const float* g, float* buf, void* p_compute, long n, const float* hyper, int dtype, aldi_stream_t stream) {
    if (!p || !g || !buf || !hyper || n <= 0) return aldi_set_error_msg(ALDI_ERR_ARG, "sgd_step_dev: bad args");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == ALDI_BF16) hipLaunchKernelGGL(sgd_dev_kernel<bf16_t>, dim3(nblocks(n)), dim3(256), 0, st, p, g, buf, (bf16_t*)p_compute, n, hyper);
    else hipLaunchKernelGGL(sgd_dev_kernel<float>, dim3(nblocks(n)), dim3(256), 0, st, p, g, buf, (float*)nullptr, n, hyper);
    ALDI_CHECK_LAUNCH();
    return ALDI_OK;
}

extern "C" int aldi_ema_update(float* teacher, const float* student, void* teacher_compute, long n, long n_compute, double alpha, int copy_only, int dtype,
                               aldi_stream_t stream) {
    if (!teacher || !student) return aldi_set_error_msg(ALDI_ERR_ARG, "ema_update: null pointer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    // python computes (1 - alpha) in DOUBLE; torch then casts each scalar to fp32 for the fp32 tensor multiply
    const float oma = (float)(1.0 - alpha);
    const float alpha_f = (float)alpha;
    const long nc = teacher_compute ? (n_compute < n ? n_compute : n) : 0;
    // ema_blocks: the tick runs beside the student's stem at the head of the step; a grid that fills every wave slot of the chip starves that
    // (latency-bound) kernel, a grid-stride loop over fewer workgroups streams as fast
    const int cap = aldi_tuning().ema_blocks > 0 ? aldi_tuning().ema_blocks : 16384;
    if (dtype == ALDI_BF16) hipLaunchKernelGGL(ema_kernel<bf16_t>, dim3(nblocks(n / 4 + 1, 256, cap)), dim3(256), 0, st, teacher, student, (bf16_t*)teacher_compute, n, nc, oma, alpha_f, copy_only);
    else hipLaunchKernelGGL(ema_kernel<float>, dim3(nblocks(n / 4 + 1, 256, cap)), dim3(256), 0, st, teacher, student, (float*)nullptr, n, 0L, oma, alpha_f, copy_only);
    ALDI_CHECK_LAUNCH();
    return ALDI_OK;
}

extern "C" int aldi_bn_fold(const float* w, const float* b, const float* mean, const float* var, float* scale, float* shift, int C, aldi_stream_t stream) {
    hipLaunchKernelGGL(bn_fold_kernel, dim3(cdiv(C, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), w, b, mean, var, scale, shift, C, 1e-5f);
    ALDI_CHECK_LAUNCH();
    return ALDI_OK;
}

extern "C" int aldi_dgrad_weights_batch(const aldi_dgw_item* items, int n_items, int total_tiles, int dtype, aldi_stream_t stream) {
    if (!items || n_items <= 0 || total_tiles <= 0) return aldi_set_error_msg(ALDI_ERR_ARG, "dgrad_weights_batch: bad args");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == ALDI_BF16) hipLaunchKernelGGL(dgrad_weights_batch_kernel<bf16_t>, dim3(total_tiles), dim3(256), 0, st, items, n_items);
    else if (dtype == ALDI_F32) hipLaunchKernelGGL(dgrad_weights_batch_kernel<float>, dim3(total_tiles), dim3(256), 0, st, items, n_items);
    else return aldi_set_error_msg(ALDI_ERR_ARG, "dgrad_weights_batch: bad dtype");
    ALDI_CHECK_LAUNCH();
    return ALDI_OK;
}

extern "C" int aldi_dgrad_weights(const float* w_master, const float* scale, void* wt, int Cout, int KH, int KW, int Cin,
                                  int dtype, aldi_stream_t stream) {
    if (!w_master || !wt) return aldi_set_error_msg(ALDI_ERR_ARG, "dgrad_weights: null pointer");
    long total = (long)Cout * KH * KW * Cin;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == ALDI_BF16) hipLaunchKernelGGL(dgrad_weights_kernel<bf16_t>, dim3(blocks), dim3(256), 0, st, w_master, scale, (bf16_t*)wt, Cout, KH, KW, Cin);
    else hipLaunchKernelGGL(dgrad_weights_kernel<float>, dim3(blocks), dim3(256), 0, st, w_master, scale, (float*)wt, Cout, KH, KW, Cin);
    ALDI_CHECK_LAUNCH();
    return ALDI_OK;
}
