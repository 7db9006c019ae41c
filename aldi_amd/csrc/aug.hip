// Strong augmentation on the device: the reference builds the strong view of every image on
// the CPU with numpy / scipy (aldi/aug.py:39-60 colour blend chain, :80-91 Gaussian blur, :103-138 random erase, :149-171 MIC
// block mask; a 2048x1024 Cityscapes frame costs tens of ms per image in scipy's gaussian_filter alone).  Here the weak
// view already sits in HBM as HWC uint8 and the strong view is derived from it by these kernels; all RANDOM DRAWS stay on
// the host (aldi_amd/aug.py consumes the numpy / python generators in the reference's order), the kernels get parameters.
//
// Results are bit-identical to the numpy/scipy arithmetic (tests/test_aug_gpu.py vs golden g9 and vs the oracle):
//   * blends: numpy-2 promotion -- a float64 scalar/array times-and-plus a float32 image is evaluated in double
//     (contrast, saturation), a python-float weight on a float32 image stays float32 (brightness); no FMA contraction
//     (-ffp-contract=off), except the 3-term BGR dot of RandomSaturation, which the BLAS behind ndarray.dot evaluates as
//     fma(c2, w2, fma(c1, w1, c0 * w0));
//   * blur: scipy.ndimage.gaussian_filter over all THREE axes of the HWC float32 image (channel axis included), each axis
//     NI_Correlate1D's symmetric form in double -- tmp = x[l]*w0; for j = -r..-1: tmp += (x[l+j] + x[l-j]) * w[j] -- with
//     'reflect' borders, rounded to float32 after every axis;
//   * every op ends with clip(0, 255) and a truncating uint8 cast.
#include "common.h"

namespace {

__device__ __forceinline__ uint8_t clip_u8(double v) { v = v < 0.0 ? 0.0 : (v > 255.0 ? 255.0 : v); return (uint8_t)v; }
__device__ __forceinline__ uint8_t clip_u8f(float v) { v = v < 0.f ? 0.f : (v > 255.f ? 255.f : v); return (uint8_t)v; }

__global__ __launch_bounds__(256) void sum_u8_kernel(const uint8_t* __restrict__ img, long n, unsigned long long* __restrict__ sum) {
    unsigned long long acc = 0;
    const long n16 = n / 16;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n16; i += (long)gridDim.x * blockDim.x) {
        const uint4 v = reinterpret_cast<const uint4*>(img)[i];
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) acc += (w[k] & 0xff) + ((w[k] >> 8) & 0xff) + ((w[k] >> 16) & 0xff) + (w[k] >> 24);
    }
    if (blockIdx.x == 0)
        for (long i = n16 * 16 + threadIdx.x; i < n; i += blockDim.x) acc += img[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0 && acc) atomicAdd(sum, acc);
}

// mode 0: contrast (src = image mean), 1: brightness (src = 0), 2: saturation (src = BGR dot [0.299, 0.587, 0.114])
__global__ __launch_bounds__(256) void blend_kernel(uint8_t* __restrict__ img, long npix, int mode, double w, const unsigned long long* __restrict__ sum) {
    const long p = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (p >= npix) return;
    uint8_t* px = img + p * 3;
    const float w32 = (float)w;                       // python float * float32 array: the product is a float32 product
    const double sw = 1.0 - w;
    if (mode == 1) {
        const float s = (float)(sw * 0.0);            // src_weight * 0 -> python float 0.0, added in float32
#pragma unroll
        for (int c = 0; c < 3; ++c) px[c] = clip_u8f(s + w32 * (float)px[c]);
        return;
    }
    double src;
    if (mode == 0) src = sw * ((double)*sum / (double)(npix * 3));
    else src = sw * fma((double)px[2], 0.114, fma((double)px[1], 0.587, (double)px[0] * 0.299));
#pragma unroll
    for (int c = 0; c < 3; ++c) px[c] = clip_u8(src + (double)(w32 * (float)px[c]));
}

__device__ __forceinline__ int reflect(int i, int n) {
    int p = i % (2 * n);
    if (p < 0) p += 2 * n;
    return p >= n ? 2 * n - 1 - p : p;
}

// one axis of gaussian_filter: `len` elements `stride` apart per line; IN is uint8 (first axis) or float (later axes)
template <typename IN>
__global__ __launch_bounds__(256) void blur_axis_kernel(const IN* __restrict__ in, float* __restrict__ out, long total, int len, long stride,
                                                        const double* __restrict__ wts, int r) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= total) return;
    // element index along the axis and the base offset of its line
    const long l = (i / stride) % len;               // position along the filtered axis
    const long base = i - l * stride;
    double tmp = (double)(float)in[i] * wts[r];
    if (l >= r && l + r < len) {                       // interior: no border arithmetic (the reflect() modulo dominated the kernel)
        for (int j = -r; j < 0; ++j) {
            const double a = (double)(float)in[i + (long)j * stride];
            const double b = (double)(float)in[i - (long)j * stride];
            tmp += (a + b) * wts[j + r];
        }
    } else {
        for (int j = -r; j < 0; ++j) {
            const double a = (double)(float)in[base + (long)reflect((int)l + j, len) * stride];
            const double b = (double)(float)in[base + (long)reflect((int)l - j, len) * stride];
            tmp += (a + b) * wts[j + r];
        }
    }
    out[i] = (float)tmp;
}

__global__ __launch_bounds__(256) void f32_to_u8_kernel(const float* __restrict__ in, uint8_t* __restrict__ out, long n) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i < n) out[i] = clip_u8f(in[i]);
}

__global__ __launch_bounds__(256) void erase_kernel(uint8_t* __restrict__ img, int W, int h0, int w0, int h, int w, const float* __restrict__ fill) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= (long)h * w * 3) return;
    const int c = (int)(i % 3);
    const long q = i / 3;
    const int x = (int)(q % w), y = (int)(q / w);
    img[((long)(h0 + y) * W + (w0 + x)) * 3 + c] = clip_u8f(fill[i] * 255.0f);
}

__global__ __launch_bounds__(256) void mic_kernel(uint8_t* __restrict__ img, int H, int W, const uint8_t* __restrict__ mask, int mh, int mw) {
    const long p = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (p >= (long)H * W) return;
    const int x = (int)(p % W), y = (int)(p / W);
    // cv2.resize INTER_NEAREST: src = min(floor(dst * (src_size / dst_size)), src_size - 1), scale in double
    int sy = (int)floor((double)y * ((double)mh / (double)H)), sx = (int)floor((double)x * ((double)mw / (double)W));
    sy = sy < mh - 1 ? sy : mh - 1;
    sx = sx < mw - 1 ? sx : mw - 1;
    if (!mask[sy * mw + sx]) { img[p * 3] = 0; img[p * 3 + 1] = 0; img[p * 3 + 2] = 0; }
}

__global__ __launch_bounds__(256) void hwc_to_chw_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, long npix) {
    const long p = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (p >= npix) return;
#pragma unroll
    for (int c = 0; c < 3; ++c) out[c * npix + p] = in[p * 3 + c];
}

inline dim3 grid1(long n) { return dim3((unsigned)((n + 255) / 256)); }

}  // namespace

extern "C" int aldi_aug_sum_u8(const unsigned char* img, long n, unsigned long long* sum, aldi_stream_t stream) {
    if (!img || !sum || n <= 0) return aldi_set_error_msg(ALDI_ERR_ARG, "aug_sum_u8: bad args");
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipError_t e = hipMemsetAsync(sum, 0, 8, st);
    if (e != hipSuccess) return aldi_set_error(e, __FILE__, __LINE__);
    long blocks = (n / 16 + 255) / 256;
    blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
    hipLaunchKernelGGL(sum_u8_kernel, dim3((unsigned)blocks), dim3(256), 0, st, img, n, sum);
    ALDI_CHECK_LAUNCH();
    return ALDI_OK;
}

extern "C" int aldi_aug_blend(unsigned char* img, int H, int W, int mode, double w, const unsigned long long* sum, aldi_stream_t stream) {
    if (!img || H <= 0 || W <= 0 || mode < 0 || mode > 2 || (mode == 0 && !sum)) return aldi_set_error_msg(ALDI_ERR_ARG, "aug_blend: bad args");
    const long npix = (long)H * W;
    hipLaunchKernelGGL(blend_kernel, grid1(npix), dim3(256), 0, static_cast<hipStream_t>(stream), img, npix, mode, w, sum);
    ALDI_CHECK_LAUNCH();
    return ALDI_OK;
}

extern "C" int aldi_aug_blur(const unsigned char* img, unsigned char* out, float* tmp0, float* tmp1, int H, int W, const double* weights, int radius,
                             aldi_stream_t stream) {
    if (!img || !out || !tmp0 || !tmp1 || !weights || H <= 0 || W <= 0 || radius < 0 || radius > 64) return aldi_set_error_msg(ALDI_ERR_ARG, "aug_blur: bad args");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const long n = (long)H * W * 3;
    hipLaunchKernelGGL(blur_axis_kernel<unsigned char>, grid1(n), dim3(256), 0, st, img, tmp0, n, H, (long)W * 3, weights, radius);     // axis 0 (rows)
    hipLaunchKernelGGL(blur_axis_kernel<float>, grid1(n), dim3(256), 0, st, (const float*)tmp0, tmp1, n, W, 3L, weights, radius);        // axis 1 (columns)
    hipLaunchKernelGGL(blur_axis_kernel<float>, grid1(n), dim3(256), 0, st, (const float*)tmp1, tmp0, n, 3, 1L, weights, radius);        // axis 2 (channels)
    hipLaunchKernelGGL(f32_to_u8_kernel, grid1(n), dim3(256), 0, st, (const float*)tmp0, out, n);
    ALDI_CHECK_LAUNCH();
    return ALDI_OK;
}

extern "C" int aldi_aug_erase(unsigned char* img, int H, int W, int h0, int w0, int h, int w, const float* fill, aldi_stream_t stream) {
    if (!img || !fill || h0 < 0 || w0 < 0 || h <= 0 || w <= 0 || h0 + h > H || w0 + w > W) return aldi_set_error_msg(ALDI_ERR_ARG, "aug_erase: rectangle outside the image");
    hipLaunchKernelGGL(erase_kernel, grid1((long)h * w * 3), dim3(256), 0, static_cast<hipStream_t>(stream), img, W, h0, w0, h, w, fill);
    ALDI_CHECK_LAUNCH();
    return ALDI_OK;
}

extern "C" int aldi_aug_mic(unsigned char* img, int H, int W, const unsigned char* mask, int mh, int mw, aldi_stream_t stream) {
    if (!img || !mask || H <= 0 || W <= 0 || mh <= 0 || mw <= 0) return aldi_set_error_msg(ALDI_ERR_ARG, "aug_mic: bad args");
    hipLaunchKernelGGL(mic_kernel, grid1((long)H * W), dim3(256), 0, static_cast<hipStream_t>(stream), img, H, W, mask, mh, mw);
    ALDI_CHECK_LAUNCH();
    return ALDI_OK;
}

extern "C" int aldi_aug_hwc_to_chw(const unsigned char* in, unsigned char* out, int H, int W, aldi_stream_t stream) {
    if (!in || !out || H <= 0 || W <= 0) return aldi_set_error_msg(ALDI_ERR_ARG, "aug_hwc_to_chw: bad args");
    hipLaunchKernelGGL(hwc_to_chw_kernel, grid1((long)H * W), dim3(256), 0, static_cast<hipStream_t>(stream), in, out, (long)H * W);
    ALDI_CHECK_LAUNCH();
    return ALDI_OK;
}
