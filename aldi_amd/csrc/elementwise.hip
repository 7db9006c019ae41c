// Small bandwidth-bound kernels around the dense path: weight re-layouts, casts, stem,
// pooling, FPN glue, optimizer and EMA streams.
#include "common.h"

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

}  // namespace

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
