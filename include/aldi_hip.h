/* libaldi_hip.so -- C ABI of the MI355X (gfx950) kernels behind the ALDI student+teacher
 * training step.
 *
 * The reference (justinkay/aldi) has no FFI of its own: its hot path runs inside the
 * un-vendored detectron2 -> torch / torchvision / cuDNN / NCCL native code, reached from the
 * call sites cited on every entry point below (paths relative to the reference root).  This
 * header is the boundary a maintainer binds with ctypes: plain pointers
 * and sizes, no torch types.  All pointers are DEVICE pointers unless the name says host;
 * every function enqueues on `stream` and returns 0 on success or a negative code
 * (aldi_last_error() gives the text).  Nothing here synchronises the device.
 *
 * Layout conventions: activations NHWC; dtype ALDI_BF16 (raw uint16 bf16) or ALDI_F32;
 * conv weights [Cout][KH][KW][Cin] in the activation dtype; accumulators, losses, gradients
 * of parameters and master weights fp32.
 */
#ifndef ALDI_HIP_H
#define ALDI_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* aldi_stream_t; /* hipStream_t */

enum { ALDI_F32 = 0, ALDI_BF16 = 1 };
enum { ALDI_OK = 0, ALDI_ERR_HIP = -1, ALDI_ERR_ARG = -2 };

const char* aldi_last_error(void);
int aldi_version(void);

/* ---------------------------------------------------------------------------------------
 * Dense path: convolution / linear as implicit GEMM on MFMA.
 * Replaces: cuDNN/MIOpen conv2d + FrozenBN affine + ReLU + residual add and torch Linear
 * inside GeneralizedRCNN.forward, reached from aldi/align.py:72, aldi/distill.py:157,162,
 * aldi/pseudolabeler.py:21; discriminators aldi/align.py:103-135.
 * ------------------------------------------------------------------------------------- */
typedef struct {
    const void* x;      /* input  [N][H][W][Cin]                                          */
    const void* w;      /* weight [Cout][KH][KW][Cin]                                     */
    void* y;            /* output in `dtype` (nullable)                                   */
    float* y_f32;       /* output in fp32 (nullable), same indexing as y                  */
    const float* scale; /* per-Cout multiplier (folded FrozenBN), nullable = 1            */
    const float* shift; /* per-Cout addend (folded FrozenBN shift or bias), nullable = 0  */
    const void* res;    /* residual added before ReLU (nullable), dtype                   */
    const void* mask;   /* if set: out = mask[idx] > 0 ? out : 0 (ReLU backward), dtype   */
    int N, H, W, Cin;
    int Cout, KH, KW, stride, pad;
    int Ho, Wo;         /* conv output grid                                               */
    int relu;           /* apply ReLU after residual                                      */
    int res_mode;       /* 0 none, 1 same index as output, 2 nearest-upsample x2 (FPN top-down:
                           res is [N][Ho/2][Wo/2][Cout])                                   */
    int out_scale;      /* 1 = dense output [N][Ho][Wo][Cout]; s>1 = scatter output pixel
                           (ho,wo) to (ho*s, wo*s) of an [N][OH][OW][Cout] tensor (dgrad of
                           a stride-s 1x1 conv); res/mask use the same scattered index       */
    int OH, OW;         /* only for out_scale > 1                                          */
    int dtype;          /* ALDI_F32 or ALDI_BF16                                           */
} aldi_conv_args;

int aldi_conv_igemm(const aldi_conv_args* a, aldi_stream_t stream);

/* Weight gradient: dw[Cout][KH][KW][Cin] (fp32) += scale[co] * sum_pixels g[p][co] * x[pix(p,kh,kw)][ci].
 * Accumulates with float atomics (split-K over pixels and over micro-steps); zero dw once
 * per optimizer step.  Replaces cuDNN wgrad / Linear weight grad reached through autograd
 * from aldi/trainer.py:79. */
typedef struct {
    const void* x;      /* forward input  [N][H][W][Cin]                     */
    const void* g;      /* grad wrt the conv output, [N][Ho][Wo][Cout]       */
    float* dw;          /* fp32 gradient accumulator                         */
    const float* scale; /* per-Cout multiplier (FrozenBN fold), nullable     */
    int N, H, W, Cin, Cout, KH, KW, stride, pad, Ho, Wo;
    int dtype;
} aldi_wgrad_args;
int aldi_conv_wgrad(const aldi_wgrad_args* a, aldi_stream_t stream);

/* db[c] += sum_m g[m][c] (bias gradients), g is [M][C] in `dtype`. */
int aldi_bias_grad(const void* g, float* db, int M, int C, int dtype, aldi_stream_t stream);

/* Data-gradient weights: wt[Cin][KH][KW][Cout] = scale[co] * w[co][KH-1-kh][KW-1-kw][ci]
 * (rotated + transposed, FrozenBN scale folded), so that dgrad is aldi_conv_igemm on g with
 * pad' = KH-1-pad.  `w_master` is the fp32 master weight, output in `dtype`. */
int aldi_dgrad_weights(const float* w_master, const float* scale, void* wt, int Cout, int KH, int KW, int Cin,
                       int dtype, aldi_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* ALDI_HIP_H */
