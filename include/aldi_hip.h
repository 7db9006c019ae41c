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
/* an empty launch (one workgroup that returns): calibration of timing harnesses (bench.py subtracts what an event pair around it reads) */
int aldi_noop(aldi_stream_t stream);

/* Tuning knobs of the kernel dispatchers (test / experiment surface; the defaults are what the benchmark runs).
 * Each knob can also be preset from the environment as ALDI_<UPPER-CASE NAME>, read once at first use.
 *   igemm_force          0 heuristics | 1 128x128 | 2 128x64 | 3 64x64 | 4 256x128 | 5 128x16 tile of aldi_conv_igemm |
 *                        6 / 7 / 8: the 128x128 / 128x64 / 64x64 tile with 128-byte K slabs (plain 1x1 / linear layers) |
 *                        9 / 10 (3x3 halo form only): 240x128 on six waves, two workgroups per CU / 256x128 role-split |
 *                        16 / 17 (3x3 halo form, bf16): 64x64 tiles / 96x64 tiles on three waves;
 *                        11 / 13 / 15 (3x3 halo form, bf16, Cin % 64 == 0): the 256x256 / 128x128 / 256x256-on-four-waves tile with 128-byte K slabs (igemm_halo64.h);
 *                        14 (plain 1x1, bf16): the weight-stationary persistent kernel (igemm_ws.h)
 *   igemm_k64_min        plain 1x1 / linear layers with K >= this (and K % 64 == 0) take the 64x64 128-byte-slab form (1024)
 *   igemm_group          1 = aldi_conv_igemm_group shares one launch (0: always n single launches)
 *   igemm_splitk_tile    tile of a split-K launch (aldi_conv_args.ksplit): 0 = 128x128 (128-byte K slabs), 1 = 256x128 (64-byte slabs), 3 = 256x128
 *                        (128-byte slabs), 2 = 3 when that still gives about one workgroup per CU, else 0 (default), 4 = the same rule with 1
 *   igemm_narrow_k       bf16 layers with K up to this many channels x taps take 128x64 tiles instead of 128x128 (512; 0 = never)
 *   igemm_direct         bit mask of the 64-channel bf16 tiles that take the DIRECT epilogue (no LDS staging, permuted channel rows, residual
 *                        prefetched into registers, scale / shift / mask bits by 4-byte LDS-DMA, one rounding): 1 = 128x64 1x1 / tap tile,
 *                        2 = 64x64 long-K tile, 4 = 128x64 halo tile, 8 = the 256x256 128-byte-slab halo tile (scale / shift / ReLU outputs
 *                        only) (15; 0 = the staged epilogue everywhere).  Applies to bf16 outputs in the
 *                        plain layout with Cout % 8 == 0, Cout >= 64, no fp32 output / `mask` tensor / split-K / scatter
 *   igemm_halo64_mid     mid-size 3x3 layers with at least this many 128x128 tiles take the 128x128 tile with 128-byte K slabs (0 = never, the
 *                        default: measured 10-20 % slower than the 128x64 tiles on res3 / res4 conv2 -- those layers are bound by workgroup count)
 *   igemm_halo_small     3x3 layers with Cin >= 512 and at most this many 128x64 tiles take 64x64 halo tiles (0 = never, the default; 320 = res5 conv2: faster alone, 0.5 % slower in the step)
 *   match_wave           1 = the anchor matcher culls the GT list per wave (64 anchors) instead of per 1024-anchor workgroup (default; same labels bit for bit)
 *   igemm_halo96         1 = 3x3 layers with 200-600 tiles of 128x64 take 96x64 three-wave tiles (0 = never, the default: 2-6 % faster alone, neutral in the step)
 *   igemm_halo_ilv       1 = the 256x256 halo64 tile runs its interleaved K loop (reads / DMA pieces between the MFMAs; 0 = the lockstep loop of r05)
 *   igemm_ws             1 = plain 1x1 bf16 layers with K = Cin in {64, 128, 256, 512}, whole groups of 256 (K >= 256: 128) output channels and at least
 *                        igemm_ws_min (40000: res3 / p2-size maps of the student; measured equal or slower below) pixels run the weight-stationary persistent kernel (igemm_ws.h: weights in registers, pixel tiles streamed
 *                        through a 3-stage LDS ring, epilogue from the accumulators); igemm_ws_wgs (512) = its workgroup count; igemm_force 14 forces it
 *   igemm_lean           1 = plain 1x1 / linear layers with K % 64 == 0 on those tiles run the lean K loop (running DMA offsets)
 *   igemm_halo           1 = 3x3/stride-1/pad-1 bf16 convs use the halo form (one pixel slab per three taps)
 *   igemm_bigtile_min    128x128-tile count from which a 3x3 conv takes the big halo tile (1024)
 *   igemm_bigtile        which one: 64 = 256x256 with 128-byte K slabs where Cin % 64 == 0 and Cout % 256 == 0 (default; else 256x128), 65 = the same tile on four waves,
 *                        4 = 256x128 lockstep, 1 = 128x128, 10 = 256x128 with the two wave halves in alternating roles
 *   igemm_lintile_min, igemm_bigtile_k   tile count / K from which a 1x1 conv or linear takes the 256x128 tile (768, 768)
 *   igemm_tile           9 = never use the 256x128 tile for 1x1 / linear; 7 = that tile with 64-byte K slabs (default: 128-byte slabs where K % 64 == 0)
 *   igemm_xcd            1 = XCD-aware workgroup -> tile order
 *   igemm_dbg            ablation bits (4 skip epilogue, 8 one K slab, 16 L1-resident loads): results are WRONG when set
 *   ema_blocks           workgroups of the EMA tick's grid-stride loop (2048: it runs beside the student's stem, a chip-filling grid starves it)
 *   nms_mask_tri         1 = the NMS suppression mask from a triangular grid, one wave per 64x64 block, column boxes by v_readlane (0: square grid + LDS)
 *   rpn_topk_fused       1 = the RPN's exact top-k (radix passes, collect, sort + decode) as ONE launch with group barriers (0: five launches)
 *   wgrad_lean           1 = lean bf16 weight-gradient kernel for 1x1 and "same" KxK convs (0: generic gather kernel)
 *   wgrad_ilv            1 = those loops with their transpose reads / DMA pieces between the MFMAs (asm MFMAs tied in place); 0 (default: equal alone, 1.5 % slower in the step)
 *   wgrad_dma64          bit mask: 1 = the 256x256 bf16 tile, 2 = the grouped 128x128 tile run the LDS-DMA (full 128-byte lines) + transpose-read
 *                        loop (3; 0: the register-staged loops)
 *   wgrad_big_min        slabs (64 pixels) per workgroup from which the 256x256 tile is used (28; 0 = never)
 *   wgrad_big_slots, wgrad_slots   target workgroup counts of the 256x256 / 128x128 forms (256, 384)
 *   wgrad_xcd            1 = XCD-aware order
 *   wgrad_group_slots    > 0: minimum workgroup count of an aldi_conv_wgrad_group launch before it stops splitting pixel ranges;
 *                        0 (default): the pixel split of the group is chosen by a model of 256-workgroup rounds
 *   wgrad_group_epi      cost of one atomic epilogue in that model, in 32-pixel slab steps (24)
 *   wgrad_ordered        1 = ordered (atomic-free, deterministic) epilogue whenever the caller passes a workspace (default)
 *   wgrad_big_group      1 = a group's layers with Cout % 256 == K % 256 == 0 run as ONE launch of 256x256 tiles (default)
 *   wgrad_big_epi        cost of one 256x256 epilogue in the group's split model, in 32-pixel slab steps (12)
 *   wgrad_big_group_min  (256x256 tiles x pixels) / 4096 a group needs for that launch (64); less: its layers join the 128x128 group
 *   wgrad_db             1 = grouped weight gradients with two LDS images and one barrier per 64-pixel slab (64 KB, two workgroups per CU)
 *   roialign_sep         1 = aldi_roialign forward in the separable form (row / column weight tables, one workgroup per ROI); 0 = per sample
 *   roialign_bwd_rows    2 = aldi_roialign_backward on bf16 pooled gradients with two feature rows per workgroup; 1 = one row (same results)
 *   wgrad_dbg            ablation bits (1 = skip the atomic epilogue): results are WRONG when set
 *   wgrad_dma            LDS-DMA + ds_read_b64_tr_b16 weight-gradient kernel: 0 off, 1 in place of the lean kernel, 2 also of the 256x256
 *   wgrad_f32_tile128    1 = fp32 weight gradients with Cout, K >= 128 on the 128x128 f32-MFMA tile (0: the 64x64 kernel)
 *   igemm_halo_f32       > 0: fp32 3x3/stride-1/pad-1 convs with at least this many 128x64 tiles take the halo form too (0 = off, the
 *                        default: it sums K in another order than the tap form)
 *   igemm_f32_tile64_max fp32 layers with fewer 128x128-tile equivalents than this run on 64x64 tiles (4096; 0 = the bf16 rules)
 *   msda_bin             1 = aldi_ms_deform_attn_backward_self with a workspace bins the samples into per-tile lists (0: the walk form)
 *   msda_bin_list        expected list entries per tile the binned form sizes its tiles for (512)
 *   msda_gather          walk form (no workspace): bit mask of the target levels that are gathered (7); 0 = the general scatter
 *   msda_gather_list     walk form: list length the tile sizes aim at (1500)
 *   colsum_blocks, colsum_minrows, colsum_nt, colsum_block_kb   aldi_bias_grad launch geometry
 *   stem_mfma            1 = MFMA stem kernel in bf16 mode
 *   sab_blocks, ln_bwd_blocks, ln_bwd_blocks_narrow   ConvNeXt scale-and-bias / LayerNorm backward launch geometry
 * aldi_last_dispatch(): name of the kernel variant chosen by the most recent aldi_conv_igemm / aldi_conv_wgrad call on
 * this thread, e.g. "igemm<bf16,256,128,4,2,flat,halo>" or "wgrad_bf16_big splits=4" (valid until the next call). */
int aldi_set_tuning(const char* name, int value);
int aldi_get_tuning(const char* name, int* value);
int aldi_reset_tuning(void);
const char* aldi_last_dispatch(void);

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
    /* split-K for long-K linear layers with few output tiles (the box head's FC1: 2048 x 1024 x 12544 is 128 tiles of 128 x 128):
     * ksplit > 1 runs the K range in `ksplit` slices on ksplit times the workgroups, every slice writing its fp32 partial tile
     * to ws[slice][M][Cout] with plain stores, and a second launch sums the slices in slice order (deterministic, no atomics),
     * applies scale / shift / ReLU and writes y.  bf16, plain 1x1 / linear, K % (64 * ksplit) == 0, no res / mask / y_f32. */
    void* ws;           /* ksplit * M * Cout floats (only read / written when ksplit > 1)   */
    int ksplit;         /* 0 / 1 = off                                                       */
    /* ReLU masks as bits (bf16, Cout % 8 == 0, plain output layout): [M][Cout / 8] bytes, bit c % 8 of byte c / 8 = (y[m][c] > 0).
     * bits_out (nullable): the forward launch writes the mask of its own output beside y; mask_bits (nullable, alternative to `mask`):
     * the backward launch multiplies by it instead of reading the 16x larger activation -- the conv1 / lateral data-gradient launches of
     * res3..res5 are HBM-bound and a quarter of their bytes was that read. */
    const void* mask_bits;
    void* bits_out;
} aldi_conv_args;

int aldi_conv_igemm(const aldi_conv_args* a, aldi_stream_t stream);
/* n (<= 12) convolutions of ONE layer shape -- same channel counts, taps, stride and padding; different tensors, batch sizes and
 * H x W: the student's and the teacher's pass through a layer (aldi/distill.py:157,162 run them as two model calls), and one layer
 * applied to several pyramid levels (FPN output convs, the RPN conv on p2..p6) -- in ONE launch; any other combination (or
 * igemm_group = 0) falls back to n single launches.  Same arithmetic as n aldi_conv_igemm calls. */
int aldi_conv_igemm_group(const aldi_conv_args* args, int n, aldi_stream_t stream);

/* Weight gradient: dw[Cout][KH][KW][Cin] (fp32) += scale[co] * sum_pixels g[p][co] * x[pix(p,kh,kw)][ci].
 * Accumulates into dw (split-K over pixels -- ordered through `ws`, or float atomics without it -- and over micro-steps); zero dw once
 * per optimizer step.  Replaces cuDNN wgrad / Linear weight grad reached through autograd
 * from aldi/trainer.py:79. */
typedef struct {
    const void* x;      /* forward input  [N][H][W][Cin]                     */
    const void* g;      /* grad wrt the conv output, [N][Ho][Wo][Cout]       */
    float* dw;          /* fp32 gradient accumulator                         */
    const float* scale; /* per-Cout multiplier (FrozenBN fold), nullable     */
    int N, H, W, Cin, Cout, KH, KW, stride, pad, Ho, Wo;
    int dtype;
    float* db;          /* nullable: db[co] += sum over pixels of g[p][co] (the layer's bias gradient) in the same call: the lean /
                           256x256 bf16 kernels add it as one more MFMA column (a constant ones fragment) instead of re-reading g in
                           aldi_bias_grad; the other kernels fall back to that launch */
    void* ws;           /* nullable workspace of the ORDERED epilogue (bf16 lean / 256x256 kernels): pixel splits write their partial
                           tiles here with plain stores and a second launch adds them to dw / db in split order (no float atomics, the
                           same bits on every run); a tile with one pixel range adds to dw with plain loads and stores.  NULL (or knob
                           wgrad_ordered = 0): float-atomic epilogue.  A group call reads ws / ws_bytes of args[0] only. */
    long ws_bytes;      /* >= aldi_conv_wgrad_group_workspace(args, n) */
} aldi_wgrad_args;
int aldi_conv_wgrad(const aldi_wgrad_args* a, aldi_stream_t stream);

/* The weight gradients of n layers in ONE launch (same arithmetic as n aldi_conv_wgrad calls).  Nothing reads a weight
 * gradient before the optimizer, so a stage's layers -- small GEMMs over the same pixels -- are launched together: hundreds
 * of output tiles fill the chip without the 11-24-way pixel splits (and their float-atomic epilogues) each layer needs alone.
 * Problems the lean bf16 kernel cannot take (fp32, strided, unpadded KxK) are forwarded to aldi_conv_wgrad one by one.
 * Knobs wgrad_group_slots / wgrad_group_epi: how far the group's pixel ranges are split (see the knob table above). */
int aldi_conv_wgrad_group(const aldi_wgrad_args* args, int n, aldi_stream_t stream);
/* Bytes of workspace the ordered epilogue of this call needs under the current knobs (n = 1: of aldi_conv_wgrad); < 0: bad arguments.
 * Layers whose Cout and K are multiples of 256 (knob wgrad_big_group, default 1) are launched together as 256x256 tiles, about one
 * workgroup per CU (knob wgrad_big_epi: cost of one epilogue in 32-pixel steps in the split model), the others as 128x128 tiles. */
long aldi_conv_wgrad_group_workspace(const aldi_wgrad_args* args, int n);

/* db[c] += sum_m g[m][c] (bias gradients), g is [M][C] in `dtype`. */
int aldi_bias_grad(const void* g, float* db, int M, int C, int dtype, aldi_stream_t stream);

/* Data-gradient weights: wt[Cin][KH][KW][Cout] = scale[co] * w[co][KH-1-kh][KW-1-kw][ci]
 * (rotated + transposed, FrozenBN scale folded), so that dgrad is aldi_conv_igemm on g with
 * pad' = KH-1-pad.  `w_master` is the fp32 master weight, output in `dtype`. */
int aldi_dgrad_weights(const float* w_master, const float* scale, void* wt, int Cout, int KH, int KW, int Cin,
                       int dtype, aldi_stream_t stream);

/* Batched form: one launch re-derives the data-gradient weights of every layer after an optimizer step.  `items` is a
 * DEVICE array of n_items descriptors; layer i owns the tiles [tile_begin_i, tile_begin_{i+1}) of the launch, one tile =
 * 32 output channels x 32 input channels of one tap: KH*KW * ceil(Cout/32) * ceil(Cin/32) tiles per layer
 * (tile_begin ascending, total_tiles = end of the last layer). */
typedef struct aldi_dgw_item {
    const float* w_master;   /* [Cout][KH][KW][Cin] fp32 */
    const float* scale;      /* [Cout] or NULL */
    void* wt;                /* [Cin][KH][KW][Cout] in `dtype` */
    int Cout, KH, KW, Cin;
    int tile_begin, reserved;
} aldi_dgw_item;
int aldi_dgrad_weights_batch(const aldi_dgw_item* items, int n_items, int total_tiles, int dtype, aldi_stream_t stream);

/* A whole ResNet bottleneck (1x1 reduce -> 3x3 -> 1x1 expand, FrozenBN folded, ReLUs, residual add) in ONE kernel, bf16, for
 * blocks that keep no activations for a backward pass -- res2 of the student (BACKBONE.FREEZE_AT = 2) and of the EMA teacher:
 * the two 64-channel intermediate maps stay in the LDS (csrc/bneck.hip).  Replaces detectron2 BottleneckBlock.forward for res2
 * (three aldi_conv_igemm launches per block, 2048 -> 1024 B of HBM traffic per pixel).
 *   y = relu( w3 . relu( w2 (*) relu( w1 . x + b1 ) + b2 ) + b3 + res ),   res = x for an identity block (Cin == Cout), the shortcut
 *   conv's output for the first block of the stage.  w1 [mid][Cin], w2 [mid][3][3][mid], w3 [Cout][mid] are bf16 WITH the FrozenBN
 *   scales folded in (aldi_fold_weights_batch), b1 / b2 / b3 the folded shifts (fp32).  Built for Cin in {64, 256}, mid 64, Cout 256. */
typedef struct aldi_bottleneck_args {
    const void* x;            /* [N][H][W][Cin] bf16 */
    const void* res;          /* [N][H][W][Cout] bf16 */
    void* y;                  /* [N][H][W][Cout] bf16 */
    const void* w1; const void* w2; const void* w3;
    const float* b1; const float* b2; const float* b3;
    int N, H, W, Cin, mid, Cout;
} aldi_bottleneck_args;
int aldi_bottleneck_fused(const aldi_bottleneck_args* a, aldi_stream_t stream);
/* out[r][c] = bf16(w[r][c] * scale[r]) for a DEVICE table of fp32 matrices (rows * cols a multiple of 8, 16-byte aligned);
 * matrix i owns the 8-element chunks [chunk_begin_i, chunk_begin_{i+1}) of the launch. */
typedef struct aldi_fold_item {
    const float* w; const float* scale; void* out;
    int rows, cols, chunk_begin, reserved;
} aldi_fold_item;
int aldi_fold_weights_batch(const aldi_fold_item* items, int n_items, int total_chunks, aldi_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Stem and glue (bandwidth-bound).
 * ------------------------------------------------------------------------------------- */
#define ALDI_MAX_IMAGES 16

/* GeneralizedRCNN.preprocess_image + BasicStem conv1/FrozenBN/ReLU (detectron2; call sites
 * aldi/align.py:72, aldi/pseudolabeler.py:21): (uint8 - mean)/std, zero pad AFTER
 * normalisation, conv 7x7 s2 p3 in fp32, y = relu(conv*scale+shift) -> [N][Hc][Wc][64]. */
typedef struct {
    const uint8_t* img;   /* [N][3][Hs][Ws] staging; image n is the top-left h[n] x w_img[n] */
    const float* w;       /* [64][7][7][3] fp32                                             */
    const float* scale; const float* shift;
    void* y;
    int N, Hs, Ws, Hc, Wc;
    int h[ALDI_MAX_IMAGES], w_img[ALDI_MAX_IMAGES];
    float mean[3], std[3];
    int dtype;
} aldi_stem_args;
int aldi_stem_forward(const aldi_stem_args* a, aldi_stream_t stream);
/* Stem + max_pool2d(k3,s2,p1) in one kernel (bf16): y_pool [N][(Hc-1)/2+1][(Wc-1)/2+1][64] == aldi_maxpool3s2(aldi_stem_forward(a))
 * bit for bit, without the 64-channel half-resolution map ever reaching HBM.  w_packed = aldi_stem_pack_weights(a->w): the fp32
 * [64][7][7][3] kernel re-laid as bf16 [64][200] in the MFMA reduction order (once per weight refresh).  a->y is not used. */
int aldi_stem_pack_weights(const float* w, void* w_packed, aldi_stream_t stream);
int aldi_stem_pool_forward(const aldi_stem_args* a, const void* w_packed, void* y_pool, aldi_stream_t stream);

/* max_pool2d(k3,s2,p1) NHWC; y is [N][(H-1)/2+1][(W-1)/2+1][C]. */
int aldi_maxpool3s2(const void* x, void* y, int N, int H, int W, int C, int dtype, aldi_stream_t stream);
/* Input staging (ImageList.from_tensors): n <= 16 uint8 CHW device images of sizes heights[i] x widths[i] (host arrays) -> rows
 * [i][c][0..h)[0..w) of the padded uint8 batch [n][C][Hs][Ws] in one launch; the buffer's padding is left as it is. */
int aldi_stage_images(const void* const* images, const int* heights, const int* widths, int n, int C, int Hs, int Ws, void* batch,
                      aldi_stream_t stream);
/* LastLevelMaxPool (k1,s2): forward y = x[:, ::2, ::2]; backward (x = grad small, y = grad big) y[::2,::2] += x. */
int aldi_subsample2(const void* x, void* y, int N, int H, int W, int C, int backward, int dtype, aldi_stream_t stream);
/* backward of FPN nearest-upsample-x2 + add: out[N][Hc][Wc][C] (=|+=) 2x2 block sums of g[N][2Hc][2Wc][C]. */
int aldi_upsample2_bwd(const void* g, void* out, int N, int Hc, int Wc, int C, int accumulate, int dtype, aldi_stream_t stream);
/* three accumulating calls of the above in one launch (the FPN's top-down backward, detectron2 FPN.forward's `prev_features = lateral + top_down`
 * reached from aldi/trainer.py:87): o3 += blocks(g2); o4 += blocks(o3); o5 += blocks(o4).  g2 [N][8 H5][8 W5][C] ... o5 [N][H5][W5][C]; same bits. */
int aldi_upsample2_bwd_chain(const void* g2, void* o3, void* o4, void* o5, int N, int H5, int W5, int C, int dtype, aldi_stream_t stream);
/* out = a + b (b fp32), optionally masked by relu_src > 0; a, relu_src nullable. */
int aldi_add_f32(const void* a, const float* b, const void* relu_src, void* out, long n, int dtype, aldi_stream_t stream);
int aldi_cast_from_f32(const float* src, void* dst, long n, int dtype, aldi_stream_t stream);

/* torch.optim.SGD step over flat fp32 buffers (detectron2 build_optimizer via aldi/trainer.py:199-208,
 * applied at aldi/dropin.py:121); refreshes the compute-dtype copy when dtype is bf16. */
int aldi_sgd_step(float* p, const float* g, float* buf, void* p_compute, long n, float lr, float momentum, float weight_decay,
                  float grad_scale, int first_step, int dtype, aldi_stream_t stream);
/* The same step (never the first one: the momentum buffer exists) over a RANGE of the flat buffers, with lr, momentum, weight decay and
 * the gradient scale read from device memory (hyper[0..3]): a launch that can be recorded in a hipGraph and replayed while the
 * learning-rate schedule moves.  The fused step issues one per layer group as soon as the group's weight gradients are complete,
 * beside the rest of the backward, instead of one pass over all parameters after it. */
int aldi_sgd_step_dev(float* p, const float* g, float* buf, void* p_compute, long n, const float* hyper, int dtype, aldi_stream_t stream);
/* EMA teacher update over the flat state (aldi/ema.py:32-57): t = s*(1-alpha) + t*alpha, or t = s.  teacher_compute (nullable,
 * dtype bf16): the compute copy of the first n_compute elements, written in the same pass. */
int aldi_ema_update(float* teacher, const float* student, void* teacher_compute, long n, long n_compute, double alpha, int copy_only, int dtype,
                    aldi_stream_t stream);
/* FrozenBatchNorm2d fold (detectron2): scale = w*rsqrt(var+1e-5), shift = b - mean*scale. */
int aldi_bn_fold(const float* w, const float* b, const float* mean, const float* var, float* scale, float* shift, int C, aldi_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * RPN / matcher / sampler / proposals (integer + box arithmetic; bit-exact index results).
 * Replaces detectron2 Matcher, subsample_labels, RPN.label_and_sample_anchors, RPN.losses,
 * RPN.predict_proposals / find_top_rpn_proposals and torchvision batched_nms; reference call
 * sites aldi/distill.py:157,162,200-202, aldi/pseudolabeler.py:21.
 * ------------------------------------------------------------------------------------- */
#define ALDI_MAX_LEVELS 5
typedef struct {
    int num_levels;
    int A;                        /* anchors per cell                                        */
    int C;                        /* channels of a head output row: [0,A) objectness logits,
                                     [A, 5A) deltas (a*4+d), rest padding                    */
    int H[ALDI_MAX_LEVELS], W[ALDI_MAX_LEVELS];
    int off[ALDI_MAX_LEVELS + 1]; /* anchor offset of each level; off[num_levels] = sum A   */
} aldi_rpn_geom;

/* Matcher: boxes [*(N)][L][4] (box_stride_n = 0 when shared by all images, else L), optional
 * per-image box_count; gt [N][Gmax][4] + gt_count[N] (device).  labels: iou<lo -> 0,
 * lo<=iou<hi -> -1, >=hi -> 1, low-quality matches -> 1; padding slots -> -2.
 * best_idx = argmax GT (first max).  gt_best_scratch: gt_best_bytes >= N * Gmax * 4 ([N][Gmax] uint32: the per-GT best IoU bits); with
 * N * Gmax * 128 bytes every GT's word gets its own 128-byte line ([N][Gmax][32]: the anchor matcher's same-line atomics no longer serialise). */
int aldi_box_match(const float* boxes, long box_stride_n, const int* box_count, int L,
                   const float* gt_boxes, const int* gt_count, int Gmax, int N,
                   float lo, float hi, int allow_low_quality,
                   float* best_iou, int* best_idx, unsigned* gt_best_scratch, size_t gt_best_bytes, int* labels, aldi_stream_t stream);
/* subsample_labels, step 1: ordered index lists. lists [N][2][L] (0: positives = not -1/-2/bg,
 * 1: negatives = bg), counts [N][2].  workspace: aldi_compact_labels_workspace(L, N) bytes (per-segment counts). */
size_t aldi_compact_labels_workspace(int L, int N);
int aldi_compact_labels(const int* labels, int L, int N, int bg_label, int* lists, int* counts, void* workspace, aldi_stream_t stream);
/* subsample_labels, step 2 (RPN): labels.fill(-1); labels[lists[0][sel[0]]] = 1; labels[lists[1][sel[1]]] = 0.
 * sel [N][2][S] are positions drawn by the host RNG (torch.randperm order), nsel [N][2]. */
int aldi_rpn_apply_sample(int* labels, int L, int N, const int* lists, const int* sel, const int* nsel, int S, aldi_stream_t stream);
/* RPN objectness BCE (sum/norm) + L1 box loss (sum/norm), and d(scale_cls*loss_cls + scale_loc*loss_loc)/d(head)
 * added into grad[level] (fp32, same layout as head; nullable).  head[level]: fp32 [N][H][W][C]. loss2 += */
int aldi_rpn_loss(const aldi_rpn_geom* gm, float* const* head, float* const* grad, const float* anchors, const int* labels, const int* matched,
                  const float* gt_boxes, const int* gt_count, int Gmax, int N, float inv_norm, float grad_scale_cls, float grad_scale_loc,
                  float* loss2, aldi_stream_t stream);
size_t aldi_rpn_proposals_workspace(int N, int num_levels);
/* per level top-k -> decode -> clip -> drop empty -> batched NMS -> post-NMS top-k.
 * out_boxes [N][post][4], out_scores [N][post], out_count [N]; err_flag |= 1 on non-finite. */
int aldi_rpn_proposals(const aldi_rpn_geom* gm, float* const* head, const float* anchors, const int* img_hw, int N,
                       int pre_nms_topk, int post_nms_topk, float nms_thresh, void* workspace,
                       float* out_boxes, float* out_scores, int* out_count, int* err_flag, aldi_stream_t stream);

/* Sparse backward of the RPN head (what autograd runs as dense convolutions over all five levels, reached from
 * aldi/trainer.py:79): d(loss)/d(head outputs) is non-zero only at the sampled anchors' pixels (<= 256 per image and sample).
 *   aldi_rpn_active_pixels   idx[0..*count) = global row (level-major pixel position, row = N*sum_{k<l}H_k W_k + (n*H_l+h)*W_l+w)
 *                            of every pixel whose C head-gradient channels are not all zero, in ASCENDING row order (two-launch
 *                            ordered compaction: the list, and with it every sum downstream, is run-to-run reproducible);
 *                            *count (device) = number of active pixels; err_flag |= 2 when more than `cap` are active (the
 *                            excess is dropped).  workspace: aldi_rpn_active_pixels_workspace(gm, N) bytes.
 *   aldi_rpn_sparse_gather   rows s < min(*count, cap): G[s][C] = grad_head (in dtype), Tm[s][Cf] = hidden[l][pixel],
 *                            X9[s][tap][Cf] = feat[l][pixel + (tap/3-1, tap%3-1)] (zero outside the image); rows >= count zero.
 *   aldi_rpn_sparse_scatter  gfeat[l][pixel + (tap/3-1, tap%3-1)][ci] += Y[s][tap][ci], Y in dtype [cap][9][Cf]; gfeat maps in grad_dtype.
 *                            No atomics: every target pixel is finished by one workgroup, which adds its (<= 9) contributions in tap
 *                            order in fp32 and updates the map once (bf16 maps: previous content + sum, rounded once); needs the
 *                            sorted list of aldi_rpn_active_pixels. */
size_t aldi_rpn_active_pixels_workspace(const aldi_rpn_geom* gm, int N);
int aldi_rpn_active_pixels(const aldi_rpn_geom* gm, float* const* grad_head, int N, int cap, int* idx, int* count, void* workspace,
                           int* err_flag, aldi_stream_t stream);
int aldi_rpn_sparse_gather(const aldi_rpn_geom* gm, float* const* grad_head, const void* const* hidden, const void* const* feat, int N, int Cf,
                           int cap, const int* idx, const int* count, void* G, void* Tm, void* X9, int dtype, aldi_stream_t stream);
int aldi_rpn_sparse_scatter(const aldi_rpn_geom* gm, void* const* gfeat, const void* Y, int N, int Cf, int cap, const int* idx,
                            const int* count, int dtype, int grad_dtype, aldi_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * ROI heads.  Replaces detectron2 StandardROIHeads.label_and_sample_proposals, ROIPooler +
 * torchvision roi_align(aligned, 7x7, adaptive sampling), FastRCNNOutputLayers.losses and
 * .inference (fast_rcnn_inference), plus the reference's pseudo-label threshold filter
 * aldi/pseudolabeler.py:51-67.  Call sites: aldi/distill.py:157,162; aldi/pseudolabeler.py:21.
 * ------------------------------------------------------------------------------------- */
typedef struct {
    const void* feat[4];  /* p2..p5, [N][H][W][C] in dtype                 */
    float* grad[4];       /* fp32 gradient accumulators (backward only)    */
    int H[4], W[4];
    float scale[4];       /* 1/4 .. 1/32                                   */
    int C;                /* 256                                           */
} aldi_roi_feats;

/* proposals [N][P][4]+pcount, GT -> cand [N][L=P+Gmax][4] (GT appended), ccount, matcher (iou>=thr fg),
 * cls [N][L]: matched gt class for fg, K for bg, -2 padding. */
int aldi_roi_prepare(const float* props, const int* pcount, int P, const float* gt_boxes, const int* gt_classes, const int* gt_count,
                     int Gmax, int N, int K, float iou_thresh, float* cand, int* ccount, float* best_iou, int* best_idx,
                     unsigned* gt_best_scratch, int* labels, int* cls, aldi_stream_t stream);
/* The same, AND the ordered foreground / background lists of aldi_compact_labels(cls, bg = K) with their lengths, in ONE launch (Gmax <= 256;
 * the ROI heads' matcher: one threshold, no low-quality rule).  lists [N][2][L], counts [N][2], L = P + Gmax.  tickets: N uint32 words that
 * are ZERO before the first call (the kernel leaves them zero).  tail_a / tail_b (nullable): two device words copied to counts[2N] and
 * counts[2N + 1] (counts then has 2N + 2 words: the list lengths and e.g. two error words leave in ONE device -> host copy).  Replaces label_and_sample_proposals' matching part (detectron2
 * StandardROIHeads, reached from aldi/trainer.py:87, aldi/distill.py:157): identical results to aldi_roi_prepare + aldi_compact_labels. */
int aldi_roi_prepare_lists(const float* props, const int* pcount, int P, const float* gt_boxes, const int* gt_classes, const int* gt_count,
                           int Gmax, int N, int K, float iou_thresh, float* cand, int* ccount, float* best_iou, int* best_idx, int* labels,
                           int* cls, int* lists, int* counts, unsigned* tickets, const int* tail_a, const int* tail_b, aldi_stream_t stream);
/* sampled rows = cat(fg_list[sel_fg], bg_list[sel_bg]) per image, packed from row_off[n]:
 * rois [R][5] (batch, x1,y1,x2,y2), r_cls [R], r_gt [R][4], r_idx [R] (index into cand). */
int aldi_roi_gather(const float* cand, const int* cls, const int* best_idx, int L, const int* lists, const int* sel, const int* nsel, int S,
                    const int* row_off, const float* gt_boxes, const int* gt_count, int Gmax, int N,
                    float* rois, int* r_cls, float* r_gt, int* r_idx, aldi_stream_t stream);
/* inference: rois [N*P][5] from all proposals (padding rows get batch = -1). */
int aldi_rois_from_proposals(const float* props, const int* pcount, int P, int N, float* rois, aldi_stream_t stream);
/* ROIAlign over 4 FPN levels (level from box size). forward: pooled [R][P][P][C] written;
 * backward: pooled holds the gradient, scattered with float atomics into feats->grad. */
int aldi_roialign(const aldi_roi_feats* f, const float* rois, int R, int P, void* pooled, int backward, int dtype, aldi_stream_t stream);
/* ROIAlign backward as a gather: OVERWRITES the four fp32 gradient maps f->grad[l] ([N][H_l][W_l][C], no zero-fill needed) with
 * d(loss)/d(feature) given g_pooled [R][P][P][C] in `dtype`; every element is written exactly once (deterministic, no atomics).
 * grad_dtype = ALDI_BF16 (with dtype = ALDI_BF16): f->grad[l] point to bf16 maps, each element rounded once from the fp32 sum.
 * `aldi_roialign(..., backward=1)` is the accumulating scatter form of the same operator.  rois_sorted = 1 promises that the ROI
 * rows are grouped by ascending image index (what label_and_sample_proposals produces): each workgroup then scans its image's rows only. */
int aldi_roialign_backward(const aldi_roi_feats* f, const float* rois, int R, int P, const void* g_pooled, int N, int rois_sorted, int dtype,
                           int grad_dtype, aldi_stream_t stream);

/* pred fp32 [R][Cp]: [0,K] class logits, then 4K class-specific deltas. loss2 += {CE mean, L1 sum/R};
 * grad (fp32 [R][Cp], nullable) += d(scale_cls*loss_cls + scale_box*loss_box_reg)/d(pred). */
int aldi_box_loss(const float* pred, int Cp, int K, int R, const float* rois, const int* cls, const float* gt_boxes,
                  const float* weights4, float grad_scale_cls, float grad_scale_box, float* grad, float* loss2, aldi_stream_t stream);
/* The box head's losses of every chunk of a fused step in ONE launch: aldi_box_loss per chunk (rows r0 .. r1 of pred / rois / cls / gt_boxes,
 * normalised by the chunk's row count, its own scales and loss2 slot), aldi_roih_distill_loss for the chunks with teacher_pred (the chunk's
 * teacher rows, row r0 first; separate gradient scales for the classification and the regression part), and grad_lo (nullable): the bf16
 * copy of the finished gradient rows [R][Cp].  grad: fp32 [R][Cp], zeroed by the caller.  The same values as the separate launches. */
typedef struct {
    int r0, r1;
    float grad_scale_cls, grad_scale_box;
    float* loss_box;                 /* [2] += {CE mean, L1 sum / rows} */
    const float* teacher_pred;       /* NULL: no distillation for this chunk */
    float cls_temperature;
    int kl, do_cls, do_reg;
    float grad_scale_distill_cls, grad_scale_distill_reg;
    float* loss_distill;             /* [2] += {loss_cls_ce, loss_roih_l1} */
} aldi_box_loss_chunk;
int aldi_box_losses_fused(const float* pred, int Cp, int K, const float* rois, const int* cls, const float* gt_boxes, const float* weights4,
                          const aldi_box_loss_chunk* chunks, int nchunks, float* grad, void* grad_lo, aldi_stream_t stream);
size_t aldi_detections_workspace(int N);
/* fast_rcnn_inference + pseudo-label filter. pred fp32 [N*P][Cp] for all proposals.
 * det_* [N][topk], pl_* [N][pl_rows >= topk] (detections with score > pl_thresh, order kept, the rest of each row cleared: the
 * rows can be the ground-truth slots the matcher reads), counts [N]. */
int aldi_detections(const float* pred, int Cp, int K, const float* props, const int* pcount, int P, int N, const int* img_hw,
                    const float* weights4, float score_thresh, float nms_thresh, int topk, float pl_thresh, void* workspace,
                    float* det_boxes, float* det_scores, int* det_cls, int* det_count,
                    float* pl_boxes, int* pl_cls, float* pl_scores, int* pl_count, int pl_rows, int* err_flag, aldi_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * ALDI-owned losses (forward + backward fused).
 * ------------------------------------------------------------------------------------- */
/* ALDIDistiller.get_rpn_losses, aldi/distill.py:193-229, including the reference's index-order
 * behaviour: mask position q of the (N, sumA) label tensor selects position q of
 * cat([flatten(raw_level)]) with raw_level = (N, A|4A, H, W).  heads are fp32 [N][H][W][C].
 * n_valid / n_fg: number of labels >= 0 / == 1 (known to the host that drew the sample); when n_valid_fg_dev is set the two
 * counts are read from that DEVICE int[2] instead (a launch recorded in a hipGraph must not bake in per-step values).
 * loss2 += {loss_obj_bce, loss_rpn_l1}; grad[level] += d(loss*grad_scale)/d(student head). */
int aldi_rpn_distill_loss(const aldi_rpn_geom* gm, float* const* student_head, float* const* teacher_head, float* const* grad,
                          const int* labels, int N, float obj_temperature, int n_valid, int n_fg, const int* n_valid_fg_dev,
                          int do_obj, int do_reg, float grad_scale, float* loss2, aldi_stream_t stream);
/* ALDIDistiller.get_roih_losses, aldi/distill.py:231-278 (kl = 0: soft CE, 1: KL batchmean).
 * pred rows fp32 [R][Cp]: [0,K] logits then 4K deltas. loss2 += {loss_cls_ce, loss_roih_l1}. */
int aldi_roih_distill_loss(const float* student_pred, const float* teacher_pred, int Cp, int K, int R, float cls_temperature,
                           int kl, int do_cls, int do_reg, float grad_scale, float* grad, float* loss2, aldi_stream_t stream);
/* AlignMixin domain loss, aldi/align.py:83-84,89-90: weight * BCEWithLogits(pred[:,0], label).mean();
 * grad [R][ld] (dtype) = d(loss*grad_scale)/d(pred) (written, other columns zero). */
int aldi_domain_bce(const float* pred, int ld, int R, float label, float weight, float grad_scale, void* grad, float* loss,
                    int dtype, aldi_stream_t stream);
/* ConvDiscriminator's AdaptiveAvgPool2d(1) (aldi/align.py:114): x [N][HW][C] -> y [N][C]; and the
 * backward through ReLU + pool: gx = act > 0 ? gy / HW : 0. */
size_t aldi_avgpool_workspace(int N, int C);      /* bytes of fp32 partial sums aldi_avgpool may use (nullable workspace: one workgroup per 64 channels) */
int aldi_avgpool(const void* x, void* y, int N, int HW, int C, int dtype, void* workspace, aldi_stream_t stream);
int aldi_avgpool_bwd(const void* gy, const void* act, void* gx, int N, int HW, int C, int dtype, aldi_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Host helper (no device work): out[0..k) = torch.randperm(n)[:k] on the CPU generator whose state blob
 * (torch.get_rng_state(), 5056 bytes) is passed in and advanced exactly as torch.randperm(n) would.
 * Replaces the torch.randperm draws of detectron2 subsample_labels (aldi/distill.py:200-202 and inside
 * model(...)) at O(k + n/624) instead of O(n) divisions. */
int aldi_torch_randperm_prefix(unsigned char* state, long n, long k, long* out);
/* A whole iteration's sampling draws in one host call: script = nops rows {kind, a, b, out_off} (longs): kind 0 = the first
 * min(b, a) entries of torch.randperm(a) -> out[out_off ...] (int32; out_off < 0: discarded), kind 1 = torch.manual_seed(a).
 * Segments between seeds are independent and run on `threads` host threads; the state blob ends as torch would leave it. */
int aldi_torch_rng_script(unsigned char* state, const long* script, int nops, int* out, int threads);
/* Pre-generates on background threads the Mersenne streams the NEXT aldi_torch_rng_script call will consume: the one that
 * continues `state` (NULL: none) and one per torch.manual_seed value of `seeds`, max_draws long each.  Issued while the
 * device runs the phase whose results size the draws; the script then indexes into the streams instead of skipping through
 * ~430 state refills per 268k-entry list.  Streams that do not match the script's engine state / seeds are ignored. */
/* The host phase of one fused iteration in one call: device list lengths (counts: [N][2] RPN positives / negatives, then [N][2]
 * ROI) -> sampling positions, their counts, ROI row offsets and the distillation normalisers in the pinned upload buffer `words`
 * (int32 word offsets word0[8] = rsel, rnsel, osel, onsel, row_off, dsel, dnsel, nvf), in the reference's draw order, micro-step by
 * micro-step (aldi/trainer.py:51-52,86-89; aldi/distill.py:148-162,200-202; aldi/helpers.py:17-26).  chunks: nch rows {kind (1 =
 * distillation), n0, n1}, any number of distillation chunks; seeds[nseeds]: the ManualSeed hook's seed before the iteration, then
 * after each distillation chunk's reset_seed; dsel / dnsel rows and the nvf pairs follow the distillation chunks in order.
 * rows_out [N]: sampled ROI rows per image. */
int aldi_step_draws(unsigned char* state, const int* counts, int N, const int* chunks, int nch, const long* seeds, int nseeds,
                    int rpn_batch, int rpn_pos_cap, int roi_batch, int roi_pos_cap, int* words, const int* word0, int* rows_out, int threads);
int aldi_torch_rng_prefetch(const unsigned char* state, const long* seeds, int nseeds, long max_draws);
int aldi_torch_rng_prefetch_hits(void);       /* script segments served from a pre-generated stream so far */
int aldi_torch_rng_prefetch_wait(void);       /* blocks until the background fillers of aldi_torch_rng_prefetch are done (the script would wait itself) */

/* ---------------------------------------------------------------------------------------
 * Strong augmentation on the device (the step next to the hot path).  Images are HWC uint8 in HBM;
 * the random parameters are drawn on the host in the reference's order (aldi_amd/aug.py) and passed in.  Bit-identical to
 * the reference's numpy / scipy arithmetic.
 * --------------------------------------------------------------------------------------- */
/* *sum = exact integer sum of n bytes (RandomContrast blends with image.mean(); detectron2 transforms reached from
 * aldi/aug.py:47-51). */
int aldi_aug_sum_u8(const unsigned char* img, long n, unsigned long long* sum, aldi_stream_t stream);
/* BlendTransform in place: mode 0 contrast (needs *sum of this image), 1 brightness, 2 saturation / grayscale (w = 0);
 * `w` is the drawn weight (dst_weight; src_weight = 1 - w).  aldi/aug.py:47-53. */
int aldi_aug_blend(unsigned char* img, int H, int W, int mode, double w, const unsigned long long* sum, aldi_stream_t stream);
/* RandomBlurTransform.apply_image (aldi/aug.py:85-91): scipy gaussian_filter(sigma) over the three axes of the HWC float32
 * image; `weights` = the 2*radius+1 normalised taps in double (DEVICE pointer), tmp0/tmp1 = H*W*3 floats of scratch. */
int aldi_aug_blur(const unsigned char* img, unsigned char* out, float* tmp0, float* tmp1, int H, int W, const double* weights, int radius,
                  aldi_stream_t stream);
/* RandomEraseTransform (value="random", aldi/aug.py:113-131): img[h0:h0+h, w0:w0+w, :] = clip(fill * 255); fill = the
 * np.random.rand(h, w, 3) draw as float32, [h][w][3]. */
int aldi_aug_erase(unsigned char* img, int H, int W, int h0, int w0, int h, int w, const float* fill, aldi_stream_t stream);
/* MICTransform (aldi/aug.py:154-171): zero every pixel whose block (nearest-resized mh x mw mask, 1 = keep) is masked. */
int aldi_aug_mic(unsigned char* img, int H, int W, const unsigned char* mask, int mh, int mw, aldi_stream_t stream);
/* HWC uint8 -> CHW uint8 (the layout `dataset_dict["image"]` has in the reference, aldi/dataloader.py). */
int aldi_aug_hwc_to_chw(const unsigned char* in, unsigned char* out, int H, int W, aldi_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * ViTDet trunk (BASELINE cfg 4).  Replaces, for the ALDI step, the torch modules that
 * aldi/backbone.py:21-43 (checkpointed_vit_forward) drives: detectron2 modeling/backbone/vit.py Block / Attention /
 * PatchEmbed and backbone/utils.py window_partition / get_rel_pos / get_abs_pos (detectron2 is not vendored in the
 * reference; transformers' VitDet* modules implement the same algorithm and pin the tests).  Linear layers run on
 * aldi_conv_igemm / aldi_conv_wgrad with H = W = 1.
 * ------------------------------------------------------------------------------------------------------------------ */

/* LayerNorm over the last dim (C % 4 == 0, C <= 2048), rows of `dtype`, fp32 affine and statistics.
 * map (nullable): output row r reads source row map[r]; map[r] < 0 writes a zero row (window padding).
 * relu != 0 applies ReLU to the output (detectron2 Conv2d(norm=LN, activation=ReLU) of the ViTDet box head). */
int aldi_layernorm_forward(const void* x, const int* map, const float* gamma, const float* beta, void* y, float* mean,
                           float* rstd, int rows, int C, float eps, int relu, int dtype, aldi_stream_t stream);
/* g is indexed like y; dx (and the optional residual gradient `res`) like x.  mask (nullable, indexed like g): g counts only
 * where mask > 0 (pass the forward output when relu was set).  dgamma / dbeta accumulate (fp32 atomics). */
int aldi_layernorm_backward(const void* g, const void* x, const int* map, const float* gamma, const float* mean, const float* rstd,
                            const void* res, const void* mask, void* dx, float* dgamma, float* dbeta, int rows, int C, int dtype,
                            aldi_stream_t stream);
/* exact (erf) GELU: g == NULL -> out = gelu(x); else out = g * gelu'(x) */
int aldi_gelu(const void* x, const void* g, void* out, long n, int dtype, aldi_stream_t stream);
/* out[r] = (a ? a[r] : 0) + s * (idx >= 0 ? b[idx] : 0), idx = map ? map[r] : r, s = scale ? scale[r / rows_per_sample] : 1
 * (residual add, drop-path scaling, window un-partition and its transpose) */
int aldi_rows_add(const void* a, const void* b, const int* map, const float* scale, void* out, int rows, int C,
                  int rows_per_sample, int dtype, aldi_stream_t stream);
/* uint8 [N][3][Hs][Ws] staging -> normalised rows [N*(Hs/P)*(Ws/P)][3*P*P] (c, ph, pw order); hw = int[2N] image sizes (device),
 * mean / std = float[3] (host) */
int aldi_patchify(const uint8_t* img, void* out, int N, int Hs, int Ws, int P, const int* hw, const float* mean, const float* std,
                  int dtype, aldi_stream_t stream);
/* F.interpolate(mode="linear", align_corners=False) of a table [L0][C] -> [L1][C]; backward: in = d[L1][C], out += d[L0][C] */
int aldi_linear_resize(const float* in, float* out, int L0, int L1, int C, int backward, aldi_stream_t stream);
/* F.interpolate(mode="bicubic", align_corners=False) of a grid [S0h][S0w][C] -> [gh][gw][C]; backward accumulates likewise */
int aldi_bicubic_resize(const float* in, float* out, int S0h, int S0w, int gh, int gw, int C, int backward, aldi_stream_t stream);
/* y[n] = x[n] + pos (pos fp32 [TC]); aldi_sum_batch: out[TC] = sum_n g[n] (gradient of pos) */
int aldi_add_pos(const void* x, const float* pos, void* y, int N, long TC, int dtype, aldi_stream_t stream);
int aldi_sum_batch(const void* g, float* out, int N, long TC, int dtype, aldi_stream_t stream);
/* 2x2 stride-2 max pool, NHWC (SimpleFeaturePyramid scale 0.5).  forward: x [N][H][W][C] -> y [N][H/2][W/2][C], idx = winning tap
 * (first maximum in window order); backward: x = d(pooled), y = d(input) [N][H][W][C], fully written. */
int aldi_maxpool2(const void* x, void* y, unsigned char* idx, int N, int H, int W, int C, int backward, int dtype, aldi_stream_t stream);
/* torch.optim.AdamW step `step` (1-based) on a flat fp32 segment; p_compute (nullable) receives the `dtype` copy */
int aldi_adamw_step(float* p, const float* g, float* m, float* v, void* p_compute, long n, float lr, float beta1, float beta2,
                    float eps, float weight_decay, int step, float grad_scale, int dtype, aldi_stream_t stream);

/* Attention with decomposed relative position bias, head dim 64, bf16.  nB = images x windows, each a gh x gw token grid
 * (L = gh*gw, Lp = L rounded up to 64); Dq from aldi_attn_layout (64 + gh + gw rounded up to 32 on small grids; 64 when rel_h == NULL),
 * at most 256.  Workspaces: Qp, Kp, dQp [nB*heads][L][Dq]; KpT [nB*heads][Dq][Lp]; VT [nB*heads][64][vt_cols]; QsT, dOT [nB*heads][64][Lp];
 * lse, delta [nB*heads][L]. */
typedef struct {
    const void* qkv;        /* [nB*L][3*heads*64]: q | k | v                                   */
    const float* rel_h;     /* [2*gh-1][64] fp32 (already resized to this grid), nullable      */
    const float* rel_w;     /* [2*gw-1][64]                                                    */
    void *Qp, *Kp, *KpT, *VT, *QsT;      /* written by aldi_attn_prepare                        */
    void* O;                /* [nB*L][heads*64]: written by forward, read by backward          */
    float* lse;             /* written by forward, read by backward                            */
    const void* dO;         /* backward input, like O                                          */
    void *dOT, *dQp;        /* backward scratch                                                */
    float* delta;           /* backward scratch                                                */
    void* dqkv;             /* backward output, like qkv                                       */
    float *drel_h, *drel_w; /* backward outputs, accumulate (fp32 atomics), nullable iff rel_h */
    int nB, gh, gw, heads, Dq;
    float scale;            /* head_dim ** -0.5                                                */
} aldi_attn_args;
/* layout the kernels use for a gh x gw grid: Dq, whether the tiled path (8x8 key blocks: Dq = 64 + ceil8(gh) + ceil8(gw) rounded to 32)
 * is taken, and the number of columns VT must provide per (image, head, channel) row (>= Lp) */
int aldi_attn_layout(int gh, int gw, int rel, int* Dq, int* tiled, long* vt_cols);
int aldi_attn_prepare(const aldi_attn_args* a, aldi_stream_t stream);
int aldi_attn_forward(const aldi_attn_args* a, aldi_stream_t stream);
int aldi_attn_backward(const aldi_attn_args* a, aldi_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Multi-scale deformable attention sampling (Deformable-DETR; BASELINE configs[4]), fp32.
 * Replaces MSDeformAttnFunction.apply(value, spatial_shapes, level_start_index, sampling_locations, attention_weights,
 * im2col_step) of the reference's absent aldi/detr/libs submodule (.gitmodules:4-6; configs/Base-DETR.yaml:1-81).
 * value [N][S][M][D] (S = sum_l H_l*W_l, D = 32 or 64); spatial_shapes int[L][2] = (H, W); level_start_index int[L];
 * sampling_loc [N][Lq][M][L][P][2] = (x, y) in [0, 1]; attn_weight [N][Lq][M][L][P]; out [N][Lq][M*D].
 * backward: grad_value is zeroed and accumulated (fp32 atomics); grad_sampling_loc / grad_attn_weight are fully written.
 * ------------------------------------------------------------------------------------------------------------------ */
int aldi_ms_deform_attn_forward(const float* value, const int* spatial_shapes, const int* level_start_index, const float* sampling_loc,
                                const float* attn_weight, float* out, int N, int S, int M, int D, int Lq, int L, int P, aldi_stream_t stream);
int aldi_ms_deform_attn_backward(const float* value, const int* spatial_shapes, const int* level_start_index, const float* sampling_loc,
                                 const float* attn_weight, const float* grad_out, float* grad_value, float* grad_sampling_loc,
                                 float* grad_attn_weight, int N, int S, int M, int D, int Lq, int L, int P, aldi_stream_t stream);

/* The same backward when the queries ARE the pyramid's positions (the encoder's self attention: Lq == S, query level_start[l] + y W_l + x
 * at pixel (x, y) of level l); spatial_shapes_host = the same int[L][2] in host memory (launch geometry).  The value gradient is GATHERED:
 * a workgroup owns an 8 x 8 tile of value pixels of one level and head, lists the (query, weight) pairs that reach each pixel and writes
 * every gradient element once; samples farther than 5 pixels from their query's position on the target level go through the atomic
 * scatter in a second launch (any offsets are handled; the sum is the same, the order of the float additions differs).
 * Tuning msda_gather = bit mask of the target levels that are gathered (default 7: the three finest; the others keep the scatter),
 * msda_gather_list = the list length the tile sizes aim at.  D != 32 or msda_gather = 0: the general form. */
int aldi_ms_deform_attn_backward_self(const float* value, const int* spatial_shapes, const int* level_start_index, const int* spatial_shapes_host,
                                      const float* sampling_loc, const float* attn_weight, const float* grad_out, float* grad_value,
                                      float* grad_sampling_loc, float* grad_attn_weight, void* workspace, size_t workspace_bytes,
                                      int N, int S, int M, int D, int L, int P, aldi_stream_t stream);
/* With a workspace of aldi_ms_deform_attn_backward_self_workspace(...) bytes (tuning msda_bin = 1) the lists are built by ONE pass over the
 * samples: a thread per sample appends it to the 1..4 tiles its footprint touches (tiles sized for >= msda_bin_list (512) expected entries on
 * every level), a workgroup per tile sums its list; no neighbourhood predicate, any offsets, all levels.  workspace NULL: the walk form. */
size_t aldi_ms_deform_attn_backward_self_workspace(const int* spatial_shapes_host, int N, int S, int M, int L, int P);

/* Deformable-DETR pieces around that op (csrc/detr.hip), fp32; the arithmetic follows oracle/deformable_detr.py.
 * GroupNorm(G) over NHWC maps x [N][HW][C] (the input projections' normalisation): mean / rstd [N][G] are written for a backward
 * pass; workspace: aldi_group_norm_workspace(N, HW, G) bytes.  Deterministic (two-stage sums in a fixed order). */
size_t aldi_group_norm_workspace(int N, int HW, int G);
int aldi_group_norm_forward(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd, void* workspace, int N, int HW,
                            int C, int G, float eps, aldi_stream_t stream);
/* From the outputs raw [T][M*L*P*3] of a deformable-attention layer's two linear maps (sampling offsets [M][L][P][2], then attention
 * logits [M][L*P]) and the tokens' reference points ref [T][L][2]: sampling_loc [T][M][L][P][2] = ref + offset / (W_l, H_l) and
 * attn_weight [T][M][L][P] = softmax over (l, p) -- the operands of aldi_ms_deform_attn_forward. */
int aldi_msda_prepare(const float* raw, const float* ref, const int* spatial_shapes, float* sampling_loc, float* attn_weight, long T, int M, int L, int P,
                      aldi_stream_t stream);
/* softmax(q k^T * scale) v for a few hundred tokens (the decoder's self attention): q / k / v [B][Q][H*D] with row strides ldq / ldk / ldv
 * floats, D = 16, 32 or 64; out [B][Q][H*D]; lse [B][H][Q] nullable. */
int aldi_mha_small_forward(const float* q, const float* k, const float* v, float* out, float* lse, int B, int Q, int H, int D, int ldq, int ldk, int ldv,
                           float scale, float drop_p, unsigned long long seed, aldi_stream_t stream);
/* Dropout without a stored mask: out[i] = (res ? res[i] : 0) + x[i] * keep(seed, i) / (1 - p), keep decided by a 64-bit mix of (seed, i).
 * The same call on a gradient is the backward pass.  aldi_mha_small_* with drop_p > 0 drop attention PROBABILITIES with index
 * ((b * H + h) * Q + i) * Q + j of the same function (nn.MultiheadAttention(dropout=p)).  Not torch's random stream (TRANSFORMER.DROPOUT 0.1 of
 * configs/Base-DETR.yaml:24 draws from CUDA Philox in the reference): same statistics, own generator. */
int aldi_dropout_add(const float* x, const float* res, float* out, long n, float p, unsigned long long seed, aldi_stream_t stream);
/* Backward passes of the three above.  msda_prepare: g_raw [T][M*L*P*3] fully written from the sampling op's gradients (g_loc, g_aw) and
 * the forward's attention weights; g_ref [T][L][2] (nullable) = the reference points' gradient, fully written.  mha_small: dq / dk / dv with
 * row strides lddq / lddk / lddv from out, d_out and the forward's lse; delta [B][H][Q] is scratch.  group_norm: dx fully written, dgamma /
 * dbeta ACCUMULATE (deterministically: fixed summation orders); workspace aldi_group_norm_backward_workspace(N, HW, C, G) bytes. */
int aldi_msda_prepare_backward(const float* g_loc, const float* g_aw, const float* attn_weight, const int* spatial_shapes, float* g_raw, float* g_ref, long T,
                               int M, int L, int P, aldi_stream_t stream);
int aldi_mha_small_backward(const float* q, const float* k, const float* v, const float* out, const float* d_out, const float* lse, float* dq, float* dk,
                            float* dv, float* delta, int B, int Q, int H, int D, int ldq, int ldk, int ldv, int lddq, int lddk, int lddv, float scale,
                            float drop_p, unsigned long long seed, aldi_stream_t stream);
size_t aldi_group_norm_backward_workspace(int N, int HW, int C, int G);
int aldi_group_norm_backward(const float* g, const float* x, const float* gamma, const float* mean, const float* rstd, float* dx, float* dgamma,
                             float* dbeta, void* workspace, int N, int HW, int C, int G, aldi_stream_t stream);
/* g_t [R][4] = g_boxes * boxes * (1 - boxes); g_ref [refs][2] (nullable) ACCUMULATES the reference points' gradient through logit(). */
int aldi_detr_box_finish_backward(const float* g_boxes, const float* boxes, const float* ref, float* g_t, float* g_ref, long R, long refs,
                                  aldi_stream_t stream);
/* The set loss of LB = (decoder layers x B) prediction sets against the B images' targets (padded to Gmax; t_count [B]), as the authors
 * compute it (oracle/deformable_detr.py; coefficients configs/Base-DETR.yaml:27-39).
 * match_cost: cost [LB][Nq][Gmax] = w_bbox * L1 + w_class * focal class cost - w_giou * GIoU (columns >= t_count are 0): what the HOST's
 *   Hungarian solver takes (scipy.optimize.linear_sum_assignment, as in the reference: a sequential algorithm on 300 x ~10 matrices).
 * set_loss: match [LB][Nq] = index of the query's target or -1.  losses [LB / B][3] = (focal x 1, L1, 1 - GIoU) sums / num_boxes per
 *   decoder layer (unweighted, rows added in order: deterministic); g_logits / g_boxes = gradients of sum_layers (c_ce * focal + c_bbox *
 *   L1 + c_giou * (1 - GIoU)); rows [LB * Nq][3] is scratch. */
int aldi_detr_match_cost(const float* logits, const float* boxes, const int* t_labels, const float* t_boxes, const int* t_count, float* cost, int LB, int B,
                         int Nq, int K, int Gmax, float w_class, float w_bbox, float w_giou, float alpha, aldi_stream_t stream);
int aldi_detr_set_loss(const float* logits, const float* boxes, const int* match, const int* t_labels, const float* t_boxes, float* rows, float* losses,
                       float* g_logits, float* g_boxes, int LB, int B, int Nq, int K, int Gmax, float alpha, float c_ce, float c_bbox, float c_giou,
                       float num_boxes, aldi_stream_t stream);
/* x [T][C] in place: rows whose keep byte is 0 become zero (the value maps of padded pixels, MSDeformAttn's masked_fill) */
int aldi_mask_rows(float* x, const unsigned char* keep, long T, int C, aldi_stream_t stream);
/* boxes [R][4] = sigmoid(t [R][4] + (logit(ref [r % refs][0..1]), 0, 0)): the box head's last step (reference points in logit space) */
int aldi_detr_box_finish(const float* t, const float* ref, float* boxes, long R, long refs, aldi_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * ConvNeXt trunk (reference aldi/backbone.py:189-352).  Depthwise 7x7 / pad 3 convolution, NHWC, C % 8 == 0; wt is [7][7][C]
 * (the reference's [C][1][7][7] transposed), bias fp32.  flip != 0 mirrors the taps and ignores bias: the data gradient.
 * aldi_dwconv7_wgrad accumulates dw [7][7][C] (fp32 atomics).  Layer scale: out = x + s(r) * gamma (.) y with s per sample
 * (stochastic depth; nullable = 1) and its backward (dy fully written, dgamma accumulated).
 * ------------------------------------------------------------------------------------------------------------------ */
int aldi_dwconv7(const void* x, const void* wt, const float* bias, void* y, int N, int H, int W, int C, int flip, int dtype, aldi_stream_t stream);
int aldi_dwconv7_wgrad(const void* x, const void* g, float* dw, int N, int H, int W, int C, int dtype, aldi_stream_t stream);
int aldi_scale_add(const void* x, const void* y, const float* gamma, const float* scale, void* out, long rows, int C, int rows_per_sample,
                   int dtype, aldi_stream_t stream);
int aldi_scale_add_backward(const void* g, const void* y, const float* gamma, const float* scale, void* dy, float* dgamma, long rows, int C,
                            int rows_per_sample, int dtype, aldi_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* ALDI_HIP_H */
