// Implicit-GEMM convolution / linear on the gfx950 matrix cores.
//
//   out[pixel][co] = epilogue( sum_{kh,kw,ci} x[n, ho*s-p+kh, wo*s-p+kw, ci] * w[co][kh][kw][ci] )
//
// GEMM view: "pixels" (N*Ho*Wo) x "channels" (Cout) x K (KH*KW*Cin).  The weight tile is the
// MFMA A operand and the gathered activation tile the B operand, so in the 16x16 accumulator
// fragment every lane owns 4 CONSECUTIVE output channels of one pixel: the epilogue (FrozenBN
// scale/shift, residual / FPN top-down add, ReLU, ReLU-backward mask) works on 4-vectors and
// stores 8 B (bf16) or 16 B (fp32) per lane.
//
// Tiles go global -> LDS directly by LDS-DMA (`buffer_load_dwordx4 ... lds`): a 32-bit per-lane byte offset into a raw
// buffer descriptor, where an out-of-range offset writes zeros (border taps, tile tails and K tails are selects of the
// offset, never branches).  LDS rows are 64 B (four 16-B chunks = one MFMA k-step per slab); the bank swizzle is applied
// on the SOURCE address because the DMA writes lane-linearly.  Three-slab LDS ring with counted `vmcnt` waits and raw
// `s_barrier`; fragments are hand-issued `ds_read_b128` into two register sets so the reads of slab s+1 run under the
// MFMAs of slab s.  3x3 / stride-1 / pad-1 convs use the "halo" form: one (BM+2)-pixel slab serves the three
// horizontal taps (see the HALO branch).
//
// dtype: bf16 -> v_mfma_f32_16x16x32_bf16 (K slab 64); fp32 -> v_mfma_f32_16x16x4_f32 x4
// (K slab 32, exact fp32 -- the parity mode).  Both share the byte geometry of the tiles.
//
// Replaces the cuDNN/MIOpen conv + FrozenBN + ReLU (+ residual) chain and torch Linear that
// the reference reaches through detectron2 (aldi/align.py:72, aldi/distill.py:157,162).
#include "common.h"
#include "tile_prims.h"
#include <stdio.h>
#include <stdlib.h>

namespace {

struct ConvDev {
    const void* x; const void* w; void* y; float* y_f32;
    const float* scale; const float* shift; const void* res; const void* mask;
    int N, H, W, Cin, Cout, KH, KW, stride, pad, Ho, Wo;
    int relu, res_mode, out_scale, OH, OW;
    int M, K, xcd, dbg;
    unsigned x_bytes, w_bytes;
    int ksplit, slabs_per_split;     // split-K (plain 1x1 / linear): blockIdx.z = slice, slabs_per_split K slabs each
    int lean;                        // plain 1x1 / linear with whole K slabs (direct-epilogue tiles): the K loop's DMA offsets are running sums
    // ReLU masks as BITS (bf16, Cout % 8 == 0, plain output layout): [M][Cout / 8] bytes, bit c % 8 of byte c / 8 = (y[m][c] > 0).
    // bits_out: written by the forward launch beside y; mask_bits: read by the backward launch instead of the 16x larger `mask` tensor.
    const unsigned char* mask_bits; unsigned char* bits_out;
};

// ---- epilogue of a BM x BN tile: lane owns pixel (lane&15), channels (lane>>4)*4 .. +3 of each 16x16 accumulator fragment.
// `stg`: the workgroup's LDS (LDS_BYTES of it), free for staging once every wave is past its last fragment read.
template <typename T, int BM, int BN, int WM, int WN, int LDS_BYTES>
__device__ __forceinline__ void igemm_epilogue(const ConvDev& p, f32x4_t (&acc)[BM / WM / 16][BN / WN / 16], const int m0, const int n0, unsigned char* stg) {
    constexpr int NT = WM * WN * 64;
    constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int fr = lane & 15, fq = lane >> 4;
    if (p.dbg & 4) return;
    T* __restrict__ Y = static_cast<T*>(p.y);
    const T* __restrict__ R = static_cast<const T*>(p.res);
    const T* __restrict__ Mk = static_cast<const T*>(p.mask);
    if constexpr (sizeof(T) == 2) {
        // bf16 fast path: the accumulator fragment gives every lane 8 B per pixel (32-B runs per pixel row) -- poor store /
        // residual-load granularity for what are mostly HBM-bound layers.  Stage scale*acc+shift through LDS (padded rows,
        // conflict-free 8-B writes) and let every lane finish 16 B of one pixel row: 256-B coalesced residual / mask
        // loads and stores.  (conv*scale+shift is rounded to bf16 before the residual add; the fp32 parity mode keeps
        // the single-rounding direct path below.)
        constexpr int ROWB = BN * 2 + 16;
        constexpr bool kStageFits = BM * ROWB <= LDS_BYTES;              // staging tile reuses the pipeline buffers
        if (kStageFits && Y && !p.y_f32 && (p.Cout & 7) == 0) {
            // The epilogue is instruction-bound if written naively (wave64 VALU ops cost 4 cycles each and a block only
            // moves 32 KB): hardware bf16 packing, per-column scale/shift hoisted, immediate LDS offsets, and packed
            // 16-bit integer ops for ReLU / mask when there is no residual to add.
            __syncthreads();                    // every wave is done reading the last slab
            {
                unsigned char* wr = stg + (wm * (BM / WM) + fr) * ROWB + (wn * (BN / WN) + fq * 4) * 2;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int c = n0 + wn * (BN / WN) + j * 16 + fq * 4;
                    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (c < p.Cout) {
                        if (p.scale) sc = *reinterpret_cast<const float4*>(p.scale + c);
                        if (p.shift) sh = *reinterpret_cast<const float4*>(p.shift + c);
                    }
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        uint2 t;
                        if (p.scale || p.shift) {
                            t.x = pack2_bf16(acc[i][j][0] * sc.x + sh.x, acc[i][j][1] * sc.y + sh.y);
                            t.y = pack2_bf16(acc[i][j][2] * sc.z + sh.z, acc[i][j][3] * sc.w + sh.w);
                        } else {
                            t.x = pack2_bf16(acc[i][j][0], acc[i][j][1]);
                            t.y = pack2_bf16(acc[i][j][2], acc[i][j][3]);
                        }
                        *reinterpret_cast<uint2*>(wr + i * 16 * ROWB + j * 32) = t;
                    }
                }
            }
            __syncthreads();
            constexpr int CPR = BN / 8;          // 16-B chunks per tile row
            constexpr int RPI = NT / CPR;        // tile rows covered per iteration
            const int rowl0 = tid / CPR, ch8 = tid % CPR;
            const int c = n0 + ch8 * 8;
            if (c >= p.Cout) return;
            const bool plain = p.out_scale == 1 && p.res_mode != 2;
            const unsigned char* rd = stg + rowl0 * ROWB + ch8 * 16;
            typedef short s16x2_t __attribute__((ext_vector_type(2)));
            // Branch-free and latency-flat: byte offsets (out-of-tile rows -> the dropped/zero-filled range), then ALL the
            // residual / mask loads of this lane in flight at once, then the arithmetic and the stores.  (A load next to a
            // store inside one loop body is serialised behind it: Y and R may alias as far as the compiler knows.)
            constexpr int NI = BM / RPI;
            constexpr unsigned OOBX = 0x80000000u;
            const __amdgpu_buffer_rsrc_t ry = make_rsrc_uniform(Y, 0x7fffffffu);
            unsigned ooff[NI], roff[NI], boff[NI];
#pragma unroll
            for (int it = 0; it < NI; ++it) {
                const int m = m0 + rowl0 + it * RPI;
                const bool ok = m < p.M;
                unsigned oidx, ridx;
                if (plain) {
                    oidx = (unsigned)m * (unsigned)p.Cout;
                    ridx = oidx;
                } else {
                    const int mm = ok ? m : 0;
                    const int n = mm / (p.Ho * p.Wo);
                    const int r = mm - n * (p.Ho * p.Wo);
                    const int ho = r / p.Wo, wo = r - ho * p.Wo;
                    if (p.out_scale == 1) oidx = (unsigned)mm * (unsigned)p.Cout;
                    else oidx = (unsigned)((((long)n * p.OH + ho * p.out_scale) * p.OW + wo * p.out_scale) * p.Cout);
                    if (p.res_mode == 2) ridx = (unsigned)((((long)n * (p.Ho >> 1) + (ho >> 1)) * (p.Wo >> 1) + (wo >> 1)) * p.Cout);
                    else ridx = oidx;
                }
                ooff[it] = ok ? (oidx + (unsigned)c) * 2u : OOBX;
                roff[it] = ok ? (ridx + (unsigned)c) * 2u : OOBX;
                boff[it] = ok ? (oidx + (unsigned)c) >> 3 : OOBX;          // this lane's 8 channels = one byte of a bit mask
            }
            u32x4_t rres[NI], rmsk[NI];
            if (p.res_mode) {
                const __amdgpu_buffer_rsrc_t rr_ = make_rsrc_uniform(R, 0x7fffffffu);
#pragma unroll
                for (int it = 0; it < NI; ++it) rres[it] = __builtin_amdgcn_raw_buffer_load_b128(rr_, roff[it], 0, 0);
            }
            if (Mk) {
                const __amdgpu_buffer_rsrc_t rm_ = make_rsrc_uniform(Mk, 0x7fffffffu);
#pragma unroll
                for (int it = 0; it < NI; ++it) rmsk[it] = __builtin_amdgcn_raw_buffer_load_b128(rm_, ooff[it], 0, 0);
            }
            unsigned rbits[NI];
            if (p.mask_bits) {
                const __amdgpu_buffer_rsrc_t rb_ = make_rsrc_uniform(p.mask_bits, 0x7fffffffu);
#pragma unroll
                for (int it = 0; it < NI; ++it) rbits[it] = __builtin_amdgcn_raw_buffer_load_b8(rb_, boff[it], 0, 0);
            }
            const __amdgpu_buffer_rsrc_t rbo = make_rsrc_uniform(p.bits_out, 0x7fffffffu);
#pragma unroll
            for (int it = 0; it < NI; ++it) {
                const uint4 raw4 = *reinterpret_cast<const uint4*>(rd + it * RPI * ROWB);
                uint32_t raw[4] = {raw4.x, raw4.y, raw4.z, raw4.w};
                if (p.res_mode) {
                    const uint32_t rs[4] = {rres[it].x, rres[it].y, rres[it].z, rres[it].w};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float lo = __uint_as_float(raw[q] << 16) + __uint_as_float(rs[q] << 16);
                        float hi = __uint_as_float(raw[q] & 0xffff0000u) + __uint_as_float(rs[q] & 0xffff0000u);
                        if (p.relu) { lo = fmaxf(lo, 0.f); hi = fmaxf(hi, 0.f); }
                        raw[q] = pack2_bf16(lo, hi);
                    }
                } else if (p.relu) {
                    // bf16 as int16: negative floats (and -0) are negative integers
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        s16x2_t v = *reinterpret_cast<s16x2_t*>(&raw[q]);
                        v = __builtin_elementwise_max(v, s16x2_t{0, 0});
                        raw[q] = *reinterpret_cast<uint32_t*>(&v);
                    }
                }
                if (Mk) {
                    // keep where the forward activation (a ReLU output, so >= 0) is > 0: multiply the bit patterns by 0 / 1
                    const uint32_t ms[4] = {rmsk[it].x, rmsk[it].y, rmsk[it].z, rmsk[it].w};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        s16x2_t k = *reinterpret_cast<const s16x2_t*>(&ms[q]);
                        k = __builtin_elementwise_min(__builtin_elementwise_max(k, s16x2_t{0, 0}), s16x2_t{1, 1});
                        s16x2_t v = *reinterpret_cast<s16x2_t*>(&raw[q]);
                        v = v * k;
                        raw[q] = *reinterpret_cast<uint32_t*>(&v);
                    }
                }
                if (p.mask_bits) {          // the same, the forward activation's sign from one bit per element
                    const unsigned b = rbits[it];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const unsigned k32 = ((b >> (2 * q)) & 1u) | (((b >> (2 * q + 1)) & 1u) << 16);
                        s16x2_t k = *reinterpret_cast<const s16x2_t*>(&k32);
                        s16x2_t v = *reinterpret_cast<s16x2_t*>(&raw[q]);
                        v = v * k;
                        raw[q] = *reinterpret_cast<uint32_t*>(&v);
                    }
                }
                const u32x4_t ov = {raw[0], raw[1], raw[2], raw[3]};
                __builtin_amdgcn_raw_buffer_store_b128(ov, ry, ooff[it], 0, 0);
                if (p.bits_out) {           // (y > 0) of this lane's 8 channels: bf16 as int16, positive floats are positive integers
                    unsigned b = 0;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        b |= ((short)(raw[q] & 0xffffu) > 0 ? 1u : 0u) << (2 * q);
                        b |= ((short)(raw[q] >> 16) > 0 ? 1u : 0u) << (2 * q + 1);
                    }
                    __builtin_amdgcn_raw_buffer_store_b8((unsigned char)b, rbo, boff[it], 0, 0);
                }
            }
            return;
        }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        int m = m0 + wm * (BM / WM) + i * 16 + fr;
        if (m >= p.M) continue;
        long oidx, ridx = 0;
        if (p.out_scale == 1 && p.res_mode != 2) {
            oidx = (long)m * p.Cout;
            ridx = oidx;
        } else {
            int n = m / (p.Ho * p.Wo);
            int r = m - n * (p.Ho * p.Wo);
            int ho = r / p.Wo, wo = r - ho * p.Wo;
            if (p.out_scale == 1) oidx = (long)m * p.Cout;
            else oidx = (((long)n * p.OH + ho * p.out_scale) * p.OW + wo * p.out_scale) * p.Cout;
            if (p.res_mode == 2) ridx = (((long)n * (p.Ho >> 1) + (ho >> 1)) * (p.Wo >> 1) + (wo >> 1)) * p.Cout;
            else ridx = oidx;
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            int c = n0 + wn * (BN / WN) + j * 16 + fq * 4;
            if (c >= p.Cout) continue;
            float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            if (p.scale) {
                float4 sc = *reinterpret_cast<const float4*>(p.scale + c);
                v[0] *= sc.x; v[1] *= sc.y; v[2] *= sc.z; v[3] *= sc.w;
            }
            if (p.shift) {
                float4 sh = *reinterpret_cast<const float4*>(p.shift + c);
                v[0] += sh.x; v[1] += sh.y; v[2] += sh.z; v[3] += sh.w;
            }
            if (p.res_mode) {
                float r4[4];
                load4(R + ridx + c, r4);
                v[0] += r4[0]; v[1] += r4[1]; v[2] += r4[2]; v[3] += r4[3];
            }
            if (p.relu) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.f ? v[r] : 0.f;
            }
            if (Mk) {
                float k4[4];
                load4(Mk + oidx + c, k4);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = k4[r] > 0.f ? v[r] : 0.f;
            }
            if (Y) store4(Y + oidx + c, v);
            if (p.y_f32) store4(p.y_f32 + oidx + c, v);
        }
    }
}


template <int N, int ROWB, int I = 0>
__device__ __forceinline__ void frag_read_each(u32x4_t* f, const unsigned* addr) {   // fragment I at addr[I] + I * 16 rows
    if constexpr (I < N) {
        f[I] = frag_read<I * 16 * ROWB>(addr[I]);
        frag_read_each<N, ROWB, I + 1>(f, addr);
    }
}

// ---- "direct" epilogue (bf16, BN = 64, plain output layout): no LDS staging, no workgroup barrier, no dependent global loads -----------
// The staged epilogue above is what bounds the short-K 1x1 layers of res3..res5 (rocprofv3: 7.3 VALU per MFMA, MFMA pipes busy 12.7 % of the
// cycles on res4 conv3, profiles/r03_pmc_conv.txt): after the K loop every wave writes its tile to the LDS, waits at a barrier, THEN requests
// the residual from memory, waits for it, adds, rounds a second time and stores.  Here
//   * the output channels are PERMUTED among the MFMA rows (`direct_perm`: the weight rows are fetched in that order, a free change of the
//     DMA source address), so that a lane's accumulators of fragments 2h, 2h+1 are 8 CONSECUTIVE channels of its pixel: one 16-byte store
//     per pixel row and 32-channel block straight from registers (the four lanes of a pixel write 64 contiguous bytes);
//   * the residual is requested in the PROLOGUE, beside the first K slabs, straight into 16 registers per lane in that same layout (the
//     data arrives while the K loop runs) and added in fp32: ONE rounding of conv * scale + shift + residual instead of two.  (Measured
//     first: the residual tile by LDS-DMA into an activation-slab image + an MFMA with a 0/1 selection fragment as the A operand -- exact,
//     no VALU -- was 10-27 % SLOWER than the staged epilogue: 16 KB more LDS per workgroup (3 instead of 4 per CU) and +17 % bytes on the
//     L2 -> LDS path, which is what bounds the K loop of these layers);
//   * FrozenBN scale / shift (or the ReLU-mask bits of a data-gradient launch) arrive the same way (one 4-byte-per-lane DMA per wave);
//   * ReLU is a packed int16 max on the rounded pairs, the mask an AND with a sign-extended bit.
template <int WNE>
__device__ __forceinline__ int direct_perm(int row) {          // MFMA row `row` of the BN-wide tile computes channel n0 + direct_perm(row)
    const int wv = row / WNE, rr = row % WNE, j = rr >> 4, r = rr & 15;
    return wv * WNE + (j >> 1) * 32 + (r >> 2) * 8 + (j & 1) * 4 + (r & 3);
}
__device__ __forceinline__ unsigned lds_read_u8(unsigned addr) {
    unsigned v;
    asm volatile("ds_read_u8 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}
template <int BM, int BN, int WM, int WN, bool RESL>
__device__ __forceinline__ void igemm_epilogue_direct(const ConvDev& p, f32x4_t (&acc)[BM / WM / 16][BN / WN / 16], const int m0, const int n0,
                                                      u32x4_t (&rres)[BM / WM / 16][BN / WN / 32], const unsigned aux_addr) {
    constexpr int TM = BM / WM / 16, TN = BN / WN / 16, WNE = BN / WN, H = WNE / 32;
    static_assert(TN == 2 * H, "whole 32-channel blocks per wave");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int fr = lane & 15, fq = lane >> 4;
    if (p.dbg & 4) return;
    // Everything below touches an accumulator ONCE (they live in the AGPR half of the register file: every separate pass over them --
    // scale, shift, residual -- would be a read and a write-back per element): the per-channel operands are fetched first, then each
    // (fragment row i, 32-channel block h) is finished in one expression chain  v = acc * scale + shift + residual -> mask -> round -> store.
    // (1) FrozenBN scale / shift from the workgroup's LDS copy of the tile's 64 scales / shifts (neutral values when the layer has none)
    // (the LDS reads and their wait are UNCONDITIONAL -- an unused operand reads whatever the aux region holds: a conditional hand-issued
    // read is a second definition of its registers, and the compiler may copy them on the joining edge before the data has landed)
    u32x4_t sc[TN], sh[TN];
    const bool has_sc = p.scale != nullptr, has_sh = p.shift != nullptr;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const unsigned a = aux_addr + (unsigned)(wn * WNE + (j >> 1) * 32 + fq * 8 + (j & 1) * 4) * 4u;
        sc[j] = frag_read<0>(a);
        sh[j] = frag_read<256>(a);
    }
    // (2) the ReLU-mask bits of a data-gradient launch (one byte = this lane's 8 channels)
    unsigned mb[TM][H];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int h = 0; h < H; ++h) mb[i][h] = lds_read_u8(aux_addr + (unsigned)((wm * (BM / WM) + i * 16 + fr) * (BN / 8) + (wn * H + h) * 4 + fq));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int j = 0; j < TN; ++j) { asm volatile("" : "+v"(sc[j])); asm volatile("" : "+v"(sh[j])); }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int h = 0; h < H; ++h) asm volatile("" : "+v"(mb[i][h]));
    const bool has_res = RESL && p.res_mode != 0;
    if constexpr (RESL) {
        if (has_res) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int h = 0; h < H; ++h) asm volatile("" : "+v"(rres[i][h]));      // (requested in the prologue; landed: the K loop's waits are behind us)
        }
    }
    const __amdgpu_buffer_rsrc_t ry = make_rsrc_uniform(p.y, 0x7fffffffu);
    const __amdgpu_buffer_rsrc_t rbo = make_rsrc_uniform(p.bits_out, 0x7fffffffu);
    typedef short s16x2_t __attribute__((ext_vector_type(2)));
    typedef unsigned short u16x2_t __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + wm * (BM / WM) + i * 16 + fr;
#pragma unroll
        for (int h = 0; h < H; ++h) {
            const int c = n0 + wn * WNE + h * 32 + fq * 8;
            const bool ok = m < p.M && c < p.Cout;
            const unsigned e0 = (unsigned)m * (unsigned)p.Cout + (unsigned)c;
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = acc[i][2 * h][e]; v[4 + e] = acc[i][2 * h + 1][e]; }
            if (has_sc) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] *= __uint_as_float(sc[2 * h][e]); v[4 + e] *= __uint_as_float(sc[2 * h + 1][e]); }
            }
            if (has_sh) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] += __uint_as_float(sh[2 * h][e]); v[4 + e] += __uint_as_float(sh[2 * h + 1][e]); }
            }
            if constexpr (RESL) {
                if (has_res) {                  // fp32 add before the single rounding
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const unsigned r2 = rres[i][h][q];
                        v[2 * q] += __uint_as_float(r2 << 16);
                        v[2 * q + 1] += __uint_as_float(r2 & 0xffff0000u);
                    }
                }
            }
            if (p.mask_bits) {
#pragma unroll
                for (int t = 0; t < 8; ++t) v[t] = __uint_as_float(__float_as_uint(v[t]) & (unsigned)__builtin_amdgcn_sbfe((int)mb[i][h], t, 1));
            }
            uint32_t d[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) d[q] = pack2_bf16(v[2 * q], v[2 * q + 1]);
            if (p.relu) {                       // bf16 as int16: negative floats (and -0) are negative integers
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    s16x2_t t = *reinterpret_cast<s16x2_t*>(&d[q]);
                    t = __builtin_elementwise_max(t, s16x2_t{0, 0});
                    d[q] = *reinterpret_cast<uint32_t*>(&t);
                }
            }
            const u32x4_t ov = {d[0], d[1], d[2], d[3]};
            __builtin_amdgcn_raw_buffer_store_b128(ov, ry, ok ? e0 * 2u : 0x80000000u, 0, 0);
            if (p.bits_out) {                   // (y > 0) of the 8 channels = one byte: halves clamped to {0, 1}, gathered by shifts
                unsigned u = 0;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    u16x2_t t = *reinterpret_cast<u16x2_t*>(&d[q]);
                    // after the ReLU every half is >= +0: (half != 0) == (y > 0); without a ReLU the sign bit decides first
                    if (!p.relu) { s16x2_t s_ = *reinterpret_cast<s16x2_t*>(&d[q]); s_ = __builtin_elementwise_max(s_, s16x2_t{0, 0}); t = *reinterpret_cast<u16x2_t*>(&s_); }
                    t = __builtin_elementwise_min(t, u16x2_t{1, 1});
                    u |= *reinterpret_cast<unsigned*>(&t) << (2 * q);
                }
                const unsigned b = (u & 0x55u) | ((u >> 15) & 0xaau);
                __builtin_amdgcn_raw_buffer_store_b8((unsigned char)b, rbo, ok ? e0 >> 3 : 0x80000000u, 0, 0);
            }
        }
    }
}

// body of one workgroup: tile `bid` of an nmt x nnt tile grid of problem p (launched alone: igemm_kernel; as one of several
// problems of the same layer shape sharing a launch: igemm_group_kernel)
// EPI: 0 = the staged epilogue; 1 = the direct epilogue (bf16, BN = 64); 2 = direct + the residual tile prefetched into the LDS
template <typename T, int BM, int BN, int WM, int WN, int KC, bool PIPE, bool HALO = false, int EPI = 0, bool LEAN = false>
__device__ __forceinline__ void igemm_body(const ConvDev& p, int bid, const int nmt, const int nnt) {
    static_assert(!LEAN || (EPI != 0 && !HALO), "lean K loop: the direct-epilogue tap tiles");
    constexpr int NT = WM * WN * 64;
    constexpr bool DIRECT = EPI != 0, RESL = EPI == 2;
    static_assert(!DIRECT || (sizeof(T) == 2 && BN == 64 && (BN / WN) % 32 == 0 && BM * BN / 32 <= NT), "direct epilogue: bf16, 64-channel tiles");
    static_assert(!RESL || (PIPE && !HALO), "residual prefetch: the register-pipelined 1x1 loop");
    constexpr int EP = Elem<T>::kPer16B;           // elements per 16-B chunk
    constexpr int BK = KC * EP;                    // K slab: KC 16-B chunks per LDS row (KC=8: 64 bf16 / 32 fp32)
    constexpr int LOG = KC == 8 ? 3 : 2;
    constexpr int A_IT = (BM * KC) / NT;           // pixel-tile chunks per thread
    constexpr int B_IT = (BN * KC + NT - 1) / NT;  // weight-tile chunks per thread
    constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
    static_assert(HALO || (BM * KC) % NT == 0, "tile/threads mismatch");

    constexpr int NBUF = 3;                        // LDS ring: two slabs of DMA in flight across the barrier
    constexpr int SLOTS = HALO ? ((BM + 2) * KC + 63) / 64 * 64 + 3 * BN * KC : (BM + BN) * KC;     // per ring stage
    constexpr int NSTAGE = HALO ? 2 : NBUF;
    // one LDS object (a second one makes hipcc drain the DMA queue before LDS reads): [ring | residual tile | scale + shift or mask bits]
    constexpr int AUX_SLOTS = DIRECT ? 64 : 0;
    constexpr int ZERO_SLOTS = HALO ? 4 : 0;                    // one all-zero 64-byte row: what a tap reads at the left / right image border
    __shared__ __attribute__((aligned(128))) uint4 lds_all[NSTAGE * SLOTS + AUX_SLOTS + ZERO_SLOTS];
    uint4 (*const lds)[SLOTS] = reinterpret_cast<uint4 (*)[SLOTS]>(&lds_all[0]);
    uint4* const aux_lds = &lds_all[NSTAGE * SLOTS];
    constexpr int N_RES = RESL ? (BM / WM / 16) * (BN / WN / 32) : 0;   // residual loads per thread (16 B each: its 8 channels of a pixel and 32-block)
    constexpr int N_PRE = DIRECT ? N_RES + 1 : 0;               // loads per wave issued once, behind the first K slabs (residual + aux)
    u32x4_t rres[BM / WM / 16][BN / WN / 32 > 0 ? BN / WN / 32 : 1];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    // XCD-aware tile order: workgroup b runs on XCD b % 8 (observed dispatch); give every XCD a contiguous range of
    // (pixel-tile, channel-tile) pairs, channel-tile fastest, so the tiles sharing an activation tile / halo rows
    // share one L2.  Pure speed: any placement computes the same result.
    if (p.xcd) {
        const int total = nmt * nnt, q = total >> 3, r = total & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int m0 = (bid / nnt) * BM, n0 = (bid % nnt) * BN;
    const T* __restrict__ X = static_cast<const T*>(p.x);
    const T* __restrict__ Wt = static_cast<const T*>(p.w);

    // per-thread gather descriptors (rows are fixed for the whole K loop).  Tiles go global -> LDS directly
    // (`buffer_load_dwordx4 ... lds`, LDS-DMA): no VGPR round trip and no ds_write pass, which is what bounds the
    // register-staged form of this loop (ds_write_b128 sustains ~79 B/clk/CU).  The DMA writes lane-linearly
    // (wave-uniform LDS base + lane*16), i.e. thread c of the tile fills LDS slot c = row*KC + (c % KC); the bank
    // swizzle therefore moves to the SOURCE: the lane fetches logical chunk swz(row, c % KC) (the swizzle is an
    // involution, the fragment reads keep using swz).  A 32-bit per-lane byte offset indexes a raw buffer
    // descriptor over the whole tensor; an out-of-range offset makes the DMA write zeros -- border taps, tail rows
    // and K tails need no branch, just a select of the offset.  Tap validity is a 16-bit mask computed once per row.
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(X), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(Wt), 0, p.w_bytes, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    const int wbase = __builtin_amdgcn_readfirstlane(tid & ~63);      // first tile slot of this wave (wave-uniform)
    // direct epilogue: what it reads besides the accumulators arrives by DMA while the K loop runs (N_PRE pieces per wave, always issued so
    // that the counted waits below are static; an unused piece is an out-of-range offset)
    auto issue_pre = [&]() {
        if constexpr (DIRECT) {
            if constexpr (RESL) {
                const __amdgpu_buffer_rsrc_t rr = make_rsrc_uniform(p.res, 0x7fffffffu);
#pragma unroll
                for (int i = 0; i < BM / WM / 16; ++i) {
                    const int m = m0 + wm * (BM / WM) + i * 16 + (lane & 15);
                    unsigned ridx = (unsigned)m;
                    if (p.res_mode == 2) {                      // FPN top-down: the coarser map's pixel (ho >> 1, wo >> 1)
                        const int mm = m < p.M ? m : 0;
                        const int n = mm / (p.Ho * p.Wo), r = mm - n * (p.Ho * p.Wo), ho = r / p.Wo, wo = r - ho * p.Wo;
                        ridx = (unsigned)((n * (p.Ho >> 1) + (ho >> 1)) * (p.Wo >> 1) + (wo >> 1));
                    }
#pragma unroll
                    for (int h = 0; h < BN / WN / 32; ++h) {
                        const int ch = n0 + wn * (BN / WN) + h * 32 + (lane >> 4) * 8;
                        const bool ok = p.res_mode && m < p.M && ch < p.Cout;
                        rres[i][h] = __builtin_amdgcn_raw_buffer_load_b128(rr, ok ? (ridx * (unsigned)p.Cout + (unsigned)ch) * 2u : OOB, 0, 0);
                    }
                }
            }
            // aux piece, 4 bytes per lane: a data-gradient launch fetches the tile's ReLU-mask bits (BN / 8 bytes per pixel row: lane = 32
            // channels of one row); a forward launch its 64 scales (even waves) / shifts (odd waves) -- the two never occur together
            if (p.mask_bits) {
                const __amdgpu_buffer_rsrc_t rb = make_rsrc_uniform(p.mask_bits, 0x7fffffffu);
                const int row = tid >> 1, m = m0 + row, ch = n0 + (tid & 1) * 32;
                const bool ok = tid < BM * 2 && m < p.M && ch < p.Cout;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (__attribute__((address_space(3))) void*)(reinterpret_cast<unsigned*>(aux_lds) + wbase), 4,
                                                         ok ? ((unsigned)m * (unsigned)p.Cout + (unsigned)ch) >> 3 : OOB, 0, 0, 0);
            } else {
                const bool odd = __builtin_amdgcn_readfirstlane(wave) & 1;
                const float* src = odd ? p.shift : p.scale;
                const __amdgpu_buffer_rsrc_t rs = make_rsrc_uniform(src, 0x7fffffffu);
                const int ch = n0 + lane;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(reinterpret_cast<unsigned*>(aux_lds) + (odd ? 64 : 0)), 4,
                                                         src && ch < p.Cout ? (unsigned)ch * 4u : OOB, 0, 0, 0);
            }
        }
    };
    f32x4_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fq = lane >> 4;
    bool pre_seen = true;                         // (direct epilogue) the once-per-tile DMA pieces are landed and visible when the K loop ends
    if constexpr (HALO) {
    // ---- 3x3 / stride 1 / pad 1 with operand reuse across the three horizontal taps -------------------------------
    // For a fixed (kh, 32-channel chunk) the pixel tiles of kw = 0,1,2 are the same BM+2 consecutive input pixels
    // shifted by one row of the LDS image: load that "halo" slab ONCE and point the three taps' fragment reads at rows
    // +0/+1/+2.  The K loop is bound by the L2 -> CU fill path, and the pixel tile is half of its bytes: a group of three
    // k-steps now moves (BM+2) + 3*BN rows instead of 3*(BM+BN).  What the shift cannot express is the zero padding at
    // the left / right image border (the neighbouring LDS row holds the previous / next image row's pixel): those
    // fragments are zeroed in registers (two of the three taps, 4 v_cndmask each).  Vertical padding and the tile's own
    // ends are out-of-range DMA offsets as before.
    static_assert(KC == 4, "halo form: one MFMA k-step per tap");
    constexpr int AS = ((BM + 2) * KC + 63) / 64 * 64;          // halo slots, padded to whole waves of DMA
    constexpr int WS = 3 * BN * KC;                              // three taps of weights
    constexpr int AH_IT = (AS + NT - 1) / NT, WH_IT = WS / NT;
    static_assert(WS % NT == 0, "weight slots per thread");
    unsigned ha_voff[AH_IT], ha_mask[AH_IT];
#pragma unroll
    for (int it = 0; it < AH_IT; ++it) {
        const int c = tid + it * NT, row = c >> 2, kce = swz_halo(row, c & 3);
        const int q0 = m0 - 1 + row;                             // input pixel under the CENTRE row (kh = 1), flat index
        const bool ok = row < BM + 2 && q0 >= 0 && q0 < p.M;
        const int qq = ok ? q0 : 0;
        const int h0 = (qq / p.W) % p.H;
        ha_voff[it] = ((unsigned)qq * (unsigned)p.Cin + (unsigned)(kce * EP)) * (unsigned)sizeof(T);
        ha_mask[it] = ok ? ((h0 >= 1 ? 1u : 0u) | 2u | (h0 <= p.H - 2 ? 4u : 0u)) : 0u;
    }
    unsigned hw_voff[WH_IT];
#pragma unroll
    for (int it = 0; it < WH_IT; ++it) {
        const int c = tid + it * NT, kwi = c / (BN * KC), rem = c - kwi * (BN * KC), row = rem >> 2, kce = swz_halo(row, rem & 3);
        const int co = n0 + (DIRECT ? direct_perm<BN / WN>(row) : row);
        hw_voff[it] = co < p.Cout ? ((unsigned)co * (unsigned)p.K + (unsigned)(kwi * p.Cin + kce * EP)) * (unsigned)sizeof(T) : OOB;
    }
    // Offsets of the group being LOADED are running sums: inside one kernel row (kh) a group's sources are the previous group's + one
    // 32-channel chunk (an out-of-range offset stays out of range), so a DMA costs one add; the per-lane validity of the halo rows (vertical
    // padding) changes with kh only -- three times per tile -- and is re-derived there.
    int gkh = 0, gci = 0;                                        // (kh, channel chunk) of the group being LOADED
    unsigned ha_cur[AH_IT], hw_cur[WH_IT];
    auto rebase = [&]() {
        const unsigned a_off = (unsigned)(((gkh - 1) * p.W * p.Cin) * (int)sizeof(T));
#pragma unroll
        for (int it = 0; it < AH_IT; ++it) ha_cur[it] = ((ha_mask[it] >> gkh) & 1u) ? ha_voff[it] + a_off : OOB;
    };
    rebase();
#pragma unroll
    for (int it = 0; it < WH_IT; ++it) hw_cur[it] = hw_voff[it];
    auto issue_group = [&](int buf) {
#pragma unroll
        for (int it = 0; it < AH_IT; ++it)
            if (wbase + it * NT < AS) {                          // wave uniform
                glds16(rx, &lds[buf][wbase + it * NT], ha_cur[it]);
                ha_cur[it] += (unsigned)(BK * sizeof(T));
            }
#pragma unroll
        for (int it = 0; it < WH_IT; ++it) glds16(rw, &lds[buf][AS + wbase + it * NT], hw_cur[it]);
        gci += BK;
        unsigned w_adv = (unsigned)(BK * sizeof(T));             // next chunk of this kernel row's three taps ...
        if (gci >= p.Cin) {
            gci = 0; ++gkh;
            w_adv = (unsigned)((2 * p.Cin + BK) * (int)sizeof(T));   // ... or the first chunk of the next kernel row (taps 3 kh .. 3 kh + 2)
            rebase();
        }
#pragma unroll
        for (int it = 0; it < WH_IT; ++it) hw_cur[it] += w_adv;
    };
    // left / right image border of this lane's fragment rows
    bool edge_l[TM], edge_r[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + wm * (BM / WM) + i * 16 + fr;
        const int wo = m % p.W;
        edge_l[i] = wo == 0;
        edge_r[i] = wo == p.W - 1;
    }
    const int xrow = wm * (BM / WM) + fr, wrow = wn * (BN / WN) + fr;
    const unsigned lbase = lds_addr(&lds[0][0]);
    unsigned x_rd[3];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) x_rd[kw] = lbase + (unsigned)((xrow + kw) * KC + swz_halo(xrow + kw, fq)) * 16u;
    const unsigned w_rd = lbase + (unsigned)((AS + wrow * KC) + swz_halo(wrow, fq)) * 16u;
    // The left / right border taps read ZEROS: the fragment's ADDRESS is redirected to the all-zero row (one select per fragment before
    // the read is issued) instead of zeroing its four registers behind the read (four selects).  frag_read_each adds i * 16 rows back.
    uint4* const zero_lds = &lds_all[NSTAGE * SLOTS + AUX_SLOTS];
    if (tid < ZERO_SLOTS) zero_lds[tid] = make_uint4(0u, 0u, 0u, 0u);        // (visible behind the loop's first barrier)
    unsigned zaddr[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) zaddr[i] = lds_addr(zero_lds) + (unsigned)fq * 16u - (unsigned)(i * 16 * KC * 16);
    constexpr unsigned GROUP_BYTES = (AS + WS) * 16;
    const int G = 3 * (p.Cin / BK);
    issue_group(0);
    issue_pre();                                 // (every wait of this loop is vmcnt(0): landed and, behind the barrier, visible from group 0 on)
    int buf = 0;
    for (int g = 0; g < G; ++g) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (g + 1 < G) issue_group(buf ^ 1);
        // the three taps of the group: fragments of tap kw+1 are read while tap kw's MFMAs run
        u32x4_t xf[2][TM], wf[2][TN];
        const unsigned gb = (unsigned)buf * GROUP_BYTES;
        {
            unsigned xa[TM];
#pragma unroll
            for (int i = 0; i < TM; ++i) xa[i] = edge_l[i] ? zaddr[i] : x_rd[0] + gb;
            frag_read_each<TM, KC * 16>(xf[0], xa);
        }
        frag_read_all<TN, KC * 16>(wf[0], w_rd + gb);
        frag_wait<TM, TN>(xf[0], wf[0]);
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int cur = kw & 1, nxt = cur ^ 1;
            if (kw < 2) {
                if (kw + 1 == 2) {
                    unsigned xa[TM];
#pragma unroll
                    for (int i = 0; i < TM; ++i) xa[i] = edge_r[i] ? zaddr[i] : x_rd[2] + gb;
                    frag_read_each<TM, KC * 16>(xf[nxt], xa);
                } else {
                    frag_read_all<TM, KC * 16>(xf[nxt], x_rd[kw + 1] + gb);
                }
                frag_read_all<TN, KC * 16>(wf[nxt], w_rd + gb + (unsigned)((kw + 1) * BN * KC * 16));
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = Mma<T>::run(wf[cur][j], xf[cur][i], acc[i][j]);
            if (kw < 2) frag_wait<TM, TN>(xf[nxt], wf[nxt]);
        }
        buf ^= 1;
    }
    } else {
    // LEAN (a template parameter: plain 1x1 / linear layers with whole K slabs on the direct-epilogue tiles = every res3..res5 1x1 at stride 1):
    // the generic issue below spends ~15 scalar / vector instructions per DMA on taps, border masks and K tails that do not exist there --
    // more issue slots than the slab's MFMAs take, and the kernels are issue-bound (measured: -10..-14 % on the 64x64 long-K tile, -3..-4 %
    // on 128x64).  A slab's offsets are the previous slab's + one slab of bytes (an out-of-range offset stays out of range: bit 31 is set
    // and K * 2 < 2^31), so a DMA costs one add, and the loop carries no tap / channel counters.
    unsigned a_voff[A_IT], a_mask[A_IT];
    int a_kce[A_IT];
    unsigned b_voff[B_IT];
    int b_kce[B_IT];
    const bool ktail = !LEAN && (p.K % BK) != 0;  // only 1x1 convs with a short / ragged K
    if constexpr (LEAN) {
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            const int c = tid + it * NT, row = c >> LOG, kce = swz<KC>(row, c & (KC - 1)), m = m0 + row;
            a_voff[it] = m < p.M ? ((unsigned)m * (unsigned)p.Cin + (unsigned)(kce * EP)) * (unsigned)sizeof(T) : OOB;
        }
#pragma unroll
        for (int it = 0; it < B_IT; ++it) {
            const int c = tid + it * NT, row = c >> LOG, kce = swz<KC>(row, c & (KC - 1));
            const int co = n0 + direct_perm<BN / WN>(row);
            b_voff[it] = (c < BN * KC) && co < p.Cout ? ((unsigned)co * (unsigned)p.K + (unsigned)(kce * EP)) * (unsigned)sizeof(T) : OOB;
        }
    } else {
    const bool ident = p.KH * p.KW == 1 && p.stride == 1 && p.pad == 0;   // 1x1: pixel index == input index, no divisions
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
        int c = tid + it * NT, row = c >> LOG, kce = swz<KC>(row, c & (KC - 1));
        int m = m0 + row;
        bool ok = m < p.M;
        a_kce[it] = kce * EP;
        if (ident) {
            a_voff[it] = ((unsigned)m * (unsigned)p.Cin + (unsigned)(kce * EP)) * (unsigned)sizeof(T);
            a_mask[it] = ok ? 1u : 0u;
            continue;
        }
        int mm = ok ? m : 0;
        int n = mm / (p.Ho * p.Wo);
        int r = mm - n * (p.Ho * p.Wo);
        int ho = r / p.Wo, wo = r - ho * p.Wo;
        int hi0 = ho * p.stride - p.pad, wi0 = wo * p.stride - p.pad;
        a_voff[it] = (unsigned)((((long)n * p.H + hi0) * p.W + wi0) * p.Cin + kce * EP) * (unsigned)sizeof(T);   // mod 2^32, tap offset added later
        unsigned mask = 0, colbits = 0;
        for (int kw_ = 0; kw_ < p.KW; ++kw_)
            if ((unsigned)(wi0 + kw_) < (unsigned)p.W) colbits |= 1u << kw_;
        for (int kh_ = 0; kh_ < p.KH; ++kh_)
            if ((unsigned)(hi0 + kh_) < (unsigned)p.H) mask |= colbits << (kh_ * p.KW);
        a_mask[it] = ok ? mask : 0u;
    }
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
        int c = tid + it * NT, row = c >> LOG, kce = swz<KC>(row, c & (KC - 1));
        int co = n0 + (DIRECT ? direct_perm<BN / WN>(row) : row);
        bool ok = (c < BN * KC) && co < p.Cout;
        b_voff[it] = ok ? (unsigned)((long)co * p.K + kce * EP) * (unsigned)sizeof(T) : OOB;
        b_kce[it] = kce * EP;
    }
    }

    int tap = 0, kh = 0, kw = 0, ci0 = 0;   // tap / channel offset of the slab being LOADED (block uniform)

    auto issue_slab = [&](int s, int buf) {
        if constexpr (LEAN) {
#pragma unroll
            for (int it = 0; it < A_IT; ++it) {
                glds16(rx, &lds[buf][wbase + it * NT], a_voff[it]);
                a_voff[it] += (unsigned)(BK * sizeof(T));
            }
#pragma unroll
            for (int it = 0; it < B_IT; ++it)
                if (wbase + it * NT < BN * KC) {                               // wave-uniform
                    glds16(rw, &lds[buf][BM * KC + wbase + it * NT], b_voff[it]);
                    b_voff[it] += (unsigned)(BK * sizeof(T));
                }
            return;
        }
        const unsigned tap_off = (unsigned)(((kh * p.W + kw) * p.Cin + ci0) * (int)sizeof(T));
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            bool ok = (a_mask[it] >> tap) & 1u;
            if (ktail) ok = ok && (ci0 + a_kce[it] < p.Cin);
            unsigned vo = ok ? a_voff[it] + tap_off : OOB;
            if (p.dbg & 16) vo &= 0xfffu;
            glds16(rx, &lds[buf][wbase + it * NT], vo);
        }
        const unsigned k_off = (unsigned)(s * BK) * (unsigned)sizeof(T);
#pragma unroll
        for (int it = 0; it < B_IT; ++it) {
            if (wbase + it * NT < BN * KC) {                               // wave-uniform
                unsigned off = b_voff[it] == OOB ? OOB : b_voff[it] + k_off;
                if (ktail && s * BK + b_kce[it] >= p.K) off = OOB;
                if (p.dbg & 16) off &= 0xfffu;
                glds16(rw, &lds[buf][BM * KC + wbase + it * NT], off);
            }
        }
        ci0 += BK;
        if (ci0 >= p.Cin) { ci0 = 0; ++tap; if (++kw == p.KW) { kw = 0; ++kh; } }
    };

    // Software pipeline, three levels deep (NBUF = 3 LDS slabs + two fragment register sets):
    //   iteration s:  wait for MY DMA pieces of slab s+1 (counted vmcnt: slab s+2 stays in flight) -> raw s_barrier (slab s+1
    //   visible, everyone is done reading slab s) -> issue the fragment reads of slab s+1 into the spare register set ->
    //   issue the DMA of slab s+3 into the buffer slab s just vacated -> 16 MFMAs on slab s from registers -> lgkmcnt wait.
    // So the LDS read latency and the DMA both run under the MFMAs of the same wave, not only under other waves'.
    // __syncthreads() would drain the DMA queue (vmcnt(0)) at every slab (cdna_hip_programming.md section 5).
    static_assert(KC == 4 || !PIPE, "the register-pipelined loop reads one k-step (4 chunks) per slab");
    constexpr int N_DMA = A_IT + (BN * KC) / NT;   // DMA instructions per slab of the wave that issues the fewest
    const int S_all = (p.dbg & 8) ? 1 : (p.K + BK - 1) / BK;
    // split-K: this workgroup's slice of the K slabs (plain 1x1 only: the running channel offset is the slab index)
    const int s_first = p.ksplit > 1 ? (int)blockIdx.z * p.slabs_per_split : 0;
    const int S = p.ksplit > 1 ? min(S_all - s_first, p.slabs_per_split) : S_all;
    ci0 = s_first * BK;
    if constexpr (PIPE) pre_seen = S > 3;         // (the register-pipelined loop waits for them together with slab 3)
    issue_slab(s_first, 0);
    if (S > 1) issue_slab(s_first + 1, 1);
    if constexpr (!PIPE) issue_pre();            // (two-level loop: behind slabs 0 and 1)
    constexpr unsigned SLAB_BYTES = (BM + BN) * KC * 16;
    const int xrow = wm * (BM / WM) + fr, wrow = wn * (BN / WN) + fr;
    const unsigned x_rd0 = lds_addr(&lds[0][0]) + (unsigned)(xrow * KC + swz<KC>(xrow, fq)) * 16u;
    const unsigned w_rd0 = lds_addr(&lds[0][0]) + (unsigned)((BM + wrow) * KC + swz<KC>(wrow, fq)) * 16u;
    if constexpr (PIPE) {
    if (S > 2) issue_slab(s_first + 2, 2);
    // the once-per-tile pieces go out behind the (up to) three slabs of the prologue: a counted wait for slab k may leave them in flight
    // as long as k <= 2 -- from the wait for slab 3 on they are older than everything still allowed to be outstanding
    __builtin_amdgcn_sched_barrier(0);           // (the counted waits below assume this issue order)
    issue_pre();
    __builtin_amdgcn_sched_barrier(0);
    u32x4_t xf0[TM], wf0[TN], xf1[TM], wf1[TN];
    if (S > 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * N_DMA + N_PRE) : "memory");
    else if (S > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_DMA + N_PRE) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_PRE) : "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    frag_read_all<TM, KC * 16>(xf0, x_rd0);
    frag_read_all<TN, KC * 16>(wf0, w_rd0);
    frag_wait<TM, TN>(xf0, wf0);
    int rbuf = 1, ibuf = 0;                         // buffer of slab s+1 (next reads) / of slab s (next DMA target)
    auto step = [&](int s, u32x4_t* xc, u32x4_t* wc, u32x4_t* xn, u32x4_t* wn_) {
        const bool more = s + 1 < S;
        if (more) {
            // waiting for slab s + 1; younger: slab s + 2 (if any) and, while s + 1 <= 2, the once-per-tile pieces
            if (N_PRE > 0 && s < 2) {
                if (s + 2 < S) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_DMA + N_PRE) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_PRE) : "memory");
            } else {
                if (s + 2 < S) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_DMA) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
        // The reads and their wait are UNCONDITIONAL (after the last slab they fetch an idle ring stage that nobody uses): the compiler takes
        // the asm's outputs for available at once, so where two definitions of a fragment register set meet (`if (more)` around the reads: a
        // phi) it may copy the registers on the incoming edge -- between the asynchronous read and frag_wait(), i.e. before the data has
        // landed.  It did: 128x64 tiles with 2 or 3 K slabs returned garbage in the fourth channel fragment (tools/isa_hazard_check.py
        // finds such copies in the generated code; tests/test_isa_hazards_cpu.py runs it over every kernel with hand-issued reads).
        frag_read_all<TM, KC * 16>(xn, x_rd0 + (unsigned)rbuf * SLAB_BYTES);
        frag_read_all<TN, KC * 16>(wn_, w_rd0 + (unsigned)rbuf * SLAB_BYTES);
        if (more && s + 3 < S) issue_slab(s_first + s + 3, ibuf);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = Mma<T>::run(wc[j], xc[i], acc[i][j]);
        frag_wait<TM, TN>(xn, wn_);
        rbuf = rbuf == NBUF - 1 ? 0 : rbuf + 1;
        ibuf = ibuf == NBUF - 1 ? 0 : ibuf + 1;
    };
    for (int s = 0; s < S; s += 2) {
        step(s, xf0, wf0, xf1, wf1);
        if (s + 1 < S) step(s + 1, xf1, wf1, xf0, wf0);
    }
    } else {
    // two-level form (no fragment double buffering): slab s+2 is issued while slab s is computed; used by the 8-wave
    // tile, where the extra fragment registers and issue slots cost more than the in-wave overlap returns
    int buf = 0, nbuf = 2;
    for (int s = 0; s < S; ++s) {
        // waiting for slab s; younger: slab s + 1 (if any) and, at s = 0, the once-per-tile pieces (issued behind slabs 0 and 1)
        if (N_PRE > 0 && s == 0 && S > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_DMA + N_PRE) : "memory");
        else if (s + 1 < S) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_DMA) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (s + 2 < S) issue_slab(s_first + s + 2, nbuf);
#pragma unroll
        for (int h = 0; h < KC / 4; ++h) {            // KC == 8: two MFMA k-steps per 128-byte slab row
            u32x4_t xf[TM], wf[TN];
            const unsigned xa = h == 0 ? x_rd0 : lds_addr(&lds[0][0]) + (unsigned)(xrow * KC + swz<KC>(xrow, fq + 4)) * 16u;
            const unsigned wa = h == 0 ? w_rd0 : lds_addr(&lds[0][0]) + (unsigned)((BM + wrow) * KC + swz<KC>(wrow, fq + 4)) * 16u;
            frag_read_all<TM, KC * 16>(xf, xa + (unsigned)buf * SLAB_BYTES);
            frag_read_all<TN, KC * 16>(wf, wa + (unsigned)buf * SLAB_BYTES);
            frag_wait<TM, TN>(xf, wf);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = Mma<T>::run(wf[j], xf[i], acc[i][j]);
        }
        buf = buf == NBUF - 1 ? 0 : buf + 1;
        nbuf = nbuf == NBUF - 1 ? 0 : nbuf + 1;
    }
    }

    }

    if constexpr (DIRECT) {
        // the once-per-tile pieces: every wave has waited for its own (the K loop's last wait is vmcnt(0)); a barrier makes the other waves'
        // visible -- the loop's own barriers did that already when it waited for a slab younger than the pieces (PIPE: slab 3; flat: slab 1)
        if (!pre_seen) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        __builtin_amdgcn_sched_barrier(0);
        igemm_epilogue_direct<BM, BN, WM, WN, RESL>(p, acc, m0, n0, rres, lds_addr(aux_lds));
        return;
    }
    if (p.ksplit > 1) {                              // this slice's partial tile, raw fp32, into its own slab of the workspace
        ConvDev q = p;
        q.y_f32 = p.y_f32 + (long)blockIdx.z * p.M * p.Cout;
        igemm_epilogue<T, BM, BN, WM, WN, NSTAGE * SLOTS * 16>(q, acc, m0, n0, reinterpret_cast<unsigned char*>(&lds_all[0]));
        return;
    }
    igemm_epilogue<T, BM, BN, WM, WN, NSTAGE * SLOTS * 16>(p, acc, m0, n0, reinterpret_cast<unsigned char*>(&lds_all[0]));
}

// ---- 3x3 / stride 1 / pad 1, halo form, with the two halves of an 8-wave workgroup in ALTERNATING ROLES ------------------------
// The plain halo loop above runs its waves in lockstep: barrier -> everyone issues DMA and fragment reads -> everyone issues
// MFMAs; on a SIMD the two resident waves want the matrix pipe at the same time and leave it idle at the same time (rocprofv3
// on the p2 conv: MFMA pipe busy 45 % of the cycles, waves parked at barriers / waitcnt 36 %, profiles/r03_pmc_conv.txt).
// Here a K step ("phase" = one horizontal tap of one (kh, 32-channel) group) is split into a LOAD segment (fragment reads of
// this phase, two DMA pieces of the group two ahead, edge fix-up) and a COMPUTE segment (16 MFMAs at raised priority) with a
// barrier after each, and waves 4..7 run ONE BARRIER BEHIND waves 0..3: while one half computes phase p its SIMD partners of the
// other half do their load segment -- matrix work beside memory work on every SIMD at all times (MI355X_MICROARCH.md
// "Two waves per SIMD", cdna_hip_programming.md T3+T4/T5).  Three LDS stages (groups g, g+1, g+2), DMA never drained inside
// the loop: the counted vmcnt in the last load segment of group g retires group g+1's pieces only.
//   RAW: a group's pieces are waited for (vmcnt) in a load segment and first read in the NEXT phase's load segment, two barriers
//        later -- one more than the lag between the halves.
//   WAR: a phase's fragment reads are retired (lgkmcnt(0)) right behind the barrier that ends its load segment, i.e. before the
//        barrier that ends its compute segment; a stage is re-targeted by DMA two phases after its last read, which for the
//        leading half is one full barrier interval after the lagging half retired its reads.

template <typename T, int BN>
__device__ __forceinline__ void igemm_halo_rs_body(const ConvDev& p, int bid, const int nmt, const int nnt) {
    static_assert(sizeof(T) == 2, "bf16 only");
    constexpr int BM = 256, WM = 4, WN = BN / 64, NT = WM * WN * 64, KC = 4, EP = 8, BK = 32;
    static_assert(NT == 512, "eight waves: two halves of four");
    constexpr int TM = BM / WM / 16, TN = BN / WN / 16;            // 4 x 4 fragments per wave
    constexpr int AS = ((BM + 2) * KC + 63) / 64 * 64;             // 1088 halo slots (16 B each)
    constexpr int WS = 3 * BN * KC;                                // three taps of weights
    constexpr int STAGE = AS + WS, NS = 3;
    constexpr int AH_IT = (AS + NT - 1) / NT, WH_IT = WS / NT;
    static_assert(AH_IT == 3 && WH_IT == 3 && WS % NT == 0, "six DMA pieces per thread and group: two per phase");
    __shared__ __attribute__((aligned(128))) uint4 lds[NS * STAGE + 4];          // (+ one all-zero 64-B row)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    if (p.xcd) {
        const int total = nmt * nnt, q = total >> 3, r = total & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int m0 = (bid / nnt) * BM, n0 = (bid % nnt) * BN;
    const T* __restrict__ X = static_cast<const T*>(p.x);
    const T* __restrict__ Wt = static_cast<const T*>(p.w);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(X), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(Wt), 0, p.w_bytes, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    const int wbase = __builtin_amdgcn_readfirstlane(tid & ~63);
    const bool half_b = __builtin_amdgcn_readfirstlane(wave) >= 4;               // the lagging half
    const bool third_pix = wbase + 2 * NT < AS;                                  // this wave owns a piece of the halo slab's tail (wave 0)
    f32x4_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fq = lane >> 4;
    if (tid < 4) lds[NS * STAGE + tid] = make_uint4(0u, 0u, 0u, 0u);             // (visible after the prologue's barrier)

    unsigned ha_voff[AH_IT], ha_mask[AH_IT];
#pragma unroll
    for (int it = 0; it < AH_IT; ++it) {
        const int c = tid + it * NT, row = c >> 2, kce = swz_halo(row, c & 3);
        const int q0 = m0 - 1 + row;
        const bool ok = row < BM + 2 && q0 >= 0 && q0 < p.M;
        const int qq = ok ? q0 : 0;
        const int h0 = (qq / p.W) % p.H;
        ha_voff[it] = ((unsigned)qq * (unsigned)p.Cin + (unsigned)(kce * EP)) * (unsigned)sizeof(T);
        ha_mask[it] = ok ? ((h0 >= 1 ? 1u : 0u) | 2u | (h0 <= p.H - 2 ? 4u : 0u)) : 0u;
    }
    unsigned hw_voff[WH_IT];
#pragma unroll
    for (int it = 0; it < WH_IT; ++it) {
        const int c = tid + it * NT, kwi = c / (BN * KC), rem = c - kwi * (BN * KC), row = rem >> 2, kce = swz_halo(row, rem & 3);
        const int co = n0 + row;
        hw_voff[it] = co < p.Cout ? ((unsigned)co * (unsigned)p.K + (unsigned)(kwi * p.Cin + kce * EP)) * (unsigned)sizeof(T) : OOB;
    }
    // Left / right image border: the neighbouring LDS row holds the previous / next image row's pixel, the tap must read zeros
    // instead.  The fragment's ADDRESS is redirected to the all-zero row (one select per fragment, before the read is issued)
    // rather than zeroing the fragment's four registers behind the read.
    const int xrow = wm * (BM / WM) + fr, wrow = wn * (BN / WN) + fr;
    const unsigned lbase = lds_addr(&lds[0]);
    const unsigned zrow = lbase + (unsigned)(NS * STAGE) * 16u + (unsigned)fq * 16u;
    bool edge_l[TM], edge_r[TM];
    unsigned zaddr[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + wm * (BM / WM) + i * 16 + fr;
        const int wo = m % p.W;
        edge_l[i] = wo == 0;
        edge_r[i] = wo == p.W - 1;
        zaddr[i] = zrow - (unsigned)(i * 16 * KC * 16);            // frag_read_each adds i * 16 rows back
    }
    unsigned x_rd[3];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) x_rd[kw] = lbase + (unsigned)((xrow + kw) * KC + swz_halo(xrow + kw, fq)) * 16u;
    const unsigned w_rd = lbase + (unsigned)((AS + wrow * KC) + swz_halo(wrow, fq)) * 16u;
    constexpr unsigned STAGE_BYTES = STAGE * 16;
    const int G = 3 * (p.Cin / BK);

    // DMA pieces of one group (six per thread: halo slab 0, 1, [2: wave 0 only], weights of taps 0, 1, 2) go out as three pairs:
    //   pair 0 = slab pieces 0, 1    pair 1 = weight taps 0, 1    pair 2 = weight tap 2 (+ the slab's tail piece)
    unsigned a_off = 0, w_off = 0, ikh = 0;                        // offsets / kh of the group being issued
    int gkh = 0, gci = 0;
    auto next_group = [&]() {
        a_off = (unsigned)(((gkh - 1) * p.W * p.Cin + gci) * (int)sizeof(T));
        w_off = (unsigned)((gkh * 3 * p.Cin + gci) * (int)sizeof(T));
        ikh = (unsigned)gkh;
        gci += BK;
        if (gci >= p.Cin) { gci = 0; ++gkh; }
    };
    const unsigned dmask = (p.dbg & 16) ? 0xfffu : 0xffffffffu;   // ablation (igemm_dbg): 16 = all DMA sources inside one 4-KB window,
    const bool no_dma = p.dbg & 32, no_mfma = p.dbg & 64;          //   32 = no DMA inside the K loop, 64 = no MFMAs, 4 = no epilogue
    auto issue_pair = [&](int pair, int st) {
        uint4* base = &lds[st * STAGE];
        if (no_dma) return;
        if (pair == 0) {
            glds16(rx, base + wbase, ((ha_mask[0] >> ikh) & 1u) ? (ha_voff[0] + a_off) & dmask : OOB);
            glds16(rx, base + wbase + NT, ((ha_mask[1] >> ikh) & 1u) ? (ha_voff[1] + a_off) & dmask : OOB);
        } else if (pair == 1) {
            glds16(rw, base + AS + wbase, hw_voff[0] == OOB ? OOB : (hw_voff[0] + w_off) & dmask);
            glds16(rw, base + AS + wbase + NT, hw_voff[1] == OOB ? OOB : (hw_voff[1] + w_off) & dmask);
        } else {
            glds16(rw, base + AS + wbase + 2 * NT, hw_voff[2] == OOB ? OOB : (hw_voff[2] + w_off) & dmask);
            if (third_pix) glds16(rx, base + wbase + 2 * NT, ((ha_mask[2] >> ikh) & 1u) ? (ha_voff[2] + a_off) & dmask : OOB);
        }
    };
    // Issue schedule (phase = (group g, tap t)):  (g, 0): pair 2 of group g+1   (g, 1): pair 0 of group g+2   (g, 2): pair 1 of g+2.
    // A stage is re-targeted two phases after its last fragment read ((g-1, 2) -> (g, 1)), and at (g, 2) exactly four pieces are
    // younger than group g+1's last one: `vmcnt(4)` there retires group g+1, which is first read one phase later.
    next_group();                                      // group 0
    issue_pair(0, 0); issue_pair(1, 0); issue_pair(2, 0);
    next_group();                                      // group 1: pairs 0, 1 now, pair 2 in phase (0, 0)
    issue_pair(0, 1); issue_pair(1, 1);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // group 0 has landed
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (half_b) {                                      // stagger: the lagging half sits out one barrier interval
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }
    int rs = 0, s1 = 1, s2 = 2;                        // stages of groups g, g + 1, g + 2
    for (int g = 0; g < G; ++g) {
        const unsigned gb = (unsigned)rs * STAGE_BYTES;
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            // ---- load segment: this phase's fragment reads go out first, then the DMA pair
            u32x4_t xf[TM], wf[TN];
            if (t == 1) {
                frag_read_all<TM, KC * 16>(xf, x_rd[1] + gb);
            } else {
                unsigned xa[TM];
#pragma unroll
                for (int i = 0; i < TM; ++i) xa[i] = (t == 0 ? edge_l[i] : edge_r[i]) ? zaddr[i] : x_rd[t] + gb;
                frag_read_each<TM, KC * 16>(xf, xa);
            }
            frag_read_all<TN, KC * 16>(wf, w_rd + gb + (unsigned)(t * BN * KC * 16));
            if (t == 0) {
                if (g + 1 < G) issue_pair(2, s1);
            } else if (g + 2 < G) {
                if (t == 1) next_group();
                issue_pair(t - 1, s2);
            }
            if (t == 2 && g + 1 < G) {
                if (g + 2 < G && !no_dma) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            // ---- compute segment
            frag_wait<TM, TN>(xf, wf);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
            if (!no_mfma) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = Mma<T>::run(wf[j], xf[i], acc[i][j]);
            }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
        const int o = rs; rs = s1; s1 = s2; s2 = o;
    }
    if (!half_b) {                                     // the leading half's matching extra barrier
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }
    igemm_epilogue<T, BM, BN, WM, WN, NS * STAGE * 16>(p, acc, m0, n0, reinterpret_cast<unsigned char*>(&lds[0]));
}

template <typename T, int BN>
__global__ __launch_bounds__(512) void igemm_halo_rs_kernel(ConvDev p) {
    igemm_halo_rs_body<T, BN>(p, (int)(blockIdx.y * gridDim.x + blockIdx.x), (int)gridDim.x, (int)gridDim.y);
}

template <typename T, int BN>
int launch_halo_rs(const ConvDev& d, hipStream_t st) {
    dim3 grid(cdiv(d.M, 256), cdiv(d.Cout, BN));
    hipLaunchKernelGGL((igemm_halo_rs_kernel<T, BN>), grid, dim3(512), 0, st, d);
    ALDI_CHECK_LAUNCH();
    char name[96];
    snprintf(name, sizeof(name), "igemm<bf16,256,%d,4,2,roles,halo>", BN);
    aldi_note_dispatch(name);
    return ALDI_OK;
}

#include "igemm_halo64.h"
#include "igemm_ws.h"

// (the 240-pixel halo tile is sized for TWO workgroups per CU: 6 waves each = 3 waves per SIMD, 80 KB of LDS each)
template <int BM, int NT, bool HALO> constexpr int min_waves_per_simd() { return HALO && BM == 240 ? 2 * NT / 256 : 1; }
template <typename T, int BM, int BN, int WM, int WN, int KC, bool PIPE, bool HALO = false, int EPI = 0, bool LEAN = false>
__global__ __launch_bounds__(WM* WN * 64, (min_waves_per_simd<BM, WM * WN * 64, HALO>())) void igemm_kernel(ConvDev p) {
    igemm_body<T, BM, BN, WM, WN, KC, PIPE, HALO, EPI, LEAN>(p, (int)(blockIdx.y * gridDim.x + blockIdx.x), (int)gridDim.x, (int)gridDim.y);
}

// Several problems of ONE layer shape in one launch -- the student's and the teacher's pass through the same layer (different
// weights, different images), and the same kind of layer on the maps of several pyramid levels (the four FPN output convs, the
// RPN conv on p2..p6: same channels and taps, different H x W): the launches' fixed costs (ramp-up, partial last wave of tiles,
// dependent-launch gap: ~10 us per 3x3 layer at these sizes) are paid once, and the small problems' tiles fill the large one's tail.
constexpr int kMaxConvGroup = 12;
struct ConvGroup {
    int n;
    int wg_begin[kMaxConvGroup + 1];          // first workgroup of each problem (multiples of 8: the XCD-aware tile order assumes it)
    int nmt[kMaxConvGroup], nnt[kMaxConvGroup];
    ConvDev p[kMaxConvGroup];
};
template <typename T, int BM, int BN, int WM, int WN, int KC, bool PIPE, bool HALO = false>
__global__ __launch_bounds__(WM* WN * 64, (min_waves_per_simd<BM, WM * WN * 64, HALO>())) void igemm_group_kernel(ConvGroup G) {
    const int bid = (int)blockIdx.x;
    int i = 0;
    for (int k = 1; k < G.n; ++k)
        if (bid >= G.wg_begin[k]) i = k;
    i = __builtin_amdgcn_readfirstlane(i);
    const int local = bid - G.wg_begin[i];
    if (local >= G.nmt[i] * G.nnt[i]) return;  // alignment padding
    igemm_body<T, BM, BN, WM, WN, KC, PIPE, HALO>(G.p[i], local, G.nmt[i], G.nnt[i]);
}

static thread_local const ConvGroup* g_group = nullptr;      // set by aldi_conv_igemm_group around dispatch<T>()

template <int BM, int BN, int WM, int WN, bool DIRECT, bool ILV = false>
__global__ __launch_bounds__(WM* WN * 64) void igemm_halo64_group_kernel(ConvGroup G) {
    const int bid = (int)blockIdx.x;
    int i = 0;
    for (int k = 1; k < G.n; ++k)
        if (bid >= G.wg_begin[k]) i = k;
    i = __builtin_amdgcn_readfirstlane(i);
    const int local = bid - G.wg_begin[i];
    if (local >= G.nmt[i] * G.nnt[i]) return;  // alignment padding
    igemm_halo64_body<BM, BN, WM, WN, DIRECT, ILV>(G.p[i], local, G.nmt[i], G.nnt[i]);
}

// the 128-byte-slab halo kernel (igemm_halo64.h), alone or over the problems of a group; the direct epilogue (igemm_direct bit 8) when every
// problem's output is plain bf16 with at most scale / shift / ReLU
inline bool halo64_direct_ok(const ConvDev& d) {
    return d.y && !d.y_f32 && d.out_scale == 1 && (d.Cout & 7) == 0 && !d.mask && !d.mask_bits && !d.bits_out && !d.res_mode;
}
template <int BM, int BN, int WM, int WN>
int launch_halo64(const ConvDev& d, hipStream_t st) {
    char name[96];
    constexpr int NT = WM * WN * 64;
    bool direct = (aldi_tuning().igemm_direct & 8) != 0;
    // igemm_halo_ilv: the interleaved K loop (reads / DMA pieces between the MFMAs of a sub-phase; igemm_halo64.h) -- the 8-wave 256 x 256 tile
    constexpr bool HAS_ILV = BM == 256 && BN == 256 && WM == 4 && WN == 2;
    const bool ilv = HAS_ILV && aldi_tuning().igemm_halo_ilv != 0;
    if (g_group) {
        ConvGroup G = *g_group;
        int wg = 0;
        for (int i = 0; i < G.n; ++i) {
            G.p[i].xcd = d.xcd; G.p[i].dbg = d.dbg;
            G.nmt[i] = cdiv(G.p[i].M, BM); G.nnt[i] = cdiv(G.p[i].Cout, BN);
            G.wg_begin[i] = wg;
            wg += (G.nmt[i] * G.nnt[i] + 7) / 8 * 8;
            direct = direct && halo64_direct_ok(G.p[i]);
        }
        for (int i = G.n; i <= kMaxConvGroup; ++i) G.wg_begin[i] = wg;
        if (ilv) {
            if (direct) hipLaunchKernelGGL((igemm_halo64_group_kernel<BM, BN, WM, WN, true, HAS_ILV>), dim3(wg), dim3(NT), 0, st, G);
            else hipLaunchKernelGGL((igemm_halo64_group_kernel<BM, BN, WM, WN, false, HAS_ILV>), dim3(wg), dim3(NT), 0, st, G);
        } else {
            if (direct) hipLaunchKernelGGL((igemm_halo64_group_kernel<BM, BN, WM, WN, true>), dim3(wg), dim3(NT), 0, st, G);
            else hipLaunchKernelGGL((igemm_halo64_group_kernel<BM, BN, WM, WN, false>), dim3(wg), dim3(NT), 0, st, G);
        }
        ALDI_CHECK_LAUNCH();
        snprintf(name, sizeof(name), "igemm_group%d<bf16,%d,%d,%d,%d,halo64%s%s>", G.n, BM, BN, WM, WN, direct ? ",direct" : "", (HAS_ILV && !ilv) ? ",lockstep" : "");
    } else {
        direct = direct && halo64_direct_ok(d);
        dim3 grid(cdiv(d.M, BM), cdiv(d.Cout, BN));
        if (ilv) {
            if (direct) hipLaunchKernelGGL((igemm_halo64_kernel<BM, BN, WM, WN, true, HAS_ILV>), grid, dim3(NT), 0, st, d);
            else hipLaunchKernelGGL((igemm_halo64_kernel<BM, BN, WM, WN, false, HAS_ILV>), grid, dim3(NT), 0, st, d);
        } else {
            if (direct) hipLaunchKernelGGL((igemm_halo64_kernel<BM, BN, WM, WN, true>), grid, dim3(NT), 0, st, d);
            else hipLaunchKernelGGL((igemm_halo64_kernel<BM, BN, WM, WN, false>), grid, dim3(NT), 0, st, d);
        }
        ALDI_CHECK_LAUNCH();
        snprintf(name, sizeof(name), "igemm<bf16,%d,%d,%d,%d,halo64%s%s>", BM, BN, WM, WN, direct ? ",direct" : "", (HAS_ILV && !ilv) ? ",lockstep" : "");
    }
    aldi_note_dispatch(name);
    return ALDI_OK;
}

template <typename T, int BM, int BN, int WM, int WN, int KC, bool PIPE = true, bool HALO = false, int EPI = 0>
int launch(const ConvDev& d, hipStream_t st) {
    if constexpr (EPI != 0) {
        if (!g_group && d.ksplit <= 1) {
            dim3 grid(cdiv(d.M, BM), cdiv(d.Cout, BN));
            bool lean = false;
            if constexpr (!HALO) lean = d.lean != 0;
            if constexpr (!HALO) {
                if (lean) hipLaunchKernelGGL((igemm_kernel<T, BM, BN, WM, WN, KC, PIPE, HALO, EPI, true>), grid, dim3(WM * WN * 64), 0, st, d);
            }
            if (!lean) hipLaunchKernelGGL((igemm_kernel<T, BM, BN, WM, WN, KC, PIPE, HALO, EPI, false>), grid, dim3(WM * WN * 64), 0, st, d);
            ALDI_CHECK_LAUNCH();
            char name[112];
            snprintf(name, sizeof(name), "igemm<%s,%d,%d,%d,%d,%s,%s%s,%s>", "bf16", BM, BN, WM, WN, PIPE ? "pipe" : "flat", HALO ? "halo" : "tap", KC == 8 ? ",k64" : "",
                     EPI == 2 ? "direct+res" : "direct");
            aldi_note_dispatch(name);
            return ALDI_OK;
        }
        return launch<T, BM, BN, WM, WN, KC, PIPE, HALO, 0>(d, st);
    }
    if (g_group) {
        ConvGroup G = *g_group;
        int wg = 0;
        for (int i = 0; i < G.n; ++i) {
            G.p[i].xcd = d.xcd; G.p[i].dbg = d.dbg;
            G.nmt[i] = cdiv(G.p[i].M, BM); G.nnt[i] = cdiv(G.p[i].Cout, BN);
            G.wg_begin[i] = wg;
            wg += (G.nmt[i] * G.nnt[i] + 7) / 8 * 8;
        }
        for (int i = G.n; i <= kMaxConvGroup; ++i) G.wg_begin[i] = wg;
        hipLaunchKernelGGL((igemm_group_kernel<T, BM, BN, WM, WN, KC, PIPE, HALO>), dim3(wg), dim3(WM * WN * 64), 0, st, G);
        ALDI_CHECK_LAUNCH();
        char name[112];
        snprintf(name, sizeof(name), "igemm_group%d<%s,%d,%d,%d,%d,%s,%s%s>", G.n, sizeof(T) == 2 ? "bf16" : "f32", BM, BN, WM, WN, PIPE ? "pipe" : "flat", HALO ? "halo" : "tap",
                 KC == 8 ? ",k64" : "");
        aldi_note_dispatch(name);
        return ALDI_OK;
    }
    dim3 grid(cdiv(d.M, BM), cdiv(d.Cout, BN), d.ksplit > 1 ? d.ksplit : 1);
    hipLaunchKernelGGL((igemm_kernel<T, BM, BN, WM, WN, KC, PIPE, HALO>), grid, dim3(WM * WN * 64), 0, st, d);
    ALDI_CHECK_LAUNCH();
    char name[112];
    snprintf(name, sizeof(name), "igemm<%s,%d,%d,%d,%d,%s,%s%s>%s", sizeof(T) == 2 ? "bf16" : "f32", BM, BN, WM, WN, PIPE ? "pipe" : "flat", HALO ? "halo" : "tap",
             KC == 8 ? ",k64" : "", d.ksplit > 1 ? " splitk" : "");
    aldi_note_dispatch(name);
    return ALDI_OK;
}

// Tile selection.  Every arm is reachable from a test through aldi_set_tuning("igemm_force", ...) and named by
// aldi_last_dispatch(); the thresholds are knobs of the same table (include/aldi_hip.h).
template <typename T>
int dispatch(ConvDev& d, hipStream_t st) {
    const AldiTuning& tn = aldi_tuning();
    d.xcd = tn.igemm_xcd;
    d.dbg = tn.igemm_dbg;
    // the N=2 micro-batch leaves the deep layers (res4/res5, FC heads) with far fewer 128x128 tiles than the
    // 256 CUs: fall back to 64x64 tiles (4x the workgroups) when the big tiling cannot fill the chip
    const long big = (long)cdiv(d.M, 128) * cdiv(d.Cout, 128);
    // igemm_bigtile_min: long-K convs with thousands of tiles are bound by the L2 -> CU path (~31 B/clk/CU measured): the
    // 256x128 tile (8 waves) moves 25 % fewer bytes per flop.
    // igemm_bigtile_k / igemm_lintile_min: plain token GEMMs (ViT / ConvNeXt linears: K >= 768, M in the thousands): the
    // 256x128 tile already pays from ~770 tiles on (+10 % at K = 768, +30 % at K = 3072 measured); the short-K 1x1 convs of
    // the R50 trunk are HBM-bound and stay on 128x128.
    // igemm_halo: 3x3 / stride 1 / pad 1 (every 3x3 of the network): halo form, the pixel tile is loaded once per three taps.
    const int force = tn.igemm_force;     // 0 = heuristics; 1 = 128x128, 2 = 128x64, 3 = 64x64, 4 = 256x128, 5 = 128x16
    // igemm_direct (bit mask: 1 = the 128x64 1x1 / tap tile, 2 = the 64x64 long-K tile, 4 = the 128x64 halo tile): the direct epilogue of
    // the 64-channel tiles (igemm_epilogue_direct) for bf16 outputs in the plain layout; a residual needs the tile that prefetches it
    const bool direct_ok = sizeof(T) == 2 && !g_group && d.y && !d.y_f32 && d.out_scale == 1 && (d.Cout & 7) == 0 && !d.mask && d.ksplit <= 1 &&
                           !(d.mask_bits && (d.scale || d.shift)) && d.Cout >= 64 &&
                           !(d.mask_bits && (d.Cout & 31));        // (the mask bits arrive by a 4-byte LDS-DMA at bit offset (m * Cout + ch): dword-aligned for Cout % 32 == 0 only)
    const int direct = direct_ok ? tn.igemm_direct : 0;
    d.lean = tn.igemm_lean && d.KH * d.KW == 1 && d.stride == 1 && d.pad == 0 && d.K % 64 == 0 && !(d.dbg & (8 | 16)) ? 1 : 0;
    {
        // (fp32 -- the parity mode and the Deformable-DETR step's trunk: the halo form is the same code, 16 channels per group; igemm_halo_f32)
        const bool same3 = d.KH == 3 && d.KW == 3 && d.stride == 1 && d.pad == 1 && d.Ho == d.H && d.Wo == d.W && d.Cin % 32 == 0 && d.out_scale == 1;
        // fp32: OFF by default (igemm_halo_f32 = 0).  From ~400 half-width tiles on the halo form is faster alone (tools/halo_f32_sweep.py:
        // 33 600 px x 256 -> 256 324 -> 228 us, 33 600 x 128 -> 128 162 -> 115; below, the 64 x 64 tap form's four-fold workgroup count wins:
        // 8400 x 256 -> 256 148 vs 158, 2100 x 2048 -> 256 488 vs 647), but it sums K in another order (kh, channels, kw) than the tap form, so
        // a layer would round differently at N = 2 and at N = 6 -- the parity mode's fused-vs-sequential comparison flips discrete decisions --
        // and the Deformable-DETR step (N = 2 maps, few eligible layers) did not move (138 vs 140 ms)
        const bool f32_halo = sizeof(T) == 4 && tn.igemm_halo_f32 > 0 && (long)cdiv(d.M, 128) * cdiv(d.Cout, 64) >= tn.igemm_halo_f32;
        if (tn.igemm_halo && (sizeof(T) == 2 || f32_halo) && same3) {
            if (force == 1) return launch<T, 128, 128, 2, 2, 4, false, true>(d, st);
            if constexpr (sizeof(T) == 2)
                if (force == 2 && (direct & 4) && !d.res_mode) return launch<T, 128, 64, 4, 1, 4, false, true, 1>(d, st);
            if (force == 2) return launch<T, 128, 64, 4, 1, 4, false, true>(d, st);
            if (force == 4) return launch<T, 256, 128, 4, 2, 4, false, true>(d, st);
            if constexpr (sizeof(T) == 2) {
                // 64 x 64 halo tiles (4 waves of 32 x 32): four times the workgroups of the 128 x 128 count -- for the layers whose 128 x 64 tile count sits just
                // above a multiple of the 256 CUs (tools/quant_probe.py: 508 -> 516 workgroups = +23 % time)
                // 96 x 64 halo tiles on THREE waves (32 x 64 per wave, as in the 128 x 64 tile): 4/3 of its workgroups -- tools/quant_probe.py: a CU runs three
                // 128 x 64 workgroups in 1.23 x the time of two, and the mid-size layers of this network give it 2.06 (528 tiles) or 1.03 (264)
                if (force == 17 && (direct & 4) && !d.res_mode) return launch<T, 96, 64, 3, 1, 4, false, true, 1>(d, st);
                if (force == 17) return launch<T, 96, 64, 3, 1, 4, false, true>(d, st);
                if (force == 16 && (direct & 4) && !d.res_mode) return launch<T, 64, 64, 2, 2, 4, false, true, 1>(d, st);
                if (force == 16) return launch<T, 64, 64, 2, 2, 4, false, true>(d, st);
                if (force == 9) return launch<T, 240, 128, 3, 2, 4, false, true>(d, st);
                if (force == 10 && !g_group) return launch_halo_rs<T, 128>(d, st);
                if (force == 11 && d.Cin % 64 == 0) return launch_halo64<256, 256, 4, 2>(d, st);
                if (force == 13 && d.Cin % 64 == 0) return launch_halo64<128, 128, 2, 2>(d, st);
                if (force == 15 && d.Cin % 64 == 0) return launch_halo64<256, 256, 2, 2>(d, st);
            }
            if (force == 0 || force == 3) {      // (no 64x64 halo form; 5 = the 128x16 tap form)
                if (d.Cout <= 64) return launch<T, 128, 64, 4, 1, 4, false, true>(d, st);
                if (big >= tn.igemm_bigtile_min) {
                    // igemm_bigtile 64: 128-byte K slabs on a 256 x 256 tile (igemm_halo64.h) where the channels fill it
                    if constexpr (sizeof(T) == 2)
                        if (tn.igemm_bigtile == 64 && d.Cin % 64 == 0 && d.Cout % 256 == 0) return launch_halo64<256, 256, 4, 2>(d, st);
                    if constexpr (sizeof(T) == 2)
                        if (tn.igemm_bigtile == 65 && d.Cin % 64 == 0 && d.Cout % 256 == 0) return launch_halo64<256, 256, 2, 2>(d, st);
                    if (tn.igemm_bigtile == 1) return launch<T, 128, 128, 2, 2, 4, false, true>(d, st);
                    if constexpr (sizeof(T) == 2)
                        if (tn.igemm_bigtile == 10 && !g_group) return launch_halo_rs<T, 128>(d, st);
                    return launch<T, 256, 128, 4, 2, 4, false, true>(d, st);
                }
                // igemm_halo64_mid: mid-size layers with at least this many 128 x 128 tiles (two workgroups per CU: res3 / res4 conv2 at N = 4,
                // res3 at N = 2) take that tile with 128-byte K slabs (igemm_halo64.h); 0 = never
                if constexpr (sizeof(T) == 2)
                    if (tn.igemm_halo64_mid > 0 && big >= tn.igemm_halo64_mid && d.Cin % 64 == 0 && d.Cout % 128 == 0) return launch_halo64<128, 128, 2, 2>(d, st);
                // below ~1000 128x128 tiles the tile count of this network sits just above a multiple of the 256 CUs (16800 pixels =
                // 131.25 row tiles: 264 / 528 tiles) and the last partial round costs as much as a full one; half-width tiles halve that
                // tail (measured 8-25 % faster on every res3..res5 / FPN p3..p6 3x3 at N = 2 and 4)
                // igemm_halo_small: long-K layers that do not even give every CU one or two 128 x 64 tiles (res5 conv2: 264 tiles at N = 4, 136 at N = 2)
                // take 64 x 64 tiles -- four waves of 32 x 32, four times the workgroups per pixel: 32.9 -> 30.4 / 27.7 -> 24.3 us, bit-identical (same K order;
                // tools/quant_probe.py, profiles/r06_quant_probe.txt).  At res4's K (16 800 px: 528 tiles) and res3's the larger tile wins (31.6 vs 38.3 us).
                // OFF by default (0; 320 selects res5 conv2): in the step, beside the other stream's workgroups, it measured 0.5 % slower (8.07 vs 8.02 ms).
                // igemm_halo96: layers with 200 .. 600 tiles of 128 x 64 (one or two per CU and a few left over: res4 conv2 at both batch sizes, res5 / res3 conv2
                // at one of them) on 96 x 64 three-wave tiles: 2-6 % faster alone, bit-identical (profiles/r06_quant_probe.txt)
                if constexpr (sizeof(T) == 2)
                    if (tn.igemm_halo96 > 0) {
                        const long t64 = (long)cdiv(d.M, 128) * cdiv(d.Cout, 64);
                        if (t64 >= 200 && t64 <= 600) {
                            if ((direct & 4) && !d.res_mode) return launch<T, 96, 64, 3, 1, 4, false, true, 1>(d, st);
                            return launch<T, 96, 64, 3, 1, 4, false, true>(d, st);
                        }
                    }
                if constexpr (sizeof(T) == 2)
                    if (tn.igemm_halo_small > 0 && (long)cdiv(d.M, 128) * cdiv(d.Cout, 64) <= tn.igemm_halo_small && d.Cin >= 512) {
                        if ((direct & 4) && !d.res_mode) return launch<T, 64, 64, 2, 2, 4, false, true, 1>(d, st);
                        return launch<T, 64, 64, 2, 2, 4, false, true>(d, st);
                    }
                if constexpr (sizeof(T) == 2)
                    if ((direct & 4) && !d.res_mode) return launch<T, 128, 64, 4, 1, 4, false, true, 1>(d, st);
                return launch<T, 128, 64, 4, 1, 4, false, true>(d, st);
            }
        }
    }
    // igemm_ws: the short-K 1x1 layers of the trunk (bottleneck expansions / reductions, their data gradients) on the weight-stationary persistent
    // kernel (igemm_ws.h; its epilogue is the direct one: igemm_direct bit 1 turns it off with that); igemm_force 14 forces it wherever it is eligible.
    // With an upsampled residual (FPN laterals) from 4 x igemm_ws_min pixels: p2's lateral 115 -> 107 us, p3's 39.7 -> 41.0 (tools/ws_ab.py)
    if constexpr (sizeof(T) == 2)
        if (!g_group && ws_ok(d) && (force == 14 || (force == 0 && tn.igemm_ws && (tn.igemm_direct & 1) && d.M >= (d.res_mode == 2 ? 4L : 1L) * tn.igemm_ws_min))) return launch_ws(d, st, tn.igemm_ws_wgs);
    if (force == 5 || (force == 0 && d.Cout <= 16)) return launch<T, 128, 16, 4, 1, 4>(d, st);
    if (force == 1) return launch<T, 128, 128, 2, 2, 4>(d, st);
    if constexpr (sizeof(T) == 2) {
        if (force == 2 && (direct & 1)) return d.res_mode ? launch<T, 128, 64, 4, 1, 4, true, false, 2>(d, st) : launch<T, 128, 64, 4, 1, 4, true, false, 1>(d, st);
    }
    if (force == 2) return launch<T, 128, 64, 4, 1, 4>(d, st);
    if (force == 3) return launch<T, 64, 64, 2, 2, 4>(d, st);
    if (force == 4) return launch<T, 256, 128, 4, 2, 4, false>(d, st);
    if constexpr (sizeof(T) == 2) {
        // 128-byte K slabs (64 channels: a full cache line per pixel row and k-step, half the barriers): plain 1x1 / linear
        // layers only (a ragged K tail is handled for those).  igemm_k64_min: long-K layers (res4/res5 reductions, their dgrads,
        // the box head's FCs) run 10-18 % faster on the 64x64 form than on any 32-channel tile (tools/igemm_sweep.py);
        // short-K layers (4 slabs) lose more to the shallower pipeline than they gain.
        const bool plain = d.KH * d.KW == 1 && d.stride == 1 && d.pad == 0;
        if (plain && force == 6) return launch<T, 128, 128, 2, 2, 8, false>(d, st);
        if (plain && force == 7) return launch<T, 128, 64, 4, 1, 8, false>(d, st);
        if (plain && force == 12 && (direct & 1) && !d.res_mode && d.K % 64 == 0) return launch<T, 128, 64, 4, 1, 8, false, false, 1>(d, st);
        const bool lin256 = tn.igemm_tile != 9 && big >= tn.igemm_lintile_min && d.K >= tn.igemm_bigtile_k;     // (the token-GEMM rule below wins)
        if (plain && (force == 8 || (force == 0 && !lin256 && d.Cout > 64 && d.K % 64 == 0 && d.K >= tn.igemm_k64_min))) {
            if ((direct & 2) && !d.res_mode) return launch<T, 64, 64, 2, 2, 8, false, false, 1>(d, st);
            return launch<T, 64, 64, 2, 2, 8, false>(d, st);
        }
    }
    // fp32 (the parity mode; the Deformable-DETR step's arithmetic): the f32-input MFMA runs at 1/16 of the bf16 rate, so a tile's K loop is
    // long and what pays is workgroups, not bytes per flop -- 64 x 64 tiles are as fast or faster than every larger tile on all of that
    // step's shapes (tools/f32_tile_sweep.py: 33 600 px x 128 -> 128 3x3 161 -> 110 us, 16 800 x 512 -> 128 77 -> 52, 44 646 x 256 -> 384
    // 100 -> 85, 44 646 x 1024 -> 256 205 -> 201); same K order per output element as the other tap-form tiles (bit-identical results)
    if (sizeof(T) == 4 && force == 0 && big < tn.igemm_f32_tile64_max) return launch<T, 64, 64, 2, 2, 4>(d, st);
    if (d.Cout <= 64) return launch<T, 128, 64, 4, 1, 4>(d, st);
    if (tn.igemm_tile != 9 && big >= tn.igemm_lintile_min && d.K >= tn.igemm_bigtile_k) {
        // 128-byte K slabs on this tile for plain bf16 layers with K % 64 == 0 (+8-17 % on the box head's FC1 dgrad and the ViT linears,
        // tools/lin_tile_ab.py; igemm_tile 7: the 64-byte slabs)
        if constexpr (sizeof(T) == 2)
            if (tn.igemm_tile != 7 && d.KH * d.KW == 1 && d.stride == 1 && d.pad == 0 && d.K % 64 == 0) return launch<T, 256, 128, 4, 2, 8, false>(d, st);
        return launch<T, 256, 128, 4, 2, 4, false>(d, st);
    }
    if (big < 200) return launch<T, 64, 64, 2, 2, 4>(d, st);
    // short-K layers (the bottlenecks' 1x1 expansions and res3's reductions: K = 128 .. 512, 4-16 slabs) are all prologue and
    // epilogue: half-width tiles (twice the workgroups, half the staging epilogue each) run them 8-13 % faster than 128x128
    // (tools/fc_dgrad_sweep.py: 16800 x 256 -> 1024: 25 -> 23 us, 67200 x 128 -> 512: 31 -> 27 us, 67200 x 512 -> 128: 26 -> 24 us)
    if constexpr (sizeof(T) == 2) {
        if (d.K <= tn.igemm_narrow_k && (direct & 1)) return d.res_mode ? launch<T, 128, 64, 4, 1, 4, true, false, 2>(d, st) : launch<T, 128, 64, 4, 1, 4, true, false, 1>(d, st);
    }
    if (sizeof(T) == 2 && d.K <= tn.igemm_narrow_k) return launch<T, 128, 64, 4, 1, 4>(d, st);
    return launch<T, 128, 128, 2, 2, 4>(d, st);
}

}  // namespace

namespace {
int fill_convdev(const aldi_conv_args* a, ConvDev& d) {
    if (!a || !a->x || !a->w || (!a->y && !a->y_f32)) return aldi_set_error_msg(ALDI_ERR_ARG, "conv_igemm: null pointer");
    const int bk = a->dtype == ALDI_BF16 ? 32 : 16;
    const int ep = a->dtype == ALDI_BF16 ? 8 : 4;
    if (a->dtype != ALDI_BF16 && a->dtype != ALDI_F32) return aldi_set_error_msg(ALDI_ERR_ARG, "conv_igemm: bad dtype");
    if (a->KH * a->KW == 1 ? (a->Cin % ep != 0) : (a->Cin % bk != 0))
        return aldi_set_error_msg(ALDI_ERR_ARG, "conv_igemm: Cin must be a multiple of 32 (bf16) / 16 (f32) for KxK convs; of a 16-B chunk for 1x1");
    if (a->Cout % 4 != 0) return aldi_set_error_msg(ALDI_ERR_ARG, "conv_igemm: Cout must be a multiple of 4");
    if (a->res_mode == 2 && ((a->Ho & 1) || (a->Wo & 1))) return aldi_set_error_msg(ALDI_ERR_ARG, "conv_igemm: upsample residual needs even Ho,Wo");
    if (a->res_mode && !a->res) return aldi_set_error_msg(ALDI_ERR_ARG, "conv_igemm: res_mode set without res");
    d.x = a->x; d.w = a->w; d.y = a->y; d.y_f32 = a->y_f32; d.scale = a->scale; d.shift = a->shift;
    d.res = a->res; d.mask = a->mask;
    d.N = a->N; d.H = a->H; d.W = a->W; d.Cin = a->Cin; d.Cout = a->Cout; d.KH = a->KH; d.KW = a->KW;
    d.stride = a->stride; d.pad = a->pad; d.Ho = a->Ho; d.Wo = a->Wo;
    d.relu = a->relu; d.res_mode = a->res_mode; d.out_scale = a->out_scale < 1 ? 1 : a->out_scale;
    d.OH = a->OH; d.OW = a->OW;
    long M = (long)a->N * a->Ho * a->Wo;
    if (M <= 0 || M > 0x7fffffffL) return aldi_set_error_msg(ALDI_ERR_ARG, "conv_igemm: bad M");
    d.M = (int)M;
    d.K = a->KH * a->KW * a->Cin;
    const size_t esz = a->dtype == ALDI_BF16 ? 2 : 4;
    const size_t xb = (size_t)a->N * a->H * a->W * a->Cin * esz, wb = (size_t)a->Cout * d.K * esz;
    const size_t yb = (size_t)a->N * (d.out_scale > 1 ? (size_t)a->OH * a->OW : (size_t)a->Ho * a->Wo) * a->Cout * esz;
    if (xb >= 0x80000000ull || wb >= 0x80000000ull || yb >= 0x80000000ull)
        return aldi_set_error_msg(ALDI_ERR_ARG, "conv_igemm: operand larger than 2 GiB (32-bit buffer offsets)");
    if (a->KH * a->KW > 16) return aldi_set_error_msg(ALDI_ERR_ARG, "conv_igemm: at most 16 taps");
    d.x_bytes = (unsigned)xb;
    d.w_bytes = (unsigned)wb;
    d.xcd = 0; d.dbg = 0;
    d.ksplit = 0; d.slabs_per_split = 0; d.lean = 0;
    d.mask_bits = static_cast<const unsigned char*>(a->mask_bits);
    d.bits_out = static_cast<unsigned char*>(a->bits_out);
    if ((a->mask_bits || a->bits_out) && (a->dtype != ALDI_BF16 || (a->Cout & 7) || d.out_scale != 1 || a->res_mode == 2 || !a->y || a->y_f32 || a->ksplit > 1))
        return aldi_set_error_msg(ALDI_ERR_ARG, "conv_igemm: bit masks take bf16 outputs with Cout % 8 == 0 in the plain layout (no fp32 output, scatter, upsampled residual or split-K)");
    if (a->mask_bits && a->mask) return aldi_set_error_msg(ALDI_ERR_ARG, "conv_igemm: mask and mask_bits are alternatives");
    return ALDI_OK;
}
}  // namespace

namespace {
// y[m][c] = act(scale[c] * sum_z ws[z][m][c] + shift[c]) -> bf16: the slices are added in slice order (deterministic)
__global__ __launch_bounds__(256) void splitk_finalize_kernel(const float* __restrict__ ws, int ks, long mc, int C, const float* __restrict__ scale,
                                                              const float* __restrict__ shift, int relu, bf16_t* __restrict__ y) {
    const long e = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (e >= mc) return;
    float4 a = *reinterpret_cast<const float4*>(ws + e);
    for (int z = 1; z < ks; ++z) {
        const float4 b = *reinterpret_cast<const float4*>(ws + z * mc + e);
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    const int c = (int)(e % C);
    if (scale) { const float4 s4 = *reinterpret_cast<const float4*>(scale + c); a.x *= s4.x; a.y *= s4.y; a.z *= s4.z; a.w *= s4.w; }
    if (shift) { const float4 s4 = *reinterpret_cast<const float4*>(shift + c); a.x += s4.x; a.y += s4.y; a.z += s4.z; a.w += s4.w; }
    if (relu) { a.x = fmaxf(a.x, 0.f); a.y = fmaxf(a.y, 0.f); a.z = fmaxf(a.z, 0.f); a.w = fmaxf(a.w, 0.f); }
    uint2 o;
    o.x = pack2_bf16(a.x, a.y); o.y = pack2_bf16(a.z, a.w);
    *reinterpret_cast<uint2*>(y + e) = o;
}

int conv_splitk(const aldi_conv_args* a, ConvDev& d, hipStream_t st) {
    const int ks = a->ksplit;
    if (a->dtype != ALDI_BF16 || a->KH * a->KW != 1 || a->stride != 1 || a->pad != 0 || a->res_mode || a->mask || a->y_f32 || !a->y || !a->ws ||
        (a->out_scale > 1) || d.K % (64 * ks) != 0 || ks > 64)
        return aldi_set_error_msg(ALDI_ERR_ARG, "conv_igemm: split-K takes bf16 plain 1x1 / linear layers with K % (64 * ksplit) == 0, a workspace, no res / mask / fp32 output");
    ConvDev s = d;
    s.y = nullptr; s.y_f32 = static_cast<float*>(a->ws); s.scale = nullptr; s.shift = nullptr; s.relu = 0;
    s.ksplit = ks;
    s.xcd = aldi_tuning().igemm_xcd; s.dbg = aldi_tuning().igemm_dbg;
    // igemm_splitk_tile: 0 = 128x128 tiles with 128-byte K slabs (4 waves); 1 = 256x128 tiles, 64-byte slabs (8 waves: two per SIMD
    // also when the launch is one workgroup per CU: FC1 at 2048 rows 86 -> 72 us, at 2000 rows 94 -> 72 us, tools/fc1_splitk_sweep.py)
    const int tile_knob = aldi_tuning().igemm_splitk_tile;       // 2 (default): 256x128 (128-byte slabs) when its launch still has ~one workgroup per CU; 4: the same rule with 64-byte slabs
    const bool big_ok = (long)cdiv(d.M, 256) * cdiv(d.Cout, 128) * ks >= 200;
    if (tile_knob == 3 || (tile_knob == 2 && big_ok)) {
        // 256x128 tiles with 128-byte K slabs (full cache lines per DMA lane group, half the barriers per MFMA of the 64-byte form: FC1 at 2048
        // rows 94 -> 79 us on cold weights, tools/fc1_cold.py)
        s.slabs_per_split = d.K / 64 / ks;
        if (int rc = launch<bf16_t, 256, 128, 4, 2, 8, false>(s, st)) return rc;
    } else if (tile_knob == 1 || (tile_knob == 4 && big_ok)) {
        s.slabs_per_split = d.K / 32 / ks;
        if (int rc = launch<bf16_t, 256, 128, 4, 2, 4, false>(s, st)) return rc;
    } else {
        s.slabs_per_split = d.K / 64 / ks;
        if (int rc = launch<bf16_t, 128, 128, 2, 2, 8, false>(s, st)) return rc;
    }
    const long mc = (long)d.M * d.Cout;
    hipLaunchKernelGGL(splitk_finalize_kernel, dim3(cdiv(mc / 4, 256)), dim3(256), 0, st, static_cast<const float*>(a->ws), ks, mc, d.Cout, d.scale, d.shift,
                       d.relu, static_cast<bf16_t*>(a->y));
    ALDI_CHECK_LAUNCH();
    return ALDI_OK;
}
}  // namespace

extern "C" int aldi_conv_igemm(const aldi_conv_args* a, aldi_stream_t stream) {
    ConvDev d;
    if (int rc = fill_convdev(a, d)) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (a->ksplit > 1) return conv_splitk(a, d, st);
    if (a->dtype == ALDI_BF16) return dispatch<bf16_t>(d, st);
    return dispatch<float>(d, st);
}

extern "C" int aldi_conv_igemm_group(const aldi_conv_args* args, int n, aldi_stream_t stream) {
    if (!args || n < 1) return aldi_set_error_msg(ALDI_ERR_ARG, "conv_igemm_group: no problems");
    bool same = n <= kMaxConvGroup && aldi_tuning().igemm_group;
    for (int i = 1; i < n && same; ++i) {
        const aldi_conv_args &a = args[0], &b = args[i];
        // one layer shape: everything that selects code paths inside the kernel template is equal; N, H x W (pyramid levels) and the
        // tensors differ -- the tile heuristics look at the pixel count, Cout, K and the conv geometry ("same" padding or not) only
        same = a.dtype == b.dtype && a.Cin == b.Cin && a.Cout == b.Cout && a.KH == b.KH && a.KW == b.KW &&
               a.stride == b.stride && a.pad == b.pad && (a.Ho == a.H) == (b.Ho == b.H) && (a.Wo == a.W) == (b.Wo == b.W) && a.out_scale == b.out_scale &&
               (a.y != nullptr) == (b.y != nullptr) && (a.y_f32 != nullptr) == (b.y_f32 != nullptr);
    }
    if (!same || n == 1) {
        for (int i = 0; i < n; ++i)
            if (int rc = aldi_conv_igemm(&args[i], stream)) return rc;
        return ALDI_OK;
    }
    static thread_local ConvGroup G;
    G.n = n;
    long Msum = 0;
    for (int i = 0; i < n; ++i) {
        if (int rc = fill_convdev(&args[i], G.p[i])) return rc;
        Msum += G.p[i].M;
    }
    for (int i = 1; i < n; ++i)         // largest problem first: the small ones' tiles fill its tail
        for (int j = i; j > 0 && G.p[j].M > G.p[j - 1].M; --j) { const ConvDev t_ = G.p[j]; G.p[j] = G.p[j - 1]; G.p[j - 1] = t_; }
    // the tile template is chosen for the COMBINED pixel count (the heuristics look at M, Cout, K and the conv geometry only)
    ConvDev d = G.p[0];
    d.M = (int)(Msum > 0x7fffffffL ? 0x7fffffffL : Msum);
    hipStream_t st = static_cast<hipStream_t>(stream);
    g_group = &G;
    const int rc = args[0].dtype == ALDI_BF16 ? dispatch<bf16_t>(d, st) : dispatch<float>(d, st);
    g_group = nullptr;
    return rc;
}
