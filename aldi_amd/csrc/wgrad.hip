// Weight-gradient GEMM on the matrix cores:
//
//   dW[co][kh][kw][ci] += scale[co] * sum_{pixels p} g[p][co] * x[pix(p,kh,kw)][ci]
//
// The reduction runs over PIXELS, which are the slow (strided) dimension of both NHWC
// operands, while an MFMA fragment wants 8 consecutive reduction elements per lane.  bf16
// path: every thread loads an 8-pixel x 8-channel block (8 x 16 B, full 128-B lines per
// wave instruction), transposes it in registers and writes 8 x 16 B to an LDS image
// [channel][64 pixels]; fragments are then plain ds_read_b128.  The 128-B LDS rows are
// XOR-swizzled (chunk ^ (row>>1)&7) so both the 8-lane write groups and the 16-lane read
// groups are conflict free.  fp32 path (parity mode): mfma_f32_16x16x4f32 takes one float
// per lane, so the LDS image stays [pixel][channel] and no transpose is needed.
//
// Split-K over pixel ranges (grid.z); partial tiles are accumulated into the fp32 gradient
// buffer with hardware float atomics, which is also how micro-steps accumulate.
//
// Replaces cuDNN/MIOpen wgrad + torch Linear weight grad reached through autograd from
// aldi/trainer.py:79 (`trainer.do_backward`).
#include "common.h"
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>
#include <stdio.h>
#include <stdlib.h>

namespace {

struct WgDev {
    const void* x; const void* g; float* dw; const float* scale; float* db;
    int N, H, W, Cin, Cout, KH, KW, stride, pad, Ho, Wo;
    int M, K, pix_per_split, ident, xcd, dbg;
    unsigned x_bytes, g_bytes, dw_bytes;
    // ordered epilogue (no float atomics): a pixel split writes its partial tile to `ws` (fragment order, 16 B per lane) and its
    // partial bias sums to `wsb`; wgrad_finalize_kernel adds the splits IN ORDER to dw / db.  splits == 1: the only owner of a tile
    // adds to dw with plain loads and stores.  ordered == 0: the float-atomic epilogue (callers without a workspace).
    float* ws; float* wsb;
    int splits, ordered;
};

__device__ __forceinline__ int swz8(int row, int c) { return c ^ ((row >> 1) & 7); }

// out[c] = pixels 0..7 of channel c, from in[p] = channels 0..7 of pixel p (16-bit elements)
__device__ __forceinline__ void transpose8x8_b16(const uint4 in[8], uint4 out[8]) {
    const uint32_t* r = reinterpret_cast<const uint32_t*>(in);   // r[p*4+d]
    uint32_t* o = reinterpret_cast<uint32_t*>(out);              // o[c*4+j]
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            uint32_t a = r[(2 * j) * 4 + d], b = r[(2 * j + 1) * 4 + d];
            o[(2 * d) * 4 + j] = (a & 0xffffu) | (b << 16);
            o[(2 * d + 1) * 4 + j] = (a >> 16) | (b & 0xffff0000u);
        }
}

// ------------------------------------------------------------------------------------ bf16
// 128 (co) x 128 (kk) tile, 64 pixels per slab, 4 waves (2x2), wave tile 64x64.
__global__ __launch_bounds__(256) void wgrad_bf16_kernel(WgDev p) {
    constexpr int BP = 64;
    __shared__ uint4 lds[2 * 128 * 8];   // [A rows 0..127 | B rows 0..127] x 8 chunks (32 KB)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // uniform loader role
    const int co0 = blockIdx.x * 128, kk0 = blockIdx.y * 128;
    const int pbeg = blockIdx.z * p.pix_per_split;
    const int pend = min(p.M, pbeg + p.pix_per_split);

    // loader role: waves 0,1 -> g tile (channels = co); waves 2,3 -> x tile (channels = kk)
    const bool isB = wave >= 2;
    const int pg = lane & 7;                         // pixel group (8 pixels) inside the slab
    const int cc = (wave & 1) * 8 + (lane >> 3);     // 8-channel chunk inside the 128-wide tile
    int ch, kh = 0, kw = 0, ci = 0;
    bool ch_ok;
    if (!isB) { ch = co0 + cc * 8; ch_ok = ch < p.Cout; }
    else {
        ch = kk0 + cc * 8; ch_ok = ch < p.K;
        int tap = ch / p.Cin; ci = ch - tap * p.Cin; kh = tap / p.KW; kw = tap - kh * p.KW;
    }

    const int wm = wave >> 1, wn = wave & 1;
    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fq = lane >> 4;

    // raw buffer loads: 32-bit byte offsets, out-of-range offsets return zeros (no branches around the loads)
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    // built from wave-uniform scalars only: a descriptor the compiler believes divergent costs a waterfall loop per load
    const uintptr_t base = reinterpret_cast<uintptr_t>(isB ? p.x : p.g);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<void*>(((uintptr_t)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(base >> 32)) << 32) |
                                (uintptr_t)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)base)),
        0, __builtin_amdgcn_readfirstlane(isB ? p.x_bytes : p.g_bytes), 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    const int row_elems = isB ? p.Cin : p.Cout;
    uint4 in[8], out[8];
    auto load_slab = [&](int p0) {
        int pp = p0 + pg * 8;
        int n = 0, ho = 0, wo = 0;
        if (isB && !p.ident) {
            int q = min(pp, p.M - 1);
            n = q / (p.Ho * p.Wo);
            int r = q - n * (p.Ho * p.Wo);
            ho = r / p.Wo; wo = r - ho * p.Wo;
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            int px = pp + r;
            bool ok = ch_ok && px < pend;
            unsigned off;
            if (!isB || p.ident) off = (unsigned)((long)px * row_elems + (isB ? ci : ch)) * 2u;
            else {
                int hi = ho * p.stride - p.pad + kh, wi = wo * p.stride - p.pad + kw;
                ok = ok && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
                off = (unsigned)((((long)n * p.H + hi) * p.W + wi) * p.Cin + ci) * 2u;
            }
            u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, ok ? off : OOB, 0, 0);
            in[r] = make_uint4(v.x, v.y, v.z, v.w);
            if (++wo == p.Wo) { wo = 0; if (++ho == p.Ho) { ho = 0; ++n; } }
        }
    };
    if (pbeg < pend) load_slab(pbeg);
    for (int p0 = pbeg; p0 < pend; p0 += BP) {
        transpose8x8_b16(in, out);
        __syncthreads();   // previous slab's fragment reads are done
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            int row = cc * 8 + c;
            lds[((isB ? 128 : 0) + row) * 8 + swz8(row, pg)] = out[c];
        }
        if (p0 + BP < pend) load_slab(p0 + BP);   // next slab's global loads fly under this slab's MFMAs
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            uint4 af[4], bfr[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int row = wm * 64 + i * 16 + fr;
                af[i] = lds[row * 8 + swz8(row, ks * 4 + fq)];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int row = wn * 64 + j * 16 + fr;
                bfr[j] = lds[(128 + row) * 8 + swz8(row, ks * 4 + fq)];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8_t*>(&af[i]),
                                                                         *reinterpret_cast<bf16x8_t*>(&bfr[j]), acc[i][j], 0, 0, 0);
        }
    }
    // D[row = co][col = kk]
    {   // split-K partial tile -> fp32 gradient: 64 fire-and-forget buffer atomics per lane, out-of-tile lanes dropped
        const __amdgpu_buffer_rsrc_t rdw = make_rsrc_uniform(p.dw, p.dw_bytes);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = co0 + wm * 64 + i * 16 + fq * 4 + r;
                const bool cok = co < p.Cout;
                const float sc = p.scale ? p.scale[cok ? co : 0] : 1.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int kk = kk0 + wn * 64 + j * 16 + fr;
                    const unsigned off = (cok && kk < p.K) ? ((unsigned)co * (unsigned)p.K + (unsigned)kk) * 4u : kBufOOB;
                    buf_atomic_add_f32(rdw, off, acc[i][j][r] * sc);
                }
            }
    }
}

// Lean form of the same kernel for the shapes the network actually has (1x1 stride-1 convs / FC, and stride-1 "same"
// KxK convs with Cin % 64 == 0).  wgrad_bf16_kernel above spends ~800 instructions per 32 MFMAs (64-bit gather addresses,
// per-load border tests, mask/shift/or transposes); wave64 VALU ops cost 4 cycles each, so it is issue bound at ~14 % of
// the MFMA peak.  Here:
//   * "same" geometry makes the tap offset linear in the pixel index: offset = (pixel + dh*W + dw) * Cin + ci, mod 2^32;
//   * border validity depends on the pixel only and the tap is wave uniform (Cin % 64 == 0), so each lane tracks ONE pixel
//     of the slab incrementally, the wave ballots, and the per-load lane mask is a byte of the ballot replicated on the
//     scalar unit -> one v_cndmask per load picks the out-of-range offset (buffer loads return zeros there);
//   * the g operand needs no tests at all: its buffer descriptor ends at the split's last pixel;
//   * the 8x8 16-bit transposes are v_perm_b32, one per output register.
// TCO x TKK output tile (channels of g x (tap, channel) of x), 64 pixels per slab; (TCO + TKK) / 64 waves, each loading 64
// channels of one operand and owning a (TCO / WM) x (TKK / WN) piece of the accumulator.
// What a workgroup does with its finished TCO x TKK partial tile (acc: the MFMA accumulators of this wave, accb: the ones-column
// product = the bias gradient, valid where do_bias).  D[row = co][col = kk]; lane (fr, fq) of fragment (i, j) holds rows
// 4 fq .. 4 fq + 3 of column fr.
template <int TCO, int TKK, int WM, int WN>
__device__ __forceinline__ void wgrad_finish_tile(const WgDev& p, const f32x4_t (&acc)[TCO / WM / 16][TKK / WN / 16], const f32x4_t (&accb)[TCO / WM / 16],
                                                  bool do_bias, int bx, int by, int bz, int wave, int lane) {
    constexpr int TM = TCO / WM / 16, TN = TKK / WN / 16, NW = WM * WN;
    const int wm = wave / WN, wn = wave % WN, fr = lane & 15, fq = lane >> 4;
    const int co0 = bx * TCO, kk0 = by * TKK;
    if (p.dbg & 1) {     // ablation (wgrad_dbg): no epilogue -- every accumulator stays live through one sum
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) t += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
        if (t == 123.456f) p.dw[0] = t;
        return;
    }
    if (p.ordered && p.splits > 1) {
        // partial tile -> workspace, one 1-KB wave store per fragment: [tile][split][wave][fragment][lane] x 16 B
        const int gx = (p.Cout + TCO - 1) / TCO;
        const long t = (long)by * gx + bx;
        float4* dst = reinterpret_cast<float4*>(p.ws) + ((t * p.splits + bz) * NW + wave) * (long)(TM * TN * 64) + lane;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) dst[(i * TN + j) * 64] = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        if (do_bias && fr == 0) {       // [co tile][split][TCO] floats
            float* b = p.wsb + ((long)bx * p.splits + bz) * TCO + wm * (TCO / WM) + fq * 4;
#pragma unroll
            for (int i = 0; i < TM; ++i) *reinterpret_cast<float4*>(b + i * 16) = make_float4(accb[i][0], accb[i][1], accb[i][2], accb[i][3]);
        }
        return;
    }
    if (do_bias && fr == 0) {     // every column of the ones product holds the sum: column 0's lanes add it to the bias gradient
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = co0 + wm * (TCO / WM) + i * 16 + fq * 4 + r;
                if (co < p.Cout) {
                    if (p.ordered) p.db[co] += accb[i][r];          // the only pixel range of this co tile
                    else unsafeAtomicAdd(p.db + co, accb[i][r]);
                }
            }
    }
    const __amdgpu_buffer_rsrc_t rdw = make_rsrc_uniform(p.dw, p.dw_bytes);
    if (p.ordered) {   // sole owner of the tile: plain read-modify-write (out-of-tile lanes read zeros and drop their store)
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            float old_[4][TN];
            unsigned off[4][TN];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = co0 + wm * (TCO / WM) + i * 16 + fq * 4 + r;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int kk = kk0 + wn * (TKK / WN) + j * 16 + fr;
                    off[r][j] = (co < p.Cout && kk < p.K) ? ((unsigned)co * (unsigned)p.K + (unsigned)kk) * 4u : kBufOOB;
                    old_[r][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rdw, off[r][j], 0, 0));
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = co0 + wm * (TCO / WM) + i * 16 + fq * 4 + r;
                const float sc = p.scale ? p.scale[co < p.Cout ? co : 0] : 1.f;
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, old_[r][j] + acc[i][j][r] * sc), rdw, off[r][j], 0, 0);
            }
        }
        return;
    }
    // split-K partial tile -> fp32 gradient: fire-and-forget buffer atomics, out-of-tile lanes dropped
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = co0 + wm * (TCO / WM) + i * 16 + fq * 4 + r;
            const bool cok = co < p.Cout;
            const float sc = p.scale ? p.scale[cok ? co : 0] : 1.f;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int kk = kk0 + wn * (TKK / WN) + j * 16 + fr;
                const unsigned off = (cok && kk < p.K) ? ((unsigned)co * (unsigned)p.K + (unsigned)kk) * 4u : kBufOOB;
                buf_atomic_add_f32(rdw, off, acc[i][j][r] * sc);
            }
        }
}

template <int TCO, int TKK, int WM, int WN, bool DB = false>
__device__ __forceinline__ void wgrad_bf16_lean_tile(const WgDev& p, uint4* lds, int bx, int by, int bz);

template <int TCO, int TKK, int WM, int WN, bool DB = false>
__device__ __forceinline__ void wgrad_bf16_lean_body(const WgDev& p, uint4* lds) {
    // XCD-aware order: workgroup b runs on XCD b % 8; give every XCD a contiguous range of (split, tile) pairs, tile
    // fastest, so the tiles of one pixel range (which re-read the same x / g rows) share one L2.
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    {
        const int gx = gridDim.x, gy = gridDim.y, total = gx * gy * gridDim.z;
        int bid = bx + gx * (by + gy * bz);
        if (p.xcd) {
            const int q = total >> 3, r = total & 7, xcd = bid & 7, idx = bid >> 3;
            bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        }
        bx = bid % gx; bid /= gx;
        by = bid % gy; bz = bid / gy;
    }
    wgrad_bf16_lean_tile<TCO, TKK, WM, WN, DB>(p, lds, bx, by, bz);
}

// DB: two LDS images (slab s in image s & 1) and ONE barrier per slab: a wave writes slab s + 1 while others still read slab s; the
// image it overwrites was last read two slabs ago, i.e. before everybody's previous barrier.
template <int TCO, int TKK, int WM, int WN, bool DB>
__device__ __forceinline__ void wgrad_bf16_lean_tile(const WgDev& p, uint4* lds, int bx, int by, int bz) {
    static_assert(WM * WN * 64 == TCO + TKK, "one loader wave per 64 channels");
    constexpr int NGW = TCO / 64;                    // g-loader waves
    constexpr int TM = TCO / WM / 16, TN = TKK / WN / 16;
    constexpr int BP = 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform role -> scalar descriptors
    const int co0 = bx * TCO, kk0 = by * TKK;
    const int pbeg = bz * p.pix_per_split;
    const int pend = min(p.M, pbeg + p.pix_per_split);
    if (pbeg >= pend) return;

    const bool isB = wave >= NGW;                    // the first TCO/64 waves load the g tile, the rest the x tile
    const int pg = lane & 7;                         // 8-pixel group inside the slab
    const int cc = (isB ? wave - NGW : wave) * 8 + (lane >> 3);     // 8-channel chunk inside this operand's tile
    constexpr unsigned OOB = 0x80000000u;
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

    // per-lane byte offset of (pixel pbeg + pg*8, this lane's channel chunk); later pixels / slabs are uniform adds
    unsigned off0;
    int dh = 0, dw = 0;
    if (!isB) {
        const int ch = co0 + cc * 8;
        off0 = ch < p.Cout ? ((unsigned)(pbeg + pg * 8) * (unsigned)p.Cout + (unsigned)ch) * 2u : OOB;
    } else {
        const int ch = kk0 + cc * 8;
        const int tap = ch / p.Cin, ci = ch - tap * p.Cin;
        const int kh = tap / p.KW, kw = tap - kh * p.KW;
        dh = kh - p.pad; dw = kw - p.pad;
        off0 = ch < p.K ? ((unsigned)(pbeg + pg * 8 + dh * p.W + dw) * (unsigned)p.Cin + (unsigned)ci) * 2u : OOB;
    }
    // everything the load instructions take from scalar registers is built from wave-uniform values only (a descriptor the
    // compiler believes divergent costs a readfirstlane "waterfall" loop around every load)
    const bool track = isB && !p.ident;             // border validity needed (KxK x-loader waves)
    const unsigned row_bytes = (unsigned)(isB ? p.Cin : p.Cout) * 2u;
    const unsigned g_lim = (unsigned)pend * (unsigned)p.Cout * 2u;    // g rows >= pend read as zeros
    const unsigned lim = isB ? p.x_bytes : (g_lim < p.g_bytes ? g_lim : p.g_bytes);
    const uintptr_t base = reinterpret_cast<uintptr_t>(isB ? p.x : p.g);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<void*>(((uintptr_t)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(base >> 32)) << 32) |
                                (uintptr_t)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)base)),
        0, __builtin_amdgcn_readfirstlane(lim), 0x00020000);
    // the pixel this lane tracks for the ballot: bit l of the ballot <-> slab pixel (l&7)*8 + (l>>3)
    int ho = 0, wo = 0;
    if (track) {
        int q = pbeg + pg * 8 + (lane >> 3);
        int r = q % (p.H * p.W);
        ho = r / p.W; wo = r - ho * p.W;
    }
    const int adv_h = BP / p.W, adv_w = BP - adv_h * p.W;

    uint4 in[8], out[8];
    unsigned off = off0;
    auto load_slab = [&]() {
        unsigned long long bal = ~0ull;
        if (track) {
            bal = __ballot((unsigned)(ho + dh) < (unsigned)p.H && (unsigned)(wo + dw) < (unsigned)p.W);
            wo += adv_w; ho += adv_h;
            if (wo >= p.W) { wo -= p.W; ++ho; }
            while (ho >= p.H) ho -= p.H;
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            unsigned o = off + (unsigned)r * row_bytes;
            if (track) {
                const unsigned half = __builtin_amdgcn_readfirstlane((unsigned)(bal >> (r < 4 ? 0 : 32)));
                const unsigned rep = ((half >> (8 * (r & 3))) & 0xffu) * 0x01010101u;                 // scalar unit
                const unsigned long long lm = ((unsigned long long)rep << 32) | rep;
                asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(o) : "v"(OOB), "v"(o), "s"(lm));
            }
            u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, o, 0, 0);
            in[r] = make_uint4(v.x, v.y, v.z, v.w);
        }
        off += BP * row_bytes;
    };

    const int wm = wave / WN, wn = wave % WN;
    f32x4_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    // Bias gradient db[co] = sum over pixels of g[p][co] in the same pass: it is the product of the g tile with a column of ONES,
    // i.e. TM more MFMAs per k-step whose B fragment is a constant register (no LDS read), in the waves of the first kk tile's first
    // column only -- instead of a separate column-sum launch that reads g again (13 launches per step).
    const bool do_bias = p.db != nullptr && by == 0 && wn == 0;
    f32x4_t accb[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) accb[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const uint4 ones4 = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);      // eight bf16 1.0
    const int fr = lane & 15, fq = lane >> 4;
    // LDS slots are loop invariant and differ only by immediates / one XOR: fragment rows i*16 apart share the swizzle
    // term ((row>>1)&7 sees only fr), the second k-step flips chunk bit 2; write rows c share pg ^ (cc&1)*4 up to c>>1.
    const int ra = wm * (TCO / WM) + fr, rb = wn * (TKK / WN) + fr;
    const int ia0 = ra * 8 + swz8(ra, fq);
    const int ib0 = (TCO + rb) * 8 + swz8(rb, fq);
    const int iw = ((isB ? TCO : 0) + cc * 8) * 8;
    const int wu = pg ^ ((cc & 1) << 2);

    load_slab();
    for (int p0 = pbeg; p0 < pend; p0 += BP) {
        {   // out[c] = pixels 0..7 of channel c
            const uint32_t* rr = reinterpret_cast<const uint32_t*>(in);
            uint32_t* oo = reinterpret_cast<uint32_t*>(out);
#pragma unroll
            for (int d = 0; d < 4; ++d)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t a = rr[(2 * j) * 4 + d], b = rr[(2 * j + 1) * 4 + d];
                    oo[(2 * d) * 4 + j] = __builtin_amdgcn_perm(b, a, 0x05040100u);       // (a & 0xffff) | (b << 16)
                    oo[(2 * d + 1) * 4 + j] = __builtin_amdgcn_perm(b, a, 0x07060302u);   // (a >> 16) | (b & 0xffff0000)
                }
        }
        if constexpr (!DB) __syncthreads();   // previous slab's fragment reads are done
        uint4* img = DB ? lds + ((p0 - pbeg) / BP & 1) * ((TCO + TKK) * 8) : lds;
#pragma unroll
        for (int c = 0; c < 8; ++c) img[iw + c * 8 + (wu ^ (c >> 1))] = out[c];
        if (p0 + BP < pend) load_slab();   // next slab's global loads fly under this slab's MFMAs
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            uint4 af[TM], bfr[TN];
            const int ia = ia0 ^ (ks * 4), ib = ib0 ^ (ks * 4);
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = img[ia + i * 128];
#pragma unroll
            for (int j = 0; j < TN; ++j) bfr[j] = img[ib + j * 128];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8_t*>(&af[i]),
                                                                         *reinterpret_cast<bf16x8_t*>(&bfr[j]), acc[i][j], 0, 0, 0);
            if (do_bias) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8_t*>(&af[i]), *reinterpret_cast<const bf16x8_t*>(&ones4), accb[i], 0, 0, 0);
            }
        }
    }
    wgrad_finish_tile<TCO, TKK, WM, WN>(p, acc, accb, do_bias, bx, by, bz, wave, lane);
}

// <= 168 VGPRs: three workgroups per CU
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) void wgrad_bf16_lean_kernel(WgDev p) {
    __shared__ uint4 lds[2 * 128 * 8];
    wgrad_bf16_lean_body<128, 128, 2, 2>(p, lds);
}
// Grouped form: the weight gradients of SEVERAL layers in one launch.  A bottleneck stage's layers are small GEMMs (16-36 output
// tiles each) over the same 16800 pixels; launched one by one each needs an 11-24-way pixel split to occupy the chip, and every
// split ends in 16 K float atomics -- 20-50 % of the kernel time (tools/wgrad_sweep.py) -- plus a launch and a tail per layer.
// The backward pass does not need them one by one (nothing reads a weight gradient before the optimizer), so the engine
// collects a stage's layers and launches them together: hundreds of tiles, (almost) no pixel split, a handful of atomics.
constexpr int kMaxGroup = 24;
struct WgGroup {
    int n;
    int wg_begin[kMaxGroup + 1];      // first workgroup of each problem; wg_begin[n] = grid size
    int gx[kMaxGroup], gy[kMaxGroup], gz[kMaxGroup]; // output tiles and pixel splits of each problem (its workgroups: tile-fastest, then split)
    WgDev p[kMaxGroup];
};
// Which (problem, tile, pixel split) does workgroup `bid` of a grouped launch work on.  XCD-aware order over the WHOLE group: workgroup b
// runs on XCD b % 8 (observed dispatch); every XCD gets one contiguous range of the group's (problem, split, tile) list, tile fastest,
// so the workgroups resident in an XCD at the same time are neighbours in that list -- the tiles of one pixel range, which re-read the
// same x / g rows, out of ONE L2 -- also for the small problems (4-16 tiles) that a per-problem order would deal out one per XCD.
__device__ __forceinline__ void wgrad_group_pick(const WgGroup& G, int& i, int& tx, int& ty, int& bz) {
    const int W = G.wg_begin[G.n];
    int g = (int)blockIdx.x;
    if (G.p[0].xcd) {
        const int q = W >> 3, r = W & 7, xcd = g & 7, idx = g >> 3;
        g = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    i = 0;
    for (int k = 1; k < G.n; ++k)
        if (g >= G.wg_begin[k]) i = k;
    i = __builtin_amdgcn_readfirstlane(i);
    const int local = g - G.wg_begin[i];
    const int tiles = G.gx[i] * G.gy[i];
    const int t = local % tiles;
    bz = local / tiles;
    tx = t % G.gx[i]; ty = t / G.gx[i];
}
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) void wgrad_bf16_lean_group_kernel(WgGroup G) {
    __shared__ uint4 lds[2 * 128 * 8];
    // longest problems first in the launch order is the host's job; here: which problem does this workgroup belong to
    int i, tx, ty, bz;
    wgrad_group_pick(G, i, tx, ty, bz);
    wgrad_bf16_lean_tile<128, 128, 2, 2>(G.p[i], lds, tx, ty, bz);
}

// the same with two LDS images and one barrier per slab (64 KB: two workgroups per CU)
__global__ __launch_bounds__(256) void wgrad_bf16_lean_group_db_kernel(WgGroup G) {
    __shared__ uint4 lds[2 * 2 * 128 * 8];
    int i, tx, ty, bz;
    wgrad_group_pick(G, i, tx, ty, bz);
    wgrad_bf16_lean_tile<128, 128, 2, 2, true>(G.p[i], lds, tx, ty, bz);
}

// 256 x 256 tile, 8 waves (128 x 64 each), one workgroup per CU: half the L2 -> CU bytes per FLOP, for layers whose
// pixel ranges are long enough to amortise the 256-KB atomic epilogue
__global__ __launch_bounds__(512) void wgrad_bf16_big_kernel(WgDev p) {
    __shared__ uint4 lds[2 * 256 * 8];
    wgrad_bf16_lean_body<256, 256, 2, 4>(p, lds);
}
// Grouped form of the 256 x 256 tile: a stage's layers whose Cout and K are multiples of 256 in ONE launch of (about) one
// workgroup per CU.  The 128 x 128 kernel stages twice the operand bytes per MFMA through registers and LDS and is bound by
// that (~450 TFLOP/s in its loop against ~850 for this tile); alone, a res4 / res5 layer has 4-36 tiles of this size and
// would need a 7-60-way pixel split to occupy the chip, together they are 100-240 tiles: one or two pixel ranges each.
__global__ __launch_bounds__(512) void wgrad_bf16_big_group_kernel(WgGroup G) {
    __shared__ uint4 lds[2 * 256 * 8];
    int i, tx, ty, bz;
    wgrad_group_pick(G, i, tx, ty, bz);
    wgrad_bf16_lean_tile<256, 256, 2, 4>(G.p[i], lds, tx, ty, bz);
}

// Second pass of the ordered epilogue: dw += scale * (split 0 + split 1 + ...), db += (...), the splits in index order -- the
// same bits on every run, and plain loads / stores.  One 256-thread workgroup per (tile, fragment, 4 waves of the producer).
struct WgFinItem {
    const float* ws; const float* wsb; float* dw; float* db; const float* scale;
    int Cout, K, gx, gy, S, big;
};
struct WgFin {
    int n;
    int wg_begin[kMaxGroup + 1];
    WgFinItem it[kMaxGroup];
};
template <int TCO, int TKK, int WM, int WN>
__device__ __forceinline__ void wgrad_finalize_part(const WgFinItem& it, int local) {
    constexpr int TM = TCO / WM / 16, TN = TKK / WN / 16, NW = WM * WN, NF = TM * TN, HALVES = NW / 4;
    const int S = it.S;
    const int nblk = it.gx * it.gy * NF * HALVES;
    if (local >= nblk) {        // the item's last workgroup: bias gradient
        if (!it.db || local > nblk) return;
        for (int co = threadIdx.x; co < it.Cout; co += 256) {
            const float* b = it.wsb + (long)(co / TCO) * S * TCO + co % TCO;
            float sum = 0.f;
            for (int k = 0; k < S; ++k) sum += b[(long)k * TCO];
            it.db[co] += sum;
        }
        return;
    }
    const int h = local % HALVES, f = (local / HALVES) % NF, t = local / (HALVES * NF);
    const int wave = h * 4 + (int)(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const float4* src = reinterpret_cast<const float4*>(it.ws) + (((long)t * S) * NW + wave) * (long)(NF * 64) + f * 64 + lane;
    float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
    int k = 0;
    for (; k + 4 <= S; k += 4) {      // four loads in flight, added in split order
        const float4 v0 = src[(long)(k + 0) * NW * NF * 64], v1 = src[(long)(k + 1) * NW * NF * 64];
        const float4 v2 = src[(long)(k + 2) * NW * NF * 64], v3 = src[(long)(k + 3) * NW * NF * 64];
        sum.x += v0.x; sum.y += v0.y; sum.z += v0.z; sum.w += v0.w;
        sum.x += v1.x; sum.y += v1.y; sum.z += v1.z; sum.w += v1.w;
        sum.x += v2.x; sum.y += v2.y; sum.z += v2.z; sum.w += v2.w;
        sum.x += v3.x; sum.y += v3.y; sum.z += v3.z; sum.w += v3.w;
    }
    for (; k < S; ++k) {
        const float4 v = src[(long)k * NW * NF * 64];
        sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
    }
    const int bx = t % it.gx, by = t / it.gx, wm = wave / WN, wn = wave % WN, i = f / TN, j = f % TN, fr = lane & 15, fq = lane >> 4;
    const int kk = by * TKK + wn * (TKK / WN) + j * 16 + fr;
    if (kk >= it.K) return;
    const float v[4] = {sum.x, sum.y, sum.z, sum.w};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int co = bx * TCO + wm * (TCO / WM) + i * 16 + fq * 4 + r;
        if (co < it.Cout) it.dw[(long)co * it.K + kk] += v[r] * (it.scale ? it.scale[co] : 1.f);
    }
}
__global__ __launch_bounds__(256) void wgrad_finalize_kernel(WgFin F) {
    const int bid = (int)blockIdx.x;
    int i = 0;
    for (int k = 1; k < F.n; ++k)
        if (bid >= F.wg_begin[k]) i = k;
    i = __builtin_amdgcn_readfirstlane(i);
    const int local = bid - F.wg_begin[i];
    if (F.it[i].big) wgrad_finalize_part<256, 256, 2, 4>(F.it[i], local);
    else wgrad_finalize_part<128, 128, 2, 2>(F.it[i], local);
}

// ------------------------------------------------------------------------------------ bf16, LDS-DMA + transpose reads
// The lean kernel above moves every operand byte global -> VGPR -> (8x8 transposes in registers) -> LDS -> VGPR: its loop is
// bound by that staging, not by the matrix cores.  gfx950 can do both halves in hardware:
//   * `buffer_load_dwordx4 ... lds` (LDS-DMA) writes 16 B per lane straight into LDS, lane-linear;
//   * `ds_read_b64_tr_b16` hands a lane four 16-bit elements that are 32 B (one LDS row) apart -- i.e. four PIXELS of one
//     channel out of an image stored pixel-major, which is exactly the MFMA operand when the reduction runs over pixels.
// LDS image of one operand for a 32-pixel slab: per 16-channel block a [32 pixels][16 channels] array (32-B rows, 1 KB): one
// wave-wide DMA instruction fills one block (lane i -> pixel i/2, channel half i%2), two transpose reads (pixels 0-15 and
// 16-31 of the block, 512 contiguous bytes each: conflict free) give a 16-channel x 32-pixel MFMA fragment.  The order of
// the 32 pixels inside a fragment is the hardware's (lane group q holds pixels 4q..4q+3 and 16+4q..16+4q+3); both operands
// use the same one, and a reduction does not care.  Three-slab LDS ring, counted vmcnt, raw barriers, as in igemm.hip.
// Same domain as the lean kernel (1x1 stride-1 and "same" KxK convs, Cin % 16 == 0 for KxK), same split-K atomics epilogue.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void glds16w(__amdgpu_buffer_rsrc_t rsrc, void* dst, unsigned voff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)dst, 16, voff, 0, 0, 0);
}
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
template <int OFF>
__device__ __forceinline__ u32x2_t tr_read(unsigned addr) {
    u32x2_t v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
template <int N, int I = 0>
__device__ __forceinline__ void tr_read_frags(u32x2_t* lo, u32x2_t* hi, unsigned addr) {      // fragments are 1 KB apart
    if constexpr (I < N) {
        lo[I] = tr_read<I * 1024>(addr);
        hi[I] = tr_read<I * 1024 + 512>(addr);
        tr_read_frags<N, I + 1>(lo, hi, addr);
    }
}

template <int TCO, int TKK, int WM, int WN>
__global__ __launch_bounds__(WM* WN * 64) void wgrad_bf16_dma_kernel(WgDev p) {
    constexpr int NW = WM * WN, NT = NW * 64;
    constexpr int BP = 32;                              // pixels per slab = one MFMA k-step
    constexpr int GB = TCO / 16, XB = TKK / 16;         // 16-channel blocks of the g / x tile
    constexpr int NBLK = GB + XB;
    static_assert(NBLK % NW == 0, "every wave issues the same number of DMA instructions per slab");
    constexpr int PER_WAVE = NBLK / NW;
    constexpr int TM = TCO / WM / 16, TN = TKK / WN / 16;
    constexpr int NBUF = 3;
    constexpr int STAGE = NBLK * 1024;                  // bytes per ring stage
    __shared__ __attribute__((aligned(1024))) unsigned char lds[NBUF * STAGE];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    {
        const int gx = gridDim.x, gy = gridDim.y, total = gx * gy * gridDim.z;
        int bid = bx + gx * (by + gy * bz);
        if (p.xcd) {
            const int q = total >> 3, r = total & 7, xcd = bid & 7, idx = bid >> 3;
            bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        }
        bx = bid % gx; bid /= gx;
        by = bid % gy; bz = bid / gy;
    }
    const int co0 = bx * TCO, kk0 = by * TKK;
    const int pbeg = bz * p.pix_per_split;
    const int pend = min(p.M, pbeg + p.pix_per_split);
    if (pbeg >= pend) return;

    constexpr unsigned OOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t rg = make_rsrc_uniform(p.g, min((unsigned)pend * (unsigned)p.Cout * 2u, p.g_bytes));   // g rows >= pend read as zeros
    const __amdgpu_buffer_rsrc_t rx = make_rsrc_uniform(p.x, p.x_bytes);

    // this wave's DMA instructions: block b = wave * PER_WAVE + k (blocks 0..GB-1: g tile, GB..: x tile)
    const int lp = lane >> 1, lh = lane & 1;            // slab pixel / channel half this lane copies
    unsigned voff[PER_WAVE];                            // byte offset of (pixel pbeg + lp, this block's channels); OOB for channel tails
    int tapbit[PER_WAVE];                               // x blocks of KxK convs: bit index of the tap in the pixel's validity mask (-1: none)
#pragma unroll
    for (int k = 0; k < PER_WAVE; ++k) {
        const int b = wave * PER_WAVE + k;
        tapbit[k] = -1;
        if (b < GB) {
            const int ch = co0 + b * 16 + lh * 8;
            voff[k] = ch < p.Cout ? ((unsigned)(pbeg + lp) * (unsigned)p.Cout + (unsigned)ch) * 2u : OOB;
        } else {
            const int ch = kk0 + (b - GB) * 16 + lh * 8;
            if (ch >= p.K) { voff[k] = OOB; continue; }
            const int tap = ch / p.Cin, ci = ch - tap * p.Cin;
            const int kh = tap / p.KW, kw = tap - kh * p.KW;
            voff[k] = ((unsigned)(pbeg + lp + (kh - p.pad) * p.W + (kw - p.pad)) * (unsigned)p.Cin + (unsigned)ci) * 2u;    // mod 2^32
            if (!p.ident) tapbit[k] = tap;
        }
    }
    // validity of this lane's pixel for each tap of a "same" KxK conv (at most 25 taps): recomputed per slab from (h, w)
    int ph = 0, pw = 0;
    if (!p.ident) {
        const int r = (pbeg + lp) % (p.H * p.W);
        ph = r / p.W; pw = r - ph * p.W;
    }
    const unsigned x_row = (unsigned)p.Cin * 2u, g_row = (unsigned)p.Cout * 2u;
    int ld_pix = pbeg;                                  // first pixel of the slab being LOADED

    auto issue_slab = [&](int buf) {
        unsigned mask = ~0u;
        bool in = true;
        if (!p.ident) {
            mask = 0u;
            for (int kh = 0; kh < p.KH; ++kh)
                for (int kw = 0; kw < p.KW; ++kw)
                    if ((unsigned)(ph + kh - p.pad) < (unsigned)p.H && (unsigned)(pw + kw - p.pad) < (unsigned)p.W) mask |= 1u << (kh * p.KW + kw);
        }
        in = ld_pix + lp < pend;
        const unsigned adv = (unsigned)(ld_pix - pbeg);
#pragma unroll
        for (int k = 0; k < PER_WAVE; ++k) {
            const int b = wave * PER_WAVE + k;
            unsigned vo;
            if (b < GB) vo = voff[k] == OOB ? OOB : voff[k] + adv * g_row;
            else {
                const bool ok = in && voff[k] != OOB && (tapbit[k] < 0 || ((mask >> tapbit[k]) & 1u));
                vo = ok ? voff[k] + adv * x_row : OOB;
            }
            if (b < GB) glds16w(rg, lds + buf * STAGE + b * 1024 + 0, vo);
            else glds16w(rx, lds + buf * STAGE + b * 1024 + 0, vo);
        }
        ld_pix += BP;
        if (!p.ident) {
            pw += BP;
            while (pw >= p.W) { pw -= p.W; ++ph; }
            while (ph >= p.H) ph -= p.H;
        }
    };

    const int wm = wave / WN, wn = wave % WN;
    f32x4_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fq = lane >> 4;
    // transpose read (measured: tools/probes/tr_probe.hip): the 16 lanes of group q cover the [4 pixels][16 channels] block of
    // pixels 4q..4q+3 -- lane l points at the 8 B of pixel row 4q + (l&15)/4, channels 4*(l&3)..+3 -- and lane l RECEIVES
    // channel l&15 of the four pixels: the MFMA operand row of that lane
    const unsigned lane_off = (unsigned)((fq * 4 + ((lane & 15) >> 2)) * 32 + (lane & 3) * 8);
    const unsigned lbase = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) void*)lds;
    const unsigned a_rd = lbase + (unsigned)(wm * TM) * 1024u + lane_off;
    const unsigned b_rd = lbase + (unsigned)(GB + wn * TN) * 1024u + lane_off;

    const int S = (pend - pbeg + BP - 1) / BP;
    issue_slab(0);
    if (S > 1) issue_slab(1);
    int buf = 0, nbuf = 2;
    for (int s = 0; s < S; ++s) {
        if (s + 1 < S) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_WAVE) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (s + 2 < S) issue_slab(nbuf);
        u32x2_t alo[TM], ahi[TM], blo[TN], bhi[TN];
        tr_read_frags<TM>(alo, ahi, a_rd + (unsigned)buf * STAGE);
        tr_read_frags<TN>(blo, bhi, b_rd + (unsigned)buf * STAGE);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < TM; ++i) asm volatile("" : "+v"(alo[i]), "+v"(ahi[i]));
#pragma unroll
        for (int j = 0; j < TN; ++j) asm volatile("" : "+v"(blo[j]), "+v"(bhi[j]));
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const u32x4 af = {alo[i][0], alo[i][1], ahi[i][0], ahi[i][1]};
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const u32x4 bf = {blo[j][0], blo[j][1], bhi[j][0], bhi[j][1]};
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, af), __builtin_bit_cast(bf16x8_t, bf), acc[i][j], 0, 0, 0);
            }
        }
        buf = buf == NBUF - 1 ? 0 : buf + 1;
        nbuf = nbuf == NBUF - 1 ? 0 : nbuf + 1;
    }
    if (p.dbg & 1) {     // ablation (wgrad_dbg): no atomics
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) t += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
        if (t == 123.456f) p.dw[0] = t;
        return;
    }
    {   // split-K partial tile -> fp32 gradient (fire-and-forget buffer atomics, out-of-tile lanes dropped)
        const __amdgpu_buffer_rsrc_t rdw = make_rsrc_uniform(p.dw, p.dw_bytes);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = co0 + (wm * TM + i) * 16 + fq * 4 + r;
                const bool cok = co < p.Cout;
                const float sc = p.scale ? p.scale[cok ? co : 0] : 1.f;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int kk = kk0 + (wn * TN + j) * 16 + fr;
                    const unsigned off = (cok && kk < p.K) ? ((unsigned)co * (unsigned)p.K + (unsigned)kk) * 4u : kBufOOB;
                    buf_atomic_add_f32(rdw, off, acc[i][j][r] * sc);
                }
            }
    }
}

// ------------------------------------------------------------------------------------ bf16, LDS-DMA in full cache lines + transpose reads
// wgrad_bf16_dma_kernel above fills the LDS in 32-byte segments (a [32 pixels][16 channels] image per 16-channel block: two lanes per pixel
// row): a QUARTER of a 128-byte line per lane group.  tools/probes/dma_rate_probe.hip: the L2 -> LDS path is paced per line touched (64-byte
// segments 16.5 TB/s, 128-byte segments 27 TB/s over the chip), which is why that kernel only tied the register-staged one.  Here the image of
// a 64-channel block is [32 pixels][64 channels] = 32 rows of 128 bytes: one DMA instruction = 8 pixel rows x one full line each.  The
// transpose read takes a per-lane address, so the row pitch is free; what it needs is that the 32 lanes of a half-wave (8 pixel rows x 32 bytes
// of one 16-channel fragment) hit 64 distinct banks: rows of equal parity are 256 bytes apart, so the 32-byte chunk of a row is XORed with
// (row >> 1) & 3 -- on the DMA's SOURCE address (16-byte chunk ^ ((row >> 1) & 3) << 1), the LDS image itself is lane-linear.
// Four-slab ring (32 pixels = one MFMA k-step per slab), counted vmcnt, one raw barrier per slab; waves 0 .. TCO/64-1 load the g blocks,
// the rest the x blocks (4 DMA instructions per wave and slab), every wave owns a (TCO / WM) x (TKK / WN) piece of the accumulator.
// Same domain and epilogues (wgrad_finish_tile: ordered partial tiles, sole-owner read-modify-write, float atomics; bias gradient as a ones
// column) as wgrad_bf16_lean_tile, with Cin % 64 == 0 for K x K convs.
template <int V> struct WgTag { static constexpr int value = V; };
template <int N, int I = 0, typename F>
__device__ __forceinline__ void wg_static_for(F&& f) {
    if constexpr (I < N) {
        f(WgTag<I>{});
        wg_static_for<N, I + 1>(f);
    }
}
// ILV (r06; wgrad_ilv 1 -- a tested arm, NOT the default: alone on the chip it measures equal (354 vs 357 us on the p2 3x3), in the step, beside the
// data-gradient stream, the lockstep loop is 1.5 % faster: profiles/r06_ab_ilv.txt): the slab loop with its transpose reads and DMA pieces BETWEEN the MFMAs.  The lockstep form below issues a slab's 24
// reads and 4 DMA pieces, waits for all of them and only then starts its 32 MFMAs -- in all 8 waves at once, behind a barrier per slab: ~500 clocks per
// slab with an idle matrix pipe against 1 024 of MFMA work per SIMD (0.36 of the peak alone on the chip).  tools/probes/mfma_issue_probe.hip: reads
// between MFMAs are free, one wave feeds 0.98 of its pipe.  Here a slab starts with the x fragments and the first TWO g fragments only, every MFMA is
// an `asm volatile` with its accumulator tied in place (hipcc re-allocates the results of the builtin when other instructions sit between them), the
// reads of g fragment i + 2 and one DMA piece sit between the MFMAs of row i, and the waits are counted (`lgkmcnt(2)`: LDS data returns in order).
// Same products in the same order per accumulator: bit-identical results.
__device__ __forceinline__ void wg_mma_ip(f32x4_t& c, const u32x4& a, const u32x4& b) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
template <int TCO, int TKK, int WM, int WN, int NBUF, bool ILV = false>
__device__ __forceinline__ void wgrad_bf16_dma64_tile(const WgDev& p, unsigned char* lds, int bx, int by, int bz) {
    constexpr int NW = WM * WN;
    static_assert(NW * 64 == TCO + TKK, "one loader wave per 64-channel block");
    static_assert(NBUF == 3 || NBUF == 4, "ring depth");
    constexpr int NGW = TCO / 64;                       // g-loader waves (= g blocks)
    constexpr int BP = 32;
    constexpr int NBLK = (TCO + TKK) / 64, STAGE = NBLK * 4096;
    constexpr int TM = TCO / WM / 16, TN = TKK / WN / 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int co0 = bx * TCO, kk0 = by * TKK;
    const int pbeg = bz * p.pix_per_split;
    const int pend = min(p.M, pbeg + p.pix_per_split);
    if (pbeg >= pend) return;
    constexpr unsigned OOB = 0x80000000u;
    const bool isB = wave >= NGW;
    const int blk = isB ? wave - NGW : wave;            // 64-channel block of this operand's tile
    const int lrow = lane >> 3;                          // pixel row of a piece (8 rows per DMA instruction)
    const int chunk = (lane & 7) ^ (((lrow >> 1) & 3) << 1);       // source chunk: row k * 8 + lrow has the same (row >> 1) & 3 for every piece k
    // per-lane byte offset of (pixel pbeg + lrow, this block's channels, this lane's chunk); pieces / slabs are uniform adds
    unsigned off0;
    int dh = 0, dw = 0;
    if (!isB) {
        const int ch = co0 + blk * 64 + chunk * 8;
        off0 = ch < p.Cout ? ((unsigned)(pbeg + lrow) * (unsigned)p.Cout + (unsigned)ch) * 2u : OOB;
    } else {
        const int c0 = kk0 + blk * 64;                  // (a block lies inside one tap: Cin % 64 == 0)
        const int tap = c0 / p.Cin, ci = c0 - tap * p.Cin + chunk * 8;
        const int kh = tap / p.KW, kw = tap - kh * p.KW;
        dh = kh - p.pad; dw = kw - p.pad;
        off0 = c0 + chunk * 8 < p.K ? ((unsigned)(pbeg + lrow + dh * p.W + dw) * (unsigned)p.Cin + (unsigned)ci) * 2u : OOB;      // mod 2^32
    }
    const bool track = isB && !p.ident;                 // border validity needed (K x K x-loader waves)
    const unsigned row_bytes = (unsigned)(isB ? p.Cin : p.Cout) * 2u;
    const unsigned g_lim = (unsigned)pend * (unsigned)p.Cout * 2u;    // g rows >= pend read as zeros (their products vanish whatever x holds)
    const unsigned lim = isB ? p.x_bytes : (g_lim < p.g_bytes ? g_lim : p.g_bytes);
    const uintptr_t base = reinterpret_cast<uintptr_t>(isB ? p.x : p.g);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<void*>(((uintptr_t)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(base >> 32)) << 32) |
                                (uintptr_t)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)base)),
        0, __builtin_amdgcn_readfirstlane(lim), 0x00020000);
    // (h, w) of this lane's pixel of each of the four pieces of the slab being LOADED (K x K convs: the tap leaves the image at the borders)
    int ph[4] = {0, 0, 0, 0}, pw[4] = {0, 0, 0, 0};
    if (track) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int r = (pbeg + k * 8 + lrow) % (p.H * p.W);
            ph[k] = r / p.W; pw[k] = r - ph[k] * p.W;
        }
    }
    unsigned off = off0;
    const int wbase_slot = (isB ? NGW + blk : blk) * 4096;          // this wave's block inside a stage (bytes)
    auto issue_piece = [&](int k, int buf) {             // piece k (8 pixel rows) of the slab being loaded into ring stage `buf`
        unsigned char* dst = lds + buf * STAGE + wbase_slot;
        unsigned o = off == OOB ? OOB : off + (unsigned)(k * 8) * row_bytes;
        if (track) {
            const bool ok = (unsigned)(ph[k] + dh) < (unsigned)p.H && (unsigned)(pw[k] + dw) < (unsigned)p.W;
            o = ok ? o : OOB;
            pw[k] += BP;
            while (pw[k] >= p.W) { pw[k] -= p.W; ++ph[k]; }
            while (ph[k] >= p.H) ph[k] -= p.H;
        }
        glds16w(rsrc, dst + k * 1024, o);
    };
    auto slab_done = [&]() { if (off != OOB) off += BP * row_bytes; };
    auto issue_slab = [&](int buf) {
#pragma unroll
        for (int k = 0; k < 4; ++k) issue_piece(k, buf);
        slab_done();
    };

    const int wm = wave / WN, wn = wave % WN;
    f32x4_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const bool do_bias = p.db != nullptr && by == 0 && wn == 0;
    f32x4_t accb[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) accb[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const u32x4 ones4 = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};      // eight bf16 1.0
    const int fq = lane >> 4;
    // transpose read (tools/probes/tr_probe.hip): the 16 lanes of group q cover the [4 pixels][16 channels] block of pixels 4q .. 4q + 3 -- lane
    // l points at the 8 bytes of pixel row 4q + (l & 15) / 4, channels 4 (l & 3) .. + 3 -- and lane l RECEIVES channel l & 15 of the four pixels.
    // Fragment f of an operand = channels 16 f .. of its tile: block f / 4, 32-byte chunk f % 4 of the 128-byte rows, swizzled by the row.
    const int trow = fq * 4 + ((lane & 15) >> 2);
    const unsigned lbase = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) void*)lds;
    unsigned rd[4];                                     // byte address of chunk c (0 .. 3) of this lane's row, block 0, stage 0
#pragma unroll
    for (int c = 0; c < 4; ++c) rd[c] = lbase + (unsigned)(trow * 128 + ((c ^ ((trow >> 1) & 3)) << 5) + (lane & 3) * 8);
    constexpr int FA0 = 0, FB0 = NGW * 4096;            // first byte of the g / x blocks inside a stage
    const unsigned a_blk = (unsigned)(FA0 + (wm * TM / 4) * 4096), b_blk = (unsigned)(FB0 + (wn * TN / 4) * 4096);
    static_assert(TM % 4 == 0 && TN % 4 == 0, "a wave's fragments are whole 64-channel blocks");

    const int S = (pend - pbeg + BP - 1) / BP;
    const bool prio = __builtin_amdgcn_readfirstlane(p.dbg & 2) == 0;       // raised priority around the MFMA block (+2-3 %; wgrad_dbg 2 turns it off for A/B runs)
    issue_slab(0);
    if (S > 1) issue_slab(1);
    if (NBUF > 3 && S > 2) issue_slab(2);
    int buf = 0, nbuf = NBUF - 1;
    if constexpr (ILV) {
        static_assert(TM >= 4 && TN == 4, "four DMA pieces ride in the first four g rows");
        // the ones column as an OPAQUE four-register value: as a known constant hipcc keeps one register and rebuilds the tuple with three v_mov in front of
        // every bias MFMA, then reuses those registers for address arithmetic right behind it -- VALU writes next to an MFMA it cannot see inside the asm
        // (measured: bias gradients wrong and different from run to run; the weight gradients, whose operands only LDS reads touch, were exact)
        u32x4 ones_t = ones4;
        asm volatile("" : "+v"(ones_t));
        const bool abl_nomfma = __builtin_amdgcn_readfirstlane(p.dbg & 4) != 0, abl_nodma = __builtin_amdgcn_readfirstlane(p.dbg & 8) != 0;   // ablation (wgrad_dbg): no MFMAs / no DMA behind the prologue
        for (int s = 0; s < S; ++s) {
            if (NBUF > 3 && s + 2 < S) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (s + 1 < S) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            u32x2_t alo[TM], ahi[TM], blo[TN], bhi[TN];
            const unsigned sb = (unsigned)buf * STAGE;
            const bool more = s + NBUF - 1 < S && !abl_nodma;
            auto read_a = [&](auto I) {
                constexpr int i = decltype(I)::value;
                const unsigned a = rd[i & 3] + sb + a_blk;
                alo[i] = tr_read<(i >> 2) * 4096>(a);
                ahi[i] = tr_read<(i >> 2) * 4096 + 2048>(a);
            };
            wg_static_for<TN>([&](auto J) {
                constexpr int j = decltype(J)::value;
                const unsigned b = rd[j & 3] + sb + b_blk;
                blo[j] = tr_read<(j >> 2) * 4096>(b);
                bhi[j] = tr_read<(j >> 2) * 4096 + 2048>(b);
            });
            read_a(WgTag<0>{});
            read_a(WgTag<1>{});
            if (prio) __builtin_amdgcn_s_setprio(1);
            wg_static_for<TM>([&](auto I) {
                constexpr int i = decltype(I)::value;
                // row i's g fragment has landed; the youngest reads (fragment i + 1) may still be in flight
                if constexpr (i + 1 < TM) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
                else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if constexpr (i == 0) {
#pragma unroll
                    for (int j = 0; j < TN; ++j) asm volatile("" : "+v"(blo[j]), "+v"(bhi[j]));
                }
                asm volatile("" : "+v"(alo[i]), "+v"(ahi[i]));
                const u32x4 af = {alo[i][0], alo[i][1], ahi[i][0], ahi[i][1]};
                wg_static_for<TN>([&](auto J) {
                    constexpr int j = decltype(J)::value;
                    const u32x4 bf = {blo[j][0], blo[j][1], bhi[j][0], bhi[j][1]};
                    if (!abl_nomfma) wg_mma_ip(acc[i][j], af, bf);
                    if constexpr (j == 0 && i + 2 < TM) read_a(WgTag<i + 2>{});
                    if constexpr (j == 1 && i < 4) { if (more) issue_piece(i, nbuf); }
                });
                if (do_bias) wg_mma_ip(accb[i], af, ones_t);
            });
            if (prio) __builtin_amdgcn_s_setprio(0);
            if (more) slab_done();
            __builtin_amdgcn_sched_barrier(0);
            buf = buf == NBUF - 1 ? 0 : buf + 1;
            nbuf = nbuf == NBUF - 1 ? 0 : nbuf + 1;
        }
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");       // the asm MFMAs have written their accumulators before the epilogue reads them
        wgrad_finish_tile<TCO, TKK, WM, WN>(p, acc, accb, do_bias, bx, by, bz, wave, lane);
        return;
    }
    for (int s = 0; s < S; ++s) {
        // slab s has landed (mine: counted; everybody's: the barrier); NBUF - 2 younger slabs stay in flight
        if (NBUF > 3 && s + 2 < S) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (s + 1 < S) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        u32x2_t alo[TM], ahi[TM], blo[TN], bhi[TN];
        const unsigned sb = (unsigned)buf * STAGE;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const unsigned a = rd[i & 3] + sb + a_blk;
            if ((i >> 2) == 0) { alo[i] = tr_read<0>(a); ahi[i] = tr_read<2048>(a); }
            else { alo[i] = tr_read<4096>(a); ahi[i] = tr_read<4096 + 2048>(a); }
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const unsigned b = rd[j & 3] + sb + b_blk;
            if ((j >> 2) == 0) { blo[j] = tr_read<0>(b); bhi[j] = tr_read<2048>(b); }
            else { blo[j] = tr_read<4096>(b); bhi[j] = tr_read<4096 + 2048>(b); }
        }
        // (the fragment reads go out first: their latency runs under the issue of the next slab's DMA pieces)
        __builtin_amdgcn_sched_barrier(0);
        if (s + NBUF - 1 < S) issue_slab(nbuf);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < TM; ++i) asm volatile("" : "+v"(alo[i]), "+v"(ahi[i]));
#pragma unroll
        for (int j = 0; j < TN; ++j) asm volatile("" : "+v"(blo[j]), "+v"(bhi[j]));
        __builtin_amdgcn_sched_barrier(0);
        if (prio) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const u32x4 af = {alo[i][0], alo[i][1], ahi[i][0], ahi[i][1]};
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const u32x4 bf = {blo[j][0], blo[j][1], bhi[j][0], bhi[j][1]};
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, af), __builtin_bit_cast(bf16x8_t, bf), acc[i][j], 0, 0, 0);
            }
            if (do_bias) accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, af), __builtin_bit_cast(bf16x8_t, ones4), accb[i], 0, 0, 0);
        }
        if (prio) __builtin_amdgcn_s_setprio(0);
        buf = buf == NBUF - 1 ? 0 : buf + 1;
        nbuf = nbuf == NBUF - 1 ? 0 : nbuf + 1;
    }
    wgrad_finish_tile<TCO, TKK, WM, WN>(p, acc, accb, do_bias, bx, by, bz, wave, lane);
}

// the 256 x 256 tile on that loop (8 waves: four g blocks, four x blocks; 128 KB of LDS: one workgroup per CU), alone and grouped
template <bool ILV>
__global__ __launch_bounds__(512) void wgrad_bf16_big64_kernel(WgDev p) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[4 * 8 * 4096];
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    {
        const int gx = gridDim.x, gy = gridDim.y, total = gx * gy * gridDim.z;
        int bid = bx + gx * (by + gy * bz);
        if (p.xcd) {
            const int q = total >> 3, r = total & 7, xcd = bid & 7, idx = bid >> 3;
            bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        }
        bx = bid % gx; bid /= gx;
        by = bid % gy; bz = bid / gy;
    }
    wgrad_bf16_dma64_tile<256, 256, 2, 4, 4, ILV>(p, lds, bx, by, bz);
}
template <bool ILV>
__global__ __launch_bounds__(512) void wgrad_bf16_big64_group_kernel(WgGroup G) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[4 * 8 * 4096];
    int i, tx, ty, bz;
    wgrad_group_pick(G, i, tx, ty, bz);
    wgrad_bf16_dma64_tile<256, 256, 2, 4, 4, ILV>(G.p[i], lds, tx, ty, bz);
}
// the 128 x 128 tile on that loop (4 waves, three-slab ring: 48 KB, three workgroups per CU), grouped (the layers with 128-channel sides)
template <bool ILV>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) void wgrad_bf16_lean64_group_kernel(WgGroup G) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[3 * 4 * 4096];
    int i, tx, ty, bz;
    wgrad_group_pick(G, i, tx, ty, bz);
    wgrad_bf16_dma64_tile<128, 128, 2, 2, 3, ILV>(G.p[i], lds, tx, ty, bz);
}

// ------------------------------------------------------------------------------------ fp32
// 64 (co) x 64 (kk) tile, 16 pixels per slab, 4 waves (2x2), wave tile 32x32.
__global__ __launch_bounds__(256) void wgrad_f32_kernel(WgDev p) {
    constexpr int BP = 16, LD = 64 + 16;           // padded row (floats)
    __shared__ float lds[2 * BP * LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int co0 = blockIdx.x * 64, kk0 = blockIdx.y * 64;
    const int pbeg = blockIdx.z * p.pix_per_split;
    const int pend = min(p.M, pbeg + p.pix_per_split);
    const float* __restrict__ X = static_cast<const float*>(p.x);
    const float* __restrict__ G = static_cast<const float*>(p.g);

    // each thread loads one 4-float chunk of g and one of x per slab: row = tid>>4, chunk = tid&15
    const int lrow = tid >> 4, lch = (tid & 15) * 4;
    const int co_l = co0 + lch, kk_l = kk0 + lch;
    int tap = 0, ci = 0, kh = 0, kw = 0;
    if (kk_l < p.K) { tap = kk_l / p.Cin; ci = kk_l - tap * p.Cin; kh = tap / p.KW; kw = tap - kh * p.KW; }

    const int wm = wave >> 1, wn = wave & 1;
    f32x4_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fq = lane >> 4;

    for (int p0 = pbeg; p0 < pend; p0 += BP) {
        int px = p0 + lrow;
        float4 gv = make_float4(0, 0, 0, 0), xv = make_float4(0, 0, 0, 0);
        if (px < pend) {
            if (co_l < p.Cout) gv = *reinterpret_cast<const float4*>(G + (long)px * p.Cout + co_l);
            if (kk_l < p.K) {
                if (p.ident) xv = *reinterpret_cast<const float4*>(X + (long)px * p.Cin + ci);
                else {
                    int n = px / (p.Ho * p.Wo);
                    int r = px - n * (p.Ho * p.Wo);
                    int ho = r / p.Wo, wo = r - ho * p.Wo;
                    int hi = ho * p.stride - p.pad + kh, wi = wo * p.stride - p.pad + kw;
                    if ((unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W)
                        xv = *reinterpret_cast<const float4*>(X + (((long)n * p.H + hi) * p.W + wi) * p.Cin + ci);
                }
            }
        }
        __syncthreads();
        *reinterpret_cast<float4*>(&lds[lrow * LD + lch]) = gv;
        *reinterpret_cast<float4*>(&lds[(BP + lrow) * LD + lch]) = xv;
        __syncthreads();
#pragma unroll
        for (int st = 0; st < BP / 4; ++st) {
            float af[2], bfv[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = lds[(st * 4 + fq) * LD + wm * 32 + i * 16 + fr];
#pragma unroll
            for (int j = 0; j < 2; ++j) bfv[j] = lds[(BP + st * 4 + fq) * LD + wn * 32 + j * 16 + fr];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bfv[j], acc[i][j], 0, 0, 0);
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int co = co0 + wm * 32 + i * 16 + fq * 4 + r;
            if (co >= p.Cout) continue;
            float sc = p.scale ? p.scale[co] : 1.f;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                int kk = kk0 + wn * 32 + j * 16 + fr;
                if (kk < p.K) unsafeAtomicAdd(p.dw + (long)co * p.K + kk, acc[i][j][r] * sc);
            }
        }
}

// fp32, 128 (co) x 128 (kk) tile, BP (32) pixels per slab, 4 waves (2x2), wave tile 64x64 = 4x4 fragments of v_mfma_f32_16x16x4_f32.
// The f32-input MFMA runs at 1/16 of the bf16 rate, so what matters is keeping it fed: per 4-pixel k-step a lane reads 4 + 4 floats
// from LDS for 16 MFMAs (the 64x64 tile above: 2 + 2 for 4), the next slab's global loads are in flight during the current slab's
// MFMAs (register staging, two LDS buffers, one barrier per slab; 16-pixel slabs left the loads exposed: 53 -> TF/s at short splits).  Row pitch 144 floats: the four k-rows of a fragment read land
// on 64 different banks.
template <int BP>
__global__ __launch_bounds__(256) void wgrad_f32_t128_kernel(WgDev p) {
    constexpr int LD = 128 + 16, HALF = BP * LD, NH = BP / 8;
    __shared__ float lds[2 * 2 * HALF];              // [buffer][g | x][pixel][channel]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int co0 = blockIdx.x * 128, kk0 = blockIdx.y * 128;
    const int pbeg = blockIdx.z * p.pix_per_split;
    const int pend = min(p.M, pbeg + p.pix_per_split);
    const float* __restrict__ X = static_cast<const float*>(p.x);
    const float* __restrict__ G = static_cast<const float*>(p.g);
    // a thread stages BP/8 4-float chunks of g and of x per slab: rows lrow + 8 h, chunk lch
    const int lrow = tid >> 5, lch = (tid & 31) * 4;
    const int co_l = co0 + lch, kk_l = kk0 + lch;
    int ci = 0, kh = 0, kw = 0;
    if (kk_l < p.K) { const int tap = kk_l / p.Cin; ci = kk_l - tap * p.Cin; kh = tap / p.KW; kw = tap - kh * p.KW; }
    const int HoWo = p.Ho * p.Wo;
    float4 gv[NH], xv[NH];
    auto fetch = [&](int p0) {
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            const int px = p0 + lrow + 8 * h;
            gv[h] = make_float4(0, 0, 0, 0); xv[h] = make_float4(0, 0, 0, 0);
            if (px < pend) {
                if (co_l < p.Cout) gv[h] = *reinterpret_cast<const float4*>(G + (long)px * p.Cout + co_l);
                if (kk_l < p.K) {
                    if (p.ident) xv[h] = *reinterpret_cast<const float4*>(X + (long)px * p.Cin + ci);
                    else {
                        const int n = px / HoWo, r = px - n * HoWo;
                        const int ho = r / p.Wo, wo = r - ho * p.Wo;
                        const int hi = ho * p.stride - p.pad + kh, wi = wo * p.stride - p.pad + kw;
                        if ((unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W)
                            xv[h] = *reinterpret_cast<const float4*>(X + (((long)n * p.H + hi) * p.W + wi) * p.Cin + ci);
                    }
                }
            }
        }
    };
    const int wm = wave >> 1, wn = wave & 1;
    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fq = lane >> 4;
    int buf = 0;
    if (pbeg < pend) fetch(pbeg);
    for (int p0 = pbeg; p0 < pend; p0 += BP) {
        float* Lg = lds + buf * 2 * HALF;
        float* Lx = Lg + HALF;
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            *reinterpret_cast<float4*>(&Lg[(lrow + 8 * h) * LD + lch]) = gv[h];
            *reinterpret_cast<float4*>(&Lx[(lrow + 8 * h) * LD + lch]) = xv[h];
        }
        __syncthreads();
        if (p0 + BP < pend) fetch(p0 + BP);
#pragma unroll
        for (int st = 0; st < BP / 4; ++st) {
            float af[4], bfv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = Lg[(st * 4 + fq) * LD + wm * 64 + i * 16 + fr];
#pragma unroll
            for (int j = 0; j < 4; ++j) bfv[j] = Lx[(st * 4 + fq) * LD + wn * 64 + j * 16 + fr];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bfv[j], acc[i][j], 0, 0, 0);
        }
        buf ^= 1;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = co0 + wm * 64 + i * 16 + fq * 4 + r;
            if (co >= p.Cout) continue;
            const float sc = p.scale ? p.scale[co] : 1.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int kk = kk0 + wn * 64 + j * 16 + fr;
                if (kk < p.K) unsafeAtomicAdd(p.dw + (long)co * p.K + kk, acc[i][j][r] * sc);
            }
        }
}

// db[c] += sum_p g[p][c]  (bias gradients).  Rows are read as full contiguous lines: a thread owns one 16-B chunk of
// channels (C/EP chunks per row), the block walks `rows_per_block` rows, partial sums are combined through LDS.
template <typename T, int NT>
__global__ __launch_bounds__(NT) void colsum_kernel(const T* __restrict__ g, float* __restrict__ db, int M, int C, int ld, int rows_per_block) {
    constexpr int EP = Elem<T>::kPer16B;
    __shared__ float red[NT * EP];
    const int chunks = C / EP;                         // 16-B chunks per row (<= 256)
    const int rl = NT / chunks;                        // row lanes per block iteration
    const int ch = threadIdx.x % chunks, lane_r = threadIdx.x / chunks;
    const int r0 = blockIdx.x * rows_per_block, r1 = min(M, r0 + rows_per_block);
    float acc[EP];
#pragma unroll
    for (int k = 0; k < EP; ++k) acc[k] = 0.f;
    auto add = [&](const uint4& v) {
        if constexpr (EP == 8) {
            const uint32_t* w = reinterpret_cast<const uint32_t*>(&v);
#pragma unroll
            for (int k = 0; k < 4; ++k) { acc[2 * k] += __uint_as_float(w[k] << 16); acc[2 * k + 1] += __uint_as_float(w[k] & 0xffff0000u); }
        } else {
            const float* w = reinterpret_cast<const float*>(&v);
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] += w[k];
        }
    };
    if (lane_r < rl) {
        // a streaming read with nothing to hide its latency behind: four independent 16-B loads in flight per lane
        const T* __restrict__ col = g + ch * EP;
        int r = r0 + lane_r;
        for (; r + 3 * rl < r1; r += 4 * rl) {
            const uint4 v0 = *reinterpret_cast<const uint4*>(col + (long)r * ld);
            const uint4 v1 = *reinterpret_cast<const uint4*>(col + (long)(r + rl) * ld);
            const uint4 v2 = *reinterpret_cast<const uint4*>(col + (long)(r + 2 * rl) * ld);
            const uint4 v3 = *reinterpret_cast<const uint4*>(col + (long)(r + 3 * rl) * ld);
            add(v0); add(v1); add(v2); add(v3);
        }
        for (; r < r1; r += rl) add(*reinterpret_cast<const uint4*>(col + (long)r * ld));
    }
#pragma unroll
    for (int k = 0; k < EP; ++k) red[threadIdx.x * EP + k] = acc[k];
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += NT) {
        const int cch = c / EP, k = c % EP;
        float s = 0.f;
        for (int l = 0; l < rl; ++l) s += red[(l * chunks + cch) * EP + k];
        if (s != 0.f) unsafeAtomicAdd(db + c, s);
    }
}

}  // namespace

namespace {
// validated device-side description of one weight-gradient problem
int fill_wgdev(const aldi_wgrad_args* a, WgDev& d) {
    if (!a || !a->x || !a->g || !a->dw) return aldi_set_error_msg(ALDI_ERR_ARG, "conv_wgrad: null pointer");
    const int ep = a->dtype == ALDI_BF16 ? 8 : 4;
    if (a->Cin % ep || a->Cout % ep) return aldi_set_error_msg(ALDI_ERR_ARG, "conv_wgrad: Cin/Cout must be multiples of a 16-B chunk");
    d.x = a->x; d.g = a->g; d.dw = a->dw; d.scale = a->scale; d.db = a->db;
    d.N = a->N; d.H = a->H; d.W = a->W; d.Cin = a->Cin; d.Cout = a->Cout; d.KH = a->KH; d.KW = a->KW;
    d.stride = a->stride; d.pad = a->pad; d.Ho = a->Ho; d.Wo = a->Wo;
    long M = (long)a->N * a->Ho * a->Wo;
    if (M <= 0 || M > 0x7fffffffL) return aldi_set_error_msg(ALDI_ERR_ARG, "conv_wgrad: bad M");
    d.M = (int)M;
    d.K = a->KH * a->KW * a->Cin;
    const size_t esz = a->dtype == ALDI_BF16 ? 2 : 4;
    const size_t xb = (size_t)a->N * a->H * a->W * a->Cin * esz, gb = (size_t)M * a->Cout * esz;
    if (a->dtype == ALDI_BF16 && (xb >= 0x80000000ull || gb >= 0x80000000ull))
        return aldi_set_error_msg(ALDI_ERR_ARG, "conv_wgrad: operand larger than 2 GiB (32-bit buffer offsets)");
    d.x_bytes = (unsigned)xb;
    d.g_bytes = (unsigned)gb;
    const size_t wb = (size_t)a->Cout * d.K * 4;
    if (wb >= 0x80000000ull) return aldi_set_error_msg(ALDI_ERR_ARG, "conv_wgrad: gradient larger than 2 GiB (32-bit buffer offsets)");
    d.dw_bytes = (unsigned)wb;
    d.ident = (a->KH == 1 && a->KW == 1 && a->stride == 1 && a->pad == 0 && a->Ho == a->H && a->Wo == a->W) ? 1 : 0;
    d.pix_per_split = d.M; d.xcd = 0; d.dbg = 0;
    d.ws = d.wsb = nullptr; d.splits = 1; d.ordered = 0;
    return ALDI_OK;
}
bool lean_eligible(const aldi_wgrad_args* a, const WgDev& d) {
    const bool same = a->stride == 1 && a->Ho == a->H && a->Wo == a->W && 2 * a->pad == a->KH - 1 && a->KH == a->KW && a->Cin % 64 == 0;
    return a->dtype == ALDI_BF16 && (d.ident || same);
}
// 256x256 tile (one 8-wave workgroup per CU) when every workgroup still gets a long pixel range
bool wants_big_tile(const WgDev& d, const AldiTuning& tn) {
    if (tn.wgrad_big_min <= 0 || d.Cout % 256 || d.K % 256) return false;
    const int tb = (d.Cout / 256) * (d.K / 256);
    const int sb = tn.wgrad_big_slots / tb;            // floor: one 8-wave workgroup per CU, never 257 of them
    return cdiv(d.M, 64) / (sb > 0 ? sb : 1) >= tn.wgrad_big_min;
}
}  // namespace

namespace {
// Workspace of the ordered epilogue, carved in 256-B units.  dry: count only (aldi_conv_wgrad_group_workspace).
struct WsCarver {
    float* base; size_t cap, used; bool dry;
    float* take(size_t n_floats) {
        n_floats = (n_floats + 63) / 64 * 64;
        float* r = dry ? nullptr : base + used;
        used += n_floats;
        return r;
    }
    bool fits() const { return dry || used <= cap; }
};
// the problems of one call that need the second pass; flushed in batches of kMaxGroup
struct FinBuilder {
    WgFin F; int wg; bool dry; hipStream_t st;
    FinBuilder(hipStream_t s, bool d) : wg(0), dry(d), st(s) { F.n = 0; }
    void add(const WgDev& d, int big) {
        if (dry) return;
        if (F.n == kMaxGroup) flush();
        const int tile = big ? 256 : 128, gx = cdiv(d.Cout, tile), gy = cdiv(d.K, tile);
        WgFinItem& it = F.it[F.n];
        it.ws = d.ws; it.wsb = d.wsb; it.dw = d.dw; it.db = d.db; it.scale = d.scale;
        it.Cout = d.Cout; it.K = d.K; it.gx = gx; it.gy = gy; it.S = d.splits; it.big = big;
        F.wg_begin[F.n] = wg;
        wg += gx * gy * (big ? 32 * 2 : 16) + 1;           // (tile, fragment, four producer waves) + the bias workgroup
        ++F.n;
    }
    void flush() {
        if (dry || F.n == 0) return;
        for (int k = F.n; k <= kMaxGroup; ++k) F.wg_begin[k] = wg;
        hipLaunchKernelGGL(wgrad_finalize_kernel, dim3(wg), dim3(256), 0, st, F);
        F.n = 0; wg = 0;
    }
};
// ordered epilogue of problem d (tile x tile output tiles, d.splits pixel ranges): workspace + second pass when split
void plan_ordered(WgDev& d, int big, WsCarver& ws, FinBuilder& fin) {
    d.ordered = 1;
    d.ws = d.wsb = nullptr;
    if (d.splits <= 1) return;
    const int tile = big ? 256 : 128;
    const size_t gx = cdiv(d.Cout, tile), gy = cdiv(d.K, tile);
    d.ws = ws.take(gx * gy * (size_t)d.splits * tile * tile);
    if (d.db) d.wsb = ws.take(gx * (size_t)d.splits * tile);
    fin.add(d, big);
}

int wgrad_single(const aldi_wgrad_args* a, hipStream_t st, WsCarver& ws, FinBuilder& fin, bool ordered, bool dry) {
    WgDev d;
    if (int rc = fill_wgdev(a, d)) return rc;
    const AldiTuning& tn = aldi_tuning();
    const int lean_env = tn.wgrad_lean, big_slots_env = tn.wgrad_big_slots;
    const bool same = a->stride == 1 && a->Ho == a->H && a->Wo == a->W && 2 * a->pad == a->KH - 1 && a->KH == a->KW && a->Cin % 64 == 0;
    const bool lean = a->dtype == ALDI_BF16 && lean_env && (d.ident || same);
    const bool big = lean && wants_big_tile(d, tn);
    const bool f32_t128 = a->dtype == ALDI_F32 && tn.wgrad_f32_tile128 && d.Cout >= 128 && d.K >= 128;
    const int tile = big ? 256 : (a->dtype == ALDI_BF16 || f32_t128 ? 128 : 64);
    const int bp = a->dtype == ALDI_BF16 ? 64 : (f32_t128 ? 32 : 16);
    int slabs = cdiv(d.M, bp);
    int tiles = cdiv(d.Cout, tile) * cdiv(d.K, tile);
    const int slots_env_ = tn.wgrad_slots;
    const int slots_env = big ? big_slots_env : (f32_t128 ? 512 : slots_env_);      // fp32 128x128: two workgroups per CU (VGPRs), MFMA-bound: whole rounds
    // the kernel is bound per CU (L2 -> CU path, LDS), not by latency: few, long splits (1-2 workgroups per CU) beat
    // many short ones, whose epilogues also contend on the same dW lines
    int splits = big ? slots_env / tiles : cdiv(slots_env, tiles);
    if (f32_t128) {
        // MFMA-bound, two workgroups per CU: the launch runs in rounds of 512 workgroups, a round lasting (slabs per split + an
        // epilogue of ~6 slabs: 16 K float atomics per workgroup).  cdiv(512, tiles) splits put 576 workgroups = two rounds on
        // res5's 3x3 (169 us against 150 for the 64x64 kernel)
        long best = -1;
        int best_sp = 1;
        for (int sp = 1; sp <= 512 && sp <= (slabs + 3) / 4; ++sp) {
            const long rounds = ((long)tiles * sp + 511) / 512;
            const long cost = rounds * (cdiv(slabs, sp) + 6);
            if (best < 0 || cost < best) { best = cost; best_sp = sp; }
        }
        splits = best_sp;
    }
    if (splits > slabs / 4) splits = slabs / 4;    // ... but at least 4 slabs of work behind every epilogue
    if (splits < 1) splits = 1;
    if (splits > 512) splits = 512;
    int slabs_per = cdiv(slabs, splits);
    d.pix_per_split = slabs_per * bp;
    splits = cdiv(d.M, d.pix_per_split);
    dim3 grid(cdiv(d.Cout, tile), cdiv(d.K, tile), splits);
    d.xcd = tn.wgrad_xcd;
    d.dbg = tn.wgrad_dbg;
    d.splits = splits;
    const char* which;
    // LDS-DMA + transpose-read form (wgrad_dma: 0 off, 1 = 128x128 tile in place of the lean kernel, 2 = also in place of the 256x256 one)
    const bool dma = lean && tn.wgrad_dma > 0 && (d.ident ? a->Cin % 8 == 0 : a->Cin % 16 == 0) && a->KH * a->KW <= 25 && !(big && tn.wgrad_dma < 2);
    if (dma) {
        if (big) {       // re-derive the split for the 128x128 tile
            tiles = cdiv(d.Cout, 128) * cdiv(d.K, 128);
            splits = cdiv(slots_env_, tiles);
            if (splits > slabs / 4) splits = slabs / 4;
            if (splits < 1) splits = 1;
            slabs_per = cdiv(slabs, splits);
            d.pix_per_split = slabs_per * bp;
            splits = cdiv(d.M, d.pix_per_split);
            grid = dim3(cdiv(d.Cout, 128), cdiv(d.K, 128), splits);
        }
        if (!dry) hipLaunchKernelGGL((wgrad_bf16_dma_kernel<128, 128, 2, 2>), grid, dim3(256), 0, st, d);
        which = "wgrad_bf16_dma";
    } else if (big || lean) {
        if (ordered) plan_ordered(d, big, ws, fin);
        if (!ws.fits()) return aldi_set_error_msg(ALDI_ERR_ARG, "conv_wgrad: workspace too small (aldi_conv_wgrad_group_workspace)");
        if (!dry) {
            if (big && (tn.wgrad_dma64 & 1)) { if (tn.wgrad_ilv) hipLaunchKernelGGL(wgrad_bf16_big64_kernel<true>, grid, dim3(512), 0, st, d); else hipLaunchKernelGGL(wgrad_bf16_big64_kernel<false>, grid, dim3(512), 0, st, d); }
            else if (big) hipLaunchKernelGGL(wgrad_bf16_big_kernel, grid, dim3(512), 0, st, d);
            else hipLaunchKernelGGL(wgrad_bf16_lean_kernel, grid, dim3(256), 0, st, d);
        }
        which = big ? ((tn.wgrad_dma64 & 1) ? "wgrad_bf16_big64" : "wgrad_bf16_big") : "wgrad_bf16_lean";
    }
    else if (a->dtype == ALDI_BF16) { if (!dry) hipLaunchKernelGGL(wgrad_bf16_kernel, grid, dim3(256), 0, st, d); which = "wgrad_bf16_generic"; }
    else if (a->dtype == ALDI_F32 && f32_t128) { if (!dry) hipLaunchKernelGGL(wgrad_f32_t128_kernel<32>, grid, dim3(256), 0, st, d); which = "wgrad_f32_t128"; }
    else if (a->dtype == ALDI_F32) { if (!dry) hipLaunchKernelGGL(wgrad_f32_kernel, grid, dim3(256), 0, st, d); which = "wgrad_f32"; }
    else return aldi_set_error_msg(ALDI_ERR_ARG, "conv_wgrad: bad dtype");
    if (dry) return ALDI_OK;
    ALDI_CHECK_LAUNCH();
    if (a->db && (dma || !(big || lean)))           // only the lean / 256x256 kernels add the bias gradient themselves
        if (int rc = aldi_bias_grad(a->g, a->db, d.M, d.Cout, a->dtype, st)) return rc;
    {
        char name[96];
        snprintf(name, sizeof(name), "%s splits=%d%s", which, splits, d.ordered ? " ordered" : "");
        aldi_note_dispatch(name);
    }
    return ALDI_OK;
}

// one grouped launch of the problems idx[0..ng) with a common pixel length per workgroup; big: 256x256 tiles, one workgroup per CU
int launch_group(const WgDev* probs, int ng, bool big, hipStream_t st, WsCarver& ws, FinBuilder& fin, bool ordered, bool dry, char* name, size_t name_len) {
    const AldiTuning& tn = aldi_tuning();
    const int tile = big ? 256 : 128;
    long tiles_of[kMaxGroup];
    int order[kMaxGroup];
    long maxM = 0;
    for (int i = 0; i < ng; ++i) {
        tiles_of[i] = (long)cdiv(probs[i].Cout, tile) * cdiv(probs[i].K, tile);
        if (probs[i].M > maxM) maxM = probs[i].M;
        order[i] = i;
    }
    auto wgs_for = [&](long T) {
        long w = 0;
        for (int i = 0; i < ng; ++i) w += tiles_of[i] * cdiv(probs[i].M, T);
        return w;
    };
    // Pixels per workgroup: ONE value T for the whole group (workgroups of equal length), chosen by a round model.  128x128: three
    // workgroups are resident per CU (168 VGPRs) and need each other to hide their LDS / DMA latency, so the chip works through
    // the launch in rounds of 768, a round lasting (T / 32 slab steps + one epilogue of ~wgrad_group_epi slab steps).  Measured on
    // the step's groups: 392 unsplit res4 tiles 603 us, three splits (1176 workgroups) 544 us; one 256-workgroup round of res3
    // 553 us against 432 us for 512 half-length workgroups.  256x256: one workgroup per CU, rounds of wgrad_big_slots.
    long T = (maxM + 63) / 64 * 64;
    const long target = big ? 0 : tn.wgrad_group_slots;
    if (target > 0) {
        while (T > 256 && wgs_for(T) < target) T = (T / 2 + 63) / 64 * 64;   // >= 4 slabs behind every epilogue
    } else {
        const long slots = big ? (tn.wgrad_big_slots > 0 ? tn.wgrad_big_slots : 256) : 768;
        const long epi = big ? tn.wgrad_big_epi : tn.wgrad_group_epi;
        const long minT = big ? 512 : 256;
        long best = -1;
        for (int sp = 1; sp <= 4096; ++sp) {
            const long Ts = ((maxM + sp - 1) / sp + 63) / 64 * 64;
            if (Ts < minT && sp > 1) break;
            const long rounds = (wgs_for(Ts) + slots - 1) / slots;
            const long cost = rounds * (Ts / 32 + epi);
            if (best < 0 || cost < best) { best = cost; T = Ts; }
        }
    }
    // longest-running problems first (K x K convs before 1x1: more k-steps per pixel do not matter, pixels per workgroup do)
    for (int i = 1; i < ng; ++i)
        for (int j = i; j > 0 && probs[order[j]].M > probs[order[j - 1]].M; --j) { int t_ = order[j]; order[j] = order[j - 1]; order[j - 1] = t_; }
    static thread_local WgGroup L_;
    L_.n = ng;
    int wg = 0;
    for (int k = 0; k < ng; ++k) {
        WgDev d = probs[order[k]];
        d.pix_per_split = (int)(T < d.M ? T : (d.M + 63) / 64 * 64);
        d.splits = cdiv(d.M, d.pix_per_split);
        if (ordered && d.ordered >= 0) plan_ordered(d, big, ws, fin);
        else d.ordered = 0;
        L_.p[k] = d;
        L_.gx[k] = cdiv(d.Cout, tile);
        L_.gy[k] = cdiv(d.K, tile);
        L_.gz[k] = d.splits;
        L_.wg_begin[k] = wg;
        wg += L_.gx[k] * L_.gy[k] * L_.gz[k];
    }
    if (!ws.fits()) return aldi_set_error_msg(ALDI_ERR_ARG, "conv_wgrad_group: workspace too small (aldi_conv_wgrad_group_workspace)");
    for (int k = ng; k <= kMaxGroup; ++k) L_.wg_begin[k] = wg;
    if (dry) return ALDI_OK;
    // wgrad_dma64 bit 2: the 128 x 128 group on the LDS-DMA + transpose-read loop when every layer's rows are whole 16-byte chunks
    bool lean64 = !big && (tn.wgrad_dma64 & 2);
    for (int k = 0; k < ng && lean64; ++k) lean64 = L_.p[k].Cin % 8 == 0 && L_.p[k].Cout % 8 == 0;
    if (big && (tn.wgrad_dma64 & 1)) { if (tn.wgrad_ilv) hipLaunchKernelGGL(wgrad_bf16_big64_group_kernel<true>, dim3(wg), dim3(512), 0, st, L_); else hipLaunchKernelGGL(wgrad_bf16_big64_group_kernel<false>, dim3(wg), dim3(512), 0, st, L_); }
    else if (big) hipLaunchKernelGGL(wgrad_bf16_big_group_kernel, dim3(wg), dim3(512), 0, st, L_);
    else if (lean64) { if (tn.wgrad_ilv) hipLaunchKernelGGL(wgrad_bf16_lean64_group_kernel<true>, dim3(wg), dim3(256), 0, st, L_); else hipLaunchKernelGGL(wgrad_bf16_lean64_group_kernel<false>, dim3(wg), dim3(256), 0, st, L_); }
    else if (tn.wgrad_db) hipLaunchKernelGGL(wgrad_bf16_lean_group_db_kernel, dim3(wg), dim3(256), 0, st, L_);
    else hipLaunchKernelGGL(wgrad_bf16_lean_group_kernel, dim3(wg), dim3(256), (size_t)tn.wgrad_lds_pad_kb << 10, st, L_);
    ALDI_CHECK_LAUNCH();
    snprintf(name, name_len, "wgrad_bf16_%s_group%s n=%d wgs=%d pix=%ld%s", big ? ((tn.wgrad_dma64 & 1) ? "big64" : "big") : (lean64 ? "lean64" : "lean"), (!big && tn.wgrad_db) ? "_db" : "", ng, wg, T, ordered ? " ordered" : "");
    return ALDI_OK;
}

int wgrad_group_impl(const aldi_wgrad_args* args, int n, hipStream_t st, bool dry, size_t* need) {
    if (!args || n < 1) return aldi_set_error_msg(ALDI_ERR_ARG, "conv_wgrad_group: no problems");
    const AldiTuning& tn = aldi_tuning();
    const bool ordered = dry || (args[0].ws != nullptr && tn.wgrad_ordered);
    WsCarver ws{static_cast<float*>(args[0].ws), dry ? 0 : (size_t)args[0].ws_bytes / 4, 0, dry};
    FinBuilder fin(st, dry);
    // problems the lean / 256x256 kernels cannot take (fp32, strided, unpadded ...) go through the single-problem dispatcher
    static thread_local WgDev lean_p[kMaxGroup], big_p[kMaxGroup];
    int nl = 0, nb = 0;
    for (int i = 0; i < n; ++i) {
        WgDev d;
        if (int rc = fill_wgdev(&args[i], d)) return rc;
        // layers that SHARE a gradient buffer inside one call (one conv applied to several pyramid levels) would race in the plain
        // read-modify-write / second pass: those keep the float-atomic epilogue
        bool shared = false;
        for (int j = 0; j < n && !shared; ++j)
            shared = j != i && (args[j].dw == args[i].dw || (args[i].db && args[j].db == args[i].db));
        const bool elig = lean_eligible(&args[i], d) && tn.wgrad_lean;
        const bool big_group = elig && tn.wgrad_big_group && d.Cout % 256 == 0 && d.K % 256 == 0 && d.M >= 512 && nb < kMaxGroup;
        if (!elig || (!big_group && (nl == kMaxGroup || wants_big_tile(d, tn)))) {      // (alone, the big tile has its own launch)
            if (int rc = wgrad_single(&args[i], st, ws, fin, ordered && !shared, dry)) return rc;
            continue;
        }
        d.dbg = tn.wgrad_dbg;
        d.xcd = tn.wgrad_xcd;
        d.splits = 1; d.ordered = shared ? -1 : 0; d.ws = d.wsb = nullptr;
        if (big_group) big_p[nb++] = d; else lean_p[nl++] = d;
    }
    if (nb) {
        // a 256x256 launch wants a CU-count of workgroups with >= 1000 pixels each; a couple of tiles would be cut into hundreds of
        // short pixel ranges (each ending in a 256-KB partial tile) just to occupy the chip: those layers stay with the 128x128 group
        double tile_pixels = 0.0;
        for (int i = 0; i < nb; ++i) tile_pixels += (double)(big_p[i].Cout / 256) * (big_p[i].K / 256) * big_p[i].M;
        if (tile_pixels < 4096.0 * tn.wgrad_big_group_min && nl + nb <= kMaxGroup) {
            for (int i = 0; i < nb; ++i) lean_p[nl++] = big_p[i];
            nb = 0;
        }
    }
    char nb_name[96] = "", nl_name[96] = "";
    if (nb) if (int rc = launch_group(big_p, nb, true, st, ws, fin, ordered, dry, nb_name, sizeof(nb_name))) return rc;
    if (nl) if (int rc = launch_group(lean_p, nl, false, st, ws, fin, ordered, dry, nl_name, sizeof(nl_name))) return rc;
    fin.flush();
    if (need) *need = ws.used * 4;
    if (!dry && (nb || nl)) {
        ALDI_CHECK_LAUNCH();
        char name[200];
        snprintf(name, sizeof(name), "%s%s%s", nl_name, (nb && nl) ? " | " : "", nb_name);
        aldi_note_dispatch(name);
    }
    return ALDI_OK;
}
}  // namespace

extern "C" int aldi_conv_wgrad_group(const aldi_wgrad_args* args, int n, aldi_stream_t stream) {
    return wgrad_group_impl(args, n, static_cast<hipStream_t>(stream), false, nullptr);
}
extern "C" long aldi_conv_wgrad_group_workspace(const aldi_wgrad_args* args, int n) {
    size_t need = 0;
    if (wgrad_group_impl(args, n, nullptr, true, &need)) return -1;
    if (n == 1) {
        // one problem may be handed to aldi_conv_wgrad instead, whose dispatcher splits the pixels by its own rule (knob wgrad_slots): the larger of the two
        WsCarver ws{nullptr, 0, 0, true};
        FinBuilder fin(nullptr, true);
        if (wgrad_single(&args[0], nullptr, ws, fin, true, true)) return -1;
        if (ws.used * 4 > need) need = ws.used * 4;
    }
    return (long)need;
}

extern "C" int aldi_conv_wgrad(const aldi_wgrad_args* a, aldi_stream_t stream) {
    if (!a) return aldi_set_error_msg(ALDI_ERR_ARG, "conv_wgrad: null pointer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool ordered = a->ws != nullptr && aldi_tuning().wgrad_ordered;
    WsCarver ws{static_cast<float*>(a->ws), (size_t)a->ws_bytes / 4, 0, false};
    FinBuilder fin(st, false);
    if (int rc = wgrad_single(a, st, ws, fin, ordered, false)) return rc;
    fin.flush();
    ALDI_CHECK_LAUNCH();
    return ALDI_OK;
}

extern "C" int aldi_bias_grad(const void* g, float* db, int M, int C, int dtype, aldi_stream_t stream) {
    if (!g || !db || M <= 0 || C <= 0) return aldi_set_error_msg(ALDI_ERR_ARG, "bias_grad: bad args");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int ep = dtype == ALDI_BF16 ? 8 : 4;
    if (C % ep) return aldi_set_error_msg(ALDI_ERR_ARG, "bias_grad: C must be a multiple of a 16-B chunk");
    // enough workgroups to keep every CU's memory pipeline busy (the old 64-row floor left 1 workgroup per CU on a 16800-row
    // matrix), bounded below so that the per-workgroup LDS reduction + C atomics stay a small part of the work
    const AldiTuning& tn = aldi_tuning();
    const int target_blocks = tn.colsum_blocks, min_rows = tn.colsum_minrows, nt_env = tn.colsum_nt, block_kb = tn.colsum_block_kb;
    const long row_bytes = (long)C * (dtype == ALDI_BF16 ? 2 : 4);
    int rows_per_block = cdiv(M, target_blocks);
    const int by_bytes = (int)(((long)block_kb << 10) / row_bytes);
    if (rows_per_block < by_bytes) rows_per_block = by_bytes;
    if (rows_per_block < min_rows) rows_per_block = min_rows;
    dim3 grid(cdiv(M, rows_per_block));
    for (int c0 = 0; c0 < C; c0 += 256 * ep) {       // a block covers at most 256 16-B chunks of a row: wider rows go in column slices
        const int cs = C - c0 < 256 * ep ? C - c0 : 256 * ep;
        if (dtype == ALDI_BF16) {
            if (nt_env == 1024) hipLaunchKernelGGL((colsum_kernel<bf16_t, 1024>), grid, dim3(1024), 0, st, (const bf16_t*)g + c0, db + c0, M, cs, C, rows_per_block);
            else hipLaunchKernelGGL((colsum_kernel<bf16_t, 256>), grid, dim3(256), 0, st, (const bf16_t*)g + c0, db + c0, M, cs, C, rows_per_block);
        } else hipLaunchKernelGGL((colsum_kernel<float, 256>), grid, dim3(256), 0, st, (const float*)g + c0, db + c0, M, cs, C, rows_per_block);
    }
    ALDI_CHECK_LAUNCH();
    return ALDI_OK;
}
