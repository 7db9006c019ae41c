// A whole ResNet bottleneck in one kernel, for the stages that keep nothing for a backward pass (res2: frozen by
// BACKBONE.FREEZE_AT = 2 in the student, and the EMA teacher never trains):
//
//   y = relu( W3 . relu( W2 (*) relu( W1 . x + b1 ) + b2 ) + b3 + r )        W1: 1x1 CX -> 64,  W2: 3x3 64 -> 64 (pad 1),  W3: 1x1 64 -> 256
//
// (FrozenBN folded: the per-channel scales live in the bf16 weights, aldi_fold_weights_batch; the shifts are the biases b.)
// Launched layer by layer the block moves 2048 B per pixel through HBM at the res2 resolution (conv1 512 + 128, conv2 128 + 128,
// conv3 128 + 512 + 512) and every one of those launches is HBM-bound (profiles/r02_dense_profile_insitu.txt: 4.5 TB/s, 0.10 of the
// MFMA peak); here a workgroup takes an 8 x 16 pixel tile of the output and the two 64-channel intermediate maps never leave the
// LDS: 512 B read (+ the halo re-reads, served by the L2) and 512 B written per pixel.
//
//   phase 1   a1 = relu(W1 x + b1) on the tile's 10 x 18 halo: GEMM [64 co] x [192 rows (180 halo pixels)] x CX, x slabs of 32 channels
//             by LDS-DMA (out-of-image halo pixels: out-of-range DMA offsets -> zeros, and a1 is forced to 0 there: conv2 pads a1,
//             not x), four-stage ring, three slabs in flight.  a1 -> LDS as bf16 [10 x 24 rows][64 ch] (row pitch 24: see the fragment addressing below).
//   phase 2   a2 = relu(W2 (*) a1 + b2): nine taps x two k-steps; the B fragments of tap (kh, kw) are a1 rows shifted by kh * 24 + kw,
//             read in place (no im2col); W2 streams tap by tap through a four-stage ring.  a2 -> LDS [128 px][64 ch].
//   phase 3   y = relu(W3 a2 + b3 + r): [256 co] x [128 px] x 64, W3 (32 KB) is DMA'd into the dead tap ring under the a2 epilogue; the
//             result goes through an LDS staging tile so that residual loads and stores are 16 B per lane, 512 B per pixel row.
// Four waves per workgroup, 78 KB of LDS: two workgroups per CU, so one's HBM phases (x slabs, residual, stores) run beside the
// other's MFMA phases.  Weights: the accumulator fragment's A operand (lane = 4 consecutive output channels of one pixel).
//
// Replaces detectron2's BottleneckBlock.forward (conv1/conv2/conv3 + FrozenBN + ReLU + shortcut add) for res2, reached from
// aldi/trainer.py:87 (student), aldi/pseudolabeler.py:21 and aldi/distill.py:162 (teacher).
#include "common.h"
#include "tile_prims.h"
#include <stdio.h>

namespace {

constexpr int MID = 64, CO = 256, TH = 8, TW = 16, HC = TW + 2, NHP = (TH + 2) * HC, CR = 192, A1P = 24;
// (phase 1's four-stage x / W1 ring is laid over all three regions: nothing else is live then)
constexpr int kRegA = 0;                            // phase 2: W2 tap ring (4 x 8 KB); phase 3: W3 (32 KB)
constexpr int kRegB = 32768;                        // a1: 240 rows x 128 B
constexpr int kRegC = kRegB + (TH + 2) * A1P * 128; // a2 (16 KB)
constexpr int kLdsBytes = kRegC + 16384;            // 79872
constexpr int kStgRow = CO * 2 + 16;                // output staging row (padded): 128 rows x 528 B <= kLdsBytes
static_assert(TH * TW * kStgRow <= kLdsBytes && kLdsBytes <= 81920, "two workgroups per CU");

struct BneckDev {
    const bf16_t* x; const bf16_t* res; bf16_t* y;
    const bf16_t* w1; const bf16_t* w2; const bf16_t* w3;
    const float* b1; const float* b2; const float* b3;
    int N, H, W, th, tw, xcd;
    unsigned x_bytes, y_bytes;
};

__device__ __forceinline__ void lds_write_b64(unsigned addr, uint2 v) {
    asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
template <int N, int STEP, int I = 0>
__device__ __forceinline__ void stg_read_all(u32x4_t* f, unsigned addr) {     // N 16-B reads STEP bytes apart, no wait
    if constexpr (I < N) {
        f[I] = frag_read<I * STEP>(addr);
        stg_read_all<N, STEP, I + 1>(f, addr);
    }
}
__device__ __forceinline__ uint2 relu_pack4(f32x4_t a, float4 sh, bool keep) {
    float v0 = fmaxf(a[0] + sh.x, 0.f), v1 = fmaxf(a[1] + sh.y, 0.f), v2 = fmaxf(a[2] + sh.z, 0.f), v3 = fmaxf(a[3] + sh.w, 0.f);
    uint2 t;
    t.x = keep ? pack2_bf16(v0, v1) : 0u;
    t.y = keep ? pack2_bf16(v2, v3) : 0u;
    return t;
}

template <int CX>
__global__ __launch_bounds__(256, 2) void bneck_kernel(BneckDev p) {
    static_assert(CX % 32 == 0, "x slabs of 32 channels");
    __shared__ __attribute__((aligned(128))) uint4 lds[kLdsBytes / 16];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    int bid = blockIdx.x;
    if (p.xcd) {                                     // a contiguous range of tiles per XCD: neighbouring tiles share halo pixels in one L2
        const int total = gridDim.x, q = total >> 3, r = total & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int per_img = p.th * p.tw;
    const int n = bid / per_img, rem = bid - n * per_img;
    const int ty = rem / p.tw, tx = rem - ty * p.tw;
    const int h0 = ty * TH, w0 = tx * TW;
    constexpr unsigned OOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.w1), 0, MID * CX * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.w2), 0, MID * 9 * MID * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw3 = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.w3), 0, CO * MID * 2, 0x00020000);
    const int wbase = __builtin_amdgcn_readfirstlane(tid & ~63);
    const unsigned lbase = lds_addr(&lds[0]);
    const long img_pix = (long)n * p.H * p.W;
    // the folded FrozenBN shifts of this lane's channels, fetched before any LDS-DMA is in flight (behind one, the compiler drains the
    // whole DMA queue in front of the first use of an ordinary load)
    float4 sh1[4], sh2[4], sh3[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        sh1[j] = *reinterpret_cast<const float4*>(p.b1 + j * 16 + fq * 4);
        sh2[j] = *reinterpret_cast<const float4*>(p.b2 + j * 16 + fq * 4);
        sh3[j] = *reinterpret_cast<const float4*>(p.b3 + (wave * 4 + j) * 16 + fq * 4);
    }

    // ---------------------------------------------------------------------------------------------------- phase 1
    // compute row cr (0..191) = halo pixel cr for cr < 180 (row-major over the 10 x 18 halo), the rest is padding
    unsigned xoff[3];
#pragma unroll
    for (int it = 0; it < 3; ++it) {
        const int cr = (tid >> 2) + it * 64, kce = swz<4>(cr, tid & 3);
        const int hr = cr / HC, hc = cr - hr * HC;
        const int gh = h0 - 1 + hr, gw = w0 - 1 + hc;
        const bool ok = cr < NHP && (unsigned)gh < (unsigned)p.H && (unsigned)gw < (unsigned)p.W;
        xoff[it] = ok ? (unsigned)(((img_pix + (long)gh * p.W + gw) * CX + kce * 8) * 2) : OOB;
    }
    const unsigned w1off = (unsigned)(((tid >> 2) * CX + swz<4>(tid >> 2, tid & 3) * 8) * 2);
    // The ring: four stages of (x slab 12 KB | W1 slab 4 KB) laid over the whole LDS (nothing else lives there yet), three slabs in
    // flight; a slab is 192 MFMA cycles of work per wave, far less than the latency of its DMA, so the depth is what hides it.
    constexpr int kStage1 = CR * 64 + MID * 64;      // 16384 B
    auto issue1 = [&](int s) {
        const int st = s & 3;
#pragma unroll
        for (int it = 0; it < 3; ++it) glds16(rx, &lds[st * (kStage1 / 16) + wbase + it * 256], xoff[it] == OOB ? OOB : xoff[it] + (unsigned)(s * 64));
        glds16(rw1, &lds[st * (kStage1 / 16) + CR * 4 + wbase], w1off + (unsigned)(s * 64));
    };
    f32x4_t acc1[3][4];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc1[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    {
        const int xrow = wave * 48 + fr;
        const unsigned x_rd = lbase + (unsigned)(xrow * 4 + swz<4>(xrow, fq)) * 16u;
        const unsigned w_rd = lbase + CR * 64 + (unsigned)(fr * 4 + swz<4>(fr, fq)) * 16u;
        constexpr int S = CX / 32;
#pragma unroll
        for (int s = 0; s < 3 && s < S; ++s) issue1(s);
#pragma unroll
        for (int s = 0; s < S; ++s) {
            constexpr int kPieces = 4;               // DMA instructions per thread and slab
            const int younger = (S - 1 - s < 2 ? S - 1 - s : 2) * kPieces;
            if (younger == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (younger == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();            // slab s visible; everyone is done reading slab s - 1, whose stage slab s + 3 takes
            __builtin_amdgcn_sched_barrier(0);
            if (s + 3 < S) issue1(s + 3);
            u32x4_t xf[3], wf[4];
            frag_read_all<3, 64>(xf, x_rd + (unsigned)((s & 3) * kStage1));
            frag_read_all<4, 64>(wf, w_rd + (unsigned)((s & 3) * kStage1));
            frag_wait<3, 4>(xf, wf);
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc1[i][j] = Mma<bf16_t>::run(wf[j], xf[i], acc1[i][j]);
        }
    }
    __builtin_amdgcn_s_barrier();                    // every wave is done with the x / W1 rings
    __builtin_amdgcn_sched_barrier(0);
    // W2 streams tap by tap through a four-stage ring in region A (three taps in flight); the first three start now
    unsigned w2off[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int c = tid + it * 256, row = c >> 3, q = c & 7;
        w2off[it] = (unsigned)(((row * 9) * MID + (q ^ ((row >> 1) & 7)) * 8) * 2);
    }
    auto issue_tap = [&](int t) {
#pragma unroll
        for (int it = 0; it < 2; ++it) glds16(rw2, &lds[kRegA / 16 + (t & 3) * 512 + wbase + it * 256], w2off[it] + (unsigned)(t * MID * 2));
    };
    issue_tap(0); issue_tap(1); issue_tap(2);
    // a1 = relu(acc1 + b1) (0 outside the image: conv2's zero padding) -> bf16 rows of the 10 x 24 LDS image
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int cr = (wave * 3 + i) * 16 + fr;
        const int hr = cr / HC, hc = cr - hr * HC;
        const int gh = h0 - 1 + hr, gw = w0 - 1 + hc;
        const bool inside = (unsigned)gh < (unsigned)p.H && (unsigned)gw < (unsigned)p.W;
        const int ar = hr * A1P + hc;
        if (cr < NHP) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int chunk = (j * 2 + (fq >> 1)) ^ ((ar >> 1) & 7);
                lds_write_b64(lbase + kRegB + (unsigned)(ar * 128 + chunk * 16 + (fq & 1) * 8), relu_pack4(acc1[i][j], sh1[j], inside));
            }
        }
    }
    // ---------------------------------------------------------------------------------------------------- phase 2
    f32x4_t acc2[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc2[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    {
        // B fragment of (pixel block b = 2 * wave + i, tap kh, kw): a1 rows (b + kh) * 24 + kw + fr.  24 = 16 + 8, so the row's
        // swizzle term ((row >> 1) & 7) depends on fr and on the residue class rho = ((i + kh) & 1) * 8 + kw only: one base address
        // per (rho, k-step) and lane, everything else is an immediate offset.
        unsigned a1b[2][3][2];                        // [parity of i + kh][kw][ks]
#pragma unroll
        for (int par = 0; par < 2; ++par)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const int row = par * 8 + kw + fr;
                    a1b[par][kw][ks] = lbase + kRegB + (unsigned)(wave * 2 * A1P * 128) + (unsigned)(row * 128 + (((ks * 4 + fq) ^ ((row >> 1) & 7)) << 4));
                }
        unsigned wb[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) wb[ks] = lbase + kRegA + (unsigned)((fr * 8 + ((ks * 4 + fq) ^ ((fr >> 1) & 7))) * 16);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // my a1 rows are written
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int kh = t / 3, kw = t - kh * 3;
            if (t <= 6) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");      // tap t landed; taps t + 1, t + 2 (two pieces each) still fly
            else if (t == 7) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();            // (t = 0: every wave's a1 rows are visible too)
            __builtin_amdgcn_sched_barrier(0);
            if (t + 3 < 9) issue_tap(t + 3);         // into the stage of tap t - 1, which every wave has finished reading
            u32x4_t af[2][2], wf[2][4];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                // (i + kh) * 24 + kw - rho is a multiple of 16: {0, 16, 48, 64} rows
                af[ks][0] = kh == 0 ? frag_read<0>(a1b[0][kw][ks]) : kh == 1 ? frag_read<16 * 128>(a1b[1][kw][ks]) : frag_read<48 * 128>(a1b[0][kw][ks]);
                af[ks][1] = kh == 0 ? frag_read<16 * 128>(a1b[1][kw][ks]) : kh == 1 ? frag_read<48 * 128>(a1b[0][kw][ks]) : frag_read<64 * 128>(a1b[1][kw][ks]);
                frag_read_all<4, 128>(wf[ks], wb[ks] + (unsigned)((t & 3) * 8192));
            }
            frag_wait<2, 4>(af[0], wf[0]);
            frag_wait<2, 4>(af[1], wf[1]);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc2[i][j] = Mma<bf16_t>::run(wf[ks][j], af[ks][i], acc2[i][j]);
        }
    }
    __builtin_amdgcn_s_barrier();                    // every wave is done with a1 and the tap ring: W3 (32 KB) takes the ring's place
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int c = tid + it * 256, row = c >> 3, q = c & 7;
        glds16(rw3, &lds[kRegA / 16 + wbase + it * 256], (unsigned)((row * MID + (q ^ ((row >> 1) & 7)) * 8) * 2));
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (wave * 2 + i) * 16 + fr;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int chunk = (j * 2 + (fq >> 1)) ^ ((row >> 1) & 7);
            lds_write_b64(lbase + kRegC + (unsigned)(row * 128 + chunk * 16 + (fq & 1) * 8), relu_pack4(acc2[i][j], sh2[j], true));
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     // W3 landed (it flew under the a2 epilogue), my a2 rows are written
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // ---------------------------------------------------------------------------------------------------- phase 3
    f32x4_t acc3[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc3[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int sw = (ks * 4 + fq) ^ ((fr >> 1) & 7);
        const unsigned ab = lbase + kRegC + (unsigned)((fr * 8 + sw) * 16);
        const unsigned w3b = lbase + kRegA + (unsigned)(((wave * 64 + fr) * 8 + sw) * 16);
        u32x4_t af[8], wf[4];
        frag_read_all<8, 128>(af, ab);
        frag_read_all<4, 128>(wf, w3b);
        frag_wait<8, 4>(af, wf);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc3[i][j] = Mma<bf16_t>::run(wf[j], af[i], acc3[i][j]);
    }
    __builtin_amdgcn_s_barrier();                    // a2 / W3 are dead: the staging tile covers them
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int co = (wave * 4 + j) * 16 + fq * 4;
        const float4 sh = sh3[j];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            uint2 t;
            t.x = pack2_bf16(acc3[i][j][0] + sh.x, acc3[i][j][1] + sh.y);
            t.y = pack2_bf16(acc3[i][j][2] + sh.z, acc3[i][j][3] + sh.w);
            lds_write_b64(lbase + (unsigned)((i * 16 + fr) * kStgRow + co * 2), t);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // y = relu(staged + residual): a lane finishes 16 B of one pixel, a wave instruction two whole pixel rows (2 x 512 B)
    {
        const __amdgpu_buffer_rsrc_t rr = make_rsrc_uniform(p.res, p.y_bytes);
        const __amdgpu_buffer_rsrc_t ry = make_rsrc_uniform(p.y, p.y_bytes);
        constexpr int NI = TH * TW * (CO / 8) / 256;  // 16
        unsigned goff[NI];
        u32x4_t rres[NI];
#pragma unroll
        for (int it = 0; it < NI; ++it) {
            const int c = tid + it * 256, row = c >> 5, ch = c & 31;
            const int gh = h0 + (row >> 4), gw = w0 + (row & 15);
            const bool ok = gh < p.H && gw < p.W;
            goff[it] = ok ? (unsigned)(((img_pix + (long)gh * p.W + gw) * CO + ch * 8) * 2) : OOB;
        }
#pragma unroll
        for (int it = 0; it < NI; ++it) rres[it] = __builtin_amdgcn_raw_buffer_load_b128(rr, goff[it], 0, 0);
        u32x4_t sv[NI];
        stg_read_all<NI, 8 * kStgRow>(sv, lbase + (unsigned)((tid >> 5) * kStgRow + (tid & 31) * 16));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int it = 0; it < NI; ++it) asm volatile("" : "+v"(sv[it]));
#pragma unroll
        for (int it = 0; it < NI; ++it) {
            u32x4_t ov;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float lo = fmaxf(__uint_as_float(sv[it][q] << 16) + __uint_as_float(rres[it][q] << 16), 0.f);
                const float hi = fmaxf(__uint_as_float(sv[it][q] & 0xffff0000u) + __uint_as_float(rres[it][q] & 0xffff0000u), 0.f);
                ov[q] = pack2_bf16(lo, hi);
            }
            __builtin_amdgcn_raw_buffer_store_b128(ov, ry, goff[it], 0, 0);
        }
    }
}

// out[r][c] = bf16(w[r][c] * scale[r]) for a table of matrices (FrozenBN scale folded into the rows of a conv weight)
struct FoldItem { const float* w; const float* scale; bf16_t* out; int rows, cols, chunk_begin, reserved; };
__global__ __launch_bounds__(256) void fold_weights_kernel(const FoldItem* __restrict__ items, int n_items, int total_chunks) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;         // one 8-element chunk of one matrix
    if (c >= total_chunks) return;
    int k = 0;
    for (int i = 1; i < n_items; ++i)
        if (c >= items[i].chunk_begin) k = i;
    const FoldItem it = items[k];
    const long e0 = (long)(c - it.chunk_begin) * 8;
    const int r = (int)(e0 / it.cols);
    const float s = it.scale ? it.scale[r] : 1.f;
    const float4 a = *reinterpret_cast<const float4*>(it.w + e0), b = *reinterpret_cast<const float4*>(it.w + e0 + 4);
    uint4 o;
    o.x = pack2_bf16(a.x * s, a.y * s); o.y = pack2_bf16(a.z * s, a.w * s);
    o.z = pack2_bf16(b.x * s, b.y * s); o.w = pack2_bf16(b.z * s, b.w * s);
    *reinterpret_cast<uint4*>(it.out + e0) = o;
}

}  // namespace

extern "C" int aldi_fold_weights_batch(const aldi_fold_item* items, int n_items, int total_chunks, aldi_stream_t stream) {
    static_assert(sizeof(aldi_fold_item) == sizeof(FoldItem), "descriptor layout");
    if (!items || n_items < 1 || total_chunks < 1) return aldi_set_error_msg(ALDI_ERR_ARG, "fold_weights_batch: bad args");
    hipLaunchKernelGGL(fold_weights_kernel, dim3(cdiv(total_chunks, 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       reinterpret_cast<const FoldItem*>(items), n_items, total_chunks);
    ALDI_CHECK_LAUNCH();
    return ALDI_OK;
}

extern "C" int aldi_bottleneck_fused(const aldi_bottleneck_args* a, aldi_stream_t stream) {
    if (!a || !a->x || !a->res || !a->y || !a->w1 || !a->w2 || !a->w3 || !a->b1 || !a->b2 || !a->b3)
        return aldi_set_error_msg(ALDI_ERR_ARG, "bottleneck_fused: null pointer");
    if (a->mid != MID || a->Cout != CO || (a->Cin != 64 && a->Cin != 256) || a->N < 1 || a->H < 1 || a->W < 1)
        return aldi_set_error_msg(ALDI_ERR_ARG, "bottleneck_fused: built for Cin in {64, 256}, 64 mid channels, 256 output channels (res2)");
    const size_t xb = (size_t)a->N * a->H * a->W * a->Cin * 2, yb = (size_t)a->N * a->H * a->W * CO * 2;
    if (xb >= 0x80000000ull || yb >= 0x80000000ull) return aldi_set_error_msg(ALDI_ERR_ARG, "bottleneck_fused: operand larger than 2 GiB (32-bit buffer offsets)");
    BneckDev d;
    d.x = static_cast<const bf16_t*>(a->x); d.res = static_cast<const bf16_t*>(a->res); d.y = static_cast<bf16_t*>(a->y);
    d.w1 = static_cast<const bf16_t*>(a->w1); d.w2 = static_cast<const bf16_t*>(a->w2); d.w3 = static_cast<const bf16_t*>(a->w3);
    d.b1 = a->b1; d.b2 = a->b2; d.b3 = a->b3;
    d.N = a->N; d.H = a->H; d.W = a->W; d.th = cdiv(a->H, TH); d.tw = cdiv(a->W, TW);
    d.xcd = aldi_tuning().igemm_xcd;
    d.x_bytes = (unsigned)xb; d.y_bytes = (unsigned)yb;
    const int tiles = a->N * d.th * d.tw;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (a->Cin == 64) hipLaunchKernelGGL(bneck_kernel<64>, dim3(tiles), dim3(256), 0, st, d);
    else hipLaunchKernelGGL(bneck_kernel<256>, dim3(tiles), dim3(256), 0, st, d);
    ALDI_CHECK_LAUNCH();
    char name[64];
    snprintf(name, sizeof(name), "bottleneck_fused<bf16,%d,64,256>", a->Cin);
    aldi_note_dispatch(name);
    return ALDI_OK;
}
