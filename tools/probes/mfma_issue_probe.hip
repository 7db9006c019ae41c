// r06: what ONE wave per SIMD can feed the matrix pipe, per MFMA shape -- the r05 probe (mfma_clock_probe.hip) let hipcc schedule a C++ loop of
// builtins and hipcc rotated its accumulators through v_accvgpr_mov / s_nop chains (10 extra issues per 2 MFMAs: the "27 clk" was the compiler's
// loop, not the pipe).  Here every block is ONE inline-asm statement: the instruction stream is exactly what is written below.
//   arms: shape {16x16x32, 32x32x16} x accumulators in {VGPR, AGPR} x workgroups per CU {1, 2} x fillers:
//     bare      : MFMAs only (16 independent 4-register / 8 independent 16-register accumulators per wave)
//     lds       : + the fragment reads of a 64 px x 128 ch per-wave tile, double-buffered in registers, interleaved one per MFMA slot
//                 (12 ds_read_b128 per 32 16x16x32 MFMAs = per 16 32x32x16 MFMAs: the same bytes per flop), one lgkmcnt(0) per block
//   build: hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_issue_probe tools/probes/mfma_issue_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;

#define M16(C) "v_mfma_f32_16x16x32_bf16 %" #C ", %[a], %[b], %" #C "\n"
#define M32(C) "v_mfma_f32_32x32x16_bf16 %" #C ", %[a], %[b], %" #C "\n"
#define M16F(C, A, B) "v_mfma_f32_16x16x32_bf16 %" #C ", %[" #A "], %[" #B "], %" #C "\n"
#define M32F(C, A, B) "v_mfma_f32_32x32x16_bf16 %" #C ", %[" #A "], %[" #B "], %" #C "\n"
#define RD(D, OFF) "ds_read_b128 %[" #D "], %[ad] offset:" #OFF "\n"

// MODE 0: 16x16x32 VGPR acc; 1: 16x16x32 AGPR acc; 2: 32x32x16 VGPR; 3: 32x32x16 AGPR
// operand data: rnd = 0 -> near-constant values (few bits toggle: the chip holds ~2.39 GHz); rnd = 1 -> every lane's eight bf16 values are hashed
// (random sign and mantissa, exponents spread over 2^-3 .. 2^0: what activations x weights look like) -- DVFS then sets the clock by the power drawn
__device__ __forceinline__ short rnd_bf16(unsigned k) {
    k ^= k >> 16; k *= 0x7feb352du; k ^= k >> 15; k *= 0x846ca68bu; k ^= k >> 16;
    return (short)(((k & 1u) << 15) | ((0x7cu + ((k >> 1) & 3u)) << 7) | ((k >> 3) & 0x7fu));
}
template <int MODE>
__global__ __launch_bounds__(256) void bare_kernel(int iters, float* out, unsigned long long* clk, int rnd) {
    bf16x8_t a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (short)(0x3f80 + threadIdx.x % 3); b[i] = (short)(0x3c00 + threadIdx.x % 5); }
    if (rnd) for (int i = 0; i < 8; ++i) { a[i] = rnd_bf16((blockIdx.x * 256 + threadIdx.x) * 16 + i); b[i] = rnd_bf16((blockIdx.x * 256 + threadIdx.x) * 16 + 8 + i); }
    float s = 0.f;
    unsigned long long c0, c1, w0, w1;
    if constexpr (MODE < 2) {
        f32x4_t c[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) c[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        c0 = __builtin_readcyclecounter(); w0 = wall_clock64();
        for (int it = 0; it < iters; ++it) {
            if constexpr (MODE == 0)
                asm volatile(M16(0) M16(1) M16(2) M16(3) M16(4) M16(5) M16(6) M16(7) M16(8) M16(9) M16(10) M16(11) M16(12) M16(13) M16(14) M16(15)
                             : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]), "+v"(c[8]), "+v"(c[9]),
                               "+v"(c[10]), "+v"(c[11]), "+v"(c[12]), "+v"(c[13]), "+v"(c[14]), "+v"(c[15])
                             : [a] "v"(a), [b] "v"(b));
            else
                asm volatile(M16(0) M16(1) M16(2) M16(3) M16(4) M16(5) M16(6) M16(7) M16(8) M16(9) M16(10) M16(11) M16(12) M16(13) M16(14) M16(15)
                             : "+a"(c[0]), "+a"(c[1]), "+a"(c[2]), "+a"(c[3]), "+a"(c[4]), "+a"(c[5]), "+a"(c[6]), "+a"(c[7]), "+a"(c[8]), "+a"(c[9]),
                               "+a"(c[10]), "+a"(c[11]), "+a"(c[12]), "+a"(c[13]), "+a"(c[14]), "+a"(c[15])
                             : [a] "v"(a), [b] "v"(b));
        }
        c1 = __builtin_readcyclecounter(); w1 = wall_clock64();
#pragma unroll
        for (int i = 0; i < 16; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
    } else {
        f32x16_t c[8];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 16; ++j) c[i][j] = 0.f;
        c0 = __builtin_readcyclecounter(); w0 = wall_clock64();
        for (int it = 0; it < iters; ++it) {
            if constexpr (MODE == 2)
                asm volatile(M32(0) M32(1) M32(2) M32(3) M32(4) M32(5) M32(6) M32(7)
                             : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7])
                             : [a] "v"(a), [b] "v"(b));
            else
                asm volatile(M32(0) M32(1) M32(2) M32(3) M32(4) M32(5) M32(6) M32(7)
                             : "+a"(c[0]), "+a"(c[1]), "+a"(c[2]), "+a"(c[3]), "+a"(c[4]), "+a"(c[5]), "+a"(c[6]), "+a"(c[7])
                             : [a] "v"(a), [b] "v"(b));
        }
        c1 = __builtin_readcyclecounter(); w1 = wall_clock64();
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 16; ++j) s += c[i][j];
    }
    if (s == 12345.678f) out[0] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}

// The per-wave tile of the p2-size 3x3 kernel: 64 pixels x 128 channels, fragments double-buffered in registers, one k-step of 32 channels per block.
// 16x16x32: 4 pixel fragments (x) + 8 channel fragments (w) = 12 reads, 32 MFMAs.  32x32x16: per 16-channel half 2 + 4 = 6 reads and 8 MFMAs; x2 halves.
// Reads of the NEXT block's fragments are interleaved one per MFMA from the block's first MFMA on.
template <int MODE>   // 0: 16x16x32, 1: 32x32x16  (accumulators "+v": 128 registers either way)
__global__ __launch_bounds__(256) void lds_kernel(int iters, float* out, unsigned long long* clk, int lds_stride, int rnd) {
    __shared__ __attribute__((aligned(128))) uint4 lds[4096];            // 64 KB: two workgroups per CU fit
    for (int i = threadIdx.x; i < 4096; i += 256) {
        lds[i] = make_uint4(0x3f803f80u, 0x3c003c00u, 0x3f803f80u, 0x3c003c00u);
        if (rnd) {
            unsigned w[4];
            for (int q = 0; q < 4; ++q) w[q] = (unsigned)(unsigned short)rnd_bf16((blockIdx.x * 4096 + i) * 8 + 2 * q) | ((unsigned)(unsigned short)rnd_bf16((blockIdx.x * 4096 + i) * 8 + 2 * q + 1) << 16);
            lds[i] = make_uint4(w[0], w[1], w[2], w[3]);
        }
    }
    __syncthreads();
    const unsigned ad = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) void*)&lds[0] + (threadIdx.x & 63) * 16u + (threadIdx.x >> 6) * (unsigned)lds_stride;
    u32x4_t x0, x1, x2, x3, w0_, w1_, w2_, w3_, w4_, w5_, w6_, w7_;         // set A
    u32x4_t y0, y1, y2, y3, v0, v1, v2, v3, v4, v5, v6, v7;                 // set B
    x0 = x1 = x2 = x3 = w0_ = w1_ = w2_ = w3_ = w4_ = w5_ = w6_ = w7_ = u32x4_t{0x3f803f80u, 0x3c003c00u, 0x3f803f80u, 0x3c003c00u};
    if (rnd) {       // (the first block's fragments; every later block reads its own from the hashed LDS image)
        x0 = x1 = x2 = x3 = *reinterpret_cast<u32x4_t*>(&lds[threadIdx.x]);
        w0_ = w1_ = w2_ = w3_ = w4_ = w5_ = w6_ = w7_ = *reinterpret_cast<u32x4_t*>(&lds[256 + threadIdx.x]);
    }
    y0 = y1 = y2 = y3 = v0 = v1 = v2 = v3 = v4 = v5 = v6 = v7 = x0;
    float s = 0.f;
    unsigned long long c0, c1, w0, w1;
    if constexpr (MODE == 0) {
        f32x4_t c[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) c[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        c0 = __builtin_readcyclecounter(); w0 = wall_clock64();
        for (int it = 0; it < iters; ++it) {
#define ACC32 "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]), "+v"(c[8]), "+v"(c[9]), "+v"(c[10]), "+v"(c[11]), \
              "+v"(c[12]), "+v"(c[13]), "+v"(c[14]), "+v"(c[15]), "+v"(c[16]), "+v"(c[17]), "+v"(c[18]), "+v"(c[19]), "+v"(c[20]), "+v"(c[21]), "+v"(c[22]), "+v"(c[23]), \
              "+v"(c[24]), "+v"(c[25]), "+v"(c[26]), "+v"(c[27]), "+v"(c[28]), "+v"(c[29]), "+v"(c[30]), "+v"(c[31])
            // block A: MFMAs on set A (x0-3, w0-7), reads into set B
            asm volatile(
                M16F(0, w0, x0) RD(y0, 0) M16F(1, w1, x0) RD(y1, 2048) M16F(2, w2, x0) RD(y2, 4096) M16F(3, w3, x0) RD(y3, 6144)
                M16F(4, w4, x0) RD(v0, 8192) M16F(5, w5, x0) RD(v1, 10240) M16F(6, w6, x0) RD(v2, 12288) M16F(7, w7, x0) RD(v3, 14336)
                M16F(8, w0, x1) RD(v4, 16384) M16F(9, w1, x1) RD(v5, 18432) M16F(10, w2, x1) RD(v6, 20480) M16F(11, w3, x1) RD(v7, 22528)
                M16F(12, w4, x1) M16F(13, w5, x1) M16F(14, w6, x1) M16F(15, w7, x1)
                M16F(16, w0, x2) M16F(17, w1, x2) M16F(18, w2, x2) M16F(19, w3, x2) M16F(20, w4, x2) M16F(21, w5, x2) M16F(22, w6, x2) M16F(23, w7, x2)
                M16F(24, w0, x3) M16F(25, w1, x3) M16F(26, w2, x3) M16F(27, w3, x3) M16F(28, w4, x3) M16F(29, w5, x3) M16F(30, w6, x3) M16F(31, w7, x3)
                "s_waitcnt lgkmcnt(0)\n"
                : ACC32, [y0] "=&v"(y0), [y1] "=&v"(y1), [y2] "=&v"(y2), [y3] "=&v"(y3), [v0] "=&v"(v0), [v1] "=&v"(v1), [v2] "=&v"(v2), [v3] "=&v"(v3),
                  [v4] "=&v"(v4), [v5] "=&v"(v5), [v6] "=&v"(v6), [v7] "=&v"(v7)
                : [x0] "v"(x0), [x1] "v"(x1), [x2] "v"(x2), [x3] "v"(x3), [w0] "v"(w0_), [w1] "v"(w1_), [w2] "v"(w2_), [w3] "v"(w3_), [w4] "v"(w4_), [w5] "v"(w5_),
                  [w6] "v"(w6_), [w7] "v"(w7_), [ad] "v"(ad));
            asm volatile(
                M16F(0, w0, x0) RD(y0, 0) M16F(1, w1, x0) RD(y1, 2048) M16F(2, w2, x0) RD(y2, 4096) M16F(3, w3, x0) RD(y3, 6144)
                M16F(4, w4, x0) RD(v0, 8192) M16F(5, w5, x0) RD(v1, 10240) M16F(6, w6, x0) RD(v2, 12288) M16F(7, w7, x0) RD(v3, 14336)
                M16F(8, w0, x1) RD(v4, 16384) M16F(9, w1, x1) RD(v5, 18432) M16F(10, w2, x1) RD(v6, 20480) M16F(11, w3, x1) RD(v7, 22528)
                M16F(12, w4, x1) M16F(13, w5, x1) M16F(14, w6, x1) M16F(15, w7, x1)
                M16F(16, w0, x2) M16F(17, w1, x2) M16F(18, w2, x2) M16F(19, w3, x2) M16F(20, w4, x2) M16F(21, w5, x2) M16F(22, w6, x2) M16F(23, w7, x2)
                M16F(24, w0, x3) M16F(25, w1, x3) M16F(26, w2, x3) M16F(27, w3, x3) M16F(28, w4, x3) M16F(29, w5, x3) M16F(30, w6, x3) M16F(31, w7, x3)
                "s_waitcnt lgkmcnt(0)\n"
                : ACC32, [y0] "=&v"(x0), [y1] "=&v"(x1), [y2] "=&v"(x2), [y3] "=&v"(x3), [v0] "=&v"(w0_), [v1] "=&v"(w1_), [v2] "=&v"(w2_), [v3] "=&v"(w3_),
                  [v4] "=&v"(w4_), [v5] "=&v"(w5_), [v6] "=&v"(w6_), [v7] "=&v"(w7_)
                : [x0] "v"(y0), [x1] "v"(y1), [x2] "v"(y2), [x3] "v"(y3), [w0] "v"(v0), [w1] "v"(v1), [w2] "v"(v2), [w3] "v"(v3), [w4] "v"(v4), [w5] "v"(v5),
                  [w6] "v"(v6), [w7] "v"(v7), [ad] "v"(ad));
        }
        c1 = __builtin_readcyclecounter(); w1 = wall_clock64();
#pragma unroll
        for (int i = 0; i < 32; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
    } else {
        f32x16_t c[8];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 16; ++j) c[i][j] = 0.f;
        c0 = __builtin_readcyclecounter(); w0 = wall_clock64();
        for (int it = 0; it < iters; ++it) {
#define ACC8 "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7])
            // one 32-channel k-step = two 16-channel halves; per half 2 x-fragments (x0, x1 | x2, x3) and 4 w-fragments (w0-3 | w4-7): 16 MFMAs, 12 reads
            asm volatile(
                M32F(0, w0, x0) RD(y0, 0) M32F(1, w1, x0) RD(y1, 2048) M32F(2, w2, x0) RD(y2, 4096) M32F(3, w3, x0) RD(y3, 6144)
                M32F(4, w0, x1) RD(v0, 8192) M32F(5, w1, x1) RD(v1, 10240) M32F(6, w2, x1) RD(v2, 12288) M32F(7, w3, x1) RD(v3, 14336)
                M32F(0, w4, x2) RD(v4, 16384) M32F(1, w5, x2) RD(v5, 18432) M32F(2, w6, x2) RD(v6, 20480) M32F(3, w7, x2) RD(v7, 22528)
                M32F(4, w4, x3) M32F(5, w5, x3) M32F(6, w6, x3) M32F(7, w7, x3)
                "s_waitcnt lgkmcnt(0)\n"
                : ACC8, [y0] "=&v"(y0), [y1] "=&v"(y1), [y2] "=&v"(y2), [y3] "=&v"(y3), [v0] "=&v"(v0), [v1] "=&v"(v1), [v2] "=&v"(v2), [v3] "=&v"(v3),
                  [v4] "=&v"(v4), [v5] "=&v"(v5), [v6] "=&v"(v6), [v7] "=&v"(v7)
                : [x0] "v"(x0), [x1] "v"(x1), [x2] "v"(x2), [x3] "v"(x3), [w0] "v"(w0_), [w1] "v"(w1_), [w2] "v"(w2_), [w3] "v"(w3_), [w4] "v"(w4_), [w5] "v"(w5_),
                  [w6] "v"(w6_), [w7] "v"(w7_), [ad] "v"(ad));
            asm volatile(
                M32F(0, w0, x0) RD(y0, 0) M32F(1, w1, x0) RD(y1, 2048) M32F(2, w2, x0) RD(y2, 4096) M32F(3, w3, x0) RD(y3, 6144)
                M32F(4, w0, x1) RD(v0, 8192) M32F(5, w1, x1) RD(v1, 10240) M32F(6, w2, x1) RD(v2, 12288) M32F(7, w3, x1) RD(v3, 14336)
                M32F(0, w4, x2) RD(v4, 16384) M32F(1, w5, x2) RD(v5, 18432) M32F(2, w6, x2) RD(v6, 20480) M32F(3, w7, x2) RD(v7, 22528)
                M32F(4, w4, x3) M32F(5, w5, x3) M32F(6, w6, x3) M32F(7, w7, x3)
                "s_waitcnt lgkmcnt(0)\n"
                : ACC8, [y0] "=&v"(x0), [y1] "=&v"(x1), [y2] "=&v"(x2), [y3] "=&v"(x3), [v0] "=&v"(w0_), [v1] "=&v"(w1_), [v2] "=&v"(w2_), [v3] "=&v"(w3_),
                  [v4] "=&v"(w4_), [v5] "=&v"(w5_), [v6] "=&v"(w6_), [v7] "=&v"(w7_)
                : [x0] "v"(y0), [x1] "v"(y1), [x2] "v"(y2), [x3] "v"(y3), [w0] "v"(v0), [w1] "v"(v1), [w2] "v"(v2), [w3] "v"(v3), [w4] "v"(v4), [w5] "v"(v5),
                  [w6] "v"(v6), [w7] "v"(v7), [ad] "v"(ad));
        }
        c1 = __builtin_readcyclecounter(); w1 = wall_clock64();
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 16; ++j) s += c[i][j];
    }
    if (s == 12345.678f) out[0] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}

template <typename F>
static void run(const char* name, F launch, int wgs_per_cu, int iters, double mfma_per_iter, double flop_per_mfma, double clk_floor, float* out, unsigned long long* clk) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int wgs = 256 * wgs_per_cu;
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        launch(wgs, iters);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    CK(hipGetLastError());
    std::vector<unsigned long long> h(2 * wgs);
    CK(hipMemcpy(h.data(), clk, wgs * 16, hipMemcpyDeviceToHost));
    double cyc = 0, wall = 0;
    for (int i = 0; i < wgs; ++i) { cyc += (double)h[2 * i]; wall += (double)h[2 * i + 1]; }
    const double flop = (double)wgs * 4 * iters * mfma_per_iter * flop_per_mfma;
    const double per = (cyc / wgs) / ((double)iters * mfma_per_iter * wgs_per_cu);
    printf("%-34s %d wg/CU  %8.1f us  %7.1f TFLOP/s  clock %.0f MHz  one MFMA per %5.1f clk and SIMD (pipe floor %.0f: %.2f of the pipe)\n",
           name, wgs_per_cu, best * 1e3, flop / (best * 1e-3) / 1e12, cyc / wall * 100.0, per, clk_floor, clk_floor / per);
}

int main() {
    float* out; unsigned long long* clk;
    CK(hipMalloc(&out, 64)); CK(hipMalloc(&clk, 4096 * 16));
    const double F16 = 2.0 * 16 * 16 * 32, F32 = 2.0 * 32 * 32 * 16;
    printf("# register-only / LDS-fed bf16 MFMA streams, inline asm (the stream is what the source says), 256-thread workgroups, 20000 iterations\n");
    for (int rnd : {0, 1}) {
        printf("# operand data: %s\n", rnd ? "hashed bf16 (random sign / mantissa, four exponents)" : "near-constant");
        for (int w : {1, 2}) {
            run("16x16x32 bare, acc VGPR", [&](int g, int it) { hipLaunchKernelGGL(bare_kernel<0>, dim3(g), dim3(256), 0, 0, it, out, clk, rnd); }, w, 20000, 16, F16, 16, out, clk);
            run("16x16x32 bare, acc AGPR", [&](int g, int it) { hipLaunchKernelGGL(bare_kernel<1>, dim3(g), dim3(256), 0, 0, it, out, clk, rnd); }, w, 20000, 16, F16, 16, out, clk);
            run("32x32x16 bare, acc VGPR", [&](int g, int it) { hipLaunchKernelGGL(bare_kernel<2>, dim3(g), dim3(256), 0, 0, it, out, clk, rnd); }, w, 20000, 8, F32, 32, out, clk);
            run("32x32x16 bare, acc AGPR", [&](int g, int it) { hipLaunchKernelGGL(bare_kernel<3>, dim3(g), dim3(256), 0, 0, it, out, clk, rnd); }, w, 20000, 8, F32, 32, out, clk);
            run("16x16x32 + 12 ds_read_b128 / 32", [&](int g, int it) { hipLaunchKernelGGL(lds_kernel<0>, dim3(g), dim3(256), 0, 0, it, out, clk, 1024, rnd); }, w, 10000, 64, F16, 16, out, clk);
            run("32x32x16 + 12 ds_read_b128 / 16", [&](int g, int it) { hipLaunchKernelGGL(lds_kernel<1>, dim3(g), dim3(256), 0, 0, it, out, clk, 1024, rnd); }, w, 10000, 32, F32, 32, out, clk);
        }
    }
    return 0;
}
