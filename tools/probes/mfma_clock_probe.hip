// What does the chip sustain when the MFMA pipes are the ONLY thing running?  A register-only loop of v_mfma_f32_16x16x32_bf16 (no LDS, no memory) on every
// SIMD, timed with HIP events; each wave also reads the shader-clock counter (s_memtime) and the 100-MHz wall counter (s_memrealtime): their ratio is the
// clock the kernel actually ran at.  The dense bf16 peak of MI355X_MICROARCH.md (2.5 PFLOP/s) is 256 CUs x 4 SIMDs x 1024 flop/clk x 2.4 GHz.
//   build: hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_clock_probe tools/probes/mfma_clock_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

template <int NACC>
__global__ __launch_bounds__(256) void mfma_kernel(int iters, float* out, unsigned long long* clk) {
    f32x4_t acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    bf16x8_t a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (short)(0x3f80 + threadIdx.x % 3); b[i] = (short)(0x3c00 + threadIdx.x % 5); }
    const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678f) out[0] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}

int main() {
    float* out; unsigned long long* clk;
    CK(hipMalloc(&out, 64)); CK(hipMalloc(&clk, 4096 * 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("# register-only bf16 MFMA loop (16 independent accumulators per wave, v_mfma_f32_16x16x32_bf16), 256-thread workgroups\n");
    for (int wgs_per_cu : {1, 2}) {
        for (int iters : {2000, 20000, 200000}) {
            const int wgs = 256 * wgs_per_cu;
            float best = 1e9f;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(mfma_kernel<16>, dim3(wgs), dim3(256), 0, 0, iters, out, clk);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            std::vector<unsigned long long> h(2 * wgs);
            CK(hipMemcpy(h.data(), clk, wgs * 16, hipMemcpyDeviceToHost));
            double cyc = 0, wall = 0;
            for (int i = 0; i < wgs; ++i) { cyc += (double)h[2 * i]; wall += (double)h[2 * i + 1]; }
            const double flop = (double)wgs * 4 * iters * 16 * (2.0 * 16 * 16 * 32);
            printf("%d workgroup(s) per CU, %6d x 16 MFMAs per wave: %9.1f us  %7.1f TFLOP/s   shader clock while running: %.0f MHz (s_memtime / s_memrealtime, 100 MHz)   MFMA issue: one per %.1f clk and SIMD\n",
                   wgs_per_cu, iters, best * 1e3, flop / (best * 1e-3) / 1e12, cyc / wall * 100.0, (cyc / wgs) / ((double)iters * 16 * wgs_per_cu));
        }
    }
    return 0;
}
