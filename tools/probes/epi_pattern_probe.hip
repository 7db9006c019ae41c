// Does the ACCESS PATTERN of the direct epilogue bound the memory-bound 1x1 layers?  y[m][c] = relu(r[m][c] + 1) over a [M][C] bf16 map
// (the residual read + output write of a bottleneck's conv3: no GEMM at all), 16 bytes per lane, three lane -> address maps:
//   pat 0: fully coalesced (a wave instruction covers 1 KB contiguous)
//   pat 1: the direct epilogue's: lane (fr = lane & 15, fq = lane >> 4) -> pixel row fr, bytes [h*64 + fq*16, +16): 16 rows x 64 B per instruction
//   pat 2: 8 lanes per pixel row: 8 rows x 128 B per instruction (whole cache lines)
//   pat 3: 16 lanes per row: 4 rows x 256 B
//   build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/build/epi_pattern_probe tools/probes/epi_pattern_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

// a wave owns a tile of 64 pixels x 128 channels (256 B per row) = the direct epilogue's unit (TM = 4, two 32-channel blocks... x2)
__global__ __launch_bounds__(256) void epi_kernel(const uint4* __restrict__ R, uint4* __restrict__ Y, int M, int C, int pat, int do_read, int do_write) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rowb = C * 2;                                  // bytes per pixel row
    const int tiles_n = rowb / 256;                          // 256-byte column tiles
    const long tile = (long)blockIdx.x * 4 + wave;
    const long mt = tile / tiles_n; const int nt = (int)(tile % tiles_n);
    const long m0 = mt * 64;
    if (m0 >= M) return;
    u32x4_t acc = {0, 0, 0, 0};
    // 64 rows x 256 B = 16 KB per wave = 16 instructions of 1 KB
#pragma unroll
    for (int it = 0; it < 16; ++it) {
        long row; int byte;
        if (pat == 0) { const int lin = it * 1024 + lane * 16; row = lin / 256; byte = lin % 256; }        // 4 rows x 256 B contiguous per instr (rows are 256-B pieces of different pixel rows)
        else if (pat == 1) { const int i = it >> 2, h = it & 3; row = i * 16 + (lane & 15); byte = h * 64 + (lane >> 4) * 16; }
        else if (pat == 2) { const int i = it >> 1, h = it & 1; row = i * 8 + (lane >> 3); byte = h * 128 + (lane & 7) * 16; }
        else { row = it * 4 + (lane >> 4); byte = (lane & 15) * 16; }
        const long off = ((m0 + row) * rowb + nt * 256 + byte) / 16;
        u32x4_t v = {1, 2, 3, 4};
        if (do_read) v = *reinterpret_cast<const u32x4_t*>(&R[off]);
        if (do_write) { v.x += 0x00010001u; *reinterpret_cast<u32x4_t*>(&Y[off]) = v; }
        else acc += v;
    }
    if (!do_write && acc.x == 0x12345u) Y[0] = make_uint4(acc.x, acc.y, acc.z, acc.w);
}

int main() {
    const int M = 67200, C = 512;
    const size_t bytes = (size_t)M * C * 2;
    uint4 *R, *Y;
    CK(hipMalloc(&R, bytes)); CK(hipMalloc(&Y, bytes));
    CK(hipMemset(R, 1, bytes)); CK(hipMemset(Y, 0, bytes));
    // a 600-MB scratch read between runs to push the maps out of the 256-MB last-level cache
    uint4* F; const size_t fb = 600u << 20; CK(hipMalloc(&F, fb)); CK(hipMemset(F, 2, fb));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int tiles = (M / 64) * (C * 2 / 256), wgs = (tiles + 3) / 4;
    printf("# y = f(r) over a %d x %d bf16 map (%.1f MB each way), 16 B per lane; cold = the maps evicted from the last-level cache before the run\n", M, C, bytes / 1e6);
    for (int mode = 0; mode < 3; ++mode)                       // 0 read+write, 1 read only, 2 write only
        for (int pat = 0; pat < 4; ++pat)
            for (int cold = 0; cold < 2; ++cold) {
                float best = 1e9f;
                for (int rep = 0; rep < 5; ++rep) {
                    if (cold) CK(hipMemsetAsync(F, rep, fb));
                    CK(hipEventRecord(e0));
                    hipLaunchKernelGGL(epi_kernel, dim3(wgs), dim3(256), 0, 0, R, Y, M, C, pat, mode != 2, mode != 1);
                    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    if (rep > 0 && ms < best) best = ms;
                }
                const double moved = bytes * (mode == 0 ? 2.0 : 1.0);
                printf("%-10s pat %d %s: %7.1f us  %5.2f TB/s\n", mode == 0 ? "read+write" : mode == 1 ? "read" : "write", pat, cold ? "cold" : "warm", best * 1e3, moved / (best * 1e-3) / 1e12);
            }
    return 0;
}
