// LDS-DMA (buffer_load_dwordx4 ... lds) fill rate per CU as a function of the contiguous SEGMENT each group of lanes fetches.
// The 3x3 halo kernels load 64-byte K slabs: four lanes per pixel row = HALF a 128-byte cache line per row.  Hypothesis: the
// L2 -> LDS path is paced per cache line touched, so 128-byte segments (64-channel K slabs) move twice the bytes per unit time.
//   build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/build/dma_rate_probe tools/probes/dma_rate_probe.hip
//   run:   tools/probes/build/dma_rate_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void dma_kernel(const unsigned short* __restrict__ X, unsigned x_bytes, int pitch_bytes, int seg_bytes, int rows_per_wg, int k_steps, int iters,
                                                         int depth, int swz, int shared_rows, unsigned long long* cycles) {
    __shared__ __attribute__((aligned(128))) uint4 lds[8 * WAVES * 64];        // ring of 8 pieces per wave
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(X), 0, x_bytes, 0x00020000);
    const int lanes_per_row = seg_bytes / 16;
    // this wave's piece: 64 lanes cover (64 / lanes_per_row) consecutive rows x one segment
    const int rows_per_piece = 64 / lanes_per_row;
    const int pieces_per_step = rows_per_wg / rows_per_piece;           // over the workgroup
    const int my_pieces = pieces_per_step / WAVES;                      // per wave and K step (>= 1)
    const int row0 = shared_rows ? 0 : blockIdx.x * rows_per_wg;
    int chunk = lane % lanes_per_row;
    const int rl = lane / lanes_per_row;
    unsigned long long t0 = __builtin_readcyclecounter();
    int issued = 0;
    for (int it = 0; it < iters; ++it) {
        for (int ks = 0; ks < k_steps; ++ks) {
            for (int p = 0; p < my_pieces; ++p) {
                const int row = row0 + (wave * my_pieces + p) * rows_per_piece + rl;
                const int c = swz ? (chunk ^ ((row >> 1) & (lanes_per_row - 1))) : chunk;
                const unsigned off = (unsigned)row * (unsigned)pitch_bytes + (unsigned)(ks * seg_bytes + c * 16);
                uint4* dst = &lds[((issued & 7) * WAVES + wave) * 64];
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (__attribute__((address_space(3))) void*)dst, 16, off, 0, 0, 0);
                ++issued;
                if (depth == 2) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
                else if (depth == 4) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                else if (depth == 8) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
                else if (depth == 16) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long long t1 = __builtin_readcyclecounter();
    if (tid == 0) cycles[blockIdx.x] = t1 - t0;
    if (lds[tid].x == 0x12345u) cycles[0] = 0;                           // keep the LDS alive
}

int main() {
    const int M = 268800, C = 256;                                        // the p2 map of the step: 4 x 200 x 336 pixels x 256 channels bf16
    std::vector<unsigned short> h((size_t)M * C, 0x3f80);
    unsigned short* X;
    CK(hipMalloc(&X, h.size() * 2));
    CK(hipMemcpy(X, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    unsigned long long* cyc;
    const int WGS = 1024;
    CK(hipMalloc(&cyc, WGS * 8));
    std::vector<unsigned long long> hc(WGS);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("# LDS-DMA fill rate: 8-wave workgroups (64 KB of LDS: 2 per CU), 256 rows per workgroup, rows %d B apart, K window kw bytes per row (the footprint per workgroup is 256 x kw)\n", C * 2);
    printf("# seg = contiguous bytes per row and piece (64 = half a cache line); depth = 1-KB pieces in flight per wave; shared = every workgroup reads the same rows (weights-like)\n");
    for (int shared_rows = 0; shared_rows < 2; ++shared_rows)
      for (int kwin : {128, 512})
        for (int depth : {4, 8, 16})
            for (int seg : {64, 128, 256}) {
                if (seg > kwin) continue;
                const int k_steps = kwin / seg, iters = 40 * 512 / kwin, rows = 256, swz = 1;
                const int wgs = WGS;
                for (int rep = 0; rep < 2; ++rep) {
                    CK(hipEventRecord(e0));
                    hipLaunchKernelGGL(dma_kernel<8>, dim3(wgs), dim3(512), 0, 0, X, (unsigned)(h.size() * 2), C * 2, seg, rows, k_steps, iters, depth, swz, shared_rows, cyc);
                    CK(hipEventRecord(e1));
                    CK(hipEventSynchronize(e1));
                }
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                CK(hipMemcpy(hc.data(), cyc, wgs * 8, hipMemcpyDeviceToHost));
                double mean = 0;
                for (int i = 0; i < wgs; ++i) mean += (double)hc[i];
                mean /= wgs;
                const double bytes_wg = (double)rows * kwin * iters;
                printf("shared %d kw %3d depth %2d seg %3d: %8.1f us  %6.2f TB/s chip   %6.1f B/clk per workgroup while resident (x2 per CU)\n", shared_rows, kwin, depth, seg, ms * 1e3,
                       bytes_wg * wgs / (ms * 1e-3) / 1e12, bytes_wg / mean);
            }
    return 0;
}
