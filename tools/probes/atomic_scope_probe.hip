// Float atomic adds of 128-B segments (32 lanes x 4 B) to random rows of a 45 MB buffer, the access pattern of the deformable
// attention's value-gradient scatter: agent scope (executes memory-side: every XCD may touch every line) against workgroup scope
// issued only by workgroups of the XCD that owns the line's head segment (executes in that XCD's L2).  Prints G segment-atomics/s
// and checks the sums.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int SCOPE>
__global__ __launch_bounds__(256) void scatter(float* buf, int rows, int M, int per_wave, unsigned* xcc_seen) {
    const int xcc = __builtin_amdgcn_s_getreg(6164) & 15;          // HW_REG_XCC_ID[3:0]
    if (threadIdx.x == 0) atomicOr(xcc_seen + (blockIdx.x & 7), 1u << xcc);
    const int lane = threadIdx.x & 63, d = lane & 31, half = lane >> 5;
    unsigned s = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 2654435761u + 12345u + half * 977u;
    const int m = SCOPE ? xcc % M : (blockIdx.x % M);
    for (int i = 0; i < per_wave; ++i) {
        s = s * 1664525u + 1013904223u;
        const int row = (s >> 8) % rows;
        float* p = buf + ((long)row * M + m) * 32 + d;
        if (SCOPE) __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

__global__ void total(const float* buf, long n, double* out) {
    double acc = 0;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) acc += buf[i];
    atomicAdd(out, acc);
}

int main() {
    const int rows = 2 * 22357, M = 8, per_wave = 256, blocks = 8192;
    const long n = (long)rows * M * 32;
    float* buf; double* sum; unsigned* seen;
    CK(hipMalloc(&buf, n * 4)); CK(hipMalloc(&sum, 8)); CK(hipMalloc(&seen, 32));
    for (int scope = 0; scope < 2; ++scope) {
        CK(hipMemset(buf, 0, n * 4)); CK(hipMemset(sum, 0, 8)); CK(hipMemset(seen, 0, 32));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int rep = 0; rep < 2; ++rep) {
            if (rep == 1) { CK(hipMemset(buf, 0, n * 4)); CK(hipEventRecord(e0)); }
            if (scope) hipLaunchKernelGGL(scatter<1>, dim3(blocks), dim3(256), 0, 0, buf, rows, M, per_wave, seen);
            else hipLaunchKernelGGL(scatter<0>, dim3(blocks), dim3(256), 0, 0, buf, rows, M, per_wave, seen);
        }
        CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        hipLaunchKernelGGL(total, dim3(1024), dim3(256), 0, 0, buf, n, sum);
        double h; CK(hipMemcpy(&h, sum, 8, hipMemcpyDeviceToHost));
        unsigned hs[8]; CK(hipMemcpy(hs, seen, 32, hipMemcpyDeviceToHost));
        const double segs = (double)blocks * 4 * 2 * per_wave;
        printf("%s scope: %.3f ms, %.1f G segment-atomics/s (%.2f TB/s of 128-B segments), sum %.0f expected %.0f %s | XCC ids seen by blockIdx%%8:", scope ? "workgroup (XCD-owned lines)" : "agent", ms,
               segs / ms / 1e6, segs * 128 / ms / 1e9, h, segs * 32, h == segs * 32 ? "OK" : "MISMATCH");
        for (int i = 0; i < 8; ++i) printf(" %x", hs[i]);
        printf("\n");
    }
    return 0;
}
