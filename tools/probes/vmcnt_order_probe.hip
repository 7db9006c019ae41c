// Does `s_waitcnt vmcnt(N)` on gfx950 retire vector-memory operations in ISSUE order when loads and stores are mixed?  The counted waits of
// igemm_ws.h / igemm_halo64.h leave younger STORES among the N operations allowed in flight while they rely on an OLDER load having landed.
// Each lane: a slow load (its own cold 128-byte line of a 1-GB buffer), then NS fast stores (a hot, L2-resident word), then `s_waitcnt vmcnt(NS)`
// and an immediate use of the loaded register (all hand-issued: the compiler knows of no outstanding operation and adds no wait of its own).
// If a store could retire ahead of the load, vmcnt would drop to NS with the load still in flight and the register would still hold the sentinel.
// Control: the same with `vmcnt(NS + 1)` (waits for nothing) must show sentinels -- it proves the probe can see a load that has not landed.
//   build: hipcc --offload-arch=gfx950 -O3 -o /tmp/vmcnt_order_probe tools/probes/vmcnt_order_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int NS, int WAIT>
__global__ __launch_bounds__(256) void probe(const unsigned* __restrict__ cold, unsigned* __restrict__ hot, unsigned* __restrict__ out, unsigned long long stride_words) {
    const unsigned long long gid = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned* src = cold + gid * stride_words;          // one line per lane, never touched before
    unsigned* dst = hot + (threadIdx.x & 63);
    unsigned v = 0xdeadbeefu, x = (unsigned)gid;
    asm volatile("global_load_dword %0, %1, off" : "+v"(v) : "v"(src) : "memory");
#pragma unroll
    for (int i = 0; i < NS; ++i) asm volatile("global_store_dword %0, %1, off" ::"v"(dst), "v"(x) : "memory");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAIT) : "memory");
    unsigned seen;
    asm volatile("v_mov_b32 %0, %1" : "=v"(seen) : "v"(v));   // what the register holds right behind the wait
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    out[gid] = seen;
}

template <int NS, int WAIT>
void run(const char* what, const unsigned* cold, unsigned* hot, unsigned* out, int wgs, unsigned long long stride_words) {
    CK(hipMemset(out, 0, (size_t)wgs * 256 * 4));
    hipLaunchKernelGGL((probe<NS, WAIT>), dim3(wgs), dim3(256), 0, 0, cold, hot, out, stride_words);
    CK(hipDeviceSynchronize());
    unsigned* h = (unsigned*)malloc((size_t)wgs * 256 * 4);
    CK(hipMemcpy(h, out, (size_t)wgs * 256 * 4, hipMemcpyDeviceToHost));
    long stale = 0, ok = 0;
    for (long i = 0; i < (long)wgs * 256; ++i) { if (h[i] == 0xdeadbeefu) ++stale; else if (h[i] == 0x01010101u) ++ok; }
    printf("%-64s lanes %8ld   loaded value present %8ld   register still holds the sentinel %8ld\n", what, (long)wgs * 256, ok, stale);
    free(h);
}

int main() {
    const int wgs = 8192;                                      // 2 M lanes x 128 B = 256 MB of distinct cold lines per run
    const unsigned long long stride_words = 32;
    const size_t cold_bytes = (size_t)wgs * 256 * stride_words * 4;
    unsigned *cold[4], *hot, *out;
    for (int i = 0; i < 4; ++i) { CK(hipMalloc(&cold[i], cold_bytes)); CK(hipMemset(cold[i], 1, cold_bytes)); }
    CK(hipMalloc(&hot, 4096)); CK(hipMalloc(&out, (size_t)wgs * 256 * 4));
    // evict: a 1-GB sweep of other memory between the memsets and the probes
    unsigned* flush; CK(hipMalloc(&flush, (size_t)1 << 30)); CK(hipMemset(flush, 3, (size_t)1 << 30)); CK(hipDeviceSynchronize());
    printf("# one cold load, then NS hot stores, then s_waitcnt vmcnt(WAIT), then the loaded register is read\n");
    run<4, 4>("load; 4 stores; vmcnt(4)  (the load must have landed)", cold[0], hot, out, wgs, stride_words);
    run<16, 16>("load; 16 stores; vmcnt(16) (the load must have landed)", cold[1], hot, out, wgs, stride_words);
    run<4, 5>("CONTROL load; 4 stores; vmcnt(5)  (waits for nothing)", cold[2], hot, out, wgs, stride_words);
    run<1, 1>("load; 1 store; vmcnt(1)", cold[3], hot, out, wgs, stride_words);
    return 0;
}
