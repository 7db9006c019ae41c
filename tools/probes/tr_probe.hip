// Probe of ds_read_b64_tr_b16 on gfx950: which LDS elements does each lane receive?
// LDS holds bf16-sized integers lds[i] = i.  Every lane passes its own byte address.
// build: hipcc --offload-arch=gfx950 -O2 tools/probes/tr_probe.hip -o /tmp/tr_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

__global__ void probe(uint16_t* out, int mode) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    const int l = threadIdx.x;
    unsigned addr;
    if (mode == 0) addr = (unsigned)l * 8u;                                   // lane-linear 8-B slots
    else if (mode == 1) addr = (unsigned)(((l & 3) * 16 + ((l & 15) >> 2) * 4 + (l >> 4) * 64) * 2);   // [4][16] block per 16 lanes: lane (s,r) -> row r, cols 4s..
    else addr = (unsigned)((((l & 15) >> 2) * 16 + (l & 3) * 4 + (l >> 4) * 64) * 2);                  // lane (r,s) the other way round
    addr += (unsigned)(uintptr_t)(const __attribute__((address_space(3))) void*)lds;
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr));
    out[l * 4 + 0] = v[0] & 0xffff; out[l * 4 + 1] = v[0] >> 16; out[l * 4 + 2] = v[1] & 0xffff; out[l * 4 + 3] = v[1] >> 16;
}

int main() {
    uint16_t* d; uint16_t h[256];
    hipMalloc(&d, sizeof(h));
    for (int mode = 0; mode < 3; ++mode) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, mode);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d (element indices received by lane: e0 e1 e2 e3)\n", mode);
        for (int l = 0; l < 64; ++l) printf("  lane %2d: %4d %4d %4d %4d%s", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3], (l & 3) == 3 ? "\n" : " |");
    }
    return 0;
}
