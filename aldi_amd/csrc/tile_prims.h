// Tile primitives shared by the MFMA kernels (igemm.hip, bneck.hip): LDS swizzles, LDS-DMA, hand-issued fragment reads, MFMA wrappers.
#pragma once
#include "common.h"

namespace {

// LDS rows of KC 16-B chunks.  KC = 8 (128-B rows): chunk ^ (row>>1)&7; KC = 4 (64-B rows): chunk ^ g[(row>>2)&3],
// g = [0,2,3,1].  Both make the 8-lane ds_write_b128 groups and the 16-lane ds_read_b128 groups hit distinct
// 16-B slots of the 256-B bank row.
template <int KC>
__device__ __forceinline__ int swz(int row, int kc) {
    if constexpr (KC == 8) return kc ^ ((row >> 1) & 7);
    else return kc ^ ((0x78 >> (((row >> 2) & 3) * 2)) & 3);
}

// The halo form of the 3x3 convs reads one slab of 64-B rows at row offsets +0 / +1 / +2 (the three horizontal taps).  A ds_read_b128 is
// served in 16-lane groups that take fragment rows {0-3, 12-15} with chunk fq and rows {4-11} with chunk fq ^ 1 (MI355X_MICROARCH.md, LDS):
// with swz<4> the four rows of one (row & 3) class land in distinct 16-B slots only when the first row is a multiple of 4 -- at +1 / +2 every
// group hits two slots twice (SQ_LDS_BANK_CONFLICT = 27 % of SQ_LDS_IDX_ACTIVE on the p2 conv, profiles/r04_pmc_conv.txt).  Keying the
// swizzle on bit 2 of the row alone (g = [0, 2, 0, 2]) is conflict-free for EVERY row offset: rows r, r+4, r+8, r+12 with logical chunks
// [a, a^1, a^1, a] (in either rotation) get physical chunks {a, a^1^2, a^1, a^2} -- a permutation.
__device__ __forceinline__ int swz_halo(int row, int kc) { return kc ^ ((row >> 1) & 2); }

// 16-B-per-lane global -> LDS DMA (`buffer_load_dwordx4 ... offen lds`): LDS address = wave-uniform `dst` + lane*16;
// an out-of-range `voff` writes zeros.
__device__ __forceinline__ void glds16(__amdgpu_buffer_rsrc_t rsrc, uint4* dst, unsigned voff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)dst, 16, voff, 0, 0, 0);
}

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));   // one 16-B LDS fragment (4 VGPRs)

// Fragment reads are hand-issued `ds_read_b128`: behind an LDS-DMA in flight the compiler cannot tell which ring slot a
// C++ LDS load aliases and drains the whole DMA queue (`s_waitcnt vmcnt(0)`) before the first read of every K slab,
// which serialises loads and MFMAs inside a wave.  The asm reads are invisible to that scoreboard; frag_wait() is the
// matching hand-placed lgkmcnt wait, tied to the fragments by "+v" so no MFMA can be scheduled above it.
__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) void*)p;
}
template <int OFF>
__device__ __forceinline__ u32x4_t frag_read(unsigned addr) {
    u32x4_t v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}

template <int N, int ROWB, int I = 0>
__device__ __forceinline__ void frag_read_all(u32x4_t* f, unsigned addr) {   // fragments of rows 16 apart
    if constexpr (I < N) {
        f[I] = frag_read<I * 16 * ROWB>(addr);
        frag_read_all<N, ROWB, I + 1>(f, addr);
    }
}
template <int NA, int NB>
__device__ __forceinline__ void frag_wait(u32x4_t* a, u32x4_t* b) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < NA; ++i) asm volatile("" : "+v"(a[i]));
#pragma unroll
    for (int i = 0; i < NB; ++i) asm volatile("" : "+v"(b[i]));
}
template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
    __device__ static __forceinline__ f32x4_t run(const u32x4_t& a, const u32x4_t& b, f32x4_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
    }
};
template <> struct Mma<float> {
    __device__ static __forceinline__ f32x4_t run(const u32x4_t& a, const u32x4_t& b, f32x4_t c) {
#pragma unroll
        for (int j = 0; j < 4; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[j]), __uint_as_float(b[j]), c, 0, 0, 0);
        return c;
    }
};

}  // namespace
