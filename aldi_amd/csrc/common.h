// Shared device/host helpers for the ALDI MI355X (gfx950) kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/aldi_hip.h"

#define ALDI_CHECK_LAUNCH()                                   \
    do {                                                      \
        hipError_t e__ = hipGetLastError();                   \
        if (e__ != hipSuccess) return aldi_set_error(e__, __FILE__, __LINE__); \
    } while (0)

int aldi_set_error(hipError_t e, const char* file, int line);
int aldi_set_error_msg(int code, const char* msg);

// run-time tuning knobs (aldi_set_tuning / ALDI_<NAME> environment defaults; core.hip)
struct AldiTuning {
    int igemm_xcd, igemm_tile, igemm_dbg, igemm_bigtile_min, igemm_bigtile_k, igemm_lintile_min, igemm_halo, igemm_force, igemm_group, igemm_k64_min, igemm_bigtile, igemm_narrow_k, igemm_splitk_tile, igemm_halo_f32, igemm_f32_tile64_max, igemm_direct, igemm_lean, igemm_halo64_mid, igemm_ws, igemm_ws_wgs, igemm_ws_min, igemm_halo_ilv, igemm_halo_small, igemm_halo96;
    int wgrad_lean, wgrad_big_min, wgrad_big_slots, wgrad_slots, wgrad_xcd, wgrad_dma, wgrad_dbg, wgrad_group_slots, wgrad_group_epi, roialign_sep, roialign_bwd_rows, wgrad_db, wgrad_ordered, wgrad_big_group, wgrad_big_epi, wgrad_big_group_min, wgrad_lds_pad_kb, wgrad_f32_tile128, wgrad_dma64, wgrad_ilv, msda_gather, msda_gather_list, msda_bin, msda_bin_list;
    int colsum_blocks, colsum_minrows, colsum_nt, colsum_block_kb;
    int stem_mfma, sab_blocks, ln_bwd_blocks, ln_bwd_blocks_narrow, rpn_topk_fused, ema_blocks, nms_mask_tri, match_wave;
};
AldiTuning& aldi_tuning();
void aldi_note_dispatch(const char* kernel);   // what aldi_last_dispatch() reports (thread local)

typedef uint16_t bf16_t;  // raw bf16 storage

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;   // MFMA bf16 A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) float f32x4_t;    // MFMA 16x16 accumulator / 16 B of fp32

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// round-to-nearest-even, NaN preserved (same rule as torch's float->bfloat16)
// (gfx950 has the conversion in hardware: v_cvt_pk_bf16_f32, two values per instruction)
typedef __bf16 hwbf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack2_bf16(float lo, float hi) {
    f32x2_t v = {lo, hi};
    hwbf16x2_t b = __builtin_convertvector(v, hwbf16x2_t);
    return *reinterpret_cast<uint32_t*>(&b);
}
__device__ __forceinline__ bf16_t f32_to_bf16(float f) { return (bf16_t)(pack2_bf16(f, 0.f) & 0xffffu); }

template <typename T> struct Elem;
template <> struct Elem<float> {
    static constexpr int kPer16B = 4;
    __device__ static __forceinline__ float ld(const float* p) { return *p; }
    __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct Elem<bf16_t> {
    static constexpr int kPer16B = 8;
    __device__ static __forceinline__ float ld(const bf16_t* p) { return bf16_to_f32(*p); }
    __device__ static __forceinline__ void st(bf16_t* p, float v) { *p = f32_to_bf16(v); }
};

// 4 consecutive elements <-> 4 floats
__device__ __forceinline__ void load4(const float* p, float v[4]) {
    float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}
__device__ __forceinline__ void load4(const bf16_t* p, float v[4]) {
    uint2 t = *reinterpret_cast<const uint2*>(p);
    v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u);
    v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
}
__device__ __forceinline__ void store4(float* p, const float v[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void store4(bf16_t* p, const float v[4]) {
    uint2 t;
    t.x = pack2_bf16(v[0], v[1]);
    t.y = pack2_bf16(v[2], v[3]);
    *reinterpret_cast<uint2*>(p) = t;
}

// fp32 atomic add through a raw buffer descriptor: fire-and-forget (no returned value, so no wait), and an offset with
// bit 31 set is out of range and dropped by the hardware -- "if (valid) atomicAdd(...)" becomes a select of the
// offset.  (A C++ atomic inside a per-element branch makes hipcc wait vmcnt(0) after every single one.)
constexpr unsigned kBufOOB = 0x80000000u;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc_uniform(const void* ptr, unsigned bytes) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(ptr);
    return __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<void*>(((uintptr_t)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(a >> 32)) << 32) |
                                (uintptr_t)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)a)),
        0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
__device__ __forceinline__ void buf_atomic_add_f32(__amdgpu_buffer_rsrc_t r, unsigned byte_off, float v) {
    __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(v, r, byte_off, 0, 0);
}

__device__ __forceinline__ float warp_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// block-wide sum; result valid in thread 0. blockDim.x multiple of 64, <= 1024.
__device__ __forceinline__ float block_sum(float v, float* smem /* >=16 floats */) {
    v = warp_sum(v);
    int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) smem[w] = v;
    __syncthreads();
    float r = 0.f;
    if (threadIdx.x == 0) {
        int nw = (blockDim.x + 63) >> 6;
        for (int i = 0; i < nw; ++i) r += smem[i];
    }
    return r;
}

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
