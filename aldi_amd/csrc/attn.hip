// Multi-head attention with decomposed relative position bias for the ViTDet trunk (gfx950, bf16 MFMA).
//
// Reference semantics: detectron2 `modeling/backbone/vit.py` Attention.forward + `utils.add_decomposed_rel_pos`, as used by
// aldi/backbone.py:21-43 (checkpointed_vit_forward) -- detectron2 itself is not vendored in the reference tree; the
// behaviour is pinned in tests against transformers' VitDetAttention (same algorithm, installed in the image).
//
//   attn[q,k] = (scale*q).k + q.Rh[qh-kh+gh-1] + q.Rw[qw-kw+gw-1];   out = softmax(attn) v
//
// The bias is decomposed, so it folds into the QK^T product: Q' = [scale*q | q.Rh[..kh..] | q.Rw[..kw..]] (64+gh+gw wide,
// padded to a multiple of 32) against K' = [k | onehot(kh) | onehot(kw)] gives attn exactly in the fp32 MFMA accumulator,
// and in the backward pass dS.K' returns the bias gradients as extra columns of dQ' without any segmented reduction.
//
// Layouts (per attention "batch" bh = b*heads + h; L tokens of a gh x gw grid; Lp = L rounded up to 64):
//   Qp, Kp  [BH][L][Dq] bf16      KpT [BH][Dq][Lp]      VT, QsT, dOT [BH][64][Lp]      (transposes are zero-padded to Lp)
//   O, dO   [nB*L][heads*64]      qkv, dqkv [nB*L][3*heads*64]   (q | k | v, head-major inside each third)
//
// MFMA convention (v_mfma_f32_16x16x32_bf16): D[i][j] = sum_k A[i][k] B[j][k]; lane l holds A row / B column l%16 with
// k = 8*(l/16)..+7, and D column l%16, rows 4*(l/16)..+3.  Scores are produced TRANSPOSED (keys = rows) so that a lane
// owns one query column: the softmax statistics are per-lane scalars, and the probabilities are already in B-operand
// position for the P.V product (the k order inside a 32-key step is permuted consistently on the V^T side).
#include "common.h"

namespace {

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));

constexpr int HD = 64;          // head dim
constexpr int TROW = 72;        // LDS row (elements) of a [.][64] bf16 tile: 144 B = 9 x 16 B (odd => conflict-free b128 reads)

__device__ __forceinline__ f32x4_t mma(u32x4_t a, u32x4_t b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
__device__ __forceinline__ u32x4_t ld16(const bf16_t* p) { return *reinterpret_cast<const u32x4_t*>(p); }
__device__ __forceinline__ u32x4_t zero16() { u32x4_t z = {0u, 0u, 0u, 0u}; return z; }
// A-operand fragment whose 8 k-slots are {c0..c0+3} and {c0+16..c0+19} of an LDS row (the permuted 32-step)
__device__ __forceinline__ u32x4_t ld_perm(const bf16_t* row, int c0) {
    u32x2_t lo = *reinterpret_cast<const u32x2_t*>(row + c0);
    u32x2_t hi = *reinterpret_cast<const u32x2_t*>(row + c0 + 16);
    u32x4_t v = {lo.x, lo.y, hi.x, hi.y};
    return v;
}
// two accumulator blocks (rows 4g..4g+3 of 16-row blocks 2s and 2s+1) -> one B-operand fragment in the permuted order
__device__ __forceinline__ u32x4_t pack_perm(f32x4_t a, f32x4_t b) {
    u32x4_t v = {pack2_bf16(a[0], a[1]), pack2_bf16(a[2], a[3]), pack2_bf16(b[0], b[1]), pack2_bf16(b[2], b[3])};
    return v;
}

// Cooperative copy of a ROWS x COLS bf16 tile (COLS % 8 == 0) into LDS in two phases: fetch() issues every 16-B global load
// of this thread back to back into registers, store() writes them to LDS.  (A rolled "load, store, next chunk" loop waits
// for each load in turn: ~16 serial L2 round trips per tile.)  Splitting the phases also lets a kernel fetch tile k+1 before
// it computes on tile k.  Rows >= valid_rows are zero-filled.
template <int ROWS, int COLS, int NT>
struct Tile {
    static constexpr int CPR = COLS / 8, TOTAL = ROWS * CPR, N = (TOTAL + NT - 1) / NT;
    u32x4_t v[N];
    __device__ __forceinline__ void fetch(const bf16_t* src, long src_row, int valid_rows) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int id = threadIdx.x + i * NT, r = id / CPR, ch = id - r * CPR;
            v[i] = (id < TOTAL && r < valid_rows) ? ld16(src + (long)r * src_row + ch * 8) : zero16();
        }
    }
    __device__ __forceinline__ void store(bf16_t* dst, int dst_row) const {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int id = threadIdx.x + i * NT, r = id / CPR, ch = id - r * CPR;
            if (id < TOTAL) *reinterpret_cast<u32x4_t*>(dst + r * dst_row + ch * 8) = v[i];
        }
    }
};
template <int ROWS, int COLS, int NT>
__device__ __forceinline__ void tile_load(bf16_t* dst, int dst_row, const bf16_t* src, long src_row, int valid_rows) {
    Tile<ROWS, COLS, NT> t;
    t.fetch(src, src_row, valid_rows);
    t.store(dst, dst_row);
}

struct AttnDev {
    const bf16_t *qkv, *Qp, *Kp, *KpT, *VT, *QsT, *dOT, *O, *dO;
    bf16_t *Ow, *dQp, *dqkv;
    float *lse; const float* lse_r; const float* delta;
    int nB, L, Lp, heads, Dq;
    int gh, gw, wofs, ntw, nt2;      // token grid; column offset of the w bias block inside Q'; 8x8 key-tile grid (tiled path)
};

// ------------------------------------------------------------------------------------------------ forward
template <int NKS, int NW>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) void attn_fwd_kernel(AttnDev a) {   // two 4-wave blocks per CU: one loads while the other computes
    constexpr int DQ = NKS * 32, KROW = DQ + 8;
    extern __shared__ __align__(16) unsigned char smem[];
    bf16_t* Ks = reinterpret_cast<bf16_t*>(smem);            // [64 keys][KROW]
    bf16_t* Vs = Ks + 64 * KROW;                              // [64 d][TROW]  (V^T tile)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
    const int bh = blockIdx.y, b = bh / a.heads, h = bh - b * a.heads;
    const int L = a.L, q0 = blockIdx.x * (NW * 32) + wave * 32;

    u32x4_t qf[2][NKS];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const int q = q0 + 16 * qb + c;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
            qf[qb][ks] = q < L ? ld16(a.Qp + ((long)bh * L + q) * DQ + ks * 32 + g * 8) : zero16();
    }
    f32x4_t o[4][2];
    float m[2], ls[2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        m[qb] = -INFINITY; ls[qb] = 0.f;
#pragma unroll
        for (int db = 0; db < 4; ++db) o[db][qb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
    const int nkt = a.Lp >> 6;
    for (int kt = 0; kt < nkt; ++kt) {
        __syncthreads();
        tile_load<64, DQ, NW * 64>(Ks, KROW, a.Kp + ((long)bh * L + kt * 64) * DQ, DQ, L - kt * 64);
        tile_load<64, 64, NW * 64>(Vs, TROW, a.VT + (long)bh * HD * a.Lp + kt * 64, a.Lp, 64);
        __syncthreads();
        f32x4_t st[4][2];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) st[kb][qb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                const u32x4_t kf = ld16(Ks + (16 * kb + c) * KROW + ks * 32 + g * 8);
#pragma unroll
                for (int qb = 0; qb < 2; ++qb) st[kb][qb] = mma(kf, qf[qb][ks], st[kb][qb]);
            }
        if (kt == nkt - 1) {
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (kt * 64 + 16 * kb + 4 * g + i >= L) { st[kb][0][i] = -INFINITY; st[kb][1][i] = -INFINITY; }
        }
        u32x4_t pf[2][2];
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            float mx = -INFINITY;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int i = 0; i < 4; ++i) mx = fmaxf(mx, st[kb][qb][i]);
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float mn = fmaxf(m[qb], mx);
            const float alpha = __expf(m[qb] - mn);      // first tile: exp(-inf) = 0
            m[qb] = mn;
            float s = 0.f;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float p = __expf(st[kb][qb][i] - mn);
                    st[kb][qb][i] = p;
                    s += p;
                }
            ls[qb] = ls[qb] * alpha + s;
#pragma unroll
            for (int db = 0; db < 4; ++db)
#pragma unroll
                for (int i = 0; i < 4; ++i) o[db][qb][i] *= alpha;
            pf[qb][0] = pack_perm(st[0][qb], st[1][qb]);
            pf[qb][1] = pack_perm(st[2][qb], st[3][qb]);
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int db = 0; db < 4; ++db) {
                const u32x4_t vf = ld_perm(Vs + (16 * db + c) * TROW, 32 * s2 + 4 * g);
#pragma unroll
                for (int qb = 0; qb < 2; ++qb) o[db][qb] = mma(vf, pf[qb][s2], o[db][qb]);
            }
    }
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        float l = ls[qb];
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        const int q = q0 + 16 * qb + c;
        if (q < L) {
            const float inv = 1.f / l;
            bf16_t* orow = a.Ow + ((long)b * L + q) * (a.heads * HD) + h * HD;
#pragma unroll
            for (int db = 0; db < 4; ++db) {
                uint2 t;
                t.x = pack2_bf16(o[db][qb][0] * inv, o[db][qb][1] * inv);
                t.y = pack2_bf16(o[db][qb][2] * inv, o[db][qb][3] * inv);
                *reinterpret_cast<uint2*>(orow + 16 * db + 4 * g) = t;
            }
            if (g == 0) a.lse[(long)bh * L + q] = m[qb] + __logf(l);
        }
    }
}

// ------------------------------------------------------------------------------------------------ backward: dQ'
// one block owns NW*32 queries and sweeps the key tiles: dQ'^T[d'][q] += K'^T[d'][keys] . dS[q][keys]
template <int NKS, int NW>
__global__ __launch_bounds__(NW * 64) void attn_bwd_dq_kernel(AttnDev a) {
    constexpr int DQ = NKS * 32, KROW = DQ + 8;
    extern __shared__ __align__(16) unsigned char smem[];
    bf16_t* Ks = reinterpret_cast<bf16_t*>(smem);            // [64 keys][KROW]
    bf16_t* KTs = Ks + 64 * KROW;                             // [DQ][TROW]      (K'^T tile)
    bf16_t* Vs = KTs + DQ * TROW;                             // [64 keys][TROW] (V rows)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
    const int bh = blockIdx.y, b = bh / a.heads, h = bh - b * a.heads;
    const int L = a.L, q0 = blockIdx.x * (NW * 32) + wave * 32, ld3 = 3 * a.heads * HD, ld1 = a.heads * HD;

    u32x4_t qf[2][NKS], dof[2][2];
    float lse[2], dl[2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const int q = q0 + 16 * qb + c;
        const bool ok = q < L;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) qf[qb][ks] = ok ? ld16(a.Qp + ((long)bh * L + q) * DQ + ks * 32 + g * 8) : zero16();
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) dof[qb][ks] = ok ? ld16(a.dO + ((long)b * L + q) * ld1 + h * HD + ks * 32 + g * 8) : zero16();
        lse[qb] = ok ? a.lse_r[(long)bh * L + q] : INFINITY;
        dl[qb] = ok ? a.delta[(long)bh * L + q] : 0.f;
    }
    f32x4_t dq[2 * NKS][2];
#pragma unroll
    for (int i = 0; i < 2 * NKS; ++i) { dq[i][0] = f32x4_t{0.f, 0.f, 0.f, 0.f}; dq[i][1] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }

    const int nkt = a.Lp >> 6;
    Tile<64, DQ, NW * 64> tK;
    Tile<DQ, 64, NW * 64> tKT;
    Tile<64, 64, NW * 64> tV;
    auto fetch = [&](int kt) {
        tK.fetch(a.Kp + ((long)bh * L + kt * 64) * DQ, DQ, L - kt * 64);
        tKT.fetch(a.KpT + (long)bh * DQ * a.Lp + kt * 64, a.Lp, DQ);
        tV.fetch(a.qkv + ((long)b * L + kt * 64) * ld3 + 2 * ld1 + h * HD, ld3, L - kt * 64);
    };
    fetch(0);
    for (int kt = 0; kt < nkt; ++kt) {
        __syncthreads();
        tK.store(Ks, KROW);
        tKT.store(KTs, TROW);
        tV.store(Vs, TROW);
        __syncthreads();
        if (kt + 1 < nkt) fetch(kt + 1);          // in flight while this tile is computed
        f32x4_t st[4][2], dp[4][2];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) { st[kb][qb] = f32x4_t{0.f, 0.f, 0.f, 0.f}; dp[kb][qb] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                const u32x4_t kf = ld16(Ks + (16 * kb + c) * KROW + ks * 32 + g * 8);
#pragma unroll
                for (int qb = 0; qb < 2; ++qb) st[kb][qb] = mma(kf, qf[qb][ks], st[kb][qb]);
            }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                const u32x4_t vf = ld16(Vs + (16 * kb + c) * TROW + ks * 32 + g * 8);
#pragma unroll
                for (int qb = 0; qb < 2; ++qb) dp[kb][qb] = mma(vf, dof[qb][ks], dp[kb][qb]);
            }
        u32x4_t dsf[2][2];
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const bool valid = kt * 64 + 16 * kb + 4 * g + i < L;
                    const float p = valid ? __expf(st[kb][qb][i] - lse[qb]) : 0.f;
                    st[kb][qb][i] = p * (dp[kb][qb][i] - dl[qb]);
                }
            dsf[qb][0] = pack_perm(st[0][qb], st[1][qb]);
            dsf[qb][1] = pack_perm(st[2][qb], st[3][qb]);
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int db = 0; db < 2 * NKS; ++db) {
                const u32x4_t kt_f = ld_perm(KTs + (16 * db + c) * TROW, 32 * s2 + 4 * g);
#pragma unroll
                for (int qb = 0; qb < 2; ++qb) dq[db][qb] = mma(kt_f, dsf[qb][s2], dq[db][qb]);
            }
    }
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const int q = q0 + 16 * qb + c;
        if (q < L) {
            bf16_t* row = a.dQp + ((long)bh * L + q) * DQ;
#pragma unroll
            for (int db = 0; db < 2 * NKS; ++db) {
                uint2 t;
                t.x = pack2_bf16(dq[db][qb][0], dq[db][qb][1]);
                t.y = pack2_bf16(dq[db][qb][2], dq[db][qb][3]);
                *reinterpret_cast<uint2*>(row + 16 * db + 4 * g) = t;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ backward: dK, dV
// one block owns 128 keys (32 per wave) and sweeps the query tiles:
//   dV^T[d][key] += dO^T[d][q] . P[q][key]        dK^T[d][key] += (scale*q)^T[d][q] . dS[q][key]
template <int NKS>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(AttnDev a) {
    constexpr int DQ = NKS * 32, KROW = DQ + 8;
    extern __shared__ __align__(16) unsigned char smem[];
    bf16_t* Qs = reinterpret_cast<bf16_t*>(smem);            // [64 q][KROW]
    bf16_t* dOs = Qs + 64 * KROW;                             // [64 q][TROW]
    bf16_t* QTs = dOs + 64 * TROW;                            // [64 d][TROW]  ((scale*q)^T tile)
    bf16_t* dOTs = QTs + 64 * TROW;                           // [64 d][TROW]
    float* lse_s = reinterpret_cast<float*>(dOTs + 64 * TROW);   // [64]
    float* dl_s = lse_s + 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
    const int bh = blockIdx.y, b = bh / a.heads, h = bh - b * a.heads;
    const int L = a.L, k0 = blockIdx.x * 128 + wave * 32, ld3 = 3 * a.heads * HD, ld1 = a.heads * HD;

    u32x4_t kf[2][NKS], vf[2][2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        const int key = k0 + 16 * kb + c;
        const bool ok = key < L;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) kf[kb][ks] = ok ? ld16(a.Kp + ((long)bh * L + key) * DQ + ks * 32 + g * 8) : zero16();
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            vf[kb][ks] = ok ? ld16(a.qkv + ((long)b * L + key) * ld3 + 2 * ld1 + h * HD + ks * 32 + g * 8) : zero16();
    }
    f32x4_t dv[4][2], dk[4][2];
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) { dv[db][kb] = f32x4_t{0.f, 0.f, 0.f, 0.f}; dk[db][kb] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }

    const int nqt = a.Lp >> 6;
    Tile<64, DQ, 256> tQ;
    Tile<64, 64, 256> tdO, tQT, tdOT;
    float lse_n = 0.f, dl_n = 0.f;
    auto fetch = [&](int qt) {
        tQ.fetch(a.Qp + ((long)bh * L + qt * 64) * DQ, DQ, L - qt * 64);
        tdO.fetch(a.dO + ((long)b * L + qt * 64) * ld1 + h * HD, ld1, L - qt * 64);
        tQT.fetch(a.QsT + (long)bh * HD * a.Lp + qt * 64, a.Lp, 64);
        tdOT.fetch(a.dOT + (long)bh * HD * a.Lp + qt * 64, a.Lp, 64);
        if (threadIdx.x < 64) {
            const int q = qt * 64 + threadIdx.x;
            lse_n = q < L ? a.lse_r[(long)bh * L + q] : INFINITY;
            dl_n = q < L ? a.delta[(long)bh * L + q] : 0.f;
        }
    };
    fetch(0);
    for (int qt = 0; qt < nqt; ++qt) {
        __syncthreads();
        tQ.store(Qs, KROW);
        tdO.store(dOs, TROW);
        tQT.store(QTs, TROW);
        tdOT.store(dOTs, TROW);
        if (threadIdx.x < 64) { lse_s[threadIdx.x] = lse_n; dl_s[threadIdx.x] = dl_n; }
        __syncthreads();
        if (qt + 1 < nqt) fetch(qt + 1);          // in flight while this tile is computed
        f32x4_t s[4][2], dp[4][2];
#pragma unroll
        for (int qb = 0; qb < 4; ++qb)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) { s[qb][kb] = f32x4_t{0.f, 0.f, 0.f, 0.f}; dp[qb][kb] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
            for (int qb = 0; qb < 4; ++qb) {
                const u32x4_t qfr = ld16(Qs + (16 * qb + c) * KROW + ks * 32 + g * 8);
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) s[qb][kb] = mma(qfr, kf[kb][ks], s[qb][kb]);
            }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int qb = 0; qb < 4; ++qb) {
                const u32x4_t dofr = ld16(dOs + (16 * qb + c) * TROW + ks * 32 + g * 8);
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) dp[qb][kb] = mma(dofr, vf[kb][ks], dp[qb][kb]);
            }
        u32x4_t pf[2][2], dsf[2][2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int qb = 0; qb < 4; ++qb) {
                const f32x4_t l4 = *reinterpret_cast<const f32x4_t*>(lse_s + 16 * qb + 4 * g);
                const f32x4_t d4 = *reinterpret_cast<const f32x4_t*>(dl_s + 16 * qb + 4 * g);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float p = __expf(s[qb][kb][i] - l4[i]);     // rows past L: lse = +inf => 0
                    s[qb][kb][i] = p;
                    dp[qb][kb][i] = p * (dp[qb][kb][i] - d4[i]);
                }
            }
            pf[kb][0] = pack_perm(s[0][kb], s[1][kb]);
            pf[kb][1] = pack_perm(s[2][kb], s[3][kb]);
            dsf[kb][0] = pack_perm(dp[0][kb], dp[1][kb]);
            dsf[kb][1] = pack_perm(dp[2][kb], dp[3][kb]);
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int db = 0; db < 4; ++db) {
                const u32x4_t dot_f = ld_perm(dOTs + (16 * db + c) * TROW, 32 * s2 + 4 * g);
                const u32x4_t qt_f = ld_perm(QTs + (16 * db + c) * TROW, 32 * s2 + 4 * g);
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    dv[db][kb] = mma(dot_f, pf[kb][s2], dv[db][kb]);
                    dk[db][kb] = mma(qt_f, dsf[kb][s2], dk[db][kb]);
                }
            }
    }
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        const int key = k0 + 16 * kb + c;
        if (key < L) {
            bf16_t* row = a.dqkv + ((long)b * L + key) * ld3 + h * HD;
#pragma unroll
            for (int db = 0; db < 4; ++db) {
                uint2 t;
                t.x = pack2_bf16(dk[db][kb][0], dk[db][kb][1]);
                t.y = pack2_bf16(dk[db][kb][2], dk[db][kb][3]);
                *reinterpret_cast<uint2*>(row + ld1 + 16 * db + 4 * g) = t;
                t.x = pack2_bf16(dv[db][kb][0], dv[db][kb][1]);
                t.y = pack2_bf16(dv[db][kb][2], dv[db][kb][3]);
                *reinterpret_cast<uint2*>(row + 2 * ld1 + 16 * db + 4 * g) = t;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ operand preparation
// One block per (bh, 64-token tile): builds Q' / K' rows and the transposed copies the MFMA kernels stream.
struct PrepDev {
    const bf16_t* qkv; const float *rel_h, *rel_w;
    bf16_t *Qp, *Kp, *KpT, *VT, *QsT;
    int nB, L, Lp, heads, Dq, gh, gw, wofs, tiled, tiles_per_block; float scale;
};
__global__ __launch_bounds__(256) void attn_prep_kernel(PrepDev a) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int gh = a.gh, gw = a.gw, nh = 2 * gh - 1, nrows = a.rel_h ? nh + 2 * gw - 1 : 0, nrb = (nrows + 15) >> 4;
    bf16_t* qb = reinterpret_cast<bf16_t*>(smem);      // [64 t][TROW]   q (unscaled)
    bf16_t* tab = qb + 64 * TROW;                       // [nrb*16][TROW] rel_h rows, then rel_w rows (bf16; zero rows pad the last block)
    bf16_t* kv = tab + nrb * 16 * TROW;                 // [2][64][64] raw k, v
    bf16_t* qp = kv + 2 * 64 * 64;                      // [64][Dq] finished Q' rows (for the transposes)
    int* ph = reinterpret_cast<int*>(qp + 64 * a.Dq);   // [64] qh(t) + gh - 1
    int* pw = ph + 64;                                  // [64] qw(t) + gw - 1
    const int bh = blockIdx.y, b = bh / a.heads, h = bh - b * a.heads;
    const int L = a.L, Dq = a.Dq, ld3 = 3 * a.heads * HD, ld1 = a.heads * HD;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
    for (int id = threadIdx.x; id < nrb * 16 * 16; id += 256) {           // tables fp32 -> bf16 (autocast would feed the einsum fp16)
        const int r = id >> 4, c4 = (id & 15) * 4;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (r < nrows) load4((r < nh ? a.rel_h + r * 64 : a.rel_w + (r - nh) * 64) + c4, v);
        store4(tab + r * TROW + c4, v);
    }
    // several 64-token tiles of one (image, head) per block: the tables are converted once
    for (int ti = 0; ti < a.tiles_per_block; ++ti) {
    const int tile = blockIdx.x * a.tiles_per_block + ti;
    if (tile * 64 >= a.Lp) break;
    const int t0 = tile * 64;
    __syncthreads();
    for (int id = threadIdx.x; id < 64 * 8; id += 256) {                 // q, k, v rows of this tile, 16 B at a time
        const int t = id >> 3, ch = id & 7;
        const bool ok = t0 + t < L;
        const bf16_t* row = a.qkv + ((long)b * L + t0 + t) * ld3 + h * HD + ch * 8;
        *reinterpret_cast<u32x4_t*>(qb + t * TROW + ch * 8) = ok ? ld16(row) : zero16();
        *reinterpret_cast<u32x4_t*>(kv + t * 64 + ch * 8) = ok ? ld16(row + ld1) : zero16();
        *reinterpret_cast<u32x4_t*>(kv + 64 * 64 + t * 64 + ch * 8) = ok ? ld16(row + 2 * ld1) : zero16();
    }
    if (threadIdx.x < 64) {
        const int tok = t0 + threadIdx.x, qh = tok / gw;
        ph[threadIdx.x] = qh + gh - 1;
        pw[threadIdx.x] = tok - qh * gw + gw - 1;
    }
    for (int id = threadIdx.x; id < 64 * (Dq >> 1); id += 256) reinterpret_cast<uint32_t*>(qp)[id] = 0u;
    __syncthreads();
    // scaled q (scale = 2^-3 for head dim 64: exact in bf16)
    for (int id = threadIdx.x; id < 64 * 64; id += 256) {
        const int t = id >> 6, d = id & 63;
        qp[t * Dq + d] = f32_to_bf16(bf16_to_f32(qb[t * TROW + d]) * a.scale);
    }
    // bias columns: E[r][t] = table_row(r) . q[t] on the matrix cores (rows = table rows, columns = tokens), then entry (r, t) lands in
    // column kh = qh(t) + gh - 1 - r (resp. kw) of Q'[t] when that is a valid key coordinate.  Wave w owns row blocks w, w+4, ...
    if (nrows) {
        u32x4_t qf[4][2];
#pragma unroll
        for (int tb = 0; tb < 4; ++tb)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) qf[tb][ks] = ld16(qb + (16 * tb + c) * TROW + 32 * ks + 8 * g);
        for (int rb = wave; rb < nrb; rb += 4) {
            const u32x4_t a0 = ld16(tab + (16 * rb + c) * TROW + 8 * g), a1 = ld16(tab + (16 * rb + c) * TROW + 32 + 8 * g);
#pragma unroll
            for (int tb = 0; tb < 4; ++tb) {
                f32x4_t e = f32x4_t{0.f, 0.f, 0.f, 0.f};
                e = mma(a0, qf[tb][0], e);
                e = mma(a1, qf[tb][1], e);
                const int t = 16 * tb + c, p_h = ph[t], p_w = pw[t];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r = 16 * rb + 4 * g + i;
                    if (r < nh) {
                        const int kh = p_h - r;
                        if (kh >= 0 && kh < gh) qp[t * Dq + 64 + kh] = f32_to_bf16(e[i]);
                    } else if (r < nrows) {
                        const int kw = p_w - (r - nh);
                        if (kw >= 0 && kw < gw) qp[t * Dq + 64 + a.wofs + kw] = f32_to_bf16(e[i]);
                    }
                }
            }
        }
    }
    __syncthreads();
    // row-major outputs: Q' and K'
    const int cpr = Dq >> 3;
    for (int id = threadIdx.x; id < 64 * cpr; id += 256) {
        const int t = id / cpr, ch = id - t * cpr, tok = t0 + t;
        if (tok >= L) continue;
        *reinterpret_cast<u32x4_t*>(a.Qp + ((long)bh * L + tok) * Dq + ch * 8) = *reinterpret_cast<const u32x4_t*>(qp + t * Dq + ch * 8);
        if (a.tiled) continue;            // the tiled kernels read k / v rows from qkv and synthesise the one-hot columns
        u32x4_t kvv;
        if (ch < 8) kvv = *reinterpret_cast<const u32x4_t*>(kv + t * 64 + ch * 8);
        else {
            const int kh = tok / gw, kw = tok - kh * gw;
            const int hot_h = a.rel_h ? 64 + kh : -1, hot_w = a.rel_h ? 64 + a.wofs + kw : -1, e0 = ch * 8;
            unsigned wv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned lo = (e0 + 2 * j == hot_h || e0 + 2 * j == hot_w) ? 0x3f80u : 0u;
                const unsigned hi = (e0 + 2 * j + 1 == hot_h || e0 + 2 * j + 1 == hot_w) ? 0x3f80u : 0u;
                wv[j] = lo | (hi << 16);
            }
            kvv = u32x4_t{wv[0], wv[1], wv[2], wv[3]};
        }
        *reinterpret_cast<u32x4_t*>(a.Kp + ((long)bh * L + tok) * Dq + ch * 8) = kvv;
    }
    // transposed outputs (8 tokens = 16 B per store): K'^T [Dq][Lp], V^T [64][Lp], (scale q)^T [64][Lp]
    for (int id = threadIdx.x; id < (Dq + 128) * 8; id += 256) {
        const int r = id >> 3, ch = id & 7;
        unsigned short v[8];
        bf16_t* dst;
        if (a.tiled && r < Dq + 64) continue;      // K'^T / V^T are produced tile-major by attn_prep2d_kernel
        if (r < Dq) {
            dst = a.KpT + ((long)bh * Dq + r) * a.Lp + t0 + ch * 8;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int t = ch * 8 + j, tok = t0 + t;
                if (r < 64) v[j] = kv[t * 64 + r];
                else {
                    const int kh = tok / gw, kw = tok - kh * gw;
                    v[j] = (a.rel_h && tok < L && (r == 64 + kh || r == 64 + a.wofs + kw)) ? 0x3f80 : 0;
                }
            }
        } else if (r < Dq + 64) {
            const int d = r - Dq;
            dst = a.VT + ((long)bh * HD + d) * a.Lp + t0 + ch * 8;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = kv[64 * 64 + (ch * 8 + j) * 64 + d];
        } else {
            const int d = r - Dq - 64;
            dst = a.QsT + ((long)bh * HD + d) * a.Lp + t0 + ch * 8;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = qp[(ch * 8 + j) * Dq + d];
        }
        u32x4_t o = {v[0] | ((unsigned)v[1] << 16), v[2] | ((unsigned)v[3] << 16), v[4] | ((unsigned)v[5] << 16), v[6] | ((unsigned)v[7] << 16)};
        *reinterpret_cast<u32x4_t*>(dst) = o;
    }
    }
}

// delta[q] = sum_d O.dO and dO^T, one block per (bh, 64-token tile)
struct BprepDev { const bf16_t *O, *dO; bf16_t* dOT; float* delta; int nB, L, Lp, heads; };
__global__ __launch_bounds__(256) void attn_bwd_prep_kernel(BprepDev a) {
    __shared__ bf16_t dos[64 * 66];
    const int bh = blockIdx.y, b = bh / a.heads, h = bh - b * a.heads, t0 = blockIdx.x * 64, ld1 = a.heads * HD;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int t = wave; t < 64; t += 4) {
        const int tok = t0 + t;
        float p = 0.f;
        bf16_t dv = 0;
        if (tok < a.L) {
            const long off = ((long)b * a.L + tok) * ld1 + h * HD + lane;
            dv = a.dO[off];
            p = bf16_to_f32(dv) * bf16_to_f32(a.O[off]);
        }
        dos[t * 66 + lane] = dv;
        p = warp_sum(p);
        if (lane == 0 && tok < a.L) a.delta[(long)bh * a.L + tok] = p;
    }
    __syncthreads();
    for (int id = threadIdx.x; id < 64 * 8; id += 256) {
        const int d = id >> 3, ch = id & 7;
        unsigned short v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = dos[(ch * 8 + j) * 66 + d];
        u32x4_t o = {v[0] | ((unsigned)v[1] << 16), v[2] | ((unsigned)v[3] << 16), v[4] | ((unsigned)v[5] << 16), v[6] | ((unsigned)v[7] << 16)};
        *reinterpret_cast<u32x4_t*>(a.dOT + ((long)bh * HD + d) * a.Lp + t0 + ch * 8) = o;
    }
}

// dQ' -> dq (the q slot of dqkv):  dq[t] = scale * dQ'[t][:64] + sum_e dQ'[t][64+e] * table_row(t, e).  One block per (bh, tile).
struct RbwdDev {
    const bf16_t *qkv, *dQp; const float *rel_h, *rel_w;
    bf16_t* dqkv; float *drel_h, *drel_w;
    int nB, L, heads, Dq, gh, gw, wofs, tiles_per_block; float scale;
};

__global__ __launch_bounds__(256) void attn_rel_dq_kernel(RbwdDev a) {
    // dq^T[c][t] = scale * dQ'[t][c] + sum_r R^T[c][r] * E[t][r], E[t][r] = dQ'[t][bias column qh(t) + gh - 1 - r] (the same skewed
    // operand as the table gradient): A = transposed tables from LDS, B = E gathered from the bias columns, 2-byte LDS reads.
    extern __shared__ __align__(16) unsigned char smem[];
    const int gh = a.gh, gw = a.gw, nh = 2 * gh - 1, nrows = a.rel_h ? nh + 2 * gw - 1 : 0, nks = (nrows + 31) >> 5;
    const int RT = nks * 32 + 8, DRB = a.Dq - 64 + 8, zero_col = a.Dq - 64;
    bf16_t* rt = reinterpret_cast<bf16_t*>(smem);       // [64 c][RT]  tables transposed (bf16), zero beyond nrows
    bf16_t* drs = rt + 64 * RT;                          // [64 t][DRB] bias columns of dQ' + a zero slot
    int* ph = reinterpret_cast<int*>(drs + 64 * DRB);
    int* pw = ph + 64;
    const int bh = blockIdx.y, b = bh / a.heads, h = bh - b * a.heads, t0 = blockIdx.x * 64;
    const int L = a.L, Dq = a.Dq, ld3 = 3 * a.heads * HD;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
    if (nrows) {
        for (int id = threadIdx.x; id < 64 * (RT >> 1); id += 256) reinterpret_cast<uint32_t*>(rt)[id] = 0u;
        __syncthreads();
        for (int id = threadIdx.x; id < nrows * 64; id += 256) {
            const int r = id >> 6, cc = id & 63;
            rt[cc * RT + r] = f32_to_bf16(r < nh ? a.rel_h[r * 64 + cc] : a.rel_w[(r - nh) * 64 + cc]);
        }
        const int cpr = DRB >> 3;
        for (int id = threadIdx.x; id < 64 * cpr; id += 256) {
            const int t = id / cpr, ch = id - t * cpr;
            u32x4_t v = (t0 + t < L && ch < cpr - 1) ? ld16(a.dQp + ((long)bh * L + t0 + t) * Dq + 64 + ch * 8) : zero16();
            *reinterpret_cast<u32x4_t*>(drs + t * DRB + ch * 8) = v;
        }
        if (threadIdx.x < 64) {
            const int tok = t0 + threadIdx.x, qh = tok / gw;
            ph[threadIdx.x] = qh + gh - 1;
            pw[threadIdx.x] = tok - qh * gw + gw - 1;
        }
    }
    __syncthreads();
    const int t = 16 * wave + c, tok = t0 + t;           // wave w owns tokens 16w .. 16w+15 (MFMA columns)
    f32x4_t acc[4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) acc[cb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if (nrows) {
        const int p_h = ph[t], p_w = pw[t];
        const bf16_t* drow = drs + t * DRB;
        for (int ks = 0; ks < nks; ++ks) {
            unsigned short e[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int r = 32 * ks + 8 * g + j;
                const bool is_h = r < nh;
                const int k = is_h ? p_h - r : p_w - (r - nh);
                const bool ok = r < nrows && k >= 0 && k < (is_h ? gh : gw);
                e[j] = drow[ok ? (is_h ? k : a.wofs + k) : zero_col];
            }
            const u32x4_t ef = {e[0] | ((unsigned)e[1] << 16), e[2] | ((unsigned)e[3] << 16), e[4] | ((unsigned)e[5] << 16),
                                e[6] | ((unsigned)e[7] << 16)};
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) acc[cb] = mma(ld16(rt + (16 * cb + c) * RT + 32 * ks + 8 * g), ef, acc[cb]);
        }
    }
    if (tok < L) {
        const bf16_t* src = a.dQp + ((long)bh * L + tok) * Dq;
        bf16_t* dst = a.dqkv + ((long)b * L + tok) * ld3 + h * HD;
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            float v[4];
            load4(src + 16 * cb + 4 * g, v);
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = a.scale * v[i] + acc[cb][i];
            store4(dst + 16 * cb + 4 * g, v);
        }
    }
}

// table gradients on the matrix cores.  d(rel_w)[r] = sum_t E[t][r] * q[t] with the skewed operand
// E[t][r] = dQ'[t][64 + gh + (qw(t) + gw - 1 - r)] (zero outside the band), likewise d(rel_h): one [rows x tokens].[tokens x 64]
// product per (image, head).  A fragments are gathered from an LDS copy of the bias columns of dQ' (8 two-byte reads per
// fragment); B fragments come from the (scale*q)^T tile the attention backward already streams, and the 1/scale is applied to
// the accumulator.  One block per (bh, group of 64-token tiles); wave w owns row blocks w, w+4, ...
constexpr int DT_MAXRB = 6;      // (2gh-1 + 2gw-1) <= 4 * 6 * 16 = 384 rows

__global__ __launch_bounds__(256) void attn_rel_dtab_kernel(RbwdDev a, const bf16_t* __restrict__ QsT, int Lp) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int gh = a.gh, gw = a.gw, nrows = (2 * gh - 1) + (2 * gw - 1), DRB = a.Dq - 64 + 8;   // bias columns + a zero slot (16 B)
    bf16_t* drs = reinterpret_cast<bf16_t*>(smem);                   // [64 t][DRB]
    bf16_t* QTs = drs + 64 * DRB;                                    // [64 c][TROW]
    int* ph = reinterpret_cast<int*>(QTs + 64 * TROW);               // [64] qh(t) + gh - 1
    int* pw = ph + 64;                                               // [64] qw(t) + gw - 1
    const int bh = blockIdx.y, L = a.L, Dq = a.Dq;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
    const int nrb = (nrows + 15) >> 4, zero_col = Dq - 64;
    f32x4_t acc[DT_MAXRB][4];
#pragma unroll
    for (int i = 0; i < DT_MAXRB; ++i)
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) acc[i][cb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const int ntiles = (L + 63) >> 6;
    for (int ti = 0; ti < a.tiles_per_block; ++ti) {
        const int tile = blockIdx.x * a.tiles_per_block + ti;
        if (tile >= ntiles) break;
        const int t0 = tile * 64;
        __syncthreads();
        {   // bias columns of dQ' (16-B chunks; rows past L and the extra slot are zeros)
            const int cpr = DRB >> 3;
            for (int id = threadIdx.x; id < 64 * cpr; id += 256) {
                const int t = id / cpr, ch = id - t * cpr;
                u32x4_t v = (t0 + t < L && ch < cpr - 1) ? ld16(a.dQp + ((long)bh * L + t0 + t) * Dq + 64 + ch * 8) : zero16();
                *reinterpret_cast<u32x4_t*>(drs + t * DRB + ch * 8) = v;
            }
        }
        tile_load<64, 64, 256>(QTs, TROW, QsT + (long)bh * HD * Lp + t0, Lp, 64);
        if (threadIdx.x < 64) {
            const int tok = t0 + threadIdx.x, qh = tok / gw;
            ph[threadIdx.x] = qh + gh - 1;
            pw[threadIdx.x] = tok - qh * gw + gw - 1;
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            int p_h[8], p_w[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { p_h[j] = ph[32 * ks + 8 * g + j]; p_w[j] = pw[32 * ks + 8 * g + j]; }
            u32x4_t qf[4];
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) qf[cb] = ld16(QTs + (16 * cb + c) * TROW + 32 * ks + 8 * g);
#pragma unroll
            for (int i = 0; i < DT_MAXRB; ++i) {
                const int rb = wave + 4 * i;
                if (rb >= nrb) break;
                const int r = 16 * rb + c;
                const bool is_h = r < 2 * gh - 1;
                const int rr = is_h ? r : r - (2 * gh - 1), kmax = (r < nrows) ? (is_h ? gh : gw) : 0, cbase = is_h ? 0 : a.wofs;
                unsigned short e[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int k = (is_h ? p_h[j] : p_w[j]) - rr;
                    const int col = (k >= 0 && k < kmax) ? cbase + k : zero_col;
                    e[j] = drs[(32 * ks + 8 * g + j) * DRB + col];
                }
                const u32x4_t af = {e[0] | ((unsigned)e[1] << 16), e[2] | ((unsigned)e[3] << 16), e[4] | ((unsigned)e[5] << 16),
                                    e[6] | ((unsigned)e[7] << 16)};
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) acc[i][cb] = mma(af, qf[cb], acc[i][cb]);
            }
        }
    }
    const float inv_scale = 1.f / a.scale;
#pragma unroll
    for (int i = 0; i < DT_MAXRB; ++i) {
        const int rb = wave + 4 * i;
        if (rb >= nrb) break;
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int r = 16 * rb + 4 * g + k, col = 16 * cb + c;
                if (r < 2 * gh - 1) atomicAdd(a.drel_h + r * 64 + col, acc[i][cb][k] * inv_scale);
                else if (r < nrows) atomicAdd(a.drel_w + (r - (2 * gh - 1)) * 64 + col, acc[i][cb][k] * inv_scale);
            }
    }
}


// ================================================================================================ tiled path (large grids)
// Key tiles are 8x8 BLOCKS of the token grid instead of 64 consecutive tokens.  Inside one tile the key coordinates take 8
// values each, so the one-hot part of K' needs 16 columns (8 for kh - 8th, 8 for kw - 8tw) instead of gh + gw: the score product
// is 2 k-steps of q.k plus ONE k-step whose A operand is a constant one-hot fragment and whose B operand is the 2 x 8 bias
// columns of Q' that belong to this tile.  On the 50 x 84 grid that is 3 k-steps instead of 7, and dQ' shrinks from 14
// accumulator blocks to 4 (+ one 16-column bias block that is accumulated in LDS at its tile's column offset).  Queries stay
// in linear order.  K / V rows are gathered from qkv through the tile's token map; k^T / V^T tiles come tile-major from
// attn_prep2d_kernel ([BH][tile][64 d][64 slots]).  Slots outside the grid (ragged last row / column of tiles) are masked.
__device__ __forceinline__ int slot_token(int th, int tw, int s, int gh, int gw) {
    const int r = 8 * th + (s >> 3), c = 8 * tw + (s & 7);
    return (r < gh && c < gw) ? r * gw + c : -1;
}
__device__ __forceinline__ u32x4_t onehot8(int j) {        // 8 bf16, 1.0 at position j (none when j is outside 0..7)
    u32x4_t v = {0u, 0u, 0u, 0u};
    const unsigned one = (j & 1) ? 0x3f800000u : 0x00003f80u;
    if (j >= 0 && j < 8) {
        if ((j >> 1) == 0) v.x = one;
        else if ((j >> 1) == 1) v.y = one;
        else if ((j >> 1) == 2) v.z = one;
        else v.w = one;
    }
    return v;
}
// A-operand of the bias k-step of S^T (rows = key slots 16kb + c): k-chunk 0 = onehot(row of the slot), 1 = onehot(column), 2, 3 = 0
__device__ __forceinline__ u32x4_t onehot_keys(int slot, int g) { return g == 0 ? onehot8(slot >> 3) : (g == 1 ? onehot8(slot & 7) : zero16()); }

// rows of a key tile gathered through the token map (row stride `ld` elements), 64 columns
template <int NT>
struct GatherTile {
    static constexpr int N = (64 * 8 + NT - 1) / NT;
    u32x4_t v[N];
    __device__ __forceinline__ void fetch(const bf16_t* base, long ld, int th, int tw, int gh, int gw) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int id = threadIdx.x + i * NT, s = id >> 3, ch = id & 7;
            const int tok = id < 512 ? slot_token(th, tw, s, gh, gw) : -1;
            v[i] = tok >= 0 ? ld16(base + (long)tok * ld + ch * 8) : zero16();
        }
    }
    __device__ __forceinline__ void store(bf16_t* dst, int dst_row) const {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int id = threadIdx.x + i * NT, s = id >> 3, ch = id & 7;
            if (id < 512) *reinterpret_cast<u32x4_t*>(dst + s * dst_row + ch * 8) = v[i];
        }
    }
};

// k^T / V^T of one key tile: [64 d][64 slots], tile-major
__global__ __launch_bounds__(256) void attn_prep2d_kernel(AttnDev a, bf16_t* __restrict__ KT2, bf16_t* __restrict__ VT2) {
    __shared__ bf16_t kv[2 * 64 * 66];
    const int bh = blockIdx.y, b = bh / a.heads, h = bh - b * a.heads, kt = blockIdx.x, th = kt / a.ntw, tw = kt - th * a.ntw;
    const int ld3 = 3 * a.heads * HD, ld1 = a.heads * HD;
    for (int id = threadIdx.x; id < 64 * 64; id += 256) {
        const int s = id >> 6, d = id & 63, tok = slot_token(th, tw, s, a.gh, a.gw);
        const bf16_t* row = a.qkv + ((long)b * a.L + tok) * ld3 + h * HD + d;
        kv[s * 66 + d] = tok >= 0 ? row[ld1] : (bf16_t)0;
        kv[64 * 66 + s * 66 + d] = tok >= 0 ? row[2 * ld1] : (bf16_t)0;
    }
    __syncthreads();
    for (int id = threadIdx.x; id < 2 * 64 * 8; id += 256) {
        const int which = id >> 9, d = (id >> 3) & 63, ch = id & 7;
        unsigned short v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = kv[which * 64 * 66 + (ch * 8 + j) * 66 + d];
        u32x4_t o = {v[0] | ((unsigned)v[1] << 16), v[2] | ((unsigned)v[3] << 16), v[4] | ((unsigned)v[5] << 16), v[6] | ((unsigned)v[7] << 16)};
        *reinterpret_cast<u32x4_t*>((which ? VT2 : KT2) + ((long)bh * a.nt2 + kt) * 4096 + d * 64 + ch * 8) = o;
    }
}

// scores of one key tile, transposed (rows = slots): 2 k-steps of k.q plus the one-hot bias step; invalid slots -> -inf
__device__ __forceinline__ void scores2d(f32x4_t st[4][2], const bf16_t* Ks, const u32x4_t qf[2][2], const u32x4_t bq[2], const u32x4_t oh[4],
                                         int c, int g) {
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) st[kb][qb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            const u32x4_t kf = ld16(Ks + (16 * kb + c) * TROW + ks * 32 + g * 8);
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) st[kb][qb] = mma(kf, qf[qb][ks], st[kb][qb]);
        }
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) st[kb][qb] = mma(oh[kb], bq[qb], st[kb][qb]);
}
__device__ __forceinline__ bool slot_valid(int th, int tw, int s, int gh, int gw) { return 8 * th + (s >> 3) < gh && 8 * tw + (s & 7) < gw; }

template <int NW>
__global__ __launch_bounds__(NW * 64, 2) void attn_fwd2d_kernel(AttnDev a) {
    __shared__ __align__(16) bf16_t Ks[64 * TROW];
    __shared__ __align__(16) bf16_t Vs[64 * TROW];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
    const int bh = blockIdx.y, b = bh / a.heads, h = bh - b * a.heads;
    const int L = a.L, Dq = a.Dq, q0 = blockIdx.x * (NW * 32) + wave * 32, ld3 = 3 * a.heads * HD, ld1 = a.heads * HD;
    u32x4_t qf[2][2], oh[4];
    const bf16_t* qrow[2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const int q = q0 + 16 * qb + c;
        qrow[qb] = q < L ? a.Qp + ((long)bh * L + q) * Dq : nullptr;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) qf[qb][ks] = qrow[qb] ? ld16(qrow[qb] + ks * 32 + g * 8) : zero16();
    }
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) oh[kb] = onehot_keys(16 * kb + c, g);
    f32x4_t o[4][2];
    float m[2], ls[2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        m[qb] = -INFINITY; ls[qb] = 0.f;
#pragma unroll
        for (int db = 0; db < 4; ++db) o[db][qb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
    for (int kt = 0; kt < a.nt2; ++kt) {
        const int th = kt / a.ntw, tw = kt - th * a.ntw;
        __syncthreads();
        {
            GatherTile<NW * 64> tk;
            tk.fetch(a.qkv + (long)b * L * ld3 + ld1 + h * HD, ld3, th, tw, a.gh, a.gw);
            tk.store(Ks, TROW);
        }
        tile_load<64, 64, NW * 64>(Vs, TROW, a.VT + ((long)bh * a.nt2 + kt) * 4096, 64, 64);
        u32x4_t bq[2];
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
            bq[qb] = (qrow[qb] && g < 2) ? ld16(qrow[qb] + 64 + (g == 0 ? 8 * th : a.wofs + 8 * tw)) : zero16();
        __syncthreads();
        f32x4_t st[4][2];
        scores2d(st, Ks, qf, bq, oh, c, g);
        if (8 * th + 8 > a.gh || 8 * tw + 8 > a.gw) {          // only the last row / column of key blocks has slots outside the grid
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (!slot_valid(th, tw, 16 * kb + 4 * g + i, a.gh, a.gw)) { st[kb][0][i] = -INFINITY; st[kb][1][i] = -INFINITY; }
        }
        u32x4_t pf[2][2];
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            float mx = -INFINITY;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int i = 0; i < 4; ++i) mx = fmaxf(mx, st[kb][qb][i]);
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float mn = fmaxf(m[qb], mx);
            const float alpha = __expf(m[qb] - mn);
            m[qb] = mn;
            float sum = 0.f;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float p = __expf(st[kb][qb][i] - mn);
                    st[kb][qb][i] = p;
                    sum += p;
                }
            ls[qb] = ls[qb] * alpha + sum;
#pragma unroll
            for (int db = 0; db < 4; ++db)
#pragma unroll
                for (int i = 0; i < 4; ++i) o[db][qb][i] *= alpha;
            pf[qb][0] = pack_perm(st[0][qb], st[1][qb]);
            pf[qb][1] = pack_perm(st[2][qb], st[3][qb]);
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int db = 0; db < 4; ++db) {
                const u32x4_t vf = ld_perm(Vs + (16 * db + c) * TROW, 32 * s2 + 4 * g);
#pragma unroll
                for (int qb = 0; qb < 2; ++qb) o[db][qb] = mma(vf, pf[qb][s2], o[db][qb]);
            }
    }
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        float l = ls[qb];
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        const int q = q0 + 16 * qb + c;
        if (q < L) {
            const float inv = 1.f / l;
            bf16_t* orow = a.Ow + ((long)b * L + q) * ld1 + h * HD;
#pragma unroll
            for (int db = 0; db < 4; ++db) {
                uint2 t;
                t.x = pack2_bf16(o[db][qb][0] * inv, o[db][qb][1] * inv);
                t.y = pack2_bf16(o[db][qb][2] * inv, o[db][qb][3] * inv);
                *reinterpret_cast<uint2*>(orow + 16 * db + 4 * g) = t;
            }
            if (g == 0) a.lse[(long)bh * L + q] = m[qb] + __logf(l);
        }
    }
}

// dQ' of the tiled path: 4 accumulator blocks for the q part; the 16 bias columns of each tile are added into an LDS row per
// query at the tile's column offsets (each (query, column) has exactly one owner lane: plain read-modify-write)
__global__ __launch_bounds__(256, 2) void attn_bwd_dq2d_kernel(AttnDev a) {
    extern __shared__ __align__(16) unsigned char smem[];
    bf16_t* Ks = reinterpret_cast<bf16_t*>(smem);            // [64 slots][TROW] k rows
    bf16_t* KTs = Ks + 64 * TROW;                             // [64 d][TROW]     k^T
    bf16_t* Vs = KTs + 64 * TROW;                             // [64 slots][TROW] v rows
    const int NB = a.Dq - 64;                                 // bias columns (h block, w block, zero padding)
    const int NBP = NB + 1;                                   // odd row stride: the 16 queries of a lane group fall into different banks
    float* accB = reinterpret_cast<float*>(Vs + 64 * TROW);  // [128 q][NBP]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
    const int bh = blockIdx.y, b = bh / a.heads, h = bh - b * a.heads;
    const int L = a.L, Dq = a.Dq, q0 = blockIdx.x * 128 + wave * 32, ld3 = 3 * a.heads * HD, ld1 = a.heads * HD;
    for (int i = threadIdx.x; i < 128 * NBP; i += 256) accB[i] = 0.f;
    u32x4_t qf[2][2], dof[2][2], oh[4], ohT[2];
    const bf16_t* qrow[2];
    float lse[2], dl[2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const int q = q0 + 16 * qb + c;
        const bool ok = q < L;
        qrow[qb] = ok ? a.Qp + ((long)bh * L + q) * Dq : nullptr;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            qf[qb][ks] = ok ? ld16(qrow[qb] + ks * 32 + g * 8) : zero16();
            dof[qb][ks] = ok ? ld16(a.dO + ((long)b * L + q) * ld1 + h * HD + ks * 32 + g * 8) : zero16();
        }
        lse[qb] = ok ? a.lse_r[(long)bh * L + q] : INFINITY;
        dl[qb] = ok ? a.delta[(long)bh * L + q] : 0.f;
    }
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) oh[kb] = onehot_keys(16 * kb + c, g);
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {      // rows j = c of the transposed one-hot block; k-slots in the permuted 32-step order
        unsigned short e[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int slot = 32 * s2 + (i < 4 ? 4 * g + i : 16 + 4 * g + (i - 4));
            const bool hit = c < 8 ? (slot >> 3) == c : (slot & 7) == c - 8;
            e[i] = hit ? 0x3f80 : 0;
        }
        ohT[s2] = u32x4_t{e[0] | ((unsigned)e[1] << 16), e[2] | ((unsigned)e[3] << 16), e[4] | ((unsigned)e[5] << 16), e[6] | ((unsigned)e[7] << 16)};
    }
    f32x4_t dq[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i) { dq[i][0] = f32x4_t{0.f, 0.f, 0.f, 0.f}; dq[i][1] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
    GatherTile<256> tK, tV;
    Tile<64, 64, 256> tKT;
    const bf16_t* kbase = a.qkv + (long)b * L * ld3 + ld1 + h * HD;
    auto fetch = [&](int kt) {
        const int th = kt / a.ntw, tw = kt - th * a.ntw;
        tK.fetch(kbase, ld3, th, tw, a.gh, a.gw);
        tV.fetch(kbase + ld1, ld3, th, tw, a.gh, a.gw);
        tKT.fetch(a.KpT + ((long)bh * a.nt2 + kt) * 4096, 64, 64);
    };
    fetch(0);
    for (int kt = 0; kt < a.nt2; ++kt) {
        const int th = kt / a.ntw, tw = kt - th * a.ntw;
        __syncthreads();
        tK.store(Ks, TROW);
        tKT.store(KTs, TROW);
        tV.store(Vs, TROW);
        __syncthreads();
        if (kt + 1 < a.nt2) fetch(kt + 1);
        u32x4_t bq[2];
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
            bq[qb] = (qrow[qb] && g < 2) ? ld16(qrow[qb] + 64 + (g == 0 ? 8 * th : a.wofs + 8 * tw)) : zero16();
        f32x4_t st[4][2], dp[4][2];
        scores2d(st, Ks, qf, bq, oh, c, g);
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) dp[kb][qb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                const u32x4_t vf = ld16(Vs + (16 * kb + c) * TROW + ks * 32 + g * 8);
#pragma unroll
                for (int qb = 0; qb < 2; ++qb) dp[kb][qb] = mma(vf, dof[qb][ks], dp[kb][qb]);
            }
        u32x4_t dsf[2][2];
        const bool ragged = 8 * th + 8 > a.gh || 8 * tw + 8 > a.gw;
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float p = __expf(st[kb][qb][i] - lse[qb]);
                    if (ragged && !slot_valid(th, tw, 16 * kb + 4 * g + i, a.gh, a.gw)) p = 0.f;
                    st[kb][qb][i] = p * (dp[kb][qb][i] - dl[qb]);
                }
            dsf[qb][0] = pack_perm(st[0][qb], st[1][qb]);
            dsf[qb][1] = pack_perm(st[2][qb], st[3][qb]);
        }
        f32x4_t db16[2] = {f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
            for (int db = 0; db < 4; ++db) {
                const u32x4_t kt_f = ld_perm(KTs + (16 * db + c) * TROW, 32 * s2 + 4 * g);
#pragma unroll
                for (int qb = 0; qb < 2; ++qb) dq[db][qb] = mma(kt_f, dsf[qb][s2], dq[db][qb]);
            }
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) db16[qb] = mma(ohT[s2], dsf[qb][s2], db16[qb]);
        }
        // rows 4g+i of the bias block: 0..7 -> h columns 8th.., 8..15 -> w columns wofs + 8tw..
        const int col0 = g < 2 ? 8 * th + 4 * g : a.wofs + 8 * tw + 4 * (g - 2);
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            float* row = accB + (wave * 32 + 16 * qb + c) * NBP + col0;
#pragma unroll
            for (int i = 0; i < 4; ++i) row[i] += db16[qb][i];
        }
    }
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const int q = q0 + 16 * qb + c;
        if (q < L) {
            bf16_t* row = a.dQp + ((long)bh * L + q) * Dq;
#pragma unroll
            for (int db = 0; db < 4; ++db) {
                uint2 t;
                t.x = pack2_bf16(dq[db][qb][0], dq[db][qb][1]);
                t.y = pack2_bf16(dq[db][qb][2], dq[db][qb][3]);
                *reinterpret_cast<uint2*>(row + 16 * db + 4 * g) = t;
            }
        }
    }
    // this wave's 32 rows of bias-column gradients (only this wave touched them)
    for (int idx = lane; idx < 32 * (NB >> 2); idx += 64) {
        const int qi = idx / (NB >> 2), c4 = (idx - qi * (NB >> 2)) * 4, q = q0 + qi;
        if (q < L) store4(a.dQp + ((long)bh * L + q) * Dq + 64 + c4, accB + (wave * 32 + qi) * NBP + c4);
    }
}

// dK, dV of the tiled path: one block per key tile (64 slots, 16 per wave), sweeping linear query tiles whose LDS rows carry
// the 64 q columns plus THIS tile's 2 x 8 bias columns
__global__ __launch_bounds__(256) void attn_bwd_dkv2d_kernel(AttnDev a) {
    constexpr int QR = 96 + 8;                                 // q (64) | bias (16) | zero (16) | pad
    extern __shared__ __align__(16) unsigned char smem[];
    bf16_t* Qs = reinterpret_cast<bf16_t*>(smem);            // [64 q][QR]
    bf16_t* dOs = Qs + 64 * QR;                               // [64 q][TROW]
    bf16_t* QTs = dOs + 64 * TROW;                            // [64 d][TROW]
    bf16_t* dOTs = QTs + 64 * TROW;                           // [64 d][TROW]
    float* lse_s = reinterpret_cast<float*>(dOTs + 64 * TROW);
    float* dl_s = lse_s + 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
    const int bh = blockIdx.y, b = bh / a.heads, h = bh - b * a.heads, kt = blockIdx.x, th = kt / a.ntw, tw = kt - th * a.ntw;
    const int L = a.L, Dq = a.Dq, ld3 = 3 * a.heads * HD, ld1 = a.heads * HD;
    const int slot = 16 * wave + c, key = slot_token(th, tw, slot, a.gh, a.gw);
    u32x4_t kf[3], vf[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        kf[ks] = key >= 0 ? ld16(a.qkv + ((long)b * L + key) * ld3 + ld1 + h * HD + ks * 32 + g * 8) : zero16();
        vf[ks] = key >= 0 ? ld16(a.qkv + ((long)b * L + key) * ld3 + 2 * ld1 + h * HD + ks * 32 + g * 8) : zero16();
    }
    kf[2] = onehot_keys(slot, g);                              // B operand of the bias step: column = slot, same chunks as the A form
    f32x4_t dv[4], dk[4];
#pragma unroll
    for (int db = 0; db < 4; ++db) { dv[db] = f32x4_t{0.f, 0.f, 0.f, 0.f}; dk[db] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
    for (int i = threadIdx.x; i < 64 * 2; i += 256)           // the zero chunk of every Qs row (columns 80..95) is written once
        *reinterpret_cast<u32x4_t*>(Qs + (i >> 1) * QR + 80 + (i & 1) * 8) = zero16();
    const int nqt = a.Lp >> 6;
    Tile<64, 64, 256> tQ, tdO, tQT, tdOT;
    u32x4_t bias_n = zero16();
    float lse_n = 0.f, dl_n = 0.f;
    auto fetch = [&](int qt) {
        tQ.fetch(a.Qp + ((long)bh * L + qt * 64) * Dq, Dq, L - qt * 64);
        tdO.fetch(a.dO + ((long)b * L + qt * 64) * ld1 + h * HD, ld1, L - qt * 64);
        tQT.fetch(a.QsT + (long)bh * HD * a.Lp + qt * 64, a.Lp, 64);
        tdOT.fetch(a.dOT + (long)bh * HD * a.Lp + qt * 64, a.Lp, 64);
        if (threadIdx.x < 128) {                               // bias chunk (h or w) of query threadIdx.x >> 1
            const int q = qt * 64 + (threadIdx.x >> 1);
            bias_n = q < L ? ld16(a.Qp + ((long)bh * L + q) * Dq + 64 + ((threadIdx.x & 1) ? a.wofs + 8 * tw : 8 * th)) : zero16();
        }
        if (threadIdx.x < 64) {
            const int q = qt * 64 + threadIdx.x;
            lse_n = q < L ? a.lse_r[(long)bh * L + q] : INFINITY;
            dl_n = q < L ? a.delta[(long)bh * L + q] : 0.f;
        }
    };
    fetch(0);
    for (int qt = 0; qt < nqt; ++qt) {
        __syncthreads();
        tQ.store(Qs, QR);
        tdO.store(dOs, TROW);
        tQT.store(QTs, TROW);
        tdOT.store(dOTs, TROW);
        if (threadIdx.x < 128) *reinterpret_cast<u32x4_t*>(Qs + (threadIdx.x >> 1) * QR + 64 + (threadIdx.x & 1) * 8) = bias_n;
        if (threadIdx.x < 64) { lse_s[threadIdx.x] = lse_n; dl_s[threadIdx.x] = dl_n; }
        __syncthreads();
        if (qt + 1 < nqt) fetch(qt + 1);
        f32x4_t s[4], dp[4];
#pragma unroll
        for (int qb = 0; qb < 4; ++qb) { s[qb] = f32x4_t{0.f, 0.f, 0.f, 0.f}; dp[qb] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int ks = 0; ks < 3; ++ks)
#pragma unroll
            for (int qb = 0; qb < 4; ++qb) s[qb] = mma(ld16(Qs + (16 * qb + c) * QR + ks * 32 + g * 8), kf[ks], s[qb]);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int qb = 0; qb < 4; ++qb) dp[qb] = mma(ld16(dOs + (16 * qb + c) * TROW + ks * 32 + g * 8), vf[ks], dp[qb]);
#pragma unroll
        for (int qb = 0; qb < 4; ++qb) {
            const f32x4_t l4 = *reinterpret_cast<const f32x4_t*>(lse_s + 16 * qb + 4 * g);
            const f32x4_t d4 = *reinterpret_cast<const f32x4_t*>(dl_s + 16 * qb + 4 * g);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float p = __expf(s[qb][i] - l4[i]);
                s[qb][i] = p;
                dp[qb][i] = p * (dp[qb][i] - d4[i]);
            }
        }
        const u32x4_t pf[2] = {pack_perm(s[0], s[1]), pack_perm(s[2], s[3])};
        const u32x4_t dsf[2] = {pack_perm(dp[0], dp[1]), pack_perm(dp[2], dp[3])};
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int db = 0; db < 4; ++db) {
                dv[db] = mma(ld_perm(dOTs + (16 * db + c) * TROW, 32 * s2 + 4 * g), pf[s2], dv[db]);
                dk[db] = mma(ld_perm(QTs + (16 * db + c) * TROW, 32 * s2 + 4 * g), dsf[s2], dk[db]);
            }
    }
    if (key >= 0) {
        bf16_t* row = a.dqkv + ((long)b * L + key) * ld3 + h * HD;
#pragma unroll
        for (int db = 0; db < 4; ++db) {
            uint2 t;
            t.x = pack2_bf16(dk[db][0], dk[db][1]);
            t.y = pack2_bf16(dk[db][2], dk[db][3]);
            *reinterpret_cast<uint2*>(row + ld1 + 16 * db + 4 * g) = t;
            t.x = pack2_bf16(dv[db][0], dv[db][1]);
            t.y = pack2_bf16(dv[db][2], dv[db][3]);
            *reinterpret_cast<uint2*>(row + 2 * ld1 + 16 * db + 4 * g) = t;
        }
    }
}

template <typename K>
int set_lds(K kernel, size_t bytes) {
    if (bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return aldi_set_error(e, __FILE__, __LINE__);
    }
    return ALDI_OK;
}

struct Layout { int tiled, ghp, gwp, wofs, Dq, ntw, nt2; long vt_cols; };
Layout layout_of(int gh, int gw, int rel) {
    Layout l{};
    const int L = gh * gw, Lp = (L + 63) / 64 * 64;
    l.tiled = rel && L > 7 * 32;                      // windows (one workgroup per window-head) keep the linear-tile kernels
    l.ghp = (gh + 7) / 8 * 8; l.gwp = (gw + 7) / 8 * 8;
    l.wofs = l.tiled ? l.ghp : gh;
    const int need = rel ? 64 + l.wofs + (l.tiled ? l.gwp : gw) : 64;
    l.Dq = (need + 31) / 32 * 32;
    l.ntw = l.gwp / 8; l.nt2 = (l.ghp / 8) * l.ntw;
    l.vt_cols = l.tiled && l.nt2 * 64 > Lp ? l.nt2 * 64 : Lp;
    return l;
}

AttnDev to_dev(const aldi_attn_args* p) {
    AttnDev a{};
    a.qkv = (const bf16_t*)p->qkv; a.Qp = (const bf16_t*)p->Qp; a.Kp = (const bf16_t*)p->Kp; a.KpT = (const bf16_t*)p->KpT;
    a.VT = (const bf16_t*)p->VT; a.QsT = (const bf16_t*)p->QsT; a.dOT = (const bf16_t*)p->dOT; a.O = (const bf16_t*)p->O;
    a.dO = (const bf16_t*)p->dO; a.Ow = (bf16_t*)p->O; a.dQp = (bf16_t*)p->dQp; a.dqkv = (bf16_t*)p->dqkv;
    a.lse = p->lse; a.lse_r = p->lse; a.delta = p->delta;
    a.nB = p->nB; a.L = p->gh * p->gw; a.Lp = (a.L + 63) / 64 * 64; a.heads = p->heads; a.Dq = p->Dq;
    const Layout l = layout_of(p->gh, p->gw, p->rel_h != nullptr);
    a.gh = p->gh; a.gw = p->gw; a.wofs = l.wofs; a.ntw = l.ntw; a.nt2 = l.nt2;
    return a;
}

int check_args(const aldi_attn_args* p) {
    if (!p || p->nB <= 0 || p->gh <= 0 || p->gw <= 0 || p->heads <= 0) return aldi_set_error_msg(ALDI_ERR_ARG, "attn: bad sizes");
    if ((p->rel_h == nullptr) != (p->rel_w == nullptr)) return aldi_set_error_msg(ALDI_ERR_ARG, "attn: rel_h and rel_w go together");
    const Layout l = layout_of(p->gh, p->gw, p->rel_h != nullptr);
    if (p->Dq != l.Dq || p->Dq > 256) return aldi_set_error_msg(ALDI_ERR_ARG, "attn: Dq must be the value aldi_attn_layout reports (<= 256)");
    return ALDI_OK;
}

template <int NKS>
int launch_fwd(const AttnDev& a, hipStream_t st) {
    constexpr int DQ = NKS * 32;
    const size_t lds = (size_t)(64 * (DQ + 8) + 64 * TROW) * 2;
    if (a.L <= 7 * 32) {        // a whole window in one block: K/V tiles are read once
        if (int e = set_lds(attn_fwd_kernel<NKS, 7>, lds)) return e;
        hipLaunchKernelGGL((attn_fwd_kernel<NKS, 7>), dim3(1, a.nB * a.heads), dim3(7 * 64), lds, st, a);
    } else {
        if (int e = set_lds(attn_fwd_kernel<NKS, 4>, lds)) return e;
        hipLaunchKernelGGL((attn_fwd_kernel<NKS, 4>), dim3(cdiv(a.L, 128), a.nB * a.heads), dim3(256), lds, st, a);
    }
    ALDI_CHECK_LAUNCH();
    return ALDI_OK;
}
template <int NKS>
int launch_bwd(const AttnDev& a, hipStream_t st) {
    constexpr int DQ = NKS * 32;
    const size_t lds_q = (size_t)(64 * (DQ + 8) + DQ * TROW + 64 * TROW) * 2;
    const size_t lds_kv = (size_t)(64 * (DQ + 8) + 3 * 64 * TROW) * 2 + 128 * 4;
    if (a.L <= 7 * 32) {
        if (int e = set_lds(attn_bwd_dq_kernel<NKS, 7>, lds_q)) return e;
        hipLaunchKernelGGL((attn_bwd_dq_kernel<NKS, 7>), dim3(1, a.nB * a.heads), dim3(7 * 64), lds_q, st, a);
    } else {
        if (int e = set_lds(attn_bwd_dq_kernel<NKS, 4>, lds_q)) return e;
        hipLaunchKernelGGL((attn_bwd_dq_kernel<NKS, 4>), dim3(cdiv(a.L, 128), a.nB * a.heads), dim3(256), lds_q, st, a);
    }
    ALDI_CHECK_LAUNCH();
    if (int e = set_lds(attn_bwd_dkv_kernel<NKS>, lds_kv)) return e;
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<NKS>), dim3(cdiv(a.L, 128), a.nB * a.heads), dim3(256), lds_kv, st, a);
    ALDI_CHECK_LAUNCH();
    return ALDI_OK;
}

#define ATTN_DISPATCH(fn, a, st)                                           \
    switch ((a).Dq / 32) {                                                 \
        case 2: return fn<2>(a, st);                                       \
        case 3: return fn<3>(a, st);                                       \
        case 4: return fn<4>(a, st);                                       \
        case 5: return fn<5>(a, st);                                       \
        case 6: return fn<6>(a, st);                                       \
        case 7: return fn<7>(a, st);                                       \
        case 8: return fn<8>(a, st);                                       \
        default: return aldi_set_error_msg(ALDI_ERR_ARG, "attn: unsupported Dq"); \
    }

int dispatch_fwd(const AttnDev& a, int tiled, hipStream_t st) {
    if (tiled) {
        hipLaunchKernelGGL((attn_fwd2d_kernel<4>), dim3(cdiv(a.L, 128), a.nB * a.heads), dim3(256), 0, st, a);
        ALDI_CHECK_LAUNCH();
        return ALDI_OK;
    }
    ATTN_DISPATCH(launch_fwd, a, st)
}
int dispatch_bwd(const AttnDev& a, int tiled, hipStream_t st) {
    if (tiled) {
        const size_t lds_q = (size_t)3 * 64 * TROW * 2 + (size_t)128 * (a.Dq - 64 + 1) * 4;
        if (int e = set_lds(attn_bwd_dq2d_kernel, lds_q)) return e;
        hipLaunchKernelGGL(attn_bwd_dq2d_kernel, dim3(cdiv(a.L, 128), a.nB * a.heads), dim3(256), lds_q, st, a);
        ALDI_CHECK_LAUNCH();
        const size_t lds_kv = (size_t)(64 * 104 + 3 * 64 * TROW) * 2 + 128 * 4;
        hipLaunchKernelGGL(attn_bwd_dkv2d_kernel, dim3(a.nt2, a.nB * a.heads), dim3(256), lds_kv, st, a);
        ALDI_CHECK_LAUNCH();
        return ALDI_OK;
    }
    ATTN_DISPATCH(launch_bwd, a, st)
}

}  // namespace

extern "C" int aldi_attn_layout(int gh, int gw, int rel, int* Dq, int* tiled, long* vt_cols) {
    if (gh <= 0 || gw <= 0 || !Dq || !tiled || !vt_cols) return aldi_set_error_msg(ALDI_ERR_ARG, "attn_layout: bad args");
    const Layout l = layout_of(gh, gw, rel);
    *Dq = l.Dq; *tiled = l.tiled; *vt_cols = l.vt_cols;
    return ALDI_OK;
}

extern "C" int aldi_attn_prepare(const aldi_attn_args* p, aldi_stream_t stream) {
    if (int e = check_args(p)) return e;
    const Layout l = layout_of(p->gh, p->gw, p->rel_h != nullptr);
    PrepDev a{};
    a.qkv = (const bf16_t*)p->qkv; a.rel_h = p->rel_h; a.rel_w = p->rel_w;
    a.Qp = (bf16_t*)p->Qp; a.Kp = (bf16_t*)p->Kp; a.KpT = (bf16_t*)p->KpT; a.VT = (bf16_t*)p->VT; a.QsT = (bf16_t*)p->QsT;
    a.nB = p->nB; a.L = p->gh * p->gw; a.Lp = (a.L + 63) / 64 * 64; a.heads = p->heads; a.Dq = p->Dq; a.gh = p->gh; a.gw = p->gw;
    a.wofs = l.wofs; a.tiled = l.tiled;
    a.scale = p->scale;
    const int nrb = p->rel_h ? (2 * p->gh - 1 + 2 * p->gw - 1 + 15) / 16 : 0;
    const size_t lds = (size_t)(64 * TROW + nrb * 16 * TROW + 2 * 64 * 64 + 64 * p->Dq) * 2 + 128 * 4;
    if (int e = set_lds(attn_prep_kernel, lds)) return e;
    a.tiles_per_block = a.Lp / 64 >= 16 ? 4 : 1;      // long sequences: amortise the table conversion (68 KB of fp32 per block on 50 x 84)
    hipLaunchKernelGGL(attn_prep_kernel, dim3(cdiv(a.Lp / 64, a.tiles_per_block), a.nB * a.heads), dim3(256), lds, (hipStream_t)stream, a);
    ALDI_CHECK_LAUNCH();
    if (l.tiled) {
        const AttnDev d = to_dev(p);
        hipLaunchKernelGGL(attn_prep2d_kernel, dim3(d.nt2, d.nB * d.heads), dim3(256), 0, (hipStream_t)stream, d, (bf16_t*)p->KpT, (bf16_t*)p->VT);
        ALDI_CHECK_LAUNCH();
    }
    return ALDI_OK;
}

extern "C" int aldi_attn_forward(const aldi_attn_args* p, aldi_stream_t stream) {
    if (int e = check_args(p)) return e;
    return dispatch_fwd(to_dev(p), layout_of(p->gh, p->gw, p->rel_h != nullptr).tiled, (hipStream_t)stream);
}

extern "C" int aldi_attn_backward(const aldi_attn_args* p, aldi_stream_t stream) {
    if (int e = check_args(p)) return e;
    if (p->rel_h && (!p->drel_h || !p->drel_w)) return aldi_set_error_msg(ALDI_ERR_ARG, "attn: drel_h / drel_w missing");
    hipStream_t st = (hipStream_t)stream;
    AttnDev a = to_dev(p);
    BprepDev bp{a.O, a.dO, (bf16_t*)p->dOT, p->delta, a.nB, a.L, a.Lp, a.heads};
    hipLaunchKernelGGL(attn_bwd_prep_kernel, dim3(a.Lp / 64, a.nB * a.heads), dim3(256), 0, st, bp);
    ALDI_CHECK_LAUNCH();
    if (int e = dispatch_bwd(a, layout_of(p->gh, p->gw, p->rel_h != nullptr).tiled, st)) return e;
    RbwdDev r{};
    r.qkv = a.qkv; r.dQp = a.dQp; r.rel_h = p->rel_h; r.rel_w = p->rel_w; r.dqkv = a.dqkv; r.drel_h = p->drel_h; r.drel_w = p->drel_w;
    r.nB = a.nB; r.L = a.L; r.heads = a.heads; r.Dq = a.Dq; r.gh = p->gh; r.gw = p->gw; r.wofs = a.wofs; r.scale = p->scale;
    const int ntiles = a.Lp / 64, ntab = 2 * p->gh - 1 + 2 * p->gw - 1;
    r.tiles_per_block = ntiles > 8 ? 8 : ntiles;
    const size_t lds_q = p->rel_h ? (size_t)(64 * ((ntab + 31) / 32 * 32 + 8) + 64 * (a.Dq - 64 + 8)) * 2 + 128 * 4 : 16;
    if (int e = set_lds(attn_rel_dq_kernel, lds_q)) return e;
    hipLaunchKernelGGL(attn_rel_dq_kernel, dim3(ntiles, a.nB * a.heads), dim3(256), lds_q, st, r);
    ALDI_CHECK_LAUNCH();
    if (p->rel_h) {
        if (ntab > 4 * DT_MAXRB * 16) return aldi_set_error_msg(ALDI_ERR_ARG, "attn: 2(gh+gw)-2 > 384 table rows");
        const size_t lds_t = (size_t)(64 * (a.Dq - 64 + 8) + 64 * TROW) * 2 + 128 * 4;
        if (int e = set_lds(attn_rel_dtab_kernel, lds_t)) return e;
        hipLaunchKernelGGL(attn_rel_dtab_kernel, dim3(cdiv(ntiles, r.tiles_per_block), a.nB * a.heads), dim3(256), lds_t, st, r, a.QsT, a.Lp);
        ALDI_CHECK_LAUNCH();
    }
    return ALDI_OK;
}
