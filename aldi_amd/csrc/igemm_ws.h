// Plain 1x1 convolutions with a SHORT reduction (K = Cin = 64 .. 512 channels), bf16: the bottlenecks' expansions / reductions of res3 and res4, their
// data gradients, the FPN laterals -- ~60 launches of the step.  Included by igemm.hip (ConvDev, the tile primitives).
//
// Why another kernel.  These layers are memory streams (25-60 flop per byte), yet the tile kernels run them at 2.7-4.3 TB/s.  Ablation of the 128 x 64
// tile on res3 conv3 (67 200 px, 128 -> 512, tools/igemm_sweep.py with igemm_dbg 4 / 8 / 12): 9.5 us of workgroup turnover (4 200 workgroups of 32
// MFMAs per wave), + 6-10 us for the K loop's operands, + 10-13 us for the stores -- and the three ADD: every workgroup re-fetches its weight tile
// (69 MB over the launch, as much as the output) and the pixel tile is fetched once per 64 output channels (8 x 17 MB), all through the same
// per-CU vector-memory path the stores and the residual loads go through.  Here the WEIGHTS STAY IN REGISTERS:
//   * a workgroup (4 waves) owns 64 x TN output channels (256; 128 for K >= 256) for the whole launch; wave w keeps the MFMA B fragments of its 16 x TN channels for ALL of K
//     in VGPRs (TN x K/32 fragments = 64-128 registers), fetched once -- the K loop reads only pixel fragments from the LDS (0.25-0.5 reads per MFMA)
//     and there is no weight traffic after the prologue;
//   * it is persistent: ~2 workgroups per CU walk the pixel tiles (BM = 16 x TM rows) in a strided order; the tiles of K x BM x 2 bytes arrive by
//     LDS-DMA as whole 128-byte lines (igemm_halo64.h: the L2 -> LDS path is paced per line) through a ring of three stages, two tiles ahead;
//   * one counted `vmcnt` and one raw `s_barrier` per tile: per iteration a wave issues the residual / mask-bit loads of tile t, then the DMA of tile
//     t + 2, runs the MFMAs of tile t, waits `vmcnt(P)` (P = its DMA pieces of tile t + 2: everything older has landed -- the residual of t and tile
//     t + 1) and finishes tile t straight from the accumulators: permuted channel rows (direct_perm) so that a lane holds 8 consecutive channels of
//     its pixel, v = acc * scale + shift + residual in fp32, mask bits, ONE rounding, ReLU, 16-byte stores, the output's ReLU bits beside it.
// K order per output element: ascending channels, as in every tap-form tile (but fp32 sums of MFMA k-steps of 32: same as the 32-channel slabs).
#pragma once

template <int KS, int TM, int TN>
__global__ __launch_bounds__(256, 2) void igemm_ws_kernel(ConvDev p, const int n_mtiles, const int ncg) {
    typedef bf16_t T;
    constexpr int BM = TM * 16, K = KS * 32, WNE = TN * 16, BN = 4 * WNE, H = TN / 2;
    constexpr int STAGE16 = BM * K * 2 / 16;                  // 16-B slots per stage
    constexpr int P = STAGE16 / 256;                          // DMA pieces (64 slots) per wave and stage
    constexpr int NBUF = 3;
    static_assert(TN % 2 == 0 && STAGE16 % 256 == 0 && P >= 1 && K % 64 == 0, "whole 32-channel blocks per wave; whole DMA pieces per wave; whole 128-byte slabs");
    __shared__ __attribute__((aligned(128))) uint4 lds[NBUF * STAGE16 + BN / 2];      // [stage 0 | 1 | 2 | BN scales | BN shifts]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;

    // workgroup -> (channel group, pixel-tile sequence): the ncg workgroups that walk the same pixel tiles sit on one XCD (ids 8 apart: the
    // dispatcher deals workgroups round-robin over the XCDs) and run in step, so a pixel tile comes from HBM once and from that XCD's L2 after
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int cg = slot % ncg, g = (slot / ncg) * 8 + xcd, G = (int)(gridDim.x >> 3) / ncg * 8;
    const int nt = g < n_mtiles ? (n_mtiles - g + G - 1) / G : 0;
    const int n0 = cg * BN + wave * WNE;                      // this wave's first channel

    // ---- the weights of this wave's channels, all of K, into registers: fragment (j, ks) row r = channel n0 + direct_perm(j * 16 + r), 8 k per lane
    u32x4_t wreg[TN][KS];
    {
        const T* __restrict__ Wp = static_cast<const T*>(p.w);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int co = n0 + (j >> 1) * 32 + (fr >> 2) * 8 + (j & 1) * 4 + (fr & 3);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) wreg[j][ks] = *reinterpret_cast<const u32x4_t*>(Wp + (size_t)co * K + ks * 32 + fq * 8);
        }
    }
    {
        float* aux = reinterpret_cast<float*>(&lds[NBUF * STAGE16]);
        for (int c = tid; c < BN; c += 256) {
            aux[c] = p.scale ? p.scale[cg * BN + c] : 1.f;
            aux[BN + c] = p.shift ? p.shift[cg * BN + c] : 0.f;
        }
    }
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(static_cast<const T*>(p.x)), 0, p.x_bytes, 0x00020000);
    const unsigned out_bytes = (unsigned)p.M * (unsigned)p.Cout * 2u;
    const bool res_up = p.res_mode == 2;                      // the residual is the coarser pyramid level's map, nearest-upsampled (FPN top-down sum): [N][Ho / 2][Wo / 2][Cout]
    const __amdgpu_buffer_rsrc_t rr = make_rsrc_uniform(p.res, res_up ? out_bytes >> 2 : out_bytes);
    const __amdgpu_buffer_rsrc_t rmb = make_rsrc_uniform(p.mask_bits, out_bytes >> 4);
    const __amdgpu_buffer_rsrc_t ry = make_rsrc_uniform(p.y, out_bytes);
    const __amdgpu_buffer_rsrc_t rbo = make_rsrc_uniform(p.bits_out, out_bytes >> 4);
    constexpr unsigned OOB = 0x80000000u;

    // ---- DMA sources.  Stage image: [128-byte slab s][row][8 chunks]; slot c of piece q: row (c >> 3) % BM of slab (c >> 3) / BM, physical chunk
    // c & 7 holds logical chunk (c & 7) ^ ((row >> 1) & 7) (swz<8>: conflict-free for the 8-lane DMA writes and the 16-lane fragment reads)
    unsigned vbase[P];
#pragma unroll
    for (int it = 0; it < P; ++it) {
        const int c = (wave + 4 * it) * 64 + lane, rr_ = c >> 3, r = rr_ % BM, s = rr_ / BM, lc = (c & 7) ^ ((r >> 1) & 7);
        vbase[it] = (unsigned)((r * K + s * 64 + lc * 8) * 2);
    }
    const int wbase = __builtin_amdgcn_readfirstlane(wave * 64);
    auto issue_tile = [&](int k, int stage) {                 // tile g + k * G -> ring stage
        const unsigned m0b = (unsigned)(g + k * G) * (unsigned)(BM * K * 2);
#pragma unroll
        for (int it = 0; it < P; ++it) glds16(rx, &lds[stage * STAGE16 + it * 256 + wbase], vbase[it] + m0b);       // (rows >= M: out of range, zeros)
    };
    const unsigned lds0 = lds_addr(&lds[0]);
    const unsigned rd0 = lds0 + (unsigned)(fr * 128 + ((fq ^ ((fr >> 1) & 7)) * 16));                 // k-step parity 0 / 1 of a slab
    const unsigned rd1 = lds0 + (unsigned)(fr * 128 + (((4 + fq) ^ ((fr >> 1) & 7)) * 16));
    const unsigned aux_rd = lds0 + (unsigned)(NBUF * STAGE16 * 16) + (unsigned)((wave * WNE + fq * 8) * 4);
    const bool has_sc = p.scale != nullptr, has_sh = p.shift != nullptr, has_res = p.res_mode != 0, has_mb = p.mask_bits != nullptr, has_bo = p.bits_out != nullptr;
    const bool relu = p.relu != 0, no_epi = (p.dbg & 4) != 0;
    const unsigned C2 = (unsigned)p.Cout * 2u, C8 = (unsigned)p.Cout >> 3;
    typedef short s16x2_t __attribute__((ext_vector_type(2)));
    typedef unsigned short u16x2_t __attribute__((ext_vector_type(2)));

    // m / d for m < 2^24 without the 40-instruction integer division (twice per fragment row and tile here): float quotient, one correction step
    const float inv_wo = 1.f / (float)p.Wo, inv_ho = 1.f / (float)p.Ho;
    auto fdiv = [](unsigned m, unsigned d, float inv) {
        unsigned q = (unsigned)(((float)m + 0.5f) * inv);
        if (q * d > m) --q;
        else if ((q + 1) * d <= m) ++q;
        return q;
    };
    // the per-pixel operands of a tile's epilogue (residual, ReLU-mask bits), straight into registers in the accumulator layout -- requested ONE TILE
    // AHEAD (two register sets, the tile loop is unrolled by two): the wait for them never includes a round trip started in the same iteration
    auto fetch_pre = [&](int k, u32x4_t (&rres)[TM][H], unsigned (&mb)[TM][H]) {
        const int m0 = (g + k * G) * BM;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int h = 0; h < H; ++h) {
                const unsigned m = (unsigned)(m0 + i * 16 + fr), c = (unsigned)(n0 + h * 32 + fq * 8);
                if (has_res) {
                    unsigned row = m;
                    if (res_up) {
                        const unsigned t_ = fdiv(m, (unsigned)p.Wo, inv_wo), wo = m - t_ * (unsigned)p.Wo, n = fdiv(t_, (unsigned)p.Ho, inv_ho), ho = t_ - n * (unsigned)p.Ho;
                        row = (n * (unsigned)(p.Ho >> 1) + (ho >> 1)) * (unsigned)(p.Wo >> 1) + (wo >> 1);
                    }
                    rres[i][h] = __builtin_amdgcn_raw_buffer_load_b128(rr, m < (unsigned)p.M ? row * C2 + c * 2u : OOB, 0, 0);      // (rows >= M: out of range, zeros)
                }
                if (has_mb) mb[i][h] = __builtin_amdgcn_raw_buffer_load_b8(rmb, m * C8 + (c >> 3), 0, 0);
            }
    };
    auto step = [&](int k, int stage, u32x4_t (&rres)[TM][H], unsigned (&mb)[TM][H], u32x4_t (&rres_n)[TM][H], unsigned (&mb_n)[TM][H]) {
        // every wave is done reading the stage tile k + 2 goes into (tile k - 1's), and its own pieces of tile k have landed (the wait below / the prologue's)
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        const int m0 = (g + k * G) * BM;
        // (1) tile k + 2 into the stage tile k - 1 vacated, (2) the epilogue operands of tile k + 1
        const bool more = k + 2 < nt;
        if (more) issue_tile(k + 2, stage == 0 ? 2 : stage - 1);
        __builtin_amdgcn_sched_barrier(0);
        if (k + 1 < nt) fetch_pre(k + 1, rres_n, mb_n);
        __builtin_amdgcn_sched_barrier(0);
        // (3) the MFMAs of tile k: pixel fragments from the LDS one k-step ahead, weights from registers
        f32x4_t acc[TM][TN];
        const unsigned sb = (unsigned)(stage * STAGE16 * 16);
        u32x4_t xa[TM], xb[TM];
        frag_read_n<TM, 2048, 0>(xa, rd0 + sb);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            u32x4_t* cur = (ks & 1) ? xb : xa;
            u32x4_t* nxt = (ks & 1) ? xa : xb;
            frag_wait1<TM>(cur);
            if (ks + 1 < KS) {
                const unsigned a = (((ks + 1) & 1) ? rd1 : rd0) + sb + (unsigned)(((ks + 1) >> 1) * BM * 128);
                frag_read_n<TM, 2048, 0>(nxt, a);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = Mma<T>::run(wreg[j][ks], cur[i], ks == 0 ? f32x4_t{0.f, 0.f, 0.f, 0.f} : acc[i][j]);
        }
        // (4) tile k + 1 has landed: at most the youngest operations may still be in flight -- this iteration's DMA (P) and, before it, the previous
        // tile's stores (TM * H; vector-memory operations retire in issue order).  (The epilogue operands of tile k + 1, the youngest, are not waited for;
        // those of tile k the compiler's own count covers.)
        if (no_epi) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P + TM * H) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(TM * H) : "memory");
        if (no_epi) return;
#pragma unroll
        for (int h = 0; h < H; ++h) {
            // scales / shifts of this lane's 8 channels (asm reads: a C++ LDS load behind an LDS-DMA in flight drains the DMA queue first)
            u32x4_t sc0 = frag_read<0>(aux_rd + h * 128), sc1 = frag_read<16>(aux_rd + h * 128);
            u32x4_t sh0 = frag_read<BN * 4>(aux_rd + h * 128), sh1 = frag_read<BN * 4 + 16>(aux_rd + h * 128);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            asm volatile("" : "+v"(sc0)); asm volatile("" : "+v"(sc1)); asm volatile("" : "+v"(sh0)); asm volatile("" : "+v"(sh1));
            const unsigned c = (unsigned)(n0 + h * 32 + fq * 8);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const unsigned m = (unsigned)(m0 + i * 16 + fr);
                const unsigned e0 = m * (unsigned)p.Cout + c;
                const bool ok = m < (unsigned)p.M;
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] = acc[i][2 * h][e]; v[4 + e] = acc[i][2 * h + 1][e]; }
                if (has_sc) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[e] *= __uint_as_float(sc0[e]); v[4 + e] *= __uint_as_float(sc1[e]); }
                }
                if (has_sh) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[e] += __uint_as_float(sh0[e]); v[4 + e] += __uint_as_float(sh1[e]); }
                }
                if (has_res) {                      // fp32 add before the single rounding
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const unsigned r2 = rres[i][h][q];
                        v[2 * q] += __uint_as_float(r2 << 16);
                        v[2 * q + 1] += __uint_as_float(r2 & 0xffff0000u);
                    }
                }
                if (has_mb) {
#pragma unroll
                    for (int t = 0; t < 8; ++t) v[t] = __uint_as_float(__float_as_uint(v[t]) & (unsigned)__builtin_amdgcn_sbfe((int)mb[i][h], t, 1));
                }
                uint32_t d[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) d[q] = pack2_bf16(v[2 * q], v[2 * q + 1]);
                if (relu) {                         // bf16 as int16: negative floats (and -0) are negative integers
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        s16x2_t t = *reinterpret_cast<s16x2_t*>(&d[q]);
                        t = __builtin_elementwise_max(t, s16x2_t{0, 0});
                        d[q] = *reinterpret_cast<uint32_t*>(&t);
                    }
                }
                const u32x4_t ov = {d[0], d[1], d[2], d[3]};
                __builtin_amdgcn_raw_buffer_store_b128(ov, ry, ok ? e0 * 2u : OOB, 0, 0);
                if (has_bo) {                       // (y > 0) of the 8 channels = one byte
                    unsigned u = 0;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        u16x2_t t = *reinterpret_cast<u16x2_t*>(&d[q]);
                        if (!relu) { s16x2_t s_ = *reinterpret_cast<s16x2_t*>(&d[q]); s_ = __builtin_elementwise_max(s_, s16x2_t{0, 0}); t = *reinterpret_cast<u16x2_t*>(&s_); }
                        t = __builtin_elementwise_min(t, u16x2_t{1, 1});
                        u |= *reinterpret_cast<unsigned*>(&t) << (2 * q);
                    }
                    const unsigned b = (u & 0x55u) | ((u >> 15) & 0xaau);
                    __builtin_amdgcn_raw_buffer_store_b8((unsigned char)b, rbo, ok ? e0 >> 3 : OOB, 0, 0);
                }
            }
        }
    };

    u32x4_t resA[TM][H], resB[TM][H];
    unsigned mbA[TM][H], mbB[TM][H];
    if (nt > 0) issue_tile(0, 0);
    if (nt > 1) issue_tile(1, 1);
    if (nt > 0) fetch_pre(0, resA, mbA);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int stage = 0;
    for (int k = 0; k < nt; k += 2) {
        step(k, stage, resA, mbA, resB, mbB);
        stage = stage == NBUF - 1 ? 0 : stage + 1;
        if (k + 1 < nt) {
            step(k + 1, stage, resB, mbB, resA, mbA);
            stage = stage == NBUF - 1 ? 0 : stage + 1;
        }
    }
}

// eligible: bf16, 1x1 / stride 1 / no padding, plain output layout, K = Cin in {64, 128, 256, 512}, whole channel groups, no full-tensor mask / fp32
// output / split-K
inline int ws_channels(int K) { return K == 64 || K == 128 ? 256 : K == 256 || K == 512 ? 128 : 0; }      // BN of the instantiation for this K
inline bool ws_ok(const ConvDev& d) {
    const int bn = ws_channels(d.K);
    return d.KH * d.KW == 1 && d.stride == 1 && d.pad == 0 && d.K == d.Cin && bn && d.Cout % bn == 0 && d.y && !d.y_f32 && !d.mask && d.out_scale == 1 &&
           d.ksplit <= 1 && (long)d.M * d.Cout * 2 < (1L << 31);
}
int launch_ws(const ConvDev& d, hipStream_t st, int wgs) {
    const int bn = ws_channels(d.K), ncg = d.Cout / bn;
    const int bm = d.K <= 256 ? 32 : 16;
    const int n_mtiles = cdiv(d.M, bm);
    // whole sets of 8 x ncg workgroups (one per XCD and channel group); no more pixel-tile sequences than pixel tiles
    int sets = wgs / (8 * ncg);
    if (sets < 1) sets = 1;
    if (sets > cdiv(n_mtiles, 8)) sets = cdiv(n_mtiles, 8);
    const dim3 grid(sets * 8 * ncg), block(256);
    if (d.K == 64) hipLaunchKernelGGL((igemm_ws_kernel<2, 2, 4>), grid, block, 0, st, d, n_mtiles, ncg);
    else if (d.K == 128) hipLaunchKernelGGL((igemm_ws_kernel<4, 2, 4>), grid, block, 0, st, d, n_mtiles, ncg);
    else if (d.K == 256) hipLaunchKernelGGL((igemm_ws_kernel<8, 2, 2>), grid, block, 0, st, d, n_mtiles, ncg);
    else hipLaunchKernelGGL((igemm_ws_kernel<16, 1, 2>), grid, block, 0, st, d, n_mtiles, ncg);
    ALDI_CHECK_LAUNCH();
    char name[112];
    snprintf(name, sizeof(name), "igemm_ws<bf16,%d,%d,k%d>", bm, bn, d.K);
    aldi_note_dispatch(name);
    return ALDI_OK;
}
