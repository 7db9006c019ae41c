// 3x3 / stride 1 / pad 1 convolution, bf16, "halo" form with 128-BYTE K slabs: the p2-size layers (FPN output convs, the RPN conv, their data
// gradients: 52 % of the step's igemm FLOPs).  Included by igemm.hip (ConvDev, igemm_epilogue, the tile primitives).
//
// Why a second halo kernel.  tools/probes/dma_rate_probe.hip: the L2 -> LDS path (LDS-DMA) is paced per 128-byte cache LINE touched, not per byte --
// L2-resident rows fetched as 64-byte segments (the 32-channel K slabs of igemm_body's HALO branch: four lanes per pixel row = half a line) fill the
// LDS at 16.5 TB/s over the chip, as 128-byte segments at 27 TB/s.  The 64 x 64-per-wave halo tiles move 33-42 KB per 144 MFMAs and workgroup: at the
// half-line rate that is 1250-1560 cycles of the path against 770-1540 cycles of MFMA work -- the K loop was bound by it (profiles/r03_halo_ablation.txt:
// "the costs add rather than overlap").  Here:
//   * a K group is (kernel row kh, 64 input channels): every DMA lane group of 8 fetches one full line; per group the workgroup loads ONE halo slab of
//     BM + 2 pixel rows (it serves the three horizontal taps, rows +0 / +1 / +2) and three 256-channel weight taps;
//   * the tile is 256 pixels x 256 channels (8 waves as 4 x 2, 64 pixels x 128 channels per wave): 129 KB of operands per 1536 MFMAs of the workgroup,
//     half the bytes per flop of the 256 x 128 tile and 0.375 fragment reads per MFMA instead of 0.5;
//   * LDS rings at PIECE granularity -- the slab double-buffered per group (2 x 33 KB), the weight taps double-buffered per TAP (2 x 32 KB): a whole
//     group per stage (81 KB) would not fit twice.  DMA never drains inside the loop: one counted `vmcnt` + one `s_barrier` per tap (64 MFMAs per
//     wave); a tap's weights are requested a full tap ahead, a group's slab a full group ahead;
//   * the swizzle of the 128-byte rows is keyed on row bits 1-2 (chunk ^ (row & 6)): conflict-free for the ds_read_b128 lane groups at EVERY row
//     offset, which the +1 / +2 tap reads of one slab need (the search is in DESIGN.md; `(row >> 1) & 7` has 2-way conflicts at odd offsets);
//   * fragments are double-buffered in registers at 16-MFMA granularity (sub-phase = one 32-channel k-step x one 64-channel half of the wave's
//     channels): the reads of sub-phase s + 1 and the tap's DMA pieces are issued in front of the MFMAs of sub-phase s.
// K order per output element: (64-channel chunk, kh, kw, 32-channel half) -- not the (kh, 32-channel chunk, kw) of the other halo tiles: results agree to
// fp32 summation order, not bit for bit.
#pragma once

template <int N, int STRIDE, int BASE, int I = 0>
__device__ __forceinline__ void frag_read_n(u32x4_t* f, unsigned addr) {      // fragment I at addr + BASE + I * STRIDE bytes
    if constexpr (I < N) {
        f[I] = frag_read<BASE + I * STRIDE>(addr);
        frag_read_n<N, STRIDE, BASE, I + 1>(f, addr);
    }
}
template <int N>
__device__ __forceinline__ void frag_wait1(u32x4_t* a) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("" : "+v"(a[i]));
}

template <int V> struct KwTag { static constexpr int value = V; };

// ---- the INTERLEAVED K loop (r06).  tools/probes/mfma_issue_probe.hip: ONE wave feeds 0.98 of a SIMD's matrix pipe when its fragment reads sit BETWEEN
// its MFMAs (16.3 clocks per 16x16x32 MFMA against a 16-clock pipe), and an LDS-DMA piece costs ~60 clocks of issue among MFMAs -- but the lockstep
// loop below issues a sub-phase's reads and DMA pieces IN FRONT of its 16 MFMAs, in both waves of a SIMD at the same time: ~270 clocks per sub-phase
// with an idle pipe (3 120 clocks per tap against 2 048 of MFMA work; profiles/r05_halo_ablation.txt).  r05 tried to move them behind the first MFMAs
// with the builtin and got accumulator copies + spills (hipcc re-allocates the results of `__builtin_amdgcn_mfma_*` when other instructions sit
// between them -- the same pathology as its own MFMA probe loop: v_accvgpr_mov chains).  Here every MFMA is an `asm volatile` with the accumulator
// tied in place ("+v"), the reads already were: volatile statements keep their source order, so the stream is what is written -- MFMA, read, MFMA,
// read, ..., MFMA, DMA piece, MFMA ... -- and nothing is copied.  The compiler's hazard recogniser does not see MFMAs inside asm: no accumulator is
// touched twice within 16 MFMAs (the block order guarantees it), and the loop is followed by 32 wait states before the epilogue reads them.
__device__ __forceinline__ void mma_ip(f32x4_t& c, const u32x4_t& a, const u32x4_t& b) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
template <int NM, int I = 0, typename FM, typename FF>
__device__ __forceinline__ void interleave(FM&& mma, FF&& fill) {             // MFMA 0, filler 0, MFMA 1, filler 1, ...
    if constexpr (I < NM) {
        mma(KwTag<I>{});
        fill(KwTag<I>{});
        interleave<NM, I + 1>(mma, fill);
    }
}

// DIRECT: the epilogue stores straight from the accumulators (bf16, plain layout, scale / shift / ReLU only): the weight rows are fetched in
// `direct_perm` order so that a lane's fragments 2h, 2h + 1 are 8 consecutive channels of its pixel (one 16-byte store); no LDS staging, no
// barrier, and nothing waits for the stores -- the workgroup ends (the next one's prologue runs) while they drain.  The staged epilogue of a
// 256 x 256 tile is 10 us per tile with one workgroup per CU and nothing beside it (16 % of the kernel).
// (A third instantiation, 256 x 256 on FOUR waves of 128 pixels x 128 channels each -- the accumulators fill the AGPR half of a 512-register
// budget, one wave per SIMD, 0.25 fragment reads per MFMA instead of 0.375 -- is igemm_force 15 / igemm_bigtile 65: 10-12 % SLOWER, one wave per
// SIMD issues an MFMA every 27 clocks, two every 17.5 (profiles/r05_mfma_clock_probe.txt))
// Two tile shapes run this body: 256 x 256 on 8 waves (64 pixels x 128 channels per wave: four 16-MFMA sub-phases per tap) for the p2-size layers,
// and 128 x 128 on 4 waves (64 x 64 per wave: two sub-phases per tap, 68 KB of LDS: two workgroups per CU) for the mid-size ones (res3 / res4
// conv2 and their data gradients), whose 128 x 64 tiles with 32-channel slabs spent half their loop on the L2 -> LDS path.  Both have NT / 64
// threads per 16-byte slot of a tap and of the slab, i.e. the same piece schedule (4 + 1 slab pieces, 4 tap pieces per thread).
template <int BM, int BN, int WM, int WN, bool DIRECT, bool ILV = false>
__device__ __forceinline__ void igemm_halo64_body(const ConvDev& p, int bid, const int nmt, const int nnt) {
    typedef bf16_t T;
    constexpr int NT = WM * WN * 64, KC = 8, EP = 8, BK = 64;
    constexpr int TM = BM / WM / 16;                       // 4 pixel fragments per wave
    constexpr int TN = BN / WN / 16;                       // 8 or 4 channel fragments per wave
    constexpr int CS = TN / 4;                             // channel blocks of 64 per wave = sub-phases per k-step
    constexpr int TH = 4;                                  // channel fragments per sub-phase
    static_assert((TM == 4 || TM == 8) && (CS == 1 || CS == 2), "64 / 128 pixels x 64 / 128 channels per wave");
    constexpr int XS = ((BM + 2) * KC + 63) / 64 * 64;     // halo slab, 16-B slots, padded to whole waves of DMA (2112)
    constexpr int WS = BN * KC;                            // one tap of weights (2048)
    constexpr int X_IT = (XS + NT - 1) / NT;               // DMA pieces per thread: slab 5 (the fifth: wave 0 only), tap 4
    constexpr int W_IT = WS / NT;
    constexpr int XQ = (X_IT - 1) / 4, WQ = W_IT / 4;      // pieces per thread and QUARTER of a slab / a tap: the schedule below issues quarters (1: 8 waves x 256 x 256 and 4 waves x 128 x 128; 2: 4 waves x 256 x 256)
    static_assert((X_IT - 1) % 4 == 0 && W_IT % 4 == 0 && WS % NT == 0 && XS - (X_IT - 1) * NT <= 64, "piece schedule below: four quarters + one tail piece (wave 0) per slab, four quarters per tap");
    constexpr int RING = 2 * XS + 2 * WS;
    constexpr int EPI_SLOTS = BM * (BN * 2 + 16) / 16;     // the staged epilogue's tile
    constexpr int LDS_SLOTS = RING + 8 > EPI_SLOTS ? RING + 8 : EPI_SLOTS;
    constexpr int FR = 16 * KC * 16;                       // bytes between fragments 16 rows apart (2048)
    __shared__ __attribute__((aligned(128))) uint4 lds_all[LDS_SLOTS];          // [slab 0 | slab 1 | tap 0 | tap 1 | one all-zero 128-B row]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    if (p.xcd) {
        const int total = nmt * nnt, q = total >> 3, r = total & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int m0 = (bid / nnt) * BM, n0 = (bid % nnt) * BN;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(static_cast<const T*>(p.x)), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(static_cast<const T*>(p.w)), 0, p.w_bytes, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    const int wbase = __builtin_amdgcn_readfirstlane(tid & ~63);
    const bool tail = wbase + (X_IT - 1) * NT < XS;        // this wave owns the slab's last, partial piece (wave 0)
    const int fr = lane & 15, fq = lane >> 4;
    // raised priority around every MFMA block (+3-4 %; igemm_dbg 128 turns it off for A/B runs).  ASYMMETRIC since r06 (igemm_dbg 256 = the old
    // symmetric level 1 for A/B runs; 512 = the halves swapped): waves w and w + NT / 128 share a SIMD; with equal priority their MFMA blocks interleave
    // instruction by instruction, both finish together and both then issue their fragment reads / DMA while the matrix pipe idles (r05: 28 % of a tap).
    // With the first half at level 2 its block runs alone and the second half's block fills the pipe while the first half reads: the two waves of a
    // SIMD drift half a sub-phase apart after every tap barrier (tools/probes/mfma_issue_probe.hip: one wave alone feeds 0.98 of the pipe).
    const int plev = __builtin_amdgcn_readfirstlane((p.dbg & 128) ? 0 : (p.dbg & 256) ? 1 : ((wave < WM * WN / 2) != ((p.dbg & 512) != 0)) ? 2 : 1);
    const bool prio = plev != 0, prio_hi = plev == 2;
    // ILV: ALL four pieces of tap u + 2 are requested behind the barrier of tap u (a full tap ahead; the lockstep loop spreads them over three sub-phases
    // to keep its DMA issue bursts short: pieces 2, 3 go out only 1-2 sub-phases before the barrier that waits for them).  igemm_dbg 1024 = the old points
    const bool early_w = ILV && __builtin_amdgcn_readfirstlane(p.dbg & 1024) == 0;
    const bool no_dma = p.dbg & 32, no_mfma = p.dbg & 64;  // ablation (igemm_dbg): 32 = no DMA inside the K loop, 64 = no MFMAs, 4 = no epilogue

    f32x4_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if (tid < 8) lds_all[RING + tid] = make_uint4(0u, 0u, 0u, 0u);               // (visible behind the prologue's barrier)

    // ---- DMA sources.  Slot c of a piece = LDS row c >> 3, 16-B chunk c & 7; the lane fetches logical chunk (c & 7) ^ (row & 6).
    unsigned xa_voff[X_IT], xa_ok = 0;                      // centre-row (kh = 1) byte offsets; validity: 3 bits (kh) per piece
#pragma unroll
    for (int it = 0; it < X_IT; ++it) {
        const int c = tid + it * NT, row = c >> 3, kce = (c & 7) ^ (row & 6);
        const int q0 = m0 - 1 + row;
        const bool ok = row < BM + 2 && q0 >= 0 && q0 < p.M;
        const int qq = ok ? q0 : 0;
        const int h0 = (qq / p.W) % p.H;
        xa_voff[it] = ((unsigned)qq * (unsigned)p.Cin + (unsigned)(kce * EP)) * 2u;
        if (ok) xa_ok |= ((h0 >= 1 ? 1u : 0u) | 2u | (h0 <= p.H - 2 ? 4u : 0u)) << (3 * it);
    }
    unsigned wa_voff[W_IT];
#pragma unroll
    for (int it = 0; it < W_IT; ++it) {
        const int c = tid + it * NT, row = c >> 3, kce = (c & 7) ^ (row & 6);
        const int co = n0 + (DIRECT ? direct_perm<BN / WN>(row) : row);      // (direct_perm: per wave block of BN / WN rows)
        wa_voff[it] = co < p.Cout ? ((unsigned)co * (unsigned)p.K + (unsigned)(kce * EP)) * 2u : OOB;
    }
    // (kh, first channel) of the slab / (kh, first channel, kw) of the tap that is issued NEXT (wave-uniform scalars)
    int xkh = 0, xci = 0, tkh = 0, tci = 0, tkw = 0;
    auto issue_x = [&](int it, int st) {                   // piece `it` of the slab of group (xkh, xci) into slab stage st
        if (no_dma) return;
        const unsigned a_off = (unsigned)(((xkh - 1) * p.W * p.Cin + xci) * 2);
        glds16(rx, &lds_all[st * XS + wbase + it * NT], ((xa_ok >> (3 * it + xkh)) & 1u) ? xa_voff[it] + a_off : OOB);
    };
    // K order of the groups: (64-channel chunk, kh) with kh INNER since r06 (r05: kh outer).  All workgroups of a launch walk the groups in step; a
    // pixel's chunk is fetched by three tiles (the one that holds it and the ones an image row above / below) in their kh = 0 / 1 / 2 groups.  With kh
    // outer those are a THIRD OF A TILE apart in time (~15 us) and an XCD streams 4 MB of slabs in between -- its whole L2: every slab read missed, the
    // launch fetched 2.5 x its input from the fabric (profiles/r05_pmc_conv.txt: FETCH_SIZE 173 MB x 2 against 137 MB).  With kh inner the three reads
    // are one group (~4 us, 1 MB per XCD) apart: 97 MB x 2 (profiles/r06_pmc_conv.txt; HBM-side bytes of the launch 1.79x -> 1.19x algorithmic).  The
    // TIME of the launch did not move (287 vs 288 us alone, step -0.3 %: profiles/r06_halo_khorder_clean.txt): the misses were already covered by the
    // group-ahead prefetch.  (A run-time switch between the two orders cost 4 more SGPRs than the kernel has: 56 bytes of scratch and a quarter of
    // its speed -- both arms of that first A/B were slowed, and read as -8 %; the order is a compile-time fact.)
    auto next_x = [&]() { if (++xkh == 3) { xkh = 0; xci += BK; } };
    auto issue_w = [&](int it, int st, unsigned w_off) {   // piece `it` of a tap into tap stage st
        if (no_dma) return;
        glds16(rw, &lds_all[2 * XS + st * WS + wbase + it * NT], wa_voff[it] == OOB ? OOB : wa_voff[it] + w_off);
    };
    auto tap_off = [&]() { return (unsigned)(((tkh * 3 + tkw) * p.Cin + tci) * 2); };
    auto next_tap = [&]() { if (++tkw == 3) { tkw = 0; if (++tkh == 3) { tkh = 0; tci += BK; } } };

    // ---- fragment read addresses (stage 0): X rows xrow + kw (the three taps), k-step h = 0 / 1; W rows wrow
    const int xrow = wm * (BM / WM) + fr, wrow = wn * (BN / WN) + fr;
    const unsigned lbase = lds_addr(&lds_all[0]);
    unsigned x_rd[3][2], w_rd[2];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
        for (int h = 0; h < 2; ++h) x_rd[kw][h] = lbase + (unsigned)((xrow + kw) * KC + ((h * 4 + fq) ^ ((xrow + kw) & 6))) * 16u;
#pragma unroll
    for (int h = 0; h < 2; ++h) w_rd[h] = lbase + (unsigned)(2 * XS + wrow * KC + ((h * 4 + fq) ^ (wrow & 6))) * 16u;
    // left / right image border: the tap reads the all-zero row instead of the neighbouring image row's pixel
    bool edge_l[TM], edge_r[TM];
    unsigned zaddr[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + wm * (BM / WM) + i * 16 + fr;
        const int wo = m % p.W;
        edge_l[i] = wo == 0;
        edge_r[i] = wo == p.W - 1;
        zaddr[i] = lbase + (unsigned)(RING * 16) + (unsigned)fq * 16u - (unsigned)(i * FR);     // frag_read_each adds i * 16 rows back
    }
    auto read_x = [&](u32x4_t* xf, auto KW, auto H, unsigned xoff) {
        constexpr int kw = decltype(KW)::value, h = decltype(H)::value;
        if constexpr (kw == 1) {
            frag_read_n<TM, FR, 0>(xf, x_rd[1][h] + xoff);
        } else {
            const unsigned a = x_rd[kw][h] + xoff;
            unsigned xa[TM];
#pragma unroll
            for (int i = 0; i < TM; ++i) xa[i] = (kw == 0 ? edge_l[i] : edge_r[i]) ? zaddr[i] : a;
            frag_read_each<TM, KC * 16>(xf, xa);
        }
    };

    const int ngroups = 3 * (p.Cin / BK), U = 3 * ngroups;
    // ---- prologue: slab 0, tap 0, the first half of tap 1
#pragma unroll
    for (int it = 0; it < X_IT - 1; ++it) issue_x(it, 0);
    if (tail) issue_x(X_IT - 1, 0);
    next_x();
    {
        const unsigned o = tap_off();
#pragma unroll
        for (int it = 0; it < W_IT; ++it) issue_w(it, 0, o);
        next_tap();
    }
    unsigned woff_a = tap_off();                            // byte offset of the tap whose pieces 2, 3 go out at points a, b of the current tap
#pragma unroll
    for (int q = 0; q < 2 * WQ; ++q) issue_w(q, 1, woff_a);
    if (early_w) {
#pragma unroll
        for (int q = 2 * WQ; q < 4 * WQ; ++q) issue_w(q, 1, woff_a);
    }
    next_tap();
    __builtin_amdgcn_sched_barrier(0);
    // slab 0 and tap 0 have landed (mine); the barrier makes everyone's visible
    if (early_w) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * WQ) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * WQ) : "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    u32x4_t x0[TM], x1[TM], wa[TH], wb[TH];
    read_x(x0, KwTag<0>{}, KwTag<0>{}, 0u);
    frag_read_n<TH, FR, 0>(wa, w_rd[0]);
    frag_wait<TM, TH>(x0, wa);
    __builtin_amdgcn_sched_barrier(0);

    unsigned xoff = 0, woff = 0;                            // LDS byte offsets of the slab / tap stage being READ
    constexpr unsigned XSB = XS * 16, WSB = WS * 16;
    // one tap (kw a compile-time constant: the border selects of read_x and the slab schedule fold; no hand-issued read sits in a branch)
    auto tap4 = [&](auto KW, const int g, const int u) {
        constexpr int kw = decltype(KW)::value;
        const bool more1 = u + 1 < U, more2 = u + 2 < U;
        // ---- sub-phase 0: (h 0, channels 0-63) from x0 / wa; reads of (h 0, channels 64-127); DMA point a
        frag_read_n<TH, FR, TH * FR>(wb, w_rd[0] + woff);
        if (more1) {
#pragma unroll
            for (int q = 0; q < WQ; ++q) issue_w(2 * WQ + q, (u + 1) & 1, woff_a);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!no_mfma) {
            if (prio) { if (prio_hi) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(1); }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TH; ++j) acc[i][j] = Mma<T>::run(wa[j], x0[i], acc[i][j]);
            if (prio) __builtin_amdgcn_s_setprio(0);
        }
        frag_wait1<TH>(wb);
        __builtin_amdgcn_sched_barrier(0);
        // ---- sub-phase 1: (h 0, channels 64-127) from x0 / wb; reads of (h 1, channels 0-63); DMA point b
        read_x(x1, KW, KwTag<1>{}, xoff);
        frag_read_n<TH, FR, 0>(wa, w_rd[1] + woff);
        if (more1) {
#pragma unroll
            for (int q = 0; q < WQ; ++q) issue_w(3 * WQ + q, (u + 1) & 1, woff_a);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!no_mfma) {
            if (prio) { if (prio_hi) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(1); }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TH; ++j) acc[i][TH + j] = Mma<T>::run(wb[j], x0[i], acc[i][TH + j]);
            if (prio) __builtin_amdgcn_s_setprio(0);
        }
        frag_wait<TM, TH>(x1, wa);
        __builtin_amdgcn_sched_barrier(0);
        // ---- sub-phase 2: (h 1, channels 0-63) from x1 / wa; reads of (h 1, channels 64-127); DMA point c: the NEXT group's slab
        frag_read_n<TH, FR, TH * FR>(wb, w_rd[1] + woff);
        int nx = 0;                                         // slab pieces issued at c (younger than everything the tap barrier waits for)
        if (kw < 2 && g + 1 < ngroups) {
            const int st = (g + 1) & 1;
            if constexpr (kw == 0) {
#pragma unroll
                for (int q = 0; q < 2 * XQ; ++q) issue_x(q, st);
                nx = 2 * XQ;
            }
            if constexpr (kw == 1) {
#pragma unroll
                for (int q = 0; q < 2 * XQ; ++q) issue_x(2 * XQ + q, st);
                nx = 2 * XQ;
                if (tail) { issue_x(X_IT - 1, st); nx = 2 * XQ + 1; }
                next_x();
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!no_mfma) {
            if (prio) { if (prio_hi) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(1); }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TH; ++j) acc[i][j] = Mma<T>::run(wa[j], x1[i], acc[i][j]);
            if (prio) __builtin_amdgcn_s_setprio(0);
        }
        frag_wait1<TH>(wb);
        __builtin_amdgcn_sched_barrier(0);
        // ---- tap barrier: tap u + 1 (and, behind kw = 2, the next slab) has landed; every read of tap u is retired (its last fragments are in
        // registers), so the tap's stage may be overwritten
        if (no_dma || nx == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (nx == 2 * XQ) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * XQ) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * XQ + 1) : "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ---- sub-phase 3: (h 1, channels 64-127) from x1 / wb; reads of tap u + 1's first sub-phase; DMA point d: pieces 0, 1 of tap u + 2
        constexpr int kw_n = kw == 2 ? 0 : kw + 1;
        const unsigned xoff_n = kw == 2 ? xoff ^ XSB : xoff, woff_n = woff ^ WSB;
        // (the reads are unconditional: behind the last tap they fetch a stage nobody uses -- see igemm_body on conditional hand-issued reads)
        read_x(x0, KwTag<kw_n>{}, KwTag<0>{}, xoff_n);
        frag_read_n<TH, FR, 0>(wa, w_rd[0] + woff_n);
        if (more2) {
            woff_a = tap_off();
#pragma unroll
            for (int q = 0; q < 2 * WQ; ++q) issue_w(q, u & 1, woff_a);
            next_tap();
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!no_mfma) {
            if (prio) { if (prio_hi) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(1); }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TH; ++j) acc[i][TH + j] = Mma<T>::run(wb[j], x1[i], acc[i][TH + j]);
            if (prio) __builtin_amdgcn_s_setprio(0);
        }
        frag_wait<TM, TH>(x0, wa);
        __builtin_amdgcn_sched_barrier(0);
        xoff = xoff_n; woff = woff_n;
    };
    // ---- the same tap with the reads / DMA pieces of every sub-phase BETWEEN its MFMAs (ILV; see mma_ip above).  Same DMA order, same counted waits,
    // same accumulation order per output element as tap4: bit-identical results.
    auto x_addr = [&](unsigned* xa, auto KW, auto H, unsigned xo) {
        constexpr int kw = decltype(KW)::value, h = decltype(H)::value;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            if constexpr (kw == 1) xa[i] = x_rd[1][h] + xo;
            else xa[i] = (kw == 0 ? edge_l[i] : edge_r[i]) ? zaddr[i] : x_rd[kw][h] + xo;
        }
    };
    auto tap4i = [&](auto KW, const int g, const int u) {
        constexpr int kw = decltype(KW)::value;
        constexpr int NM = TM * TH;
        static_assert(TM + TH + 1 + 2 * 2 * WQ <= NM && TH + 1 + 2 * (2 * XQ + 1) <= NM && TM + TH + 2 * 2 * WQ < NM, "the fillers of a sub-phase fit between its MFMAs");
        const bool more1 = u + 1 < U, more2 = u + 2 < U;
        // ---- sub-phase 0: (h 0, channels 0-63) from x0 / wa; reads of (h 0, channels 64-127); DMA point a
        {
            const unsigned wr = w_rd[0] + woff;
            if (prio) { if (prio_hi) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(1); }
            interleave<NM>([&](auto M) { constexpr int m = decltype(M)::value; mma_ip(acc[m / TH][m % TH], wa[m % TH], x0[m / TH]); },
                           [&](auto M) {
                               constexpr int m = decltype(M)::value;
                               if constexpr (m < TH) wb[m] = frag_read<(TH + m) * FR>(wr);
                               if constexpr (m == TH + 1) {
                                   if (more1 && !early_w) {
#pragma unroll
                                       for (int q = 0; q < WQ; ++q) issue_w(2 * WQ + q, (u + 1) & 1, woff_a);
                                   }
                               }
                           });
            if (prio) __builtin_amdgcn_s_setprio(0);
            frag_wait1<TH>(wb);
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- sub-phase 1: (h 0, channels 64-127) from x0 / wb; reads of (h 1, channels 0-63); DMA point b
        {
            unsigned xa[TM];
            x_addr(xa, KW, KwTag<1>{}, xoff);
            const unsigned wr = w_rd[1] + woff;
            if (prio) { if (prio_hi) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(1); }
            interleave<NM>([&](auto M) { constexpr int m = decltype(M)::value; mma_ip(acc[m / TH][TH + m % TH], wb[m % TH], x0[m / TH]); },
                           [&](auto M) {
                               constexpr int m = decltype(M)::value;
                               if constexpr (m < TM) x1[m] = frag_read<m * FR>(xa[m]);
                               else if constexpr (m < TM + TH) wa[m - TM] = frag_read<(m - TM) * FR>(wr);
                               if constexpr (m == TM + TH + 1) {
                                   if (more1 && !early_w) {
#pragma unroll
                                       for (int q = 0; q < WQ; ++q) issue_w(3 * WQ + q, (u + 1) & 1, woff_a);
                                   }
                               }
                           });
            if (prio) __builtin_amdgcn_s_setprio(0);
            frag_wait<TM, TH>(x1, wa);
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- sub-phase 2: (h 1, channels 0-63) from x1 / wa; reads of (h 1, channels 64-127); DMA point c: the NEXT group's slab
        const bool slab = kw < 2 && g + 1 < ngroups;
        const int sst = (g + 1) & 1;
        int nx = 0;
        if (slab) nx = (kw == 1 && tail) ? 2 * XQ + 1 : 2 * XQ;
        {
            const unsigned wr = w_rd[1] + woff;
            if (prio) { if (prio_hi) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(1); }
            interleave<NM>([&](auto M) { constexpr int m = decltype(M)::value; mma_ip(acc[m / TH][m % TH], wa[m % TH], x1[m / TH]); },
                           [&](auto M) {
                               constexpr int m = decltype(M)::value;
                               if constexpr (m < TH) wb[m] = frag_read<(TH + m) * FR>(wr);
                               if constexpr (kw < 2 && m > TH && (m - TH) % 2 == 1 && (m - TH) / 2 <= 2 * XQ) {
                                   constexpr int q = (m - TH) / 2;                 // slab piece q of this tap's half (the last slot: the tail piece)
                                   if (slab) {
                                       if constexpr (q < 2 * XQ) issue_x(kw * 2 * XQ + q, sst);
                                       else if constexpr (kw == 1) { if (tail) issue_x(X_IT - 1, sst); }
                                   }
                               }
                           });
            if (prio) __builtin_amdgcn_s_setprio(0);
            if constexpr (kw == 1) { if (slab) next_x(); }
            frag_wait1<TH>(wb);
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- tap barrier (as in tap4)
        if (no_dma || nx == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (nx == 2 * XQ) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * XQ) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * XQ + 1) : "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ---- sub-phase 3: (h 1, channels 64-127) from x1 / wb; reads of tap u + 1's first sub-phase; DMA point d: pieces 0, 1 of tap u + 2
        {
            constexpr int kw_n = kw == 2 ? 0 : kw + 1;
            const unsigned xoff_n = kw == 2 ? xoff ^ XSB : xoff, woff_n = woff ^ WSB;
            unsigned xa[TM];
            x_addr(xa, KwTag<kw_n>{}, KwTag<0>{}, xoff_n);
            const unsigned wr = w_rd[0] + woff_n;
            if (more2) woff_a = tap_off();
            if (prio) { if (prio_hi) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(1); }
            interleave<NM>([&](auto M) { constexpr int m = decltype(M)::value; mma_ip(acc[m / TH][TH + m % TH], wb[m % TH], x1[m / TH]); },
                           [&](auto M) {
                               constexpr int m = decltype(M)::value;
                               if constexpr (m < TM) x0[m] = frag_read<m * FR>(xa[m]);
                               else if constexpr (m < TM + TH) wa[m - TM] = frag_read<(m - TM) * FR>(wr);
                               if constexpr (m > TM + TH && (m - TM - TH) % 2 == 1 && (m - TM - TH) / 2 < 2 * WQ) {
                                   if (more2) issue_w((m - TM - TH) / 2, u & 1, woff_a);
                               }
                               if constexpr (m > TM + TH && (m - TM - TH) % 2 == 0 && (m - TM - TH) / 2 - 1 < 2 * WQ) {
                                   if (more2 && early_w) issue_w(2 * WQ + (m - TM - TH) / 2 - 1, u & 1, woff_a);
                               }
                           });
            if (prio) __builtin_amdgcn_s_setprio(0);
            if (more2) next_tap();
            frag_wait<TM, TH>(x0, wa);
            __builtin_amdgcn_sched_barrier(0);
            xoff = xoff_n; woff = woff_n;
        }
    };
    // the same for 64 channels per wave (CS = 1): two sub-phases per tap, one per 32-channel k-step
    auto tap2 = [&](auto KW, const int g, const int u) {
        constexpr int kw = decltype(KW)::value;
        const bool more1 = u + 1 < U, more2 = u + 2 < U;
        // ---- sub-phase 0: k-step 0 from x0 / wa; reads of k-step 1; DMA points a, b (tap u + 1's pieces 2, 3) and c (the next group's slab)
        read_x(x1, KW, KwTag<1>{}, xoff);
        frag_read_n<TH, FR, 0>(wb, w_rd[1] + woff);
        if (more1) {
#pragma unroll
            for (int q = 0; q < 2 * WQ; ++q) issue_w(2 * WQ + q, (u + 1) & 1, woff_a);
        }
        int nx = 0;
        if (kw < 2 && g + 1 < ngroups) {
            const int st = (g + 1) & 1;
            if constexpr (kw == 0) {
#pragma unroll
                for (int q = 0; q < 2 * XQ; ++q) issue_x(q, st);
                nx = 2 * XQ;
            }
            if constexpr (kw == 1) {
#pragma unroll
                for (int q = 0; q < 2 * XQ; ++q) issue_x(2 * XQ + q, st);
                nx = 2 * XQ;
                if (tail) { issue_x(X_IT - 1, st); nx = 2 * XQ + 1; }
                next_x();
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!no_mfma) {
            if (prio) { if (prio_hi) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(1); }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TH; ++j) acc[i][j] = Mma<T>::run(wa[j], x0[i], acc[i][j]);
            if (prio) __builtin_amdgcn_s_setprio(0);
        }
        frag_wait<TM, TH>(x1, wb);
        __builtin_amdgcn_sched_barrier(0);
        if (no_dma || nx == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (nx == 2 * XQ) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * XQ) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * XQ + 1) : "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ---- sub-phase 1: k-step 1 from x1 / wb; reads of tap u + 1's k-step 0; DMA point d
        constexpr int kw_n = kw == 2 ? 0 : kw + 1;
        const unsigned xoff_n = kw == 2 ? xoff ^ XSB : xoff, woff_n = woff ^ WSB;
        read_x(x0, KwTag<kw_n>{}, KwTag<0>{}, xoff_n);
        frag_read_n<TH, FR, 0>(wa, w_rd[0] + woff_n);
        if (more2) {
            woff_a = tap_off();
#pragma unroll
            for (int q = 0; q < 2 * WQ; ++q) issue_w(q, u & 1, woff_a);
            next_tap();
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!no_mfma) {
            if (prio) { if (prio_hi) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(1); }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TH; ++j) acc[i][j] = Mma<T>::run(wb[j], x1[i], acc[i][j]);
            if (prio) __builtin_amdgcn_s_setprio(0);
        }
        frag_wait<TM, TH>(x0, wa);
        __builtin_amdgcn_sched_barrier(0);
        xoff = xoff_n; woff = woff_n;
    };
    auto tap = [&](auto KW, const int g, const int u) {
        if constexpr (CS == 2 && ILV) tap4i(KW, g, u);
        else if constexpr (CS == 2) tap4(KW, g, u);
        else tap2(KW, g, u);
    };
    for (int g = 0; g < ngroups; ++g) {
        tap(KwTag<0>{}, g, 3 * g);
        tap(KwTag<1>{}, g, 3 * g + 1);
        tap(KwTag<2>{}, g, 3 * g + 2);
    }
    if constexpr (ILV) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");       // the last asm MFMAs have written their accumulators before anything reads them
    if constexpr (DIRECT) {
        if (p.dbg & 4) return;
        // per-channel operands of this lane's 8 channels of every 32-channel block (the fragment registers are dead: room for them)
        constexpr int H = TN / 2;
        float4 sc[H][2], sh[H][2];
        const bool has_sc = p.scale != nullptr, has_sh = p.shift != nullptr;
#pragma unroll
        for (int h = 0; h < H; ++h) {
            const int c = n0 + wn * (BN / WN) + h * 32 + fq * 8;
            const bool okc = c < p.Cout;
            sc[h][0] = sc[h][1] = make_float4(1.f, 1.f, 1.f, 1.f);
            sh[h][0] = sh[h][1] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (has_sc && okc) { sc[h][0] = *reinterpret_cast<const float4*>(p.scale + c); sc[h][1] = *reinterpret_cast<const float4*>(p.scale + c + 4); }
            if (has_sh && okc) { sh[h][0] = *reinterpret_cast<const float4*>(p.shift + c); sh[h][1] = *reinterpret_cast<const float4*>(p.shift + c + 4); }
        }
        const __amdgpu_buffer_rsrc_t ry = make_rsrc_uniform(p.y, 0x7fffffffu);
        typedef short s16x2_t __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = m0 + wm * (BM / WM) + i * 16 + fr;
#pragma unroll
            for (int h = 0; h < H; ++h) {
                const int c = n0 + wn * (BN / WN) + h * 32 + fq * 8;
                const bool ok = m < p.M && c < p.Cout;
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] = acc[i][2 * h][e]; v[4 + e] = acc[i][2 * h + 1][e]; }
                if (has_sc) {
                    v[0] *= sc[h][0].x; v[1] *= sc[h][0].y; v[2] *= sc[h][0].z; v[3] *= sc[h][0].w;
                    v[4] *= sc[h][1].x; v[5] *= sc[h][1].y; v[6] *= sc[h][1].z; v[7] *= sc[h][1].w;
                }
                if (has_sh) {
                    v[0] += sh[h][0].x; v[1] += sh[h][0].y; v[2] += sh[h][0].z; v[3] += sh[h][0].w;
                    v[4] += sh[h][1].x; v[5] += sh[h][1].y; v[6] += sh[h][1].z; v[7] += sh[h][1].w;
                }
                uint32_t d[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) d[q] = pack2_bf16(v[2 * q], v[2 * q + 1]);
                if (p.relu) {                       // bf16 as int16: negative floats (and -0) are negative integers
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        s16x2_t t = *reinterpret_cast<s16x2_t*>(&d[q]);
                        t = __builtin_elementwise_max(t, s16x2_t{0, 0});
                        d[q] = *reinterpret_cast<uint32_t*>(&t);
                    }
                }
                const u32x4_t ov = {d[0], d[1], d[2], d[3]};
                __builtin_amdgcn_raw_buffer_store_b128(ov, ry, ok ? ((unsigned)m * (unsigned)p.Cout + (unsigned)c) * 2u : 0x80000000u, 0, 0);
            }
        }
        return;
    }
    // every wave has waited for its DMA (the last tap barrier is vmcnt(0)) and retired its reads; the staging tile reuses the rings
    igemm_epilogue<T, BM, BN, WM, WN, LDS_SLOTS * 16>(p, acc, m0, n0, reinterpret_cast<unsigned char*>(&lds_all[0]));
}

template <int BM, int BN, int WM, int WN, bool DIRECT, bool ILV = false>
__global__ __launch_bounds__(WM* WN * 64) void igemm_halo64_kernel(ConvDev p) {
    igemm_halo64_body<BM, BN, WM, WN, DIRECT, ILV>(p, (int)(blockIdx.y * gridDim.x + blockIdx.x), (int)gridDim.x, (int)gridDim.y);
}
