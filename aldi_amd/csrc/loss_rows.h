// One ROI row of the box head's losses, shared by the per-loss kernels (box_loss_kernel, roih_distill_kernel) and the launch that runs
// both for every chunk of a fused step (box_losses_fused_kernel): the same expressions in the same order, so the results are the same bits.
// grow: this row's Cp fp32 gradient entries (accumulated into; nullable).
#pragma once
#include "common.h"

// FastRCNNOutputLayers.losses: cross-entropy over K + 1 classes (mean over the chunk's R rows) + L1 on the ground-truth class' deltas of
// foreground rows / R.  p: the row's predictions ([0, K] logits, then 4 K deltas); y: its class (K = background); rp: its ROI (b, x1 .. y2)
__device__ __forceinline__ void box_loss_row(const float* __restrict__ p, int K, int y, const float* __restrict__ rp, const float4 t,
                                             float wx, float wy, float ww, float wh, float invR, float gs_cls, float gs_box, float* grow,
                                             float& l_cls, float& l_box) {
    float m = p[0];
    for (int k = 1; k <= K; ++k) m = fmaxf(m, p[k]);
    float s = 0.f;
    for (int k = 0; k <= K; ++k) s += expf(p[k] - m);
    const float lse = m + logf(s);
    l_cls = lse - p[y];
    if (grow && gs_cls != 0.f)
        for (int k = 0; k <= K; ++k) grow[k] += (expf(p[k] - lse) - (k == y ? 1.f : 0.f)) * invR * gs_cls;
    if (y >= 0 && y < K) {
        float src_w = rp[3] - rp[1], src_h = rp[4] - rp[2];
        float sx = rp[1] + 0.5f * src_w, sy = rp[2] + 0.5f * src_h;
        float tw = t.z - t.x, th = t.w - t.y;
        float tx = t.x + 0.5f * tw, ty = t.y + 0.5f * th;
        float d[4] = {wx * (tx - sx) / src_w, wy * (ty - sy) / src_h, ww * logf(tw / src_w), wh * logf(th / src_h)};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float df = p[K + 1 + y * 4 + k] - d[k];
            l_box += fabsf(df);
            if (grow && gs_box != 0.f) grow[K + 1 + y * 4 + k] += (df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f)) * invR * gs_box;
        }
    }
}

// ALDIDistiller.get_roih_losses (aldi/distill.py:231-278): soft cross-entropy / KL against the teacher's softmax at temperature T, L1 between
// the deltas of the teacher's arg-max class (foreground only).  s / t: the student's / teacher's prediction rows.
__device__ __forceinline__ void roih_distill_row(const float* __restrict__ s, const float* __restrict__ t, int K, float inv_T, int kl, int do_cls,
                                                 int do_reg, float invR, float gs_cls, float gs_reg, float* grow, float& l_cls, float& l_reg) {
    // teacher softmax at temperature T, student log-softmax
    float tm = t[0] * inv_T, sm = s[0];
    int amax = 0;
    float tbest = t[0];
    for (int k = 1; k <= K; ++k) {
        tm = fmaxf(tm, t[k] * inv_T);
        sm = fmaxf(sm, s[k]);
        if (t[k] > tbest) { tbest = t[k]; amax = k; }
    }
    float ts = 0.f, ss = 0.f;
    for (int k = 0; k <= K; ++k) { ts += expf(t[k] * inv_T - tm); ss += expf(s[k] - sm); }
    const float tl = tm + logf(ts), sl = sm + logf(ss);
    if (do_cls) {
        float psum = 0.f;
        for (int k = 0; k <= K; ++k) {
            const float lt = t[k] * inv_T - tl;
            const float pk = expf(lt);
            const float ls = s[k] - sl;
            l_cls += kl ? pk * (lt - ls) : -pk * ls;
            psum += pk;
        }
        if (grow && gs_cls != 0.f)
            for (int k = 0; k <= K; ++k) grow[k] += (expf(s[k] - sl) * psum - expf(t[k] * inv_T - tl)) * invR * gs_cls;
    }
    if (do_reg && amax != K) {
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const int c = K + 1 + amax * 4 + d;
            const float df = s[c] - t[c];
            l_reg += fabsf(df);
            if (grow && gs_reg != 0.f) grow[c] += (df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f)) * invR * gs_reg;
        }
    }
}
