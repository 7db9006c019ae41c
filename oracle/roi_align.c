/* CPU restatement of torchvision.ops.roi_align (aligned=True, sampling_ratio=0 -> adaptive
 * ceil(roi/pooled) samples per bin), forward and backward, NCHW fp32.
 *
 * TEST INFRASTRUCTURE ONLY (oracle/__init__.py).  PARITY UNPINNED: torchvision is absent from
 * /root/reference and from this image; this follows the published algorithm of
 * torchvision/csrc/ops/cpu/roi_align_kernel.cpp (bilinear_interpolate, roi_align_forward_kernel_impl,
 * roi_align_backward_kernel_impl), anchored on the reference's use of ROIAlignV2 through
 * detectron2's ROIPooler (reference tools/visualize_featurespace.py:87; configs/detectron2/Base-RCNN-FPN.yaml:24-27).
 *
 * Build: make -C oracle   ->  oracle/_build/liboracle.so
 */
#include <math.h>
#include <stddef.h>

static void prep(float v, int size, int* lo, int* hi, float* l, float* h, int* dead) {
    *dead = (v < -1.0f || v > (float)size);
    if (v <= 0.f) v = 0.f;
    int a = (int)v, b;
    if (a >= size - 1) { b = a = size - 1; v = (float)a; }
    else b = a + 1;
    *lo = a; *hi = b;
    *l = v - (float)a;
    *h = 1.f - *l;
}

/* feat [N][C][H][W]; rois [R][5] = (batch, x1, y1, x2, y2); out [R][C][P][P] */
void oracle_roi_align_forward(const float* feat, int N, int C, int H, int W, const float* rois, int R, int P, float scale, float* out) {
    (void)N;
#pragma omp parallel for schedule(dynamic, 4)
    for (int r = 0; r < R; ++r) {
        const float* rp = rois + (size_t)r * 5;
        int b = (int)rp[0];
        float x1 = rp[1] * scale - 0.5f, y1 = rp[2] * scale - 0.5f, x2 = rp[3] * scale - 0.5f, y2 = rp[4] * scale - 0.5f;
        float rw = x2 - x1, rh = y2 - y1;
        float bw = rw / (float)P, bh = rh / (float)P;
        int gh = (int)ceilf(rh / (float)P), gw = (int)ceilf(rw / (float)P);
        float count = (float)(gh * gw > 1 ? gh * gw : 1);
        for (int c = 0; c < C; ++c) {
            const float* f = feat + ((size_t)b * C + c) * H * W;
            for (int ph = 0; ph < P; ++ph)
                for (int pw = 0; pw < P; ++pw) {
                    float acc = 0.f;
                    for (int iy = 0; iy < gh; ++iy) {
                        float y = y1 + (float)ph * bh + ((float)iy + 0.5f) * bh / (float)gh;
                        int ylo, yhi, yd; float ly, hy;
                        prep(y, H, &ylo, &yhi, &ly, &hy, &yd);
                        for (int ix = 0; ix < gw; ++ix) {
                            float x = x1 + (float)pw * bw + ((float)ix + 0.5f) * bw / (float)gw;
                            int xlo, xhi, xd; float lx, hx;
                            prep(x, W, &xlo, &xhi, &lx, &hx, &xd);
                            if (yd || xd) continue;
                            float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
                            acc += w1 * f[ylo * W + xlo] + w2 * f[ylo * W + xhi] + w3 * f[yhi * W + xlo] + w4 * f[yhi * W + xhi];
                        }
                    }
                    out[(((size_t)r * C + c) * P + ph) * P + pw] = acc / count;
                }
        }
    }
}

/* gfeat [N][C][H][W] += scatter of gout [R][C][P][P] (serial over ROIs: no atomics needed) */
void oracle_roi_align_backward(const float* gout, int N, int C, int H, int W, const float* rois, int R, int P, float scale, float* gfeat) {
    (void)N;
    for (int r = 0; r < R; ++r) {
        const float* rp = rois + (size_t)r * 5;
        int b = (int)rp[0];
        float x1 = rp[1] * scale - 0.5f, y1 = rp[2] * scale - 0.5f, x2 = rp[3] * scale - 0.5f, y2 = rp[4] * scale - 0.5f;
        float rw = x2 - x1, rh = y2 - y1;
        float bw = rw / (float)P, bh = rh / (float)P;
        int gh = (int)ceilf(rh / (float)P), gw = (int)ceilf(rw / (float)P);
        float count = (float)(gh * gw > 1 ? gh * gw : 1);
#pragma omp parallel for schedule(static)
        for (int c = 0; c < C; ++c) {
            float* g = gfeat + ((size_t)b * C + c) * H * W;
            for (int ph = 0; ph < P; ++ph)
                for (int pw = 0; pw < P; ++pw) {
                    float go = gout[(((size_t)r * C + c) * P + ph) * P + pw] / count;
                    for (int iy = 0; iy < gh; ++iy) {
                        float y = y1 + (float)ph * bh + ((float)iy + 0.5f) * bh / (float)gh;
                        int ylo, yhi, yd; float ly, hy;
                        prep(y, H, &ylo, &yhi, &ly, &hy, &yd);
                        for (int ix = 0; ix < gw; ++ix) {
                            float x = x1 + (float)pw * bw + ((float)ix + 0.5f) * bw / (float)gw;
                            int xlo, xhi, xd; float lx, hx;
                            prep(x, W, &xlo, &xhi, &lx, &hx, &xd);
                            if (yd || xd) continue;
                            g[ylo * W + xlo] += go * hy * hx;
                            g[ylo * W + xhi] += go * hy * lx;
                            g[yhi * W + xlo] += go * ly * hx;
                            g[yhi * W + xhi] += go * ly * lx;
                        }
                    }
                }
        }
    }
}

/* torchvision nms (CPU kernel semantics) on boxes already sorted by descending score:
 * keep[i] = 1 if box i survives.  IoU = inter / (area_i + area_j - inter) > thresh suppresses. */
void oracle_nms_sorted(const float* boxes, int n, float thresh, unsigned char* keep) {
    for (int i = 0; i < n; ++i) keep[i] = 1;
    for (int i = 0; i < n; ++i) {
        if (!keep[i]) continue;
        const float* a = boxes + (size_t)i * 4;
        float aa = (a[2] - a[0]) * (a[3] - a[1]);
        for (int j = i + 1; j < n; ++j) {
            if (!keep[j]) continue;
            const float* b = boxes + (size_t)j * 4;
            float w = fminf(a[2], b[2]) - fmaxf(a[0], b[0]);
            float h = fminf(a[3], b[3]) - fmaxf(a[1], b[1]);
            w = w > 0.f ? w : 0.f;
            h = h > 0.f ? h : 0.f;
            float inter = w * h;
            float ab = (b[2] - b[0]) * (b[3] - b[1]);
            if (inter / (aa + ab - inter) > thresh) keep[j] = 0;
        }
    }
}
