/* Plain-C restatement of the raster tail of the hot path.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
 * Paths are relative to /root/reference.
 *   ref_mask_union        roll/distributed/strategy/seg_strategy.py:58-60  (logical_or -> uint8 {0,1})
 *   ref_resize_nearest_u8 cv2.INTER_NEAREST at seg_strategy.py:65 and
 *                         roll/pipeline/rlvr/rlvr_socioseg_vlm_pipeline_infer.py:399
 *                         (cv2 absent here: OpenCV's resizeNN rule restated -- inverse scale 1 / ((double)dw / sw) formed the way
 *                         cv::resize forms it, sx = min(floor(dx * that), sw - 1); the reciprocal matters: floor(dx * sw / dw)
 *                         differs from it for rare size pairs, e.g. 768 -> 1148)
 *   ref_iou_counts        rlvr_socioseg_vlm_pipeline_infer.py:45-58
 *   ref_render_overlay    rlvr_socioseg_vlm_pipeline_infer.py:383-452 (PIL ImageDraw.rectangle + alpha_composite,
 *                         integer formulas verified against PIL 12.2 by tools/make_golden.py)
 *   ref_normalize_lut     hf:image_transforms.py:89-124, 384-440 (rescale in double, float32 (x-mean)/std)
 */
#include <stdint.h>
#include <stddef.h>
#include <math.h>

void ref_mask_union(uint8_t *acc, const uint8_t *m, size_t n) {
    for (size_t i = 0; i < n; ++i) acc[i] = (uint8_t)((acc[i] != 0) || (m[i] != 0));
}

void ref_resize_nearest_u8(const uint8_t *src, int sh, int sw, uint8_t *dst, int dh, int dw) {
    double fy = 1.0 / ((double)dh / sh), fx = 1.0 / ((double)dw / sw);
    for (int y = 0; y < dh; ++y) {
        int sy = (int)floor(y * fy);
        if (sy > sh - 1) sy = sh - 1;
        for (int x = 0; x < dw; ++x) {
            int sx = (int)floor(x * fx);
            if (sx > sw - 1) sx = sw - 1;
            dst[(size_t)y * dw + x] = src[(size_t)sy * sw + sx];
        }
    }
}

void ref_iou_counts(const uint8_t *p, const uint8_t *g, size_t n, int64_t out[2]) {
    int64_t inter = 0, uni = 0;
    for (size_t i = 0; i < n; ++i) {
        int a = p[i] > 0, b = g[i] > 0;
        inter += a & b;
        uni += a | b;
    }
    out[0] = inter;
    out[1] = uni;
}

/* PIL 12.2 ImageDraw.rectangle(outline, width=2) on integer coordinates (libImaging Draw.c ImagingDrawRectangle, outline
   branch, probed against PIL itself by tools/make_golden.py -> tests/golden/render_boxes.json):
     for i in 0..width-1:  hline(x0, y0+i, x1); hline(x0, y1-i, x1); line(x1-i, y0+width, x1-i, y1-width+1);
                           line(x0+i, y0+width, x0+i, y1-width+1)
   hline covers [min(xa,xb), max(xa,xb)] of one row; the vertical line loop draws |dy| points from its start towards its
   end (end point excluded) -- so a box thinner than 3 px spills outside itself exactly as PIL does.  Clipped to the image. */
static void put(uint8_t *img, int h, int w, int x, int y) {
    if (x < 0 || y < 0 || x >= w || y >= h) return;
    uint8_t *px = img + ((size_t)y * w + x) * 3;
    px[0] = 0; px[1] = 0; px[2] = 255;
}
static void hline(uint8_t *img, int h, int w, int xa, int y, int xb) {
    if (y < 0 || y >= h) return;
    if (xa > xb) { int t = xa; xa = xb; xb = t; }
    if (xa < 0) xa = 0;
    if (xb > w - 1) xb = w - 1;
    for (int x = xa; x <= xb; ++x) put(img, h, w, x, y);
}
static void vline(uint8_t *img, int h, int w, int x, int ya, int yb) {
    /* |yb - ya| points from ya towards yb: the end point is not drawn */
    if (x < 0 || x >= w || ya == yb) return;
    int lo = yb > ya ? ya : yb + 1, hi = yb > ya ? yb - 1 : ya;
    if (lo < 0) lo = 0;
    if (hi > h - 1) hi = h - 1;
    for (int y = lo; y <= hi; ++y) put(img, h, w, x, y);
}

/* img: RGB u8 [h,w,3] in place; mask u8 [mh,mw] (nearest-resized to h,w; may be NULL); boxes int32 [nb,4]: already
   validated and truncated the way PIL's _draw_rectangle does it (oracle/host_ref.py pil_box) */
void ref_render_overlay(uint8_t *img, int h, int w, const uint8_t *mask, int mh, int mw,
                        const int32_t *boxes, int nb) {
    const int width = 2;
    for (int b = 0; b < nb; ++b) {
        int x0 = boxes[4 * b], y0 = boxes[4 * b + 1], x1 = boxes[4 * b + 2], y1 = boxes[4 * b + 3];
        if (y0 > y1) { int t = y0; y0 = y1; y1 = t; }
        for (int i = 0; i < width; ++i) {
            hline(img, h, w, x0, y0 + i, x1);
            hline(img, h, w, x0, y1 - i, x1);
            vline(img, h, w, x1 - i, y0 + width, y1 - width + 1);
            vline(img, h, w, x0 + i, y0 + width, y1 - width + 1);
        }
    }
    if (!mask) return;
    const int a = 102;
    const int col[3] = {255, 0, 0};
    double fy = 1.0 / ((double)h / mh), fx = 1.0 / ((double)w / mw);
    for (int y = 0; y < h; ++y) {
        int sy = (int)floor(y * fy);
        if (sy > mh - 1) sy = mh - 1;
        for (int x = 0; x < w; ++x) {
            int sx = (int)floor(x * fx);
            if (sx > mw - 1) sx = mw - 1;
            if (mask[(size_t)sy * mw + sx] == 0) continue;
            uint8_t *px = img + ((size_t)y * w + x) * 3;
            for (int c = 0; c < 3; ++c) {
                uint32_t t = (uint32_t)(col[c] * a + px[c] * (255 - a)) * 128u + (128u << 7);
                px[c] = (uint8_t)((((t >> 8) + t) >> 8) >> 7);
            }
        }
    }
}

/* lut[c*256+u] = float32 normalised value of byte u in channel c */
void ref_normalize_lut(float *lut) {
    const float mean[3] = {0.48145466f, 0.4578275f, 0.40821073f};
    const float std_[3] = {0.26862954f, 0.26130258f, 0.27577711f};
    for (int c = 0; c < 3; ++c)
        for (int u = 0; u < 256; ++u) {
            float x = (float)((double)u * (1.0 / 255.0));
            lut[c * 256 + u] = (x - mean[c]) / std_[c];
        }
}
