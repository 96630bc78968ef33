// Integer-exact raster tail of the hot path, all HBM-bound byte kernels.
// Reference behaviour (paths relative to /root/reference):
//   mask union      roll/distributed/strategy/seg_strategy.py:58-60   np.logical_or(mask, best).astype(uint8)
//   nearest resize  seg_strategy.py:65, roll/pipeline/rlvr/rlvr_socioseg_vlm_pipeline_infer.py:399 (cv2.INTER_NEAREST:
//                   sx = min(floor(dx * ifx), sw - 1), ifx = 1 / ((double)dw / sw): OpenCV's resizeNN forms the inverse scale as a
//                   reciprocal; floor(dx * sw / dw) differs from it for rare size pairs)
//   IoU counts      rlvr_socioseg_vlm_pipeline_infer.py:45-58
//   render          rlvr_socioseg_vlm_pipeline_infer.py:383-452 (2-px blue ImageDraw.rectangle outlines, then
//                   Image.alpha_composite of (255,0,0,102) where mask>0; PIL's fixed-point formula)
#include "kernels.h"

namespace {

__global__ __launch_bounds__(256) void k_mask_union(uint8_t* acc, const uint8_t* m, size_t n) {
    const size_t nv = n / 16;
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < nv; i += stride) {
        uint4 a = reinterpret_cast<uint4*>(acc)[i], b = reinterpret_cast<const uint4*>(m)[i];
        uint32_t av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w}, o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            uint32_t x = av[k] | bv[k];
            // per byte: 1 if any bit set.  (x | x>>1 | ... | x>>7) & 0x01010101 without crossing bytes
            x |= (x >> 4) & 0x0F0F0F0Fu; x |= (x >> 2) & 0x3F3F3F3Fu; x |= (x >> 1) & 0x7F7F7F7Fu;
            o[k] = x & 0x01010101u;
        }
        reinterpret_cast<uint4*>(acc)[i] = uint4{o[0], o[1], o[2], o[3]};
    }
    for (size_t i = nv * 16 + blockIdx.x * 256ull + threadIdx.x; i < n; i += stride)
        acc[i] = (uint8_t)((acc[i] != 0) || (m[i] != 0));
}

__global__ __launch_bounds__(256) void k_resize_nearest(const uint8_t* src, int sh, int sw, uint8_t* dst, int dh, int dw) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= dw) return;
    const double fy = 1.0 / ((double)dh / sh), fx = 1.0 / ((double)dw / sw);
    int sy = (int)floor(y * fy), sx = (int)floor(x * fx);
    sy = min(sy, sh - 1);
    sx = min(sx, sw - 1);
    dst[(size_t)y * dw + x] = src[(size_t)sy * sw + sx];
}

__global__ __launch_bounds__(256) void k_iou_counts(const uint8_t* p, const uint8_t* g, size_t n, unsigned long long* out2) {
    unsigned long long inter = 0, uni = 0;
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += stride) {
        const int a = p[i] > 0, b = g[i] > 0;
        inter += a & b;
        uni += a | b;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        inter += __shfl_xor(inter, o, 64);
        uni += __shfl_xor(uni, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&out2[0], inter);
        atomicAdd(&out2[1], uni);
    }
}

// The same counts for n_items (prediction, ground truth) pairs of n bytes each in ONE launch (the raster tail of a batch of tiles:
// rlvr_socioseg_vlm_pipeline_infer.py:45-58 per sample): grid (blocks per item, items), 16 bytes per lane and load when n % 16 == 0.
__global__ __launch_bounds__(256) void k_iou_counts_batched(const uint8_t* p, const uint8_t* g, size_t n, unsigned long long* out) {
    const size_t item = blockIdx.y;
    const uint8_t* pp = p + item * n;
    const uint8_t* gg = g + item * n;
    unsigned inter = 0, uni = 0;              // <= n / (blocks * 256) * 16 per lane: far below 2^32
    const size_t nv = (n % 16 == 0 && ((uintptr_t)pp | (uintptr_t)gg) % 16 == 0) ? n / 16 : 0;
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < nv; i += stride) {
        const uint4 a = reinterpret_cast<const uint4*>(pp)[i], b = reinterpret_cast<const uint4*>(gg)[i];
        const uint32_t aw[4] = {a.x, a.y, a.z, a.w}, bw[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            // byte > 0 -> bit 0 of the byte: OR the byte's bits down (x | x >> 1 | ... ) without crossing byte borders
            uint32_t x = aw[q], y = bw[q];
            x |= (x >> 4) & 0x0f0f0f0fu; x |= (x >> 2) & 0x03030303u; x |= (x >> 1) & 0x01010101u; x &= 0x01010101u;
            y |= (y >> 4) & 0x0f0f0f0fu; y |= (y >> 2) & 0x03030303u; y |= (y >> 1) & 0x01010101u; y &= 0x01010101u;
            inter += __popc(x & y);
            uni += __popc(x | y);
        }
    }
    for (size_t i = nv * 16 + blockIdx.x * 256ull + threadIdx.x; i < n; i += stride) {
        const int a = pp[i] > 0, b = gg[i] > 0;
        inter += a & b;
        uni += a | b;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        inter += __shfl_xor(inter, o, 64);
        uni += __shfl_xor(uni, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&out[item * 2 + 0], (unsigned long long)inter);
        atomicAdd(&out[item * 2 + 1], (unsigned long long)uni);
    }
}

// one thread per pixel: rectangles first (reference draws them before compositing), then the overlay
__global__ __launch_bounds__(256) void k_render(uint8_t* img, int h, int w, const uint8_t* mask, int mh, int mw,
                                                const int* boxes, int nb) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    uint8_t* px = img + ((size_t)y * w + x) * 3;
    int c0 = px[0], c1 = px[1], c2 = px[2];
    for (int b = 0; b < nb; ++b) {
        // PIL ImageDraw.rectangle(outline, width = 2) on already validated / truncated coordinates (raster.py pil_rect):
        // rows y0, y0+1, y1-1, y1 over [x0, x1]; columns x0, x0+1, x1-1, x1 over the |dy| points that start at y0+2 and step
        // towards y1-1 (end excluded) -- which is what makes boxes thinner than 3 px spill outside themselves in PIL
        const int x0 = boxes[4 * b], x1 = boxes[4 * b + 2];
        const int y0 = min(boxes[4 * b + 1], boxes[4 * b + 3]), y1 = max(boxes[4 * b + 1], boxes[4 * b + 3]);
        bool hit = x >= min(x0, x1) && x <= max(x0, x1) && (y == y0 || y == y0 + 1 || y == y1 || y == y1 - 1);
        if (x == x0 || x == x0 + 1 || x == x1 || x == x1 - 1) {
            const int ya = y0 + 2, yb = y1 - 1;
            hit |= (yb > ya) ? (y >= ya && y < yb) : (y > yb && y <= ya && yb != ya);
        }
        if (hit) { c0 = 0; c1 = 0; c2 = 255; }
    }
    if (mask) {
        const double fy = 1.0 / ((double)h / mh), fx = 1.0 / ((double)w / mw);
        const int sy = min((int)floor(y * fy), mh - 1), sx = min((int)floor(x * fx), mw - 1);
        if (mask[(size_t)sy * mw + sx] != 0) {
            const int a = 102, col[3] = {255, 0, 0};
            int c[3] = {c0, c1, c2};
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const uint32_t t = (uint32_t)(col[k] * a + c[k] * (255 - a)) * 128u + (128u << 7);
                c[k] = (int)((((t >> 8) + t) >> 8) >> 7);
            }
            c0 = c[0]; c1 = c[1]; c2 = c[2];
        }
    }
    px[0] = (uint8_t)c0; px[1] = (uint8_t)c1; px[2] = (uint8_t)c2;
}

}  // namespace

int launch_mask_union(hipStream_t s, uint8_t* acc, const uint8_t* m, size_t n) {
    if (n == 0) return 0;
    if ((reinterpret_cast<uintptr_t>(acc) | reinterpret_cast<uintptr_t>(m)) & 15) return -22;
    size_t blocks = (n / 16 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(k_mask_union, dim3((unsigned)blocks), dim3(256), 0, s, acc, m, n);
    SR_CHECK_LAUNCH();
    return 0;
}
int launch_resize_nearest_u8(hipStream_t s, const uint8_t* src, int sh, int sw, uint8_t* dst, int dh, int dw) {
    if (dh <= 0 || dw <= 0) return 0;
    hipLaunchKernelGGL(k_resize_nearest, dim3(cdiv(dw, 256), dh), dim3(256), 0, s, src, sh, sw, dst, dh, dw);
    SR_CHECK_LAUNCH();
    return 0;
}
int launch_iou_counts(hipStream_t s, const uint8_t* p, const uint8_t* g, size_t n, long long* out2) {
    hipError_t e = hipMemsetAsync(out2, 0, 2 * sizeof(long long), s);
    if (e != hipSuccess) return (int)e;
    if (n == 0) return 0;
    size_t blocks = (n + 256 * 16 - 1) / (256 * 16);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(k_iou_counts, dim3((unsigned)blocks), dim3(256), 0, s, p, g, n, reinterpret_cast<unsigned long long*>(out2));
    SR_CHECK_LAUNCH();
    return 0;
}
int launch_iou_counts_batched(hipStream_t s, const uint8_t* p, const uint8_t* g, size_t n, int n_items, long long* out) {
    if (n_items <= 0) return 0;
    hipError_t e = hipMemsetAsync(out, 0, (size_t)n_items * 2 * sizeof(long long), s);
    if (e != hipSuccess) return (int)e;
    if (n == 0) return 0;
    size_t blocks = (n + 256 * 64 - 1) / (256 * 64);          // 4 x 16 bytes per lane
    if (blocks > 64) blocks = 64;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_iou_counts_batched, dim3((unsigned)blocks, (unsigned)n_items), dim3(256), 0, s, p, g, n, reinterpret_cast<unsigned long long*>(out));
    SR_CHECK_LAUNCH();
    return 0;
}
int launch_render_overlay(hipStream_t s, uint8_t* img, int h, int w, const uint8_t* mask, int mh, int mw, const int* boxes,
                          int nb) {
    if (h <= 0 || w <= 0) return 0;
    hipLaunchKernelGGL(k_render, dim3(cdiv(w, 256), h), dim3(256), 0, s, img, h, w, mask, mh, mw, boxes, nb);
    SR_CHECK_LAUNCH();
    return 0;
}
