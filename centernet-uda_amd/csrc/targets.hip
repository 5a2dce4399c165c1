// CenterNet target encoding on the GPU: the per-image loop of datasets/coco.py:191-221 and
// the gaussian splat of utils/image.py:8-57, for a whole batch in one launch.
//
// One workgroup per (image, object slot).  Box arithmetic in double exactly like the reference's numpy code
// (clip, ceil, gaussian_radius, centre -> float32 -> int truncation), so `ind`, `reg_mask` and the peak cells are
// identical; the splat is exp() in double, cast to float32 and merged with an integer atomicMax (non-negative
// floats order like their bit patterns), which makes overlapping objects order-independent like np.maximum.
#include "common.h"

namespace cnuda {
namespace {

__device__ __forceinline__ double gaussian_radius(double height, double width) {
    const double min_overlap = 0.7;
    const double b1 = height + width;
    const double c1 = width * height * (1 - min_overlap) / (1 + min_overlap);
    const double r1 = (b1 + sqrt(b1 * b1 - 4 * c1)) / 2;
    const double b2 = 2 * (height + width);
    const double c2 = (1 - min_overlap) * width * height;
    const double r2 = (b2 + sqrt(b2 * b2 - 16 * c2)) / 2;
    const double a3 = 4 * min_overlap;
    const double b3 = -2 * min_overlap * (height + width);
    const double c3 = (min_overlap - 1) * width * height;
    const double r3 = (b3 + sqrt(b3 * b3 - 4 * a3 * c3)) / 2;
    return fmin(r1, fmin(r2, r3));
}

__global__ __launch_bounds__(256) void encode_targets_kernel(
    const double* __restrict__ boxes, const int* __restrict__ classes, const int* __restrict__ counts,
    float* __restrict__ hm, unsigned char* __restrict__ reg_mask, long long* __restrict__ ind,
    float* __restrict__ wh, float* __restrict__ reg, float* __restrict__ gt_dets, float* __restrict__ gt_areas,
    int C, int H, int W, int M) {
    const int b = blockIdx.y, k = blockIdx.x;
    if (k >= counts[b]) return;
    const double* bx = boxes + ((size_t)b * M + k) * 4;
    const double x1 = fmin(fmax(bx[0], 0.0), (double)(W - 1)), x2 = fmin(fmax(bx[2], 0.0), (double)(W - 1));
    const double y1 = fmin(fmax(bx[1], 0.0), (double)(H - 1)), y2 = fmin(fmax(bx[3], 0.0), (double)(H - 1));
    const double h = y2 - y1, w = x2 - x1;
    if (!(h > 0 && w > 0)) return;
    const int cls = classes[(size_t)b * M + k];
    if (cls < 0 || cls >= C) return;
    int radius = (int)gaussian_radius(ceil(h), ceil(w));     // int(): truncation (the radius is never negative)
    if (radius < 0) radius = 0;
    const float ctx = (float)((x1 + x2) / 2), cty = (float)((y1 + y2) / 2);
    const int cx = (int)ctx, cy = (int)cty;
    if (threadIdx.x == 0) {
        const size_t o = (size_t)b * M + k;
        wh[o * 2] = (float)w; wh[o * 2 + 1] = (float)h;
        ind[o] = (long long)cy * W + cx;
        reg[o * 2] = ctx - (float)cx; reg[o * 2 + 1] = cty - (float)cy;
        reg_mask[o] = 1;
        // gt_det is assigned as a float64 tuple and cast to float32 (coco.py:219-220)
        gt_dets[o * 6 + 0] = (float)((double)ctx - w / 2); gt_dets[o * 6 + 1] = (float)((double)cty - h / 2);
        gt_dets[o * 6 + 2] = (float)((double)ctx + w / 2); gt_dets[o * 6 + 3] = (float)((double)cty + h / 2);
        gt_dets[o * 6 + 4] = 1.0f; gt_dets[o * 6 + 5] = (float)cls;
        gt_areas[o] = (float)(w * h);
    }
    // draw_umich_gaussian: window [cx-left, cx+right) x [cy-top, cy+bottom), sigma = diameter / 6
    const int left = min(cx, radius), right = min(W - cx, radius + 1);
    const int top = min(cy, radius), bottom = min(H - cy, radius + 1);
    const int ww = left + right, hh = top + bottom;
    if (ww <= 0 || hh <= 0) return;
    const double sigma = (2 * radius + 1) / 6.0;
    const double eps_cut = 2.220446049250313e-16;           // np.finfo(float64).eps * h.max(), h.max() == 1
    int* plane = reinterpret_cast<int*>(hm + ((size_t)b * C + cls) * H * W);
    for (int i = threadIdx.x; i < ww * hh; i += blockDim.x) {
        const int yy = i / ww, xx = i - yy * ww;
        const double dx = (double)(xx - left), dy = (double)(yy - top);
        double gv = exp(-(dx * dx + dy * dy) / (2 * sigma * sigma));
        if (gv < eps_cut) gv = 0.0;
        const float f = (float)gv;
        if (f > 0.0f) atomicMax(plane + (size_t)(cy - top + yy) * W + (cx - left + xx), __float_as_int(f));
    }
}

}  // namespace
}  // namespace cnuda

using namespace cnuda;

extern "C" int cnuda_encode_targets(const double* boxes, const int* classes, const int* counts, float* hm,
                                    unsigned char* reg_mask, long long* ind, float* wh, float* reg, float* gt_dets,
                                    float* gt_areas, int B, int C, int H, int W, int M, cnuda_stream_t stream) {
    CNUDA_REQUIRE(boxes && classes && counts && hm && reg_mask && ind && wh && reg && gt_dets && gt_areas,
                  "cnuda_encode_targets: null pointer");
    CNUDA_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && M > 0 && B <= 65535, "cnuda_encode_targets: bad sizes");
    hipStream_t st = (hipStream_t)stream;
    // the outputs start from zeros like the reference's np.zeros (coco.py:168-174)
    (void)hipMemsetAsync(hm, 0, (size_t)B * C * H * W * sizeof(float), st);
    (void)hipMemsetAsync(reg_mask, 0, (size_t)B * M, st);
    (void)hipMemsetAsync(ind, 0, (size_t)B * M * sizeof(long long), st);
    (void)hipMemsetAsync(wh, 0, (size_t)B * M * 2 * sizeof(float), st);
    (void)hipMemsetAsync(reg, 0, (size_t)B * M * 2 * sizeof(float), st);
    (void)hipMemsetAsync(gt_dets, 0, (size_t)B * M * 6 * sizeof(float), st);
    (void)hipMemsetAsync(gt_areas, 0, (size_t)B * M * sizeof(float), st);
    CNUDA_LAUNCH(encode_targets_kernel, dim3(M, B), dim3(256), 0, st, boxes, classes, counts, hm, reg_mask, ind,
                       wh, reg, gt_dets, gt_areas, C, H, W, M);
    return check_launch("cnuda_encode_targets");
}
