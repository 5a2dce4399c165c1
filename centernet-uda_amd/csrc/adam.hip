// Fused Adam step over one flat fp32 parameter arena (one launch for all 233
// tensors of DLA-34 instead of torch.optim.Adam's per-tensor loops,
// train.py:88-90).  Arithmetic follows torch.optim.Adam (L2 weight decay folded
// into the gradient, bias-corrected, eps added after the sqrt):
//   g += wd*p;  m += (g-m)*(1-b1);  v = v*b2 + g*g*(1-b2)
//   p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
#include "common.h"

namespace cnuda {
namespace {
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, long long n, float step_size, float beta1, float beta2,
                            float bc2_sqrt, float eps, float wd) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float pi = p[i];
        float gi = g[i];
        if (wd != 0.0f) gi = gi + wd * pi;
        float mi = m[i], vi = v[i];
        mi = mi + (gi - mi) * (1.0f - beta1);
        vi = vi * beta2 + (gi * gi) * (1.0f - beta2);
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = pi - step_size * (mi / denom);
    }
}
}  // namespace
}  // namespace cnuda

using namespace cnuda;

extern "C" int cnuda_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long long n,
                               float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                               cnuda_stream_t stream) {
    CNUDA_REQUIRE(param && grad && exp_avg && exp_avg_sq && n > 0 && step >= 1, "cnuda_adam_step: bad arguments");
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    CNUDA_LAUNCH(adam_kernel, dim3(stream_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg,
                       exp_avg_sq, n, (float)((double)lr / bc1), beta1, beta2, (float)sqrt(bc2), eps, weight_decay);
    return check_launch("cnuda_adam_step");
}
