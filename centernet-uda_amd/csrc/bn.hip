// BatchNorm2d (NCHW fp32) for gfx950, fused with the residual add and ReLU that
// follow it everywhere in DLA-34 (backends/dla.py:48-62 BasicBlock, :150-168
// Root, :351-372 DeformConv, :277-287 conv levels).  Replaces
// nn.BatchNorm2d(momentum=0.1) + `out += residual` + nn.ReLU(inplace=True).
//
// Training forward : (1) per-channel sum / sum-of-squares partials over a
//                    (channel, split) grid, fp64 accumulation; (2) one streaming
//                    pass y = act(xhat*g + b [+ res]) whose workgroups first fold the
//                    partials into mean / invstd (and one of them updates the
//                    running statistics: unbiased var, momentum).
// Backward         : same two-step shape for (sum dy, sum dy*xhat).
// All three passes are HBM-bound streaming kernels with 16-byte accesses when
// the plane size allows.
#include "common.h"

namespace cnuda {
namespace {

constexpr int kBnThreads = 256;

struct Split {
    int per_plane;     // splits of one (b, c) plane
    int ips;           // images per split / per apply workgroup (> 1 only with per_plane == 1: small planes)
    int S;             // partials per channel = B / ips * per_plane
    long long chunk;   // elements of a plane per split (multiple of 4)
};
// Bg = images per statistics group (a split never straddles two groups)
Split pick_split(int B, int C, long long HW, int Bg) {
    // aim for ~2048 workgroups in total, at least 4096 elements each
    long long s = 2048 / ((long long)B * C > 0 ? (long long)B * C : 1);
    const long long max_s = (HW + 4095) / 4096;
    if (s > max_s) s = max_s;
    if (s < 1) s = 1;
    Split r;
    r.chunk = ((HW + s - 1) / s + 3) / 4 * 4;
    r.per_plane = (int)((HW + r.chunk - 1) / r.chunk);
    // many small planes (the 256- and 512-channel levels: 8,192 / 16,384 planes of 1,024 / 256 values): a workgroup per
    // plane is bound by the workgroup launch rate (16,384 workgroups: 38 us for 17 MB), so one workgroup takes several
    // images of its channel -- at least 1,024 values per workgroup while more than 2,048 workgroups per statistics
    // group remain (a function of the GROUP's size only: a batch of two domains sums like the two batches alone)
    r.ips = 1;
    if (r.per_plane == 1)
        while (r.ips * 2 <= Bg && Bg % (r.ips * 2) == 0 && (long long)Bg * C / r.ips > 2048 && HW * r.ips < 4096) r.ips *= 2;
    r.S = B / r.ips * r.per_plane;
    return r;
}

// fused activation code `relu`: 0 none, 1 ReLU, 2 ReLU6 (torchvision MobileNetV2's nn.ReLU6 = hardtanh(0, 6):
// gradient passes strictly inside (0, 6))
__device__ __forceinline__ float bn_act(float v, int relu) {
    v = fmaxf(v, 0.f);
    return relu == 2 ? fminf(v, 6.f) : v;
}
__device__ __forceinline__ bool bn_pass(float y, int relu) { return y > 0.0f && (relu != 2 || y < 6.0f); }

// The normalisation as the forward applies it, y = x * sc + sh with sc = invstd * gamma and sh = beta - mean * sc: ONE
// rounded multiply and ONE rounded add, spelled with the explicit-rounding intrinsics so that no compiler flag or
// optimisation level can contract them into an fma in one kernel and not in another.  The backward kernels recompute
// the activation's gate from x with these same two functions: the sign they see is the forward's, bit for bit.
__device__ __forceinline__ float bn_shift(float beta, float mean, float sc) { return __fsub_rn(beta, __fmul_rn(mean, sc)); }
__device__ __forceinline__ float bn_affine(float x, float sc, float sh) { return __fadd_rn(__fmul_rn(x, sc), sh); }

// partial[(c*S + s)*2 + {0,1}] = sum(v), sum(v*w) over one chunk of one (b, c) plane
// MODE 0: v = x, w = x                       (forward statistics)
// MODE 1: v = dy', w = xhat                   (backward), dy' = dy * (y > 0) when relu
// The activation's gate: when no residual entered the forward (`beta` given), y = act(x * sc + sh) and the gate is
// RECOMPUTED from x with the forward's own two operations (bn_apply_kernel: sc = invstd * gamma, sh = beta - mean * sc,
// one multiply, one add) instead of read from y -- one tensor less per backward pass.
template <int MODE>
__global__ __launch_bounds__(kBnThreads) void bn_reduce_kernel(
    const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ y,
    const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ gamma,
    const float* __restrict__ beta, double* __restrict__ partial,
    int C, long long HW, long long chunk, int per_plane, int ips, int S, int relu, int imgs_per_group) {
    __shared__ double red[16];
    const int c = blockIdx.x, s = blockIdx.y;
    const int b0 = s / per_plane * ips, part = s - s / per_plane * per_plane;
    const int grp = b0 / imgs_per_group;
    const long long e0 = (long long)part * chunk;
    long long e1 = e0 + chunk;
    if (e1 > HW) e1 = HW;
    double a0 = 0.0, a1 = 0.0;
    float mu = 0.f, is = 0.f, sc = 0.f, sh = 0.f;
    if (MODE == 1) { mu = mean[grp * C + c]; is = invstd[grp * C + c]; }
    const bool regate = MODE == 1 && relu && beta != nullptr;
    if (regate) { sc = __fmul_rn(is, gamma[c]); sh = bn_shift(beta[c], mu, sc); }
    auto acc = [&](float xv, float gv, float yv) {
        if (MODE == 0) {
            const double v = (double)xv;
            a0 += v;
            a1 += v * v;
        } else {
            float g = gv;
            if (relu && !bn_pass(regate ? bn_affine(xv, sc, sh) : yv, relu)) g = 0.0f;
            const float xh = (xv - mu) * is;
            a0 += (double)g;
            a1 += (double)g * (double)xh;
        }
    };
    for (int b = b0; b < b0 + ips; ++b) {
        const size_t base = ((size_t)b * C + c) * HW;
        if ((HW & 3) == 0) {
            for (long long e = e0 + threadIdx.x * 4; e < e1; e += kBnThreads * 4) {
                const float4 xv = *reinterpret_cast<const float4*>(x + base + e);
                float4 gv = make_float4(0, 0, 0, 0), yv = make_float4(1, 1, 1, 1);
                if (MODE == 1) {
                    gv = *reinterpret_cast<const float4*>(dy + base + e);
                    if (relu && !regate) yv = *reinterpret_cast<const float4*>(y + base + e);
                }
                acc(xv.x, gv.x, yv.x); acc(xv.y, gv.y, yv.y); acc(xv.z, gv.z, yv.z); acc(xv.w, gv.w, yv.w);
            }
        } else {
            for (long long e = e0 + threadIdx.x; e < e1; e += kBnThreads)
                acc(x[base + e], MODE == 1 ? dy[base + e] : 0.f, (MODE == 1 && relu && !regate) ? y[base + e] : 1.f);
        }
    }
    a0 = block_sum(a0, red);
    a1 = block_sum(a1, red);
    if (threadIdx.x == 0) {
        partial[((size_t)c * S + s) * 2 + 0] = a0;
        partial[((size_t)c * S + s) * 2 + 1] = a1;
    }
}

// Every workgroup of the apply passes re-derives its channel's statistics from the S fp64 partials (S <= a few
// hundred loads, one wave) instead of waiting for a separate finalize launch; the workgroup that handles the
// first chunk of image 0 also writes them out (saved mean / invstd, running statistics, batch counter, or the
// gamma / beta gradients).
// (s_begin, s_end): the partials of one statistics group -- the images [g*Bg, (g+1)*Bg) of a batch that carries
// several domains (source | target), each normalised by its own statistics like two separate forward calls
__device__ __forceinline__ void bn_sum_partials(const double* __restrict__ partial, int c, int S, int s_begin, int s_end,
                                                double& s0, double& s1, double* red) {
    double a0 = 0.0, a1 = 0.0;
    for (int s = s_begin + threadIdx.x; s < s_end; s += kBnThreads) {
        a0 += partial[((size_t)c * S + s) * 2 + 0];
        a1 += partial[((size_t)c * S + s) * 2 + 1];
    }
    s0 = block_sum(a0, red);
    s1 = block_sum(a1, red);
}

// y = act((x - mean) * invstd * gamma + beta [+ residual]); grid (plane, splits)
__global__ __launch_bounds__(kBnThreads) void bn_apply_kernel(
    const float* __restrict__ x, const double* __restrict__ partial, int S, long long count, float momentum, float eps,
    float* __restrict__ save_mean, float* __restrict__ save_invstd, float* __restrict__ running_mean,
    float* __restrict__ running_var, long long* __restrict__ num_batches_tracked,
    const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ residual,
    float* __restrict__ y, int C, long long HW, int relu, int groups, int imgs_per_group, int per_plane, int ips,
    int spg /* partials per group when they do not come from bn_reduce_kernel<0> (bn_fold_stats_kernel), else 0 */) {
    __shared__ double red[16];
    __shared__ float stat[2];
    const long long plane = blockIdx.x;               // (block of ips images, channel)
    const int c = (int)(plane % C);
    const int b = (int)(plane / C) * ips, grp = b / imgs_per_group;
    const int s_per_group = spg ? spg : imgs_per_group / ips * per_plane;
    const double n = (double)count;                        // values per channel and GROUP
    double s0, s1;
    bn_sum_partials(partial, c, S, grp * s_per_group, (grp + 1) * s_per_group, s0, s1, red);
    if (threadIdx.x == 0) {
        const double mu = s0 / n;
        // biased.  With partials from bn_reduce_kernel<0> (squares and sums formed in fp64) the subtraction is safe for any
        // float32 input.  With partials folded from a GEMM epilogue (bn_fold_stats_kernel: per-block f32 sums of f32 squares,
        // fp64 only across blocks) it loses ~2 log10(|mean| / std) of float32's seven digits: fine for the zero-centred
        // outputs of bias-free convolutions in front of a BatchNorm (|mean| <~ 10 std: 1e-5), NOT for a channel whose
        // mean dwarfs its spread (`epilogue statistics`); such a layer must not request them.
        double var = s1 / n - mu * mu;
        if (var < 0.0) var = 0.0;
        const float m = (float)mu, is = (float)(1.0 / sqrt(var + (double)eps));
        stat[0] = m;
        stat[1] = is;
        if (b == grp * imgs_per_group && blockIdx.y == 0) {   // first image of the group, first chunk: single writer
            save_mean[grp * C + c] = m;
            save_invstd[grp * C + c] = is;
        }
    }
    if (plane < C && blockIdx.y == 0 && (running_mean || num_batches_tracked)) {
        // image 0: the channel's single writer of the running statistics -- one momentum update per group, in
        // group order (source first, then target: the reference forwards the two domains one after the other)
        double rm = 0.0, rv = 0.0;
        if (threadIdx.x == 0 && running_mean) { rm = (double)running_mean[c]; rv = (double)running_var[c]; }
        for (int g = 0; g < groups; ++g) {
            double t0, t1;
            if (g == grp) { t0 = s0; t1 = s1; }
            else bn_sum_partials(partial, c, S, g * s_per_group, (g + 1) * s_per_group, t0, t1, red);
            if (threadIdx.x == 0) {
                const double mu = t0 / n;
                double var = t1 / n - mu * mu;
                if (var < 0.0) var = 0.0;
                const double unbiased = count > 1 ? var * n / (n - 1.0) : var;
                rm = (1.0 - momentum) * (double)(float)rm + momentum * mu;       // rounded to float between
                rv = (1.0 - momentum) * (double)(float)rv + momentum * unbiased; // updates, like two calls
            }
        }
        if (threadIdx.x == 0) {
            if (running_mean) { running_mean[c] = (float)rm; running_var[c] = (float)rv; }
            if (c == 0 && num_batches_tracked) *num_batches_tracked += groups;     // nn.BatchNorm2d's counter
        }
    }
    if (y == nullptr) return;       // statistics only: the consumer of x applies the normalisation while it stages x (apply on load)
    __syncthreads();
    const float sc = __fmul_rn(stat[1], gamma[c]);
    const float sh = bn_shift(beta[c], stat[0], sc);
    const long long start = (long long)blockIdx.y * kBnThreads * 4 + threadIdx.x * 4;
    const long long stride = (long long)gridDim.y * kBnThreads * 4;
    for (int img = b; img < b + ips; ++img) {
    const size_t base = ((size_t)img * C + c) * HW;
    if ((HW & 3) == 0) {
        for (long long i = start; i < HW; i += stride) {
            float4 v = *reinterpret_cast<const float4*>(x + base + i);
            v.x = bn_affine(v.x, sc, sh); v.y = bn_affine(v.y, sc, sh); v.z = bn_affine(v.z, sc, sh); v.w = bn_affine(v.w, sc, sh);
            if (residual) {
                const float4 r = *reinterpret_cast<const float4*>(residual + base + i);
                v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
            }
            if (relu) { v.x = bn_act(v.x, relu); v.y = bn_act(v.y, relu); v.z = bn_act(v.z, relu); v.w = bn_act(v.w, relu); }
            *reinterpret_cast<float4*>(y + base + i) = v;
        }
    } else {
        for (long long i0 = start; i0 < HW; i0 += stride)
            for (long long i = i0; i < i0 + 4 && i < HW; ++i) {
                float v = bn_affine(x[base + i], sc, sh);
                if (residual) v += residual[base + i];
                if (relu) v = bn_act(v, relu);
                y[base + i] = v;
            }
    }    }
}

// gx = gamma*invstd*(dy' - sum_dy/n - xhat*sum_dy_xhat/n);  gres = dy'
__global__ __launch_bounds__(kBnThreads) void bn_bwd_apply_kernel(
    const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ y,
    const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ gamma,
    const float* __restrict__ beta /* nullable: see bn_reduce_kernel */,
    const double* __restrict__ partial, int S, float* __restrict__ ggamma, float* __restrict__ gbeta,
    float* __restrict__ gx, float* __restrict__ gres, int C, long long HW, long long count, int relu, int groups,
    int imgs_per_group, int per_plane, int ips) {
    __shared__ double red[16];
    __shared__ float stat[2];
    const long long plane = blockIdx.x;               // (block of ips images, channel)
    const int c = (int)(plane % C);
    const int b = (int)(plane / C) * ips, grp = b / imgs_per_group;
    const int s_per_group = imgs_per_group / ips * per_plane;
    double s0, s1;
    bn_sum_partials(partial, c, S, grp * s_per_group, (grp + 1) * s_per_group, s0, s1, red);
    if (threadIdx.x == 0) {
        stat[0] = (float)s0;
        stat[1] = (float)s1;
    }
    if (plane < C && blockIdx.y == 0) {      // the channel's single writer: gamma / beta gradients over all groups
        double g0 = s0, g1 = s1;
        for (int g = 0; g < groups; ++g) {
            if (g == grp) continue;
            double t0, t1;
            bn_sum_partials(partial, c, S, g * s_per_group, (g + 1) * s_per_group, t0, t1, red);
            g0 += t0;
            g1 += t1;
        }
        if (threadIdx.x == 0) { gbeta[c] = (float)g0; ggamma[c] = (float)g1; }
    }
    __syncthreads();
    const float mu = mean[grp * C + c], is = invstd[grp * C + c];
    const float k = gamma[c] * is;
    const bool regate = relu && beta != nullptr;
    const float sc = __fmul_rn(is, gamma[c]), sh = regate ? bn_shift(beta[c], mu, sc) : 0.0f;
    const float inv_n = 1.0f / (float)count;
    const float m0 = stat[0] * inv_n, m1 = stat[1] * inv_n;
    for (int img = b; img < b + ips; ++img) {
    const size_t base = ((size_t)img * C + c) * HW;
    if ((HW & 3) == 0) {
        const long long start = (long long)blockIdx.y * kBnThreads * 4 + threadIdx.x * 4;
        const long long stride = (long long)gridDim.y * kBnThreads * 4;
        for (long long i = start; i < HW; i += stride) {
            float4 g = *reinterpret_cast<const float4*>(dy + base + i);
            const float4 xv = *reinterpret_cast<const float4*>(x + base + i);
            if (relu) {
                float4 yv;
                if (regate) yv = make_float4(bn_affine(xv.x, sc, sh), bn_affine(xv.y, sc, sh), bn_affine(xv.z, sc, sh), bn_affine(xv.w, sc, sh));
                else yv = *reinterpret_cast<const float4*>(y + base + i);
                if (!bn_pass(yv.x, relu)) g.x = 0.0f;
                if (!bn_pass(yv.y, relu)) g.y = 0.0f;
                if (!bn_pass(yv.z, relu)) g.z = 0.0f;
                if (!bn_pass(yv.w, relu)) g.w = 0.0f;
            }
            float4 o;
            o.x = k * (g.x - m0 - (xv.x - mu) * is * m1);
            o.y = k * (g.y - m0 - (xv.y - mu) * is * m1);
            o.z = k * (g.z - m0 - (xv.z - mu) * is * m1);
            o.w = k * (g.w - m0 - (xv.w - mu) * is * m1);
            *reinterpret_cast<float4*>(gx + base + i) = o;
            if (gres) *reinterpret_cast<float4*>(gres + base + i) = g;
        }
        continue;
    }
    for (long long i = (long long)blockIdx.y * kBnThreads + threadIdx.x; i < HW; i += (long long)gridDim.y * kBnThreads) {
        float g = dy[base + i];
        if (relu && !bn_pass(regate ? bn_affine(x[base + i], sc, sh) : y[base + i], relu)) g = 0.0f;
        const float xh = (x[base + i] - mu) * is;
        gx[base + i] = k * (g - m0 - xh * m1);
        if (gres) gres[base + i] = g;
    }
    }
}

__global__ __launch_bounds__(kBnThreads) void bn_eval_kernel(
    const float* __restrict__ x, const float* __restrict__ rmean, const float* __restrict__ rvar,
    const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ residual,
    float* __restrict__ y, int C, long long HW, float eps, int relu) {
    const long long plane = blockIdx.x;
    const int c = (int)(plane % C);
    const float sc = gamma[c] / sqrtf(rvar[c] + eps);
    const float sh = beta[c] - rmean[c] * sc;
    const size_t base = (size_t)plane * HW;
    for (long long i = (long long)blockIdx.y * kBnThreads + threadIdx.x; i < HW; i += (long long)gridDim.y * kBnThreads) {
        float v = x[base + i] * sc + sh;
        if (residual) v += residual[base + i];
        if (relu) v = bn_act(v, relu);
        y[base + i] = v;
    }
}

// Statistics that arrive from the producing GEMM's epilogue (igemm.cuh, "BatchNorm statistics"): stats[block][rows][2]
// floats, one block per `blk_px` consecutive pixels of the flattened (image, pixel) axis.  One workgroup adds the blocks
// of ONE (group, split) range for 16 channels -- thread (channel, lane of 16) walks its blocks in order in double
// precision, the 16 lanes are added in lane order -- and leaves partial[(c * S + s) * 2 + {0, 1}] exactly where
// bn_reduce_kernel<0> would: bn_apply_kernel does not know the difference.  Fixed order: bit-reproducible.
__global__ __launch_bounds__(256) void bn_fold_stats_kernel(const float* __restrict__ stats, int rows, long long blocks_per_group,
                                                           int spl, double* __restrict__ partial, int C, int S) {
    __shared__ double red[2][16][17];
    const int cl = threadIdx.x & 15, bl = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    const int grp = blockIdx.y / spl, sp = blockIdx.y - grp * spl;
    const long long per = (blocks_per_group + spl - 1) / spl;
    const long long b0 = grp * blocks_per_group + sp * per;
    long long b1 = b0 + per;
    if (b1 > (grp + 1) * blocks_per_group) b1 = (grp + 1) * blocks_per_group;
    double a0 = 0.0, a1 = 0.0;
    if (c < C) {
        const float2* src = reinterpret_cast<const float2*>(stats) + c;
        for (long long b = b0 + bl; b < b1; b += 64) {
            float2 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = b + 16 * u < b1 ? src[(size_t)(b + 16 * u) * rows] : make_float2(0.f, 0.f);
#pragma unroll
            for (int u = 0; u < 4; ++u) { a0 += (double)v[u].x; a1 += (double)v[u].y; }
        }
    }
    red[0][bl][cl] = a0;
    red[1][bl][cl] = a1;
    __syncthreads();
    if (bl == 0 && c < C) {
        double t0 = 0.0, t1 = 0.0;
        for (int i = 0; i < 16; ++i) { t0 += red[0][i][cl]; t1 += red[1][i][cl]; }
        partial[((size_t)c * S + blockIdx.y) * 2 + 0] = t0;
        partial[((size_t)c * S + blockIdx.y) * 2 + 1] = t1;
    }
}

int plane_splits(long long planes, long long HW, int per_block) {
    long long want = 2048 / (planes > 0 ? planes : 1);
    if (want < 1) want = 1;
    const long long max_s = (HW + per_block - 1) / per_block;
    if (want > max_s) want = max_s;
    return (int)(want < 1 ? 1 : want);
}

}  // namespace
}  // namespace cnuda

using namespace cnuda;

extern "C" size_t cnuda_bn_workspace_bytes(int B, int C, long long HW) {
    const Split sp = pick_split(B, C, HW, 1);          // (ips = 1: the most partials any grouping needs)
    const size_t s_max = sp.S > 64 * 8 ? (size_t)sp.S : (size_t)64 * 8;   // (or 64 folded partials for each of up to 8 groups)
    return (size_t)C * s_max * 2 * sizeof(double) + (size_t)C * 2 * sizeof(float) + 512;
}

extern "C" int cnuda_bn_train_forward(const float* x, const float* gamma, const float* beta, const float* residual,
                                      float* y, float* save_mean, float* save_invstd, float* running_mean,
                                      float* running_var, long long* num_batches_tracked, float momentum, float eps,
                                      int relu, int B, int C, long long HW, int groups, void* workspace,
                                      size_t workspace_bytes, cnuda_stream_t stream) {
    CNUDA_REQUIRE(x && gamma && beta && y && save_mean && save_invstd, "cnuda_bn_train_forward: null pointer");
    CNUDA_REQUIRE(B > 0 && C > 0 && HW > 0, "cnuda_bn_train_forward: empty tensor");
    CNUDA_REQUIRE(groups >= 1 && B % groups == 0, "cnuda_bn_train_forward: batch %d not divisible into %d groups", B, groups);
    CNUDA_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "cnuda_bn_train_forward: running stats");
    const int Bg = B / groups;
    const long long count = (long long)Bg * HW;
    // nn.BatchNorm2d raises for a single value per channel in training mode
    CNUDA_REQUIRE(count > 1, "Expected more than 1 value per channel when training, got input size [%d, %d, %lld]", Bg, C,
                  HW);
    CNUDA_REQUIRE(workspace && workspace_bytes >= cnuda_bn_workspace_bytes(B, C, HW),
                  "cnuda_bn_train_forward: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const Split sp = pick_split(B, C, HW, Bg);
    double* partial = reinterpret_cast<double*>(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    CNUDA_LAUNCH(bn_reduce_kernel<0>, dim3(C, sp.S), dim3(kBnThreads), 0, st, x, (const float*)nullptr,
                       (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr,
                       (const float*)nullptr, partial, C, HW,
                       sp.chunk, sp.per_plane, sp.ips, sp.S, 0, Bg);
    const long long planes = (long long)B * C / sp.ips;        // workgroups of the apply pass (x its plane splits)
    CNUDA_LAUNCH(bn_apply_kernel, dim3((unsigned)planes, plane_splits(planes, HW, kBnThreads * 4)),
                       dim3(kBnThreads), 0, st, x, partial, sp.S, count, momentum, eps, save_mean, save_invstd,
                       running_mean, running_var, num_batches_tracked, gamma, beta, residual, y, C, HW, relu, groups, Bg,
                       sp.per_plane, sp.ips, 0);
    return check_launch("cnuda_bn_train_forward");
}

// The same layer with sum(x) / sum(x^2) already taken by the kernel that produced x (cnuda_conv2d_forward_stats,
// cnuda_dcn_v2_forward_stats): `stats` = [blocks][rows][2] floats, channel c in row c, the blocks numbered image-major;
// statistics group g is the blocks [g, g + 1) * blocks_per_group -- the caller has checked that a group is whole blocks
// (cnuda_conv2d_stats_block / cnuda_dcn_v2_stats_block say how a geometry is cut).  The pass over x that
// bn_reduce_kernel<0> makes is replaced by a fold of blocks * C pairs.
extern "C" int cnuda_bn_train_forward_stats(const float* x, const float* stats, long long blocks_per_group, int rows, const float* gamma,
                                            const float* beta, const float* residual, float* y, float* save_mean,
                                            float* save_invstd, float* running_mean, float* running_var,
                                            long long* num_batches_tracked, float momentum, float eps, int relu, int B,
                                            int C, long long HW, int groups, void* workspace, size_t workspace_bytes,
                                            cnuda_stream_t stream) {
    // (y == nullptr: statistics, saved mean / invstd and the running statistics only -- no pass over x)
    CNUDA_REQUIRE(x && stats && gamma && beta && save_mean && save_invstd, "cnuda_bn_train_forward_stats: null pointer");
    CNUDA_REQUIRE(y || !residual, "cnuda_bn_train_forward_stats: a deferred apply has no residual");
    CNUDA_REQUIRE(B > 0 && C > 0 && HW > 0 && blocks_per_group > 0 && rows >= C, "cnuda_bn_train_forward_stats: bad geometry");
    CNUDA_REQUIRE(groups >= 1 && B % groups == 0, "cnuda_bn_train_forward_stats: batch %d not divisible into %d groups", B, groups);
    CNUDA_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "cnuda_bn_train_forward_stats: running stats");
    const int Bg = B / groups;
    const long long count = (long long)Bg * HW;
    CNUDA_REQUIRE(count > 1, "Expected more than 1 value per channel when training, got input size [%d, %d, %lld]", Bg, C, HW);
    CNUDA_REQUIRE(workspace && workspace_bytes >= cnuda_bn_workspace_bytes(B, C, HW),
                  "cnuda_bn_train_forward_stats: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const long long bpg = blocks_per_group;                   // (the caller: every group is exactly this many whole blocks)
    int spl = (int)((bpg + 255) / 256);                       // <= 256 blocks (16 per thread) per workgroup ...
    if (spl > 64) spl = 64;                                   // ... and at most 64 partials per group for the apply pass
    const int S = groups * spl;
    CNUDA_REQUIRE((size_t)C * S * 2 * sizeof(double) + 512 <= workspace_bytes, "cnuda_bn_train_forward_stats: workspace");
    double* partial = reinterpret_cast<double*>(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    CNUDA_LAUNCH(bn_fold_stats_kernel, dim3((C + 15) / 16, S), dim3(256), 0, st, stats, rows, bpg, spl, partial, C, S);
    const Split sp = pick_split(B, C, HW, Bg);                // (the apply pass's own grid: unchanged)
    const long long planes = (long long)B * C / sp.ips;
    CNUDA_LAUNCH(bn_apply_kernel, dim3((unsigned)planes, y ? plane_splits(planes, HW, kBnThreads * 4) : 1),
                       dim3(kBnThreads), 0, st, x, partial, S, count, momentum, eps, save_mean, save_invstd,
                       running_mean, running_var, num_batches_tracked, gamma, beta, residual, y, C, HW, relu, groups, Bg,
                       sp.per_plane, sp.ips, spl);
    return check_launch("cnuda_bn_train_forward_stats");
}

extern "C" int cnuda_bn_eval_forward(const float* x, const float* gamma, const float* beta, const float* running_mean,
                                     const float* running_var, const float* residual, float* y, float eps, int relu,
                                     int B, int C, long long HW, cnuda_stream_t stream) {
    CNUDA_REQUIRE(x && gamma && beta && running_mean && running_var && y, "cnuda_bn_eval_forward: null pointer");
    CNUDA_REQUIRE(B > 0 && C > 0 && HW > 0, "cnuda_bn_eval_forward: empty tensor");
    const long long planes = (long long)B * C;
    CNUDA_LAUNCH(bn_eval_kernel, dim3((unsigned)planes, plane_splits(planes, HW, kBnThreads)), dim3(kBnThreads), 0,
                       (hipStream_t)stream, x, running_mean, running_var, gamma, beta, residual, y, C, HW, eps, relu);
    return check_launch("cnuda_bn_eval_forward");
}

extern "C" int cnuda_bn_backward(const float* grad_y, const float* x, const float* y, const float* gamma,
                                 const float* beta, const float* save_mean, const float* save_invstd, float* grad_x,
                                 float* grad_residual,
                                 float* grad_gamma, float* grad_beta, int relu, int B, int C, long long HW, int groups,
                                 void* workspace, size_t workspace_bytes, cnuda_stream_t stream) {
    CNUDA_REQUIRE(grad_y && x && gamma && save_mean && save_invstd && grad_x && grad_gamma && grad_beta,
                  "cnuda_bn_backward: null pointer");
    CNUDA_REQUIRE(!relu || y || beta, "cnuda_bn_backward: relu backward needs the forward output (or beta)");
    CNUDA_REQUIRE(!(beta && grad_residual), "cnuda_bn_backward: the gate is recomputed from x only without a residual");
    CNUDA_REQUIRE(B > 0 && C > 0 && HW > 0, "cnuda_bn_backward: empty tensor");
    CNUDA_REQUIRE(groups >= 1 && B % groups == 0, "cnuda_bn_backward: batch %d not divisible into %d groups", B, groups);
    CNUDA_REQUIRE(workspace && workspace_bytes >= cnuda_bn_workspace_bytes(B, C, HW),
                  "cnuda_bn_backward: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const int Bg = B / groups;
    const long long count = (long long)Bg * HW;
    const Split sp = pick_split(B, C, HW, Bg);
    uintptr_t base = ((uintptr_t)workspace + 255) & ~(uintptr_t)255;
    double* partial = reinterpret_cast<double*>(base);
    CNUDA_LAUNCH(bn_reduce_kernel<1>, dim3(C, sp.S), dim3(kBnThreads), 0, st, x, grad_y, y, save_mean,
                       save_invstd, gamma, beta, partial, C, HW, sp.chunk, sp.per_plane, sp.ips, sp.S, relu, Bg);
    const long long planes = (long long)B * C / sp.ips;
    CNUDA_LAUNCH(bn_bwd_apply_kernel, dim3((unsigned)planes, plane_splits(planes, HW, kBnThreads * 4)),
                       dim3(kBnThreads), 0, st, grad_y, x, y, save_mean, save_invstd, gamma, beta, partial, sp.S, grad_gamma,
                       grad_beta, grad_x, grad_residual, C, HW, count, relu, groups, Bg, sp.per_plane, sp.ips);
    return check_launch("cnuda_bn_backward");
}
