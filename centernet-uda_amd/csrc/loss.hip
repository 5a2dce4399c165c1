// Detection and UDA losses for gfx950 (fp32 values, fp64 block reductions).
// Replaces, on device and without host synchronisation:
//   utils/tensor.py:5-7         _sigmoid (clamp(sigmoid, 1e-4, 1-1e-4))
//   losses/centernet.py:69-95   FocalLoss._neg_loss (incl. the num_pos == 0 branch, Q11)
//   losses/centernet.py:98-133  RegL1Loss (+ rotated), :192-223 PeriodicRegL1Loss, :136-189 KPSL1Loss
//   losses/entropy.py:10-28     EntropyLoss      losses/max_square.py:6-14 MaxSquareLoss
//   utils/image.py:121-124      entropy_map      losses/advent.py:10-18    BCE-with-logits vs constant
// Every loss is a pair (forward -> scalar(s) in device memory, backward ->
// gradient w.r.t. the logits given a device-resident upstream scalar).
#include "common.h"

namespace cnuda {
namespace {

constexpr int kT = 256;
constexpr float kLo = 1e-4f, kHi = 1.0f - 1e-4f;

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// ---------------- focal ----------------
// partial[blk*3 + {0,1,2}] = sum pos_loss, sum neg_loss, num_pos
__global__ __launch_bounds__(kT) void focal_fwd_kernel(const float* __restrict__ logits, const float* __restrict__ gt,
                                                       float* __restrict__ prob, double* __restrict__ partial,
                                                       long long n) {
    __shared__ double red[16];
    double ps = 0.0, ns = 0.0, np = 0.0;
    for (long long i = (long long)blockIdx.x * kT + threadIdx.x; i < n; i += (long long)gridDim.x * kT) {
        const float p = fminf(fmaxf(sigmoidf_(logits[i]), kLo), kHi);
        prob[i] = p;
        const float g = gt[i];
        if (g == 1.0f) {
            const float q = 1.0f - p;
            ps += (double)(logf(p) * (q * q));
            np += 1.0;
        } else if (g < 1.0f) {
            const float w = 1.0f - g;
            const float w2 = w * w;
            ns += (double)(logf(1.0f - p) * (p * p) * (w2 * w2));
        }
    }
    ps = block_sum(ps, red);
    ns = block_sum(ns, red);
    np = block_sum(np, red);
    if (threadIdx.x == 0) {
        partial[(size_t)blockIdx.x * 3 + 0] = ps;
        partial[(size_t)blockIdx.x * 3 + 1] = ns;
        partial[(size_t)blockIdx.x * 3 + 2] = np;
    }
}
// out[0] = loss, out[1] = num_pos (kept for backward)
__global__ void focal_finalize_kernel(const double* __restrict__ partial, int blocks, float weight,
                                      float* __restrict__ out) {
    __shared__ double red[16];
    double ps = 0.0, ns = 0.0, np = 0.0;
    for (int i = threadIdx.x; i < blocks; i += blockDim.x) {
        ps += partial[(size_t)i * 3];
        ns += partial[(size_t)i * 3 + 1];
        np += partial[(size_t)i * 3 + 2];
    }
    ps = block_sum(ps, red);
    ns = block_sum(ns, red);
    np = block_sum(np, red);
    if (threadIdx.x == 0) {
        const float pos = (float)ps, neg = (float)ns, n = (float)np;
        const float loss = (n == 0.0f) ? (0.0f - neg) : (0.0f - (pos + neg) / n);
        out[0] = loss * weight;
        out[1] = n;
    }
}
// dlogits = upstream * weight * dL/dp * dp/dx
__global__ void focal_bwd_kernel(const float* __restrict__ logits, const float* __restrict__ gt,
                                 const float* __restrict__ fwd_out, const float* __restrict__ upstream, float weight,
                                 float* __restrict__ grad, long long n) {
    const float np = fwd_out[1];
    const float scale = upstream[0] * weight * (np == 0.0f ? -1.0f : -1.0f / np);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float s = sigmoidf_(logits[i]);
        float gval = 0.0f;
        if (s >= kLo && s <= kHi) {   // clamp passes gradient inside [min, max] only
            const float p = s, q = 1.0f - s, g = gt[i];
            float dldp = 0.0f;
            if (g == 1.0f) {
                if (np != 0.0f) dldp = (q * q) / p - 2.0f * q * logf(p);
            } else if (g < 1.0f) {
                const float w = 1.0f - g, w2 = w * w;
                dldp = (2.0f * p * logf(q) - (p * p) / q) * (w2 * w2);
            }
            gval = scale * dldp * (s * q);
        }
        grad[i] = gval;
    }
}

// ---------------- gather + masked L1 (wh / reg / angle) ----------------
// mode 0: plain (ch 2 or 3: with 3 channels = rotated non-periodic: angle through clamped sigmoid)
// mode 1: periodic angle
// out[0] = loss, out[1] = denominator (expanded-mask sum + 1e-4)
// Single workgroup: B*M is a few thousand at most (M = max_detections).
// `ind` comes from the data loader (datasets/coco.py:211).  The reference's torch.gather device-asserts on an index
// outside [0, HW) (e.g. targets encoded for another output size); here such a row never touches memory: the forward
// kernels read cell 0 instead and poison the loss with NaN (loud, no host sync), the backward kernels skip the row.
__device__ __forceinline__ bool ind_ok(long long id, long long HW) { return id >= 0 && id < HW; }

__global__ __launch_bounds__(kT) void regl1_fwd_kernel(const float* __restrict__ feat, const unsigned char* __restrict__ mask,
                                                       const long long* __restrict__ ind, float* __restrict__ target,
                                                       int B, int M, int ch, int HW, int mode, float weight,
                                                       float angle_weight, float* __restrict__ out) {
    __shared__ double red[16];
    double s_wh = 0.0, s_a = 0.0, s_m = 0.0;
    for (int i = threadIdx.x; i < B * M; i += kT) {
        const int b = i / M;
        const float m = mask[i] ? 1.0f : 0.0f;
        long long id = ind[i];
        if (!ind_ok(id, HW)) { id = 0; s_wh = (double)NAN; }
        for (int c = 0; c < ch; ++c) {
            const float pred = feat[((size_t)b * ch + c) * HW + id] * m;
            const float tg = target[(size_t)i * ch + c] * m;
            target[(size_t)i * ch + c] = tg;   // in-place masking of the batch tensor (Q2)
            s_m += (double)m;
            if (ch == 3 && c == 2) {
                if (mode == 0) {
                    const float ps = fminf(fmaxf(sigmoidf_(pred), kLo), kHi);
                    const float tsig = sigmoidf_(tg);
                    target[(size_t)i * ch + c] = tsig;   // target[...,2:3].sigmoid_() is in place too (Q2)
                    const float ts = fminf(fmaxf(tsig, kLo), kHi);
                    s_a += (double)fabsf(ps - ts);
                } else {
                    const float pi = 3.14159265358979323846f;
                    const float pa = fminf(fmaxf(sigmoidf_(pred), kLo), kHi) * 2.0f * pi - pi;
                    const float ta = tg * (pi / 180.0f);
                    const float d = (pa - ta) - pi / 2.0f;
                    float r = fmodf(d, pi);
                    if (r != 0.0f && r < 0.0f) r += pi;   // torch.remainder: sign of the divisor
                    s_a += (double)fabsf(r - pi / 2.0f);
                }
            } else {
                s_wh += (double)fabsf(pred - tg);
            }
        }
    }
    s_wh = block_sum(s_wh, red);
    s_a = block_sum(s_a, red);
    s_m = block_sum(s_m, red);
    if (threadIdx.x == 0) {
        const float denom = (float)s_m + 1e-4f;
        float loss = (float)s_wh / denom * weight;
        if (ch == 3) loss += (float)s_a / denom * angle_weight;
        out[0] = loss;
        out[1] = denom;
    }
}
// scatter-add into a zero-initialised grad [B, ch, HW]; `target` is the (already masked / sigmoided) tensor
__global__ void regl1_bwd_kernel(const float* __restrict__ feat, const unsigned char* __restrict__ mask,
                                 const long long* __restrict__ ind, const float* __restrict__ target,
                                 const float* __restrict__ fwd_out, const float* __restrict__ upstream, int B, int M,
                                 int ch, int HW, int mode, float weight, float angle_weight, float* __restrict__ grad) {
    const float up = upstream[0] / fwd_out[1];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B * M * ch; i += gridDim.x * blockDim.x) {
        const int c = i % ch, bm = i / ch, b = bm / M;
        if (!mask[bm]) continue;
        const long long id = ind[bm];
        if (!ind_ok(id, HW)) continue;
        const float pred = feat[((size_t)b * ch + c) * HW + id];
        const float tg = target[i];
        float g;
        if (ch == 3 && c == 2) {
            const float s = sigmoidf_(pred);
            const float ds = (s >= kLo && s <= kHi) ? s * (1.0f - s) : 0.0f;
            const float ps = fminf(fmaxf(s, kLo), kHi);
            if (mode == 0) {
                const float ts = fminf(fmaxf(tg, kLo), kHi);   // target already holds sigmoid(target)
                const float d = ps - ts;
                g = (d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f)) * ds * angle_weight;
            } else {
                const float pi = 3.14159265358979323846f;
                const float pa = ps * 2.0f * pi - pi, ta = tg * (pi / 180.0f);
                float r = fmodf((pa - ta) - pi / 2.0f, pi);
                if (r != 0.0f && r < 0.0f) r += pi;
                const float e = r - pi / 2.0f;
                g = (e > 0.0f ? 1.0f : (e < 0.0f ? -1.0f : 0.0f)) * (2.0f * pi) * ds * angle_weight;
            }
        } else {
            const float d = pred - tg;
            g = (d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f)) * weight;
        }
        atomicAdd(grad + ((size_t)b * ch + c) * HW + id, g * up);
    }
}

// ---------------- softmax-over-channels losses ----------------
// f(v) = v*log2(v + 1e-30);   f'(v) = log2(v + eps) + v / ((v + eps) ln 2)
__device__ __forceinline__ float fent(float v) { return v * log2f(v + 1e-30f); }
__device__ __forceinline__ float dfent(float v) { return log2f(v + 1e-30f) + v / ((v + 1e-30f) * 0.6931471805599453f); }

// kind 0: entropy  sum_c f(v_c)         kind 1: max-squares  sum_c v_c^2
__global__ __launch_bounds__(kT) void softmax_loss_fwd_kernel(const float* __restrict__ x, double* __restrict__ partial,
                                                              int B, int C, long long HW, int kind) {
    __shared__ double red[16];
    double acc = 0.0;
    const long long total = (long long)B * HW;
    for (long long i = (long long)blockIdx.x * kT + threadIdx.x; i < total; i += (long long)gridDim.x * kT) {
        const long long b = i / HW, hw = i - b * HW;
        const float* px = x + (size_t)b * C * HW + hw;
        float mx = px[0];
        for (int c = 1; c < C; ++c) mx = fmaxf(mx, px[(size_t)c * HW]);
        float den = 0.0f;
        for (int c = 0; c < C; ++c) den += expf(px[(size_t)c * HW] - mx);
        float s = 0.0f;
        for (int c = 0; c < C; ++c) {
            const float v = expf(px[(size_t)c * HW] - mx) / den;
            s += kind == 0 ? fent(v) : v * v;
        }
        acc += (double)s;
    }
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = acc;
}
__global__ void scalar_finalize_kernel(const double* __restrict__ partial, int blocks, double scale,
                                       float* __restrict__ out) {
    __shared__ double red[16];
    double s = 0.0;
    for (int i = threadIdx.x; i < blocks; i += blockDim.x) s += partial[i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) out[0] = (float)(s * scale);
}
// dx_j = up*scale * v_j * (h_j - sum_i v_i h_i),  h = f'(v) (entropy) or 2v (max-squares)
__global__ void softmax_loss_bwd_kernel(const float* __restrict__ x, const float* __restrict__ upstream, float scale,
                                        float* __restrict__ grad, int B, int C, long long HW, int kind) {
    const float up = upstream[0] * scale;
    const long long total = (long long)B * HW;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long b = i / HW, hw = i - b * HW;
        const float* px = x + (size_t)b * C * HW + hw;
        float* pg = grad + (size_t)b * C * HW + hw;
        float mx = px[0];
        for (int c = 1; c < C; ++c) mx = fmaxf(mx, px[(size_t)c * HW]);
        float den = 0.0f;
        for (int c = 0; c < C; ++c) den += expf(px[(size_t)c * HW] - mx);
        float dot = 0.0f;
        for (int c = 0; c < C; ++c) {
            const float v = expf(px[(size_t)c * HW] - mx) / den;
            dot += v * (kind == 0 ? dfent(v) : 2.0f * v);
        }
        for (int c = 0; c < C; ++c) {
            const float v = expf(px[(size_t)c * HW] - mx) / den;
            pg[(size_t)c * HW] = up * v * ((kind == 0 ? dfent(v) : 2.0f * v) - dot);
        }
    }
}
// entropy_map: out_c = -f(v_c) / log2(C)
__global__ void entropy_map_fwd_kernel(const float* __restrict__ x, float* __restrict__ out, int B, int C,
                                       long long HW) {
    const float inv = 1.0f / log2f((float)C);
    const long long total = (long long)B * HW;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long b = i / HW, hw = i - b * HW;
        const float* px = x + (size_t)b * C * HW + hw;
        float* po = out + (size_t)b * C * HW + hw;
        float mx = px[0];
        for (int c = 1; c < C; ++c) mx = fmaxf(mx, px[(size_t)c * HW]);
        float den = 0.0f;
        for (int c = 0; c < C; ++c) den += expf(px[(size_t)c * HW] - mx);
        for (int c = 0; c < C; ++c) {
            const float v = expf(px[(size_t)c * HW] - mx) / den;
            po[(size_t)c * HW] = -fent(v) * inv;
        }
    }
}
// gx_j = -(1/log2 C) * v_j * (g_j f'_j - sum_i g_i f'_i v_i)
__global__ void entropy_map_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gout,
                                       float* __restrict__ grad, int B, int C, long long HW) {
    const float inv = 1.0f / log2f((float)C);
    const long long total = (long long)B * HW;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long b = i / HW, hw = i - b * HW;
        const size_t base = (size_t)b * C * HW + hw;
        const float* px = x + base;
        float mx = px[0];
        for (int c = 1; c < C; ++c) mx = fmaxf(mx, px[(size_t)c * HW]);
        float den = 0.0f;
        for (int c = 0; c < C; ++c) den += expf(px[(size_t)c * HW] - mx);
        float dot = 0.0f;
        for (int c = 0; c < C; ++c) {
            const float v = expf(px[(size_t)c * HW] - mx) / den;
            dot += gout[base + (size_t)c * HW] * dfent(v) * v;
        }
        for (int c = 0; c < C; ++c) {
            const float v = expf(px[(size_t)c * HW] - mx) / den;
            grad[base + (size_t)c * HW] = -inv * v * (gout[base + (size_t)c * HW] * dfent(v) - dot);
        }
    }
}

// ---------------- BCE with logits against a constant label, mean reduction ----------------
__global__ __launch_bounds__(kT) void bce_const_fwd_kernel(const float* __restrict__ x, float label, long long n,
                                                           float* __restrict__ out) {
    __shared__ double red[16];
    double acc = 0.0;
    for (long long i = threadIdx.x; i < n; i += kT) {
        const float v = x[i];
        acc += (double)(fmaxf(v, 0.0f) - v * label + log1pf(expf(-fabsf(v))));
    }
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) out[0] = (float)(acc / (double)n);
}
__global__ void bce_const_bwd_kernel(const float* __restrict__ x, float label, const float* __restrict__ upstream,
                                     long long n, float* __restrict__ grad) {
    const float up = upstream[0] / (float)n;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        grad[i] = (sigmoidf_(x[i]) - label) * up;
}

// y = clamp(sigmoid(x)); x <- sigmoid(x) in place (x.sigmoid_())
__global__ void sigmoid_clamp_kernel(float* __restrict__ x, float* __restrict__ y, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float s = sigmoidf_(x[i]);
        x[i] = s;
        y[i] = fminf(fmaxf(s, kLo), kHi);
    }
}

// ---------------- keypoint L1 (+ pair-distance term) ----------------
// losses/centernet.py:136-189 (KPSL1Loss).  feat [B,2J,HW]; mask [B,M,2J] u8; target [B,M,2J] is masked in
// place like the reference's `target *= mask`; pairs [P][2] keypoint indices (kps_weight_indices) or P = 0.
//   loss  = sum |pred*m - tg*m| / (sum m + 1e-4) * weight
//         + sum_{b,m,p} |d(pred_a, pred_b) - d(tg_a, tg_b)| / (sum m + 1e-4) * distance_weight
//   d = L1 norm (use_l1) or sqrt(|.|^2 + 1e4)  (the literal 1e4 of :178-179)
// out[0] = loss, out[1] = sum m + 1e-4.  Single workgroup (B*M*2J is tens of thousands at most).
__device__ __forceinline__ float kps_dist(float ax, float ay, float bx, float by, int use_l1) {
    const float dx = ax - bx, dy = ay - by;
    return use_l1 ? fabsf(dx) + fabsf(dy) : sqrtf((dx * dx + dy * dy) + 1e4f);
}
__global__ __launch_bounds__(kT) void kpsl1_fwd_kernel(const float* __restrict__ feat, const unsigned char* __restrict__ mask,
                                                       const long long* __restrict__ ind, float* __restrict__ target,
                                                       const int* __restrict__ pairs, int B, int M, int J, int HW, int P,
                                                       int use_l1, float weight, float distance_weight,
                                                       float* __restrict__ out) {
    __shared__ double red[16];
    const int ch = 2 * J;
    double s_l1 = 0.0, s_m = 0.0, s_d = 0.0;
    for (int i = threadIdx.x; i < B * M * ch; i += kT) {
        const int c = i % ch, bm = i / ch, b = bm / M;
        const float m = mask[i] ? 1.0f : 0.0f;
        long long id = ind[bm];
        if (!ind_ok(id, HW)) { id = 0; s_l1 = (double)NAN; }
        const float pred = feat[((size_t)b * ch + c) * HW + id] * m;
        const float tg = target[i] * m;
        target[i] = tg;                                   // in place, like the reference
        s_m += (double)m;
        s_l1 += (double)fabsf(pred - tg);
    }
    __syncthreads();                                      // the pair term reads the masked targets
    for (int i = threadIdx.x; i < B * M * P; i += kT) {
        const int pr = i % P, bm = i / P, b = bm / M;
        const int ja = pairs[2 * pr], jb = pairs[2 * pr + 1];
        const size_t fb = (size_t)b * ch * HW + (ind_ok(ind[bm], HW) ? ind[bm] : 0), tb = (size_t)bm * ch;
        auto pm = [&](int c) { return feat[fb + (size_t)c * HW] * (mask[tb + c] ? 1.0f : 0.0f); };
        const float pd = kps_dist(pm(2 * ja), pm(2 * ja + 1), pm(2 * jb), pm(2 * jb + 1), use_l1);
        const float td = kps_dist(target[tb + 2 * ja], target[tb + 2 * ja + 1], target[tb + 2 * jb], target[tb + 2 * jb + 1], use_l1);
        s_d += (double)fabsf(pd - td);
    }
    s_l1 = block_sum(s_l1, red);
    s_m = block_sum(s_m, red);
    s_d = block_sum(s_d, red);
    if (threadIdx.x == 0) {
        const float denom = (float)s_m + 1e-4f;
        float loss = (float)s_l1 / denom * weight;
        if (P > 0) loss += (float)s_d / denom * distance_weight;
        out[0] = loss;
        out[1] = denom;
    }
}
__device__ __forceinline__ float sgnf(float v) { return v > 0.0f ? 1.0f : (v < 0.0f ? -1.0f : 0.0f); }
// scatter-add into a zero-initialised grad [B, 2J, HW]; `target` already masked
__global__ void kpsl1_bwd_kernel(const float* __restrict__ feat, const unsigned char* __restrict__ mask,
                                 const long long* __restrict__ ind, const float* __restrict__ target,
                                 const int* __restrict__ pairs, const float* __restrict__ fwd_out,
                                 const float* __restrict__ upstream, int B, int M, int J, int HW, int P, int use_l1,
                                 float weight, float distance_weight, float* __restrict__ grad) {
    const int ch = 2 * J;
    const float up = upstream[0] / fwd_out[1];
    const int n1 = B * M * ch, n2 = B * M * P;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n1 + n2; i += gridDim.x * blockDim.x) {
        if (i < n1) {
            const int c = i % ch, bm = i / ch, b = bm / M;
            if (!mask[i] || !ind_ok(ind[bm], HW)) continue;
            const float pred = feat[((size_t)b * ch + c) * HW + ind[bm]];
            atomicAdd(grad + ((size_t)b * ch + c) * HW + ind[bm], sgnf(pred - target[i]) * weight * up);
        } else {
            const int k = i - n1, pr = k % P, bm = k / P, b = bm / M;
            const int ja = pairs[2 * pr], jb = pairs[2 * pr + 1];
            if (!ind_ok(ind[bm], HW)) continue;
            const size_t fb = (size_t)b * ch * HW + ind[bm], tb = (size_t)bm * ch;
            const int cs[4] = {2 * ja, 2 * ja + 1, 2 * jb, 2 * jb + 1};
            float mk[4], pv[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) { mk[t] = mask[tb + cs[t]] ? 1.0f : 0.0f; pv[t] = feat[fb + (size_t)cs[t] * HW] * mk[t]; }
            const float pd = kps_dist(pv[0], pv[1], pv[2], pv[3], use_l1);
            const float td = kps_dist(target[tb + cs[0]], target[tb + cs[1]], target[tb + cs[2]], target[tb + cs[3]], use_l1);
            const float e = sgnf(pd - td) * distance_weight * up;
            const float dx = pv[0] - pv[2], dy = pv[1] - pv[3];
            const float gx = use_l1 ? sgnf(dx) : dx / pd, gy = use_l1 ? sgnf(dy) : dy / pd;   // d(pd)/d(a) = -d(pd)/d(b)
            const float g[4] = {gx, gy, -gx, -gy};
#pragma unroll
            for (int t = 0; t < 4; ++t)
                if (mk[t] != 0.0f && e * g[t] != 0.0f) atomicAdd(grad + fb + (size_t)cs[t] * HW, e * g[t]);
        }
    }
}
// keypoint branch of the decode (backends/decode.py:44-51,69-74)
__global__ void decode_kps_kernel(const float* __restrict__ kps, const float* __restrict__ reg,
                                  const long long* __restrict__ inds, float* __restrict__ out, int B, int J, int K,
                                  int H, int W) {
    const int HW = H * W;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B * K * J; i += gridDim.x * blockDim.x) {
        const int j = i % J, bk = i / J, b = bk / K;
        const long long id = inds[bk];
        float xs = (float)(int)(id % W), ys = (float)(int)(id / W);
        if (reg) { xs += reg[((size_t)b * 2 + 0) * HW + id]; ys += reg[((size_t)b * 2 + 1) * HW + id]; }
        else { xs += 0.5f; ys += 0.5f; }
        out[(size_t)i * 2 + 0] = kps[((size_t)b * 2 * J + 2 * j) * HW + id] + xs;
        out[(size_t)i * 2 + 1] = kps[((size_t)b * 2 * J + 2 * j + 1) * HW + id] + ys;
    }
}
// feat [B,ch,HW], ind [B,M] -> out [B,M,ch]
__global__ void gather_feat_kernel(const float* __restrict__ feat, const long long* __restrict__ ind,
                                   float* __restrict__ out, int B, int M, int ch, long long HW) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B * M * ch; i += gridDim.x * blockDim.x) {
        const int c = i % ch, bm = i / ch, b = bm / M;
        out[i] = ind_ok(ind[bm], HW) ? feat[((size_t)b * ch + c) * HW + ind[bm]] : NAN;
    }
}

constexpr int kLossBlocks = 1024;

}  // namespace
}  // namespace cnuda

using namespace cnuda;

extern "C" size_t cnuda_loss_workspace_bytes(void) { return (size_t)kLossBlocks * 3 * sizeof(double) + 512; }

static double* ws_ptr(void* ws) { return reinterpret_cast<double*>(((uintptr_t)ws + 255) & ~(uintptr_t)255); }

extern "C" int cnuda_focal_loss_forward(const float* logits, const float* gt, float* prob, float* out2, long long n,
                                        float weight, void* workspace, size_t workspace_bytes, cnuda_stream_t stream) {
    CNUDA_REQUIRE(logits && gt && prob && out2 && n > 0, "cnuda_focal_loss_forward: bad arguments");
    CNUDA_REQUIRE(workspace && workspace_bytes >= cnuda_loss_workspace_bytes(), "cnuda_focal_loss_forward: workspace");
    hipStream_t st = (hipStream_t)stream;
    int blocks = stream_grid(n, kT);
    if (blocks > kLossBlocks) blocks = kLossBlocks;
    double* partial = ws_ptr(workspace);
    CNUDA_LAUNCH(focal_fwd_kernel, dim3(blocks), dim3(kT), 0, st, logits, gt, prob, partial, n);
    CNUDA_LAUNCH(focal_finalize_kernel, dim3(1), dim3(kT), 0, st, partial, blocks, weight, out2);
    return check_launch("cnuda_focal_loss_forward");
}
extern "C" int cnuda_focal_loss_backward(const float* logits, const float* gt, const float* out2,
                                         const float* upstream, float* grad_logits, long long n, float weight,
                                         cnuda_stream_t stream) {
    CNUDA_REQUIRE(logits && gt && out2 && upstream && grad_logits && n > 0, "cnuda_focal_loss_backward: bad arguments");
    CNUDA_LAUNCH(focal_bwd_kernel, dim3(stream_grid(n, kT)), dim3(kT), 0, (hipStream_t)stream, logits, gt, out2,
                       upstream, weight, grad_logits, n);
    return check_launch("cnuda_focal_loss_backward");
}

extern "C" int cnuda_reg_l1_forward(const float* feat, const uint8_t* mask, const int64_t* ind, float* target,
                                    float* out2, int B, int M, int ch, long long HW, int periodic, float weight,
                                    float angle_weight, cnuda_stream_t stream) {
    CNUDA_REQUIRE(feat && mask && ind && target && out2 && B > 0 && M > 0 && HW > 0, "cnuda_reg_l1_forward: bad arguments");
    CNUDA_REQUIRE(ch >= 1 && (!periodic || ch == 3), "cnuda_reg_l1_forward: periodic loss needs 3 channels, got %d", ch);
    CNUDA_LAUNCH(regl1_fwd_kernel, dim3(1), dim3(kT), 0, (hipStream_t)stream, feat, mask,
                       (const long long*)ind, target, B, M, ch, (int)HW, periodic ? 1 : 0, weight, angle_weight, out2);
    return check_launch("cnuda_reg_l1_forward");
}
extern "C" int cnuda_reg_l1_backward(const float* feat, const uint8_t* mask, const int64_t* ind, const float* target,
                                     const float* out2, const float* upstream, float* grad_feat, int B, int M, int ch,
                                     long long HW, int periodic, float weight, float angle_weight,
                                     cnuda_stream_t stream) {
    CNUDA_REQUIRE(feat && mask && ind && target && out2 && upstream && grad_feat && B > 0 && M > 0 && HW > 0,
                  "cnuda_reg_l1_backward: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(grad_feat, 0, (size_t)B * ch * HW * sizeof(float), st) != hipSuccess)
        return check_launch("cnuda_reg_l1_backward(memset)");
    CNUDA_LAUNCH(regl1_bwd_kernel, dim3(stream_grid((long long)B * M * ch, kT)), dim3(kT), 0, st, feat, mask,
                       (const long long*)ind, target, out2, upstream, B, M, ch, (int)HW, periodic ? 1 : 0, weight,
                       angle_weight, grad_feat);
    return check_launch("cnuda_reg_l1_backward");
}

extern "C" int cnuda_kps_l1_forward(const float* feat, const uint8_t* mask, const int64_t* ind, float* target,
                                    const int32_t* pairs, float* out2, int B, int M, int J, long long HW, int n_pairs,
                                    int use_l1, float weight, float distance_weight, cnuda_stream_t stream) {
    CNUDA_REQUIRE(feat && mask && ind && target && out2 && B > 0 && M > 0 && J > 0 && HW > 0 && n_pairs >= 0 &&
                      (n_pairs == 0 || pairs), "cnuda_kps_l1_forward: bad arguments");
    CNUDA_LAUNCH(kpsl1_fwd_kernel, dim3(1), dim3(kT), 0, (hipStream_t)stream, feat, mask, (const long long*)ind,
                       target, (const int*)pairs, B, M, J, (int)HW, n_pairs, use_l1 ? 1 : 0, weight, distance_weight, out2);
    return check_launch("cnuda_kps_l1_forward");
}
extern "C" int cnuda_kps_l1_backward(const float* feat, const uint8_t* mask, const int64_t* ind, const float* target,
                                     const int32_t* pairs, const float* out2, const float* upstream, float* grad_feat,
                                     int B, int M, int J, long long HW, int n_pairs, int use_l1, float weight,
                                     float distance_weight, cnuda_stream_t stream) {
    CNUDA_REQUIRE(feat && mask && ind && target && out2 && upstream && grad_feat && B > 0 && M > 0 && J > 0 && HW > 0 &&
                      n_pairs >= 0 && (n_pairs == 0 || pairs), "cnuda_kps_l1_backward: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(grad_feat, 0, (size_t)B * 2 * J * HW * sizeof(float), st) != hipSuccess)
        return check_launch("cnuda_kps_l1_backward(memset)");
    CNUDA_LAUNCH(kpsl1_bwd_kernel, dim3(stream_grid((long long)B * M * (2 * J + n_pairs), kT)), dim3(kT), 0, st, feat,
                       mask, (const long long*)ind, target, (const int*)pairs, out2, upstream, B, M, J, (int)HW, n_pairs,
                       use_l1 ? 1 : 0, weight, distance_weight, grad_feat);
    return check_launch("cnuda_kps_l1_backward");
}
extern "C" int cnuda_decode_keypoints(const float* kps, const float* reg, const int64_t* inds, float* out, int B, int J,
                                      int K, int H, int W, cnuda_stream_t stream) {
    CNUDA_REQUIRE(kps && inds && out && B > 0 && J > 0 && K > 0 && H > 0 && W > 0, "cnuda_decode_keypoints: bad arguments");
    CNUDA_LAUNCH(decode_kps_kernel, dim3(stream_grid((long long)B * K * J, kT)), dim3(kT), 0, (hipStream_t)stream,
                       kps, reg, (const long long*)inds, out, B, J, K, H, W);
    return check_launch("cnuda_decode_keypoints");
}

// kind 0: EntropyLoss  = -sum f(v) / (n*h*w*log2 c);  kind 1: MaxSquareLoss = -mean(v^2)/2
static double softmax_loss_scale(int kind, int B, int C, long long HW) {
    if (kind == 0) return -1.0 / ((double)B * (double)HW * (double)log2f((float)C));
    return -1.0 / (2.0 * (double)B * (double)C * (double)HW);
}
extern "C" int cnuda_softmax_loss_forward(const float* logits, float* out1, int B, int C, long long HW, int kind,
                                          void* workspace, size_t workspace_bytes, cnuda_stream_t stream) {
    CNUDA_REQUIRE(logits && out1 && B > 0 && C > 0 && HW > 0 && (kind == 0 || kind == 1),
                  "cnuda_softmax_loss_forward: bad arguments");
    CNUDA_REQUIRE(workspace && workspace_bytes >= cnuda_loss_workspace_bytes(), "cnuda_softmax_loss_forward: workspace");
    hipStream_t st = (hipStream_t)stream;
    int blocks = stream_grid((long long)B * HW, kT);
    if (blocks > kLossBlocks) blocks = kLossBlocks;
    double* partial = ws_ptr(workspace);
    CNUDA_LAUNCH(softmax_loss_fwd_kernel, dim3(blocks), dim3(kT), 0, st, logits, partial, B, C, HW, kind);
    CNUDA_LAUNCH(scalar_finalize_kernel, dim3(1), dim3(kT), 0, st, partial, blocks,
                       softmax_loss_scale(kind, B, C, HW), out1);
    return check_launch("cnuda_softmax_loss_forward");
}
extern "C" int cnuda_softmax_loss_backward(const float* logits, const float* upstream, float* grad_logits, int B, int C,
                                           long long HW, int kind, cnuda_stream_t stream) {
    CNUDA_REQUIRE(logits && upstream && grad_logits && B > 0 && C > 0 && HW > 0 && (kind == 0 || kind == 1),
                  "cnuda_softmax_loss_backward: bad arguments");
    CNUDA_LAUNCH(softmax_loss_bwd_kernel, dim3(stream_grid((long long)B * HW, kT)), dim3(kT), 0,
                       (hipStream_t)stream, logits, upstream, (float)softmax_loss_scale(kind, B, C, HW), grad_logits, B,
                       C, HW, kind);
    return check_launch("cnuda_softmax_loss_backward");
}
extern "C" int cnuda_entropy_map_forward(const float* logits, float* out, int B, int C, long long HW,
                                         cnuda_stream_t stream) {
    CNUDA_REQUIRE(logits && out && B > 0 && C > 0 && HW > 0, "cnuda_entropy_map_forward: bad arguments");
    CNUDA_LAUNCH(entropy_map_fwd_kernel, dim3(stream_grid((long long)B * HW, kT)), dim3(kT), 0,
                       (hipStream_t)stream, logits, out, B, C, HW);
    return check_launch("cnuda_entropy_map_forward");
}
extern "C" int cnuda_entropy_map_backward(const float* logits, const float* grad_out, float* grad_logits, int B, int C,
                                          long long HW, cnuda_stream_t stream) {
    CNUDA_REQUIRE(logits && grad_out && grad_logits && B > 0 && C > 0 && HW > 0, "cnuda_entropy_map_backward: bad arguments");
    CNUDA_LAUNCH(entropy_map_bwd_kernel, dim3(stream_grid((long long)B * HW, kT)), dim3(kT), 0,
                       (hipStream_t)stream, logits, grad_out, grad_logits, B, C, HW);
    return check_launch("cnuda_entropy_map_backward");
}
extern "C" int cnuda_bce_const_forward(const float* logits, float label, float* out1, long long n,
                                       cnuda_stream_t stream) {
    CNUDA_REQUIRE(logits && out1 && n > 0, "cnuda_bce_const_forward: bad arguments");
    CNUDA_LAUNCH(bce_const_fwd_kernel, dim3(1), dim3(kT), 0, (hipStream_t)stream, logits, label, n, out1);
    return check_launch("cnuda_bce_const_forward");
}
extern "C" int cnuda_bce_const_backward(const float* logits, float label, const float* upstream, float* grad_logits,
                                        long long n, cnuda_stream_t stream) {
    CNUDA_REQUIRE(logits && upstream && grad_logits && n > 0, "cnuda_bce_const_backward: bad arguments");
    CNUDA_LAUNCH(bce_const_bwd_kernel, dim3(stream_grid(n, kT)), dim3(kT), 0, (hipStream_t)stream, logits, label,
                       upstream, n, grad_logits);
    return check_launch("cnuda_bce_const_backward");
}
extern "C" int cnuda_sigmoid_clamp_(float* x, float* y, long long n, cnuda_stream_t stream) {
    CNUDA_REQUIRE(x && y && n > 0, "cnuda_sigmoid_clamp_: bad arguments");
    CNUDA_LAUNCH(sigmoid_clamp_kernel, dim3(stream_grid(n, kT)), dim3(kT), 0, (hipStream_t)stream, x, y, n);
    return check_launch("cnuda_sigmoid_clamp_");
}
extern "C" int cnuda_gather_feat(const float* feat, const int64_t* ind, float* out, int B, int M, int ch, long long HW,
                                 cnuda_stream_t stream) {
    CNUDA_REQUIRE(feat && ind && out && B > 0 && M > 0 && ch > 0 && HW > 0, "cnuda_gather_feat: bad arguments");
    CNUDA_LAUNCH(gather_feat_kernel, dim3(stream_grid((long long)B * M * ch, kT)), dim3(kT), 0,
                       (hipStream_t)stream, feat, (const long long*)ind, out, B, M, ch, HW);
    return check_launch("cnuda_gather_feat");
}
