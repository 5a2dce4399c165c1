// HBM-bound spatial / elementwise kernels of the DLA-34 graph (NCHW fp32):
//   max-pool k=s (Tree.downsample, backends/dla.py:202-203), depthwise
//   ConvTranspose2d k=2f s=f p=f/2 (IDAUp.up, dla.py:385-388), channel
//   concat / slice (Root, dla.py:162), elementwise add (IDAUp, dla.py:399),
//   ReLU / LeakyReLU gradients, and the offset|mask split + sigmoid of
//   DCN.forward (libs/DCNv2/dcn_v2.py:119-122).
#include "common.h"

namespace cnuda {
namespace {

constexpr int kT = 256;

// ---- max pool, window k x k, stride k, no padding ---------------------------------
__global__ void maxpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long long planes, int H, int W,
                                   int Ho, int Wo, int k) {
    const long long total = planes * Ho * Wo;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int ox = (int)(i % Wo), oy = (int)((i / Wo) % Ho);
        const long long p = i / ((long long)Wo * Ho);
        const float* src = x + (size_t)p * H * W + (size_t)(oy * k) * W + ox * k;
        float m = src[0];
        for (int dy = 0; dy < k; ++dy)
            for (int dx = 0; dx < k; ++dx) {
                const float v = src[dy * W + dx];
                if (v > m || v != v) m = v;   // first maximum wins, NaN propagates (ATen's rule)
            }
        y[i] = m;
    }
}
// gx is written completely (zeros where not the arg-max); one thread per output cell
// accumulate: gx already holds another consumer's share of x's gradient -- only the arg-max cell is touched (+= g)
__global__ void maxpool_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gy, float* __restrict__ gx,
                                   long long planes, int H, int W, int Ho, int Wo, int k, int accumulate) {
    const long long total = planes * Ho * Wo;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int ox = (int)(i % Wo), oy = (int)((i / Wo) % Ho);
        const long long p = i / ((long long)Wo * Ho);
        const size_t base = (size_t)p * H * W + (size_t)(oy * k) * W + ox * k;
        float m = x[base];
        int arg = 0;
        for (int dy = 0; dy < k; ++dy)
            for (int dx = 0; dx < k; ++dx) {
                const float v = x[base + dy * W + dx];
                if (v > m || v != v) { m = v; arg = dy * W + dx; }
            }
        const float g = gy[i];
        if (accumulate) {
            gx[base + arg] += g;
            continue;
        }
        for (int dy = 0; dy < k; ++dy)
            for (int dx = 0; dx < k; ++dx) gx[base + dy * W + dx] = (dy * W + dx == arg) ? g : 0.0f;
    }
}
// The 2 x 2 pools of DLA-34 (Tree.downsample: every level's input, 134 MB at 128 x 128 x 64 x 32) with 16-byte accesses: a
// thread owns TWO neighbouring windows -- one float4 of each of the two input rows -- instead of four 4-byte loads per window
// at an 8-byte lane stride (round 6: 1.6 -> 4+ TB/s).  W % 4 == 0, H even, 16-byte aligned tensors.  Same scan order and
// NaN rule as the scalar kernels: (0,0), (0,1), (1,0), (1,1), first maximum wins, NaN propagates.
__device__ __forceinline__ void pool2_pick(float a, float b, float c, float d, float& m, int& arg) {
    m = a; arg = 0;
    if (b > m || b != b) { m = b; arg = 1; }
    if (c > m || c != c) { m = c; arg = 2; }
    if (d > m || d != d) { m = d; arg = 3; }
}
__global__ void maxpool2_fwd_vec_kernel(const float4* __restrict__ x, float2* __restrict__ y, long long total /* planes * Ho * Wo / 2 */,
                                        int Wq /* W / 4 */) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        // (plane, output row) flattened: input rows 2 row, 2 row + 1.  32-bit division where the index allows (always, in DLA-34)
        const long long row = total < (1ll << 32) ? (long long)((unsigned)i / (unsigned)Wq) : i / Wq;
        const int q = (int)(i - row * Wq);
        const float4 r0 = x[(2 * row) * Wq + q], r1 = x[(2 * row + 1) * Wq + q];
        float m0, m1;
        int a0, a1;
        pool2_pick(r0.x, r0.y, r1.x, r1.y, m0, a0);
        pool2_pick(r0.z, r0.w, r1.z, r1.w, m1, a1);
        y[i] = make_float2(m0, m1);
    }
}
__global__ void maxpool2_bwd_vec_kernel(const float4* __restrict__ x, const float2* __restrict__ gy, float4* __restrict__ gx,
                                        long long total, int Wq, int accumulate) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long row = total < (1ll << 32) ? (long long)((unsigned)i / (unsigned)Wq) : i / Wq;
        const int q = (int)(i - row * Wq);
        const long long i0 = (2 * row) * Wq + q, i1 = (2 * row + 1) * Wq + q;
        const float4 r0 = x[i0], r1 = x[i1];
        const float2 g = gy[i];
        float m;
        int a0, a1;
        pool2_pick(r0.x, r0.y, r1.x, r1.y, m, a0);
        pool2_pick(r0.z, r0.w, r1.z, r1.w, m, a1);
        float4 o0 = make_float4(a0 == 0 ? g.x : 0.0f, a0 == 1 ? g.x : 0.0f, a1 == 0 ? g.y : 0.0f, a1 == 1 ? g.y : 0.0f);
        float4 o1 = make_float4(a0 == 2 ? g.x : 0.0f, a0 == 3 ? g.x : 0.0f, a1 == 2 ? g.y : 0.0f, a1 == 3 ? g.y : 0.0f);
        if (accumulate) {          // (gx holds another consumer's share: only the arg-max cells change)
            const float4 c0 = gx[i0], c1 = gx[i1];
            o0 = make_float4(a0 == 0 ? c0.x + g.x : c0.x, a0 == 1 ? c0.y + g.x : c0.y, a1 == 0 ? c0.z + g.y : c0.z, a1 == 1 ? c0.w + g.y : c0.w);
            o1 = make_float4(a0 == 2 ? c1.x + g.x : c1.x, a0 == 3 ? c1.y + g.x : c1.y, a1 == 2 ? c1.z + g.y : c1.z, a1 == 3 ? c1.w + g.y : c1.w);
        }
        gx[i0] = o0;
        gx[i1] = o1;
    }
}
inline bool pool2_vec_ok(const void* a, const void* b, const void* c, int H, int W, int k) {
    return k == 2 && (W & 3) == 0 && (H & 1) == 0 && ((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)c)) & 15) == 0;
}

// rows/cols of x not covered by any window (H % k != 0) get zero gradient
__global__ void maxpool_bwd_tail_kernel(float* __restrict__ gx, long long planes, int H, int W, int Hc, int Wc) {
    const long long total = planes * H * W;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int xx = (int)(i % W), yy = (int)((i / W) % H);
        if (yy >= Hc || xx >= Wc) gx[i] = 0.0f;
    }
}

// ---- max pool, window k x k, stride s, padding p (-inf): torchvision resnet stem pool k3 s2 p1
//      (backends/resnet.py:27-30 keeps it in `base`) --------------------------------------------
__device__ __forceinline__ int window_argmax(const float* __restrict__ src, int H, int W, int oy, int ox, int k, int s,
                                             int p, float& m) {
    // scan order (dy, dx) ascending, first maximum wins, NaN propagates: ATen's max_pool2d rule
    int arg = -1;
    m = -INFINITY;
    for (int dy = 0; dy < k; ++dy) {
        const int iy = oy * s - p + dy;
        if (iy < 0 || iy >= H) continue;
        for (int dx = 0; dx < k; ++dx) {
            const int ix = ox * s - p + dx;
            if (ix < 0 || ix >= W) continue;
            const float v = src[iy * W + ix];
            if (arg < 0 || v > m || v != v) { m = v; arg = iy * W + ix; }
        }
    }
    return arg;
}
__global__ void maxpool_win_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long long planes, int H,
                                       int W, int Ho, int Wo, int k, int s, int p) {
    const long long total = planes * Ho * Wo;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int ox = (int)(i % Wo), oy = (int)((i / Wo) % Ho);
        const long long pl = i / ((long long)Wo * Ho);
        float m;
        window_argmax(x + (size_t)pl * H * W, H, W, oy, ox, k, s, p, m);
        y[i] = m;
    }
}
// gather form (windows overlap when s < k): each input cell sums grad_y of the windows whose arg-max it is,
// in ascending window order -- deterministic, no atomics
__global__ void maxpool_win_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                       float* __restrict__ gx, long long planes, int H, int W, int Ho, int Wo, int k,
                                       int s, int p) {
    const long long total = planes * H * W;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int ix = (int)(i % W), iy = (int)((i / W) % H);
        const long long pl = i / ((long long)W * H);
        const float* src = x + (size_t)pl * H * W;
        const float* g = gy + (size_t)pl * Ho * Wo;
        // windows with oy*s - p <= iy <= oy*s - p + k - 1
        int oy0 = (iy + p - k + s) / s, ox0 = (ix + p - k + s) / s;   // ceil((iy+p-k+1)/s) for non-negative numerators
        if (iy + p - k + 1 <= 0) oy0 = 0;
        if (ix + p - k + 1 <= 0) ox0 = 0;
        const int oy1 = min((iy + p) / s, Ho - 1), ox1 = min((ix + p) / s, Wo - 1);
        float acc = 0.0f;
        for (int oy = oy0; oy <= oy1; ++oy)
            for (int ox = ox0; ox <= ox1; ++ox) {
                float m;
                if (window_argmax(src, H, W, oy, ox, k, s, p, m) == iy * W + ix) acc += g[oy * Wo + ox];
            }
        gx[i] = acc;
    }
}

// ---- depthwise transposed conv: k = 2f, stride f, padding f/2 (any k,s,p accepted) ----
// y[b,c,oy,ox] = sum_{ky,kx} x[b,c,(oy+p-ky)/s,(ox+p-kx)/s] * w[c,ky,kx]   (exact divisions only)
// grid = (output tiles of 256, planes): 32-bit index math, the channel's kernel in LDS; store-bound
__global__ __launch_bounds__(kT) void dwconvt_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ skip, float* __restrict__ y, int C,
                                                         int H, int W, int Ho, int Wo, int k, int s, int p) {
    __shared__ float ws[1024];
    const int pl = blockIdx.y, c = pl % C;
    for (int i = threadIdx.x; i < k * k; i += kT) ws[i] = w[(size_t)c * k * k + i];
    __syncthreads();
    const int o = blockIdx.x * kT + threadIdx.x;
    if (o >= Ho * Wo) return;
    const int oy = o / Wo, ox = o - oy * Wo;
    const float* xp = x + (size_t)pl * H * W;
    float acc = 0.0f;
    // valid ky: (oy + p - ky) % s == 0  ->  ky = (oy + p) % s + t*s
    for (int ky = (oy + p) % s; ky < k; ky += s) {
        const int iy = (oy + p - ky) / s;
        if (iy < 0 || iy >= H) continue;
        for (int kx = (ox + p) % s; kx < k; kx += s) {
            const int ix = (ox + p - kx) / s;
            if (ix < 0 || ix >= W) continue;
            acc += xp[iy * W + ix] * ws[ky * k + kx];
        }
    }
    y[(size_t)pl * Ho * Wo + o] = skip ? acc + skip[(size_t)pl * Ho * Wo + o] : acc;
}
// k = 2f, s = f, p = f/2 (IDAUp.up, dla.py:385-388) with compile-time f: every output pixel has exactly 2 x 2 taps and
// all index arithmetic is shifts; one thread produces four consecutive outputs of a row (Wo % 4 == 0).  `skip` (may be
// null) is IDAUp's other summand (dla.py:400-401 node(up(project(x)) + layers[i-1])): added after the taps, in the
// rounding order of the separate add, so the upsampled tensor is written once and never re-read
// ROWS consecutive output rows per thread (Ho % ROWS == 0): 8 KB of traffic per workgroup and row is little enough for
// the workgroup launch rate to bound the kernel (32,768 workgroups for a 134 MB map: 2.7 TB/s)
template <int F, int ROWS>
__global__ __launch_bounds__(kT) void dwconvt_fwd_f_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ skip, float* __restrict__ y, int C,
                                                           int H, int W) {
    constexpr int K = 2 * F, P = F / 2;
    __shared__ float ws[K * K];
    const int pl = blockIdx.y, c = pl % C;
    for (int i = threadIdx.x; i < K * K; i += kT) ws[i] = w[(size_t)c * K * K + i];
    __syncthreads();
    const int Ho = H * F, Wo = W * F;                       // (H-1)*F - 2*(F/2) + 2F
    const int q = blockIdx.x * kT + threadIdx.x;           // quad of outputs in a band of ROWS rows
    const int qw = Wo >> 2;
    if (q >= qw * (Ho / ROWS)) return;
    const int band = q / qw, ox0 = (q - band * qw) * 4;
    const float* xp = x + (size_t)pl * H * W;
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const int oy = band * ROWS + r;
        // rows: ky = (oy + P) % F + {0, F}  ->  iy = (oy + P) / F - {0, 1}
        const int ky0 = (oy + P) % F, iy0 = (oy + P) / F;
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ox = ox0 + j;
            const int kx0 = (ox + P) % F, ix0 = (ox + P) / F;
            float acc = 0.0f;
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const int iy = iy0 - a;
                if (iy < 0 || iy >= H) continue;
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const int ix = ix0 - b;
                    if (ix < 0 || ix >= W) continue;
                    acc += xp[iy * W + ix] * ws[(ky0 + a * F) * K + kx0 + b * F];
                }
            }
            o[j] = acc;
        }
        const size_t at = (size_t)pl * Ho * Wo + (size_t)oy * Wo + ox0;
        if (skip) {
            const float4 k4 = *reinterpret_cast<const float4*>(skip + at);
            o[0] += k4.x; o[1] += k4.y; o[2] += k4.z; o[3] += k4.w;
        }
        *reinterpret_cast<float4*>(y + at) = make_float4(o[0], o[1], o[2], o[3]);
    }
}
// k = 4, stride 2, padding 1 (every IDAUp layer of DLA-34 but one), H and W even: a thread produces a 4 x 4 block of
// outputs from the 4 x 4 block of inputs around it -- twelve loads (per input row one aligned 8-byte load and two edge
// scalars) instead of sixty-four bounds-checked ones; the taps of an output are added in the generic kernels' order
// (a = 0, 1 over rows, b = 0, 1 over columns), so the result is theirs bit for bit.
__global__ __launch_bounds__(kT) void dwconvt_fwd_k4s2_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                              const float* __restrict__ skip, float* __restrict__ y, int C,
                                                              int H, int W) {
    __shared__ float ws[16];
    const int pl = blockIdx.y, c = pl % C;
    if (threadIdx.x < 16) ws[threadIdx.x] = w[(size_t)c * 16 + threadIdx.x];
    __syncthreads();
    const int Wo = 2 * W, qw = W >> 1;                     // blocks of 4 x 4 outputs: (H / 2) x (W / 2)
    const int q = blockIdx.x * kT + threadIdx.x;
    if (q >= qw * (H >> 1)) return;
    const int br = q / qw, bc = q - br * qw;
    const float* xp = x + (size_t)pl * H * W;
    float xin[4][4];                                        // input rows 2 br - 1 .. + 2, columns 2 bc - 1 .. + 2
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int iy = 2 * br - 1 + a;
        xin[a][0] = xin[a][1] = xin[a][2] = xin[a][3] = 0.0f;
        if (iy >= 0 && iy < H) {
            const float* row = xp + (size_t)iy * W + 2 * bc;
            const float2 m = *reinterpret_cast<const float2*>(row);
            xin[a][1] = m.x; xin[a][2] = m.y;
            if (bc > 0) xin[a][0] = row[-1];
            if (2 * bc + 2 < W) xin[a][3] = row[2];
        }
    }
    // output j of the block (row or column alike): first tap input index r0[j] with kernel index k0[j], second tap
    // input index r0[j] - 1 with kernel index k0[j] + 2
    constexpr int r0[4] = {1, 2, 2, 3}, k0[4] = {1, 0, 1, 0};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float acc = 0.0f;
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    acc += xin[r0[j] - a][r0[i] - b] * ws[(k0[j] + 2 * a) * 4 + k0[i] + 2 * b];
            o[i] = acc;
        }
        const size_t at = (size_t)pl * (2 * H) * Wo + (size_t)(4 * br + j) * Wo + 4 * bc;
        if (skip) {
            const float4 k4 = *reinterpret_cast<const float4*>(skip + at);
            o[0] += k4.x; o[1] += k4.y; o[2] += k4.z; o[3] += k4.w;
        }
        *reinterpret_cast<float4*>(y + at) = make_float4(o[0], o[1], o[2], o[3]);
    }
}
// Both gradients read the same K x K window of gy around an input pixel:
//   gx[b,c,iy,ix]  = sum_{ky,kx} gy[b,c,iy*s-p+ky, ix*s-p+kx] * w[c,ky,kx]
//   gw[c,ky,kx]    = sum_{b,iy,ix} x[b,c,iy,ix] * gy[b,c,iy*s-p+ky, ix*s-p+kx]
// one workgroup per plane (c, b): gx written directly, K*K partial sums reduced in the block (fixed order)
// into part[c][b][K*K]; dwconvt_wsum_kernel adds the B partials in order (reproducible, no atomics).
template <int K>
__global__ __launch_bounds__(kT) void dwconvt_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ gy, float* __restrict__ gx,
                                                         float* __restrict__ part, int B, int C, int H, int W, int Ho,
                                                         int Wo, int s, int p) {
    __shared__ float ws[K * K];
    __shared__ float red[4][K * K];
    const int c = blockIdx.x, b = blockIdx.y;
    const size_t pl = (size_t)b * C + c;
    for (int i = threadIdx.x; i < K * K; i += kT) ws[i] = w[(size_t)c * K * K + i];
    __syncthreads();
    const float* xp = x + pl * H * W;
    const float* gp = gy + pl * Ho * Wo;
    float acc[K * K];
#pragma unroll
    for (int t = 0; t < K * K; ++t) acc[t] = 0.0f;
    for (int i = threadIdx.x; i < H * W; i += kT) {
        const int iy = i / W, ix = i - iy * W;
        const float xv = part ? xp[i] : 0.0f;
        const int oy0 = iy * s - p, ox0 = ix * s - p;
        float g = 0.0f;
#pragma unroll
        for (int ky = 0; ky < K; ++ky) {
            const int oy = oy0 + ky;
            const bool rok = oy >= 0 && oy < Ho;
            const float* row = gp + (rok ? oy : 0) * Wo;
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
                const int ox = ox0 + kx;
                const bool ok = rok && ox >= 0 && ox < Wo;
                const float v = ok ? row[ox] : 0.0f;
                g += v * ws[ky * K + kx];
                acc[ky * K + kx] += xv * v;
            }
        }
        if (gx) gx[pl * H * W + i] = g;
    }
    if (!part) return;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int t = 0; t < K * K; ++t) {
        const float v = wave_sum(acc[t]);
        if (lane == 0) red[wid][t] = v;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < K * K; t += kT)
        part[((size_t)c * B + b) * K * K + t] = (red[0][t] + red[1][t]) + (red[2][t] + red[3][t]);
}
// The same for IDAUp's commonest layer (k = 4, stride 2, padding 1; W even): a thread takes TWO neighbouring input pixels
// (ix even).  Their windows are the six output columns 2 ix - 1 .. 2 ix + 4 of four output rows: per row one aligned
// 16-byte load and two edge scalars instead of eight bounds-checked scalar loads -- the generic kernel above spends its
// time on index math (about 160 vector instructions per input pixel, 2 TB/s).
__global__ __launch_bounds__(kT) void dwconvt_bwd_k4s2_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                              const float* __restrict__ gy, float* __restrict__ gx,
                                                              float* __restrict__ part, int B, int C, int H, int W) {
    constexpr int K = 4;
    __shared__ float ws[K * K];
    __shared__ float red[4][K * K];
    const int c = blockIdx.x, b = blockIdx.y;
    const size_t pl = (size_t)b * C + c;
    for (int i = threadIdx.x; i < K * K; i += kT) ws[i] = w[(size_t)c * K * K + i];
    __syncthreads();
    const int Ho = 2 * H, Wo = 2 * W, W2 = W >> 1;
    const float* xp = x + pl * H * W;
    const float* gp = gy + pl * Ho * Wo;
    float acc[K * K];
#pragma unroll
    for (int t = 0; t < K * K; ++t) acc[t] = 0.0f;
    for (int i = threadIdx.x; i < H * W2; i += kT) {
        const int iy = i / W2, ix = (i - iy * W2) * 2;
        float x0 = 0.0f, x1 = 0.0f;
        if (part) { const float2 xv = *reinterpret_cast<const float2*>(xp + iy * W + ix); x0 = xv.x; x1 = xv.y; }
        float g0 = 0.0f, g1 = 0.0f;
#pragma unroll
        for (int ky = 0; ky < K; ++ky) {
            const int oy = 2 * iy - 1 + ky;
            float v[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};          // output columns 2 ix - 1 .. 2 ix + 4
            if (oy >= 0 && oy < Ho) {
                const float* row = gp + (size_t)oy * Wo + 2 * ix;
                const float4 m = *reinterpret_cast<const float4*>(row);
                v[1] = m.x; v[2] = m.y; v[3] = m.z; v[4] = m.w;
                if (ix > 0) v[0] = row[-1];
                if (2 * ix + 4 < Wo) v[5] = row[4];
            }
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {                       // pixel ix sees v[kx], pixel ix + 1 sees v[kx + 2]
                const float wk = ws[ky * K + kx];
                g0 += v[kx] * wk;
                g1 += v[kx + 2] * wk;
                acc[ky * K + kx] += x0 * v[kx];
                acc[ky * K + kx] += x1 * v[kx + 2];
            }
        }
        if (gx) *reinterpret_cast<float2*>(gx + pl * H * W + iy * W + ix) = make_float2(g0, g1);
    }
    if (!part) return;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int t = 0; t < K * K; ++t) {
        const float v = wave_sum(acc[t]);
        if (lane == 0) red[wid][t] = v;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < K * K; t += kT)
        part[((size_t)c * B + b) * K * K + t] = (red[0][t] + red[1][t]) + (red[2][t] + red[3][t]);
}
__global__ void dwconvt_wsum_kernel(const float* __restrict__ part, float* __restrict__ gw, int B, int C, int T) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= C * T) return;
    const int c = i / T, t = i - c * T;
    float acc = 0.0f;
    for (int b = 0; b < B; ++b) acc += part[((size_t)c * B + b) * T + t];
    gw[i] = acc;
}
// generic kernel sizes: one thread per input pixel / one workgroup per (c, tap)
__global__ void dwconvt_bwd_data_kernel(const float* __restrict__ gy, const float* __restrict__ w,
                                        float* __restrict__ gx, int B, int C, int H, int W, int Ho, int Wo, int k,
                                        int s, int p) {
    const long long total = (long long)B * C * H * W;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int ix = (int)(i % W), iy = (int)((i / W) % H);
        const long long pl = i / ((long long)W * H);
        const int c = (int)(pl % C);
        const float* gp = gy + (size_t)pl * Ho * Wo;
        const float* wp = w + (size_t)c * k * k;
        float acc = 0.0f;
        for (int ky = 0; ky < k; ++ky) {
            const int oy = iy * s - p + ky;
            if (oy < 0 || oy >= Ho) continue;
            for (int kx = 0; kx < k; ++kx) {
                const int ox = ix * s - p + kx;
                if (ox < 0 || ox >= Wo) continue;
                acc += gp[oy * Wo + ox] * wp[ky * k + kx];
            }
        }
        gx[i] = acc;
    }
}
__global__ __launch_bounds__(kT) void dwconvt_bwd_weight_kernel(const float* __restrict__ x,
                                                                const float* __restrict__ gy, float* __restrict__ gw,
                                                                int B, int C, int H, int W, int Ho, int Wo, int k, int s,
                                                                int p) {
    __shared__ float red[16];
    const int c = blockIdx.x, tap = blockIdx.y;
    const int ky = tap / k, kx = tap - ky * k;
    float acc = 0.0f;
    for (int b = 0; b < B; ++b) {
        const float* xp = x + ((size_t)b * C + c) * H * W;
        const float* gp = gy + ((size_t)b * C + c) * Ho * Wo;
        for (int i = threadIdx.x; i < H * W; i += kT) {
            const int iy = i / W, ix = i - iy * W;
            const int oy = iy * s - p + ky, ox = ix * s - p + kx;
            if (oy >= 0 && oy < Ho && ox >= 0 && ox < Wo) acc += xp[i] * gp[oy * Wo + ox];
        }
    }
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) gw[(size_t)c * k * k + tap] = acc;
}

// ---- elementwise ------------------------------------------------------------
// (no __restrict__: out may be a or b -- hip_runtime.fork sums gradients in place)
__global__ void add_kernel(const float* a, const float* b, float* out, long long n4, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        const float4 u = reinterpret_cast<const float4*>(a)[i], v = reinterpret_cast<const float4*>(b)[i];
        reinterpret_cast<float4*>(out)[i] = make_float4(u.x + v.x, u.y + v.y, u.z + v.z, u.w + v.w);
    }
    for (long long i = n4 * 4 + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x)
        out[i] = a[i] + b[i];
}
// gx = gy * (y > 0 ? 1 : slope)      (ReLU: slope 0; LeakyReLU(0.2): slope 0.2)
__global__ void act_bwd_kernel(const float* __restrict__ gy, const float* __restrict__ y, float* __restrict__ gx,
                               long long n, float slope) {
    const long long n4 = n >> 2;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        const float4 g = reinterpret_cast<const float4*>(gy)[i], v = reinterpret_cast<const float4*>(y)[i];
        reinterpret_cast<float4*>(gx)[i] = make_float4(v.x > 0.0f ? g.x : g.x * slope, v.y > 0.0f ? g.y : g.y * slope,
                                                       v.z > 0.0f ? g.z : g.z * slope, v.w > 0.0f ? g.w : g.w * slope);
    }
    for (long long i = n4 * 4 + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x)
        gx[i] = y[i] > 0.0f ? gy[i] : gy[i] * slope;
}
// The input gradient of a 1x1 convolution with a handful of output channels (a detection head's last layer,
// dla.py:474-483: 256 -> classes / 2 / 2) fused with the backward of the activation in front of it:
//   gh[b][m][p] = (hid[b][m][p] > 0 ? 1 : slope) * sum_c w[c][m] * go[b][c][p]            (c in increasing order)
// K = CO <= 8 is no GEMM: the kernel is one read of the hidden map (for the gate) and one write of its gradient, the
// CO gradient planes of a pixel quad stay in registers across the channel loop and the weights are scalar loads.
// grid (quads of pixels / kT, channel chunks, B)
template <int CO>
__global__ __launch_bounds__(kT) void conv1x1_dgrad_act_kernel(const float* __restrict__ go, const float* __restrict__ w,
                                                               const float* __restrict__ hid, float* __restrict__ gh,
                                                               int Ch, int ch_chunk, long long HW, float slope) {
    const long long q = (long long)blockIdx.x * kT + threadIdx.x;
    if (q * 4 >= HW) return;
    const int b = blockIdx.z, m0 = blockIdx.y * ch_chunk, m1 = min(m0 + ch_chunk, Ch);
    float4 g[CO];
#pragma unroll
    for (int c = 0; c < CO; ++c) g[c] = *reinterpret_cast<const float4*>(go + ((size_t)b * CO + c) * HW + q * 4);
    for (int m = m0; m < m1; ++m) {
        const size_t at = ((size_t)b * Ch + m) * HW + q * 4;
        const float4 h = *reinterpret_cast<const float4*>(hid + at);
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int c = 0; c < CO; ++c) {
            const float wc = w[(size_t)c * Ch + m];
            o.x += wc * g[c].x; o.y += wc * g[c].y; o.z += wc * g[c].z; o.w += wc * g[c].w;
        }
        *reinterpret_cast<float4*>(gh + at) = make_float4(h.x > 0.0f ? o.x : o.x * slope, h.y > 0.0f ? o.y : o.y * slope,
                                                          h.z > 0.0f ? o.z : o.z * slope, h.w > 0.0f ? o.w : o.w * slope);
    }
}
// Row-wise kernels: grid (chunks of a row, rows); a row is one (image, channel) plane of HW floats, so the
// channel bookkeeping is per workgroup and the inner loop is 16-byte copies when HW % 4 == 0.
// copy a [B, Cn, HW] block between tensors with Csrc / Cdst channels at channel offsets
__global__ void copy_channels_kernel(const float* __restrict__ src, float* __restrict__ dst, int rows, int Cn,
                                     long long HW, int Csrc, int src_off, int Cdst, int dst_off) {
    for (int row = blockIdx.y; row < rows; row += gridDim.y) {
        const int b = row / Cn, c = row - b * Cn;
        const float* s = src + ((size_t)b * Csrc + src_off + c) * HW;
        float* d = dst + ((size_t)b * Cdst + dst_off + c) * HW;
        if ((HW & 3) == 0) {
            for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < (HW >> 2); i += (long long)gridDim.x * blockDim.x)
                reinterpret_cast<float4*>(d)[i] = reinterpret_cast<const float4*>(s)[i];
        } else {
            for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += (long long)gridDim.x * blockDim.x)
                d[i] = s[i];
        }
    }
}
// om [B, 3T, HW] -> offset [B, 2T, HW] (channels 0..2T-1 unchanged) and mask = sigmoid(channels 2T..3T-1)
__global__ void split_offset_mask_kernel(const float* __restrict__ om, float* __restrict__ offset,
                                         float* __restrict__ mask, int rows, int T, long long HW) {
  for (int row = blockIdx.y; row < rows; row += gridDim.y) {
    const int b = row / (3 * T), c = row - b * 3 * T;
    const float* s = om + (size_t)row * HW;
    const bool is_mask = c >= 2 * T;
    float* d = is_mask ? mask + ((size_t)b * T + (c - 2 * T)) * HW : offset + ((size_t)b * 2 * T + c) * HW;
    if ((HW & 3) == 0) {
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < (HW >> 2); i += (long long)gridDim.x * blockDim.x) {
            float4 v = reinterpret_cast<const float4*>(s)[i];
            if (is_mask) {
                v.x = 1.0f / (1.0f + expf(-v.x)); v.y = 1.0f / (1.0f + expf(-v.y));
                v.z = 1.0f / (1.0f + expf(-v.z)); v.w = 1.0f / (1.0f + expf(-v.w));
            }
            reinterpret_cast<float4*>(d)[i] = v;
        }
    } else {
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += (long long)gridDim.x * blockDim.x)
            d[i] = is_mask ? 1.0f / (1.0f + expf(-s[i])) : s[i];
    }
  }
}
__global__ void split_offset_mask_bwd_kernel(const float* __restrict__ goff, const float* __restrict__ gmask,
                                             const float* __restrict__ mask, float* __restrict__ gom, int rows,
                                             int T, long long HW) {
  for (int row = blockIdx.y; row < rows; row += gridDim.y) {
    const int b = row / (3 * T), c = row - b * 3 * T;
    float* d = gom + (size_t)row * HW;
    const bool is_mask = c >= 2 * T;
    const float* g = is_mask ? gmask + ((size_t)b * T + (c - 2 * T)) * HW : goff + ((size_t)b * 2 * T + c) * HW;
    const float* m = mask + ((size_t)b * T + (is_mask ? c - 2 * T : 0)) * HW;
    if ((HW & 3) == 0) {
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < (HW >> 2); i += (long long)gridDim.x * blockDim.x) {
            float4 v = reinterpret_cast<const float4*>(g)[i];
            if (is_mask) {
                const float4 mm = reinterpret_cast<const float4*>(m)[i];
                v.x = v.x * mm.x * (1.0f - mm.x); v.y = v.y * mm.y * (1.0f - mm.y);
                v.z = v.z * mm.z * (1.0f - mm.z); v.w = v.w * mm.w * (1.0f - mm.w);
            }
            reinterpret_cast<float4*>(d)[i] = v;
        }
    } else {
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += (long long)gridDim.x * blockDim.x)
            d[i] = is_mask ? g[i] * m[i] * (1.0f - m[i]) : g[i];
    }
  }
}


// ---------------- depthwise convolution (groups == channels) ----------------
// torchvision MobileNetV2's `ConvBNReLU(hidden, hidden, stride, groups=hidden)`: y[b,c,oy,ox] =
// sum_{r,t} w[c,r,t] * x[b,c,oy*s-p+r, ox*s-p+t].  HBM-streaming: one thread per output / input element, the k*k
// weights of the plane in registers.  Weight gradient: one workgroup per (channel, image) reduces its plane to
// k*k partial sums (fp64 block reduction), a second kernel adds the images in order (reproducible).
template <int K>
__global__ __launch_bounds__(kT) void dwconv_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        float* __restrict__ y, int C, int H, int W, int Ho, int Wo,
                                                        int s, int p) {
    const int c = blockIdx.x % C;
    const size_t plane = blockIdx.x;
    float wk[K * K];
#pragma unroll
    for (int i = 0; i < K * K; ++i) wk[i] = w[(size_t)c * K * K + i];
    const float* xp = x + plane * H * W;
    float* yp = y + plane * Ho * Wo;
    for (int o = blockIdx.y * kT + threadIdx.x; o < Ho * Wo; o += gridDim.y * kT) {
        const int oy = o / Wo, ox = o - oy * Wo;
        float acc = 0.0f;
#pragma unroll
        for (int r = 0; r < K; ++r) {
            const int iy = oy * s - p + r;
            if (iy < 0 || iy >= H) continue;
#pragma unroll
            for (int t = 0; t < K; ++t) {
                const int ix = ox * s - p + t;
                if (ix >= 0 && ix < W) acc += wk[r * K + t] * xp[iy * W + ix];
            }
        }
        yp[o] = acc;
    }
}
template <int K>
__global__ __launch_bounds__(kT) void dwconv_bwd_data_kernel(const float* __restrict__ gy, const float* __restrict__ w,
                                                             float* __restrict__ gx, int C, int H, int W, int Ho, int Wo,
                                                             int s, int p) {
    const int c = blockIdx.x % C;
    const size_t plane = blockIdx.x;
    float wk[K * K];
#pragma unroll
    for (int i = 0; i < K * K; ++i) wk[i] = w[(size_t)c * K * K + i];
    const float* gp = gy + plane * Ho * Wo;
    float* xp = gx + plane * H * W;
    for (int i = blockIdx.y * kT + threadIdx.x; i < H * W; i += gridDim.y * kT) {
        const int iy = i / W, ix = i - iy * W;
        float acc = 0.0f;
#pragma unroll
        for (int r = 0; r < K; ++r) {
            const int ty = iy + p - r;
            if (ty < 0 || ty % s) continue;
            const int oy = ty / s;
            if (oy >= Ho) continue;
#pragma unroll
            for (int t = 0; t < K; ++t) {
                const int tx = ix + p - t;
                if (tx < 0 || tx % s) continue;
                const int ox = tx / s;
                if (ox < Wo) acc += wk[r * K + t] * gp[oy * Wo + ox];
            }
        }
        xp[i] = acc;
    }
}
// part[(c*B + b)*K*K + tap]
template <int K>
__global__ __launch_bounds__(kT) void dwconv_bwd_weight_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                               float* __restrict__ part, int B, int C, int H, int W,
                                                               int Ho, int Wo, int s, int p) {
    __shared__ double red[16];
    const int c = blockIdx.x, b = blockIdx.y;
    const float* xp = x + ((size_t)b * C + c) * H * W;
    const float* gp = gy + ((size_t)b * C + c) * Ho * Wo;
    float acc[K * K];
#pragma unroll
    for (int i = 0; i < K * K; ++i) acc[i] = 0.0f;
    for (int o = threadIdx.x; o < Ho * Wo; o += kT) {
        const int oy = o / Wo, ox = o - oy * Wo;
        const float g = gp[o];
#pragma unroll
        for (int r = 0; r < K; ++r) {
            const int iy = oy * s - p + r;
            if (iy < 0 || iy >= H) continue;
#pragma unroll
            for (int t = 0; t < K; ++t) {
                const int ix = ox * s - p + t;
                if (ix >= 0 && ix < W) acc[r * K + t] += g * xp[iy * W + ix];
            }
        }
    }
#pragma unroll
    for (int i = 0; i < K * K; ++i) {
        const double v = block_sum((double)acc[i], red);
        if (threadIdx.x == 0) part[((size_t)c * B + b) * K * K + i] = (float)v;
    }
}
}  // namespace
}  // namespace cnuda

using namespace cnuda;

extern "C" int cnuda_maxpool2d_forward(const float* x, float* y, int B, int C, int H, int W, int k,
                                       cnuda_stream_t stream) {
    CNUDA_REQUIRE(x && y && B > 0 && C > 0 && k > 0 && H >= k && W >= k, "cnuda_maxpool2d_forward: bad arguments");
    const int Ho = H / k, Wo = W / k;
    const long long planes = (long long)B * C;
    if (pool2_vec_ok(x, y, nullptr, H, W, k)) {
        const long long total = planes * Ho * (Wo / 2);
        CNUDA_LAUNCH(maxpool2_fwd_vec_kernel, dim3(stream_grid(total, kT)), dim3(kT), 0, (hipStream_t)stream,
                     reinterpret_cast<const float4*>(x), reinterpret_cast<float2*>(y), total, W / 4);
        return check_launch("cnuda_maxpool2d_forward");
    }
    CNUDA_LAUNCH(maxpool_fwd_kernel, dim3(stream_grid(planes * Ho * Wo, kT)), dim3(kT), 0, (hipStream_t)stream, x,
                       y, planes, H, W, Ho, Wo, k);
    return check_launch("cnuda_maxpool2d_forward");
}
extern "C" int cnuda_maxpool2d_backward(const float* x, const float* grad_y, float* grad_x, int B, int C, int H, int W,
                                        int k, cnuda_stream_t stream) {
    return cnuda_maxpool2d_backward_acc(x, grad_y, grad_x, 0, B, C, H, W, k, stream);
}
extern "C" int cnuda_maxpool2d_backward_acc(const float* x, const float* grad_y, float* grad_x, int accumulate, int B, int C,
                                            int H, int W, int k, cnuda_stream_t stream) {
    CNUDA_REQUIRE(x && grad_y && grad_x && B > 0 && C > 0 && k > 0 && H >= k && W >= k,
                  "cnuda_maxpool2d_backward: bad arguments");
    const int Ho = H / k, Wo = W / k;
    const long long planes = (long long)B * C;
    hipStream_t st = (hipStream_t)stream;
    if (pool2_vec_ok(x, grad_y, grad_x, H, W, k)) {
        const long long total = planes * Ho * (Wo / 2);
        CNUDA_LAUNCH(maxpool2_bwd_vec_kernel, dim3(stream_grid(total, kT)), dim3(kT), 0, st, reinterpret_cast<const float4*>(x),
                     reinterpret_cast<const float2*>(grad_y), reinterpret_cast<float4*>(grad_x), total, W / 4, accumulate);
        return check_launch("cnuda_maxpool2d_backward");
    }
    if (!accumulate && (Ho * k != H || Wo * k != W))
        CNUDA_LAUNCH(maxpool_bwd_tail_kernel, dim3(stream_grid(planes * H * W, kT)), dim3(kT), 0, st, grad_x,
                           planes, H, W, Ho * k, Wo * k);
    CNUDA_LAUNCH(maxpool_bwd_kernel, dim3(stream_grid(planes * Ho * Wo, kT)), dim3(kT), 0, st, x, grad_y, grad_x,
                       planes, H, W, Ho, Wo, k, accumulate);
    return check_launch("cnuda_maxpool2d_backward");
}

extern "C" int cnuda_maxpool2d_window_forward(const float* x, float* y, int B, int C, int H, int W, int k, int s, int p,
                                              cnuda_stream_t stream) {
    CNUDA_REQUIRE(x && y && B > 0 && C > 0 && k > 0 && s > 0 && p >= 0 && 2 * p <= k && H + 2 * p >= k && W + 2 * p >= k,
                  "cnuda_maxpool2d_window_forward: bad arguments (need 2*padding <= kernel <= padded size)");
    const int Ho = (H + 2 * p - k) / s + 1, Wo = (W + 2 * p - k) / s + 1;
    const long long planes = (long long)B * C;
    CNUDA_LAUNCH(maxpool_win_fwd_kernel, dim3(stream_grid(planes * Ho * Wo, kT)), dim3(kT), 0,
                       (hipStream_t)stream, x, y, planes, H, W, Ho, Wo, k, s, p);
    return check_launch("cnuda_maxpool2d_window_forward");
}
extern "C" int cnuda_maxpool2d_window_backward(const float* x, const float* grad_y, float* grad_x, int B, int C, int H,
                                               int W, int k, int s, int p, cnuda_stream_t stream) {
    CNUDA_REQUIRE(x && grad_y && grad_x && B > 0 && C > 0 && k > 0 && s > 0 && p >= 0 && 2 * p <= k && H + 2 * p >= k &&
                      W + 2 * p >= k,
                  "cnuda_maxpool2d_window_backward: bad arguments");
    const int Ho = (H + 2 * p - k) / s + 1, Wo = (W + 2 * p - k) / s + 1;
    const long long planes = (long long)B * C;
    CNUDA_LAUNCH(maxpool_win_bwd_kernel, dim3(stream_grid(planes * H * W, kT)), dim3(kT), 0, (hipStream_t)stream,
                       x, grad_y, grad_x, planes, H, W, Ho, Wo, k, s, p);
    return check_launch("cnuda_maxpool2d_window_backward");
}

extern "C" int cnuda_dwconvt2d_add_forward(const float* x, const float* w, const float* skip, float* y, int B, int C,
                                           int H, int W, int k, int s, int p, cnuda_stream_t stream) {
    CNUDA_REQUIRE(x && w && y && B > 0 && C > 0 && H > 0 && W > 0 && k > 0 && k <= 32 && s > 0 && p >= 0,
                  "cnuda_dwconvt2d_forward: bad arguments (kernel size 1..32)");
    const int Ho = (H - 1) * s - 2 * p + k, Wo = (W - 1) * s - 2 * p + k;
    CNUDA_REQUIRE(Ho > 0 && Wo > 0, "cnuda_dwconvt2d_forward: empty output");
    CNUDA_REQUIRE((long long)B * C <= 65535, "cnuda_dwconvt2d_forward: more than 65535 planes");
    // the vector kernels read `skip` and write `y` 16 bytes at a time (and read x 8 bytes at a time): a view at an
    // odd offset takes the scalar kernel
    const bool aligned16 = (((uintptr_t)x | (uintptr_t)y | (uintptr_t)skip) & 15) == 0;
    const bool upsample = k == 2 * s && p == s / 2 && (s == 2 || s == 4) && Wo % 4 == 0 && aligned16;   // IDAUp's bilinear-style layers
    const int rows = (upsample && Ho % 4 == 0 && (long long)Ho * Wo >= 4096) ? 4 : 1;
    const dim3 fgrid(ceil_div((long long)Ho * Wo / 4 / rows, kT), B * C);
    if (upsample && s == 2 && rows == 4 && (H & 1) == 0 && (W & 1) == 0)
        CNUDA_LAUNCH(dwconvt_fwd_k4s2_kernel, dim3(ceil_div((long long)(H / 2) * (W / 2), kT), B * C), dim3(kT), 0,
                           (hipStream_t)stream, x, w, skip, y, C, H, W);
    else if (upsample && s == 2 && rows == 4)
        CNUDA_LAUNCH((dwconvt_fwd_f_kernel<2, 4>), fgrid, dim3(kT), 0, (hipStream_t)stream, x, w, skip, y, C, H, W);
    else if (upsample && s == 2)
        CNUDA_LAUNCH((dwconvt_fwd_f_kernel<2, 1>), fgrid, dim3(kT), 0, (hipStream_t)stream, x, w, skip, y, C, H, W);
    else if (upsample && rows == 4)
        CNUDA_LAUNCH((dwconvt_fwd_f_kernel<4, 4>), fgrid, dim3(kT), 0, (hipStream_t)stream, x, w, skip, y, C, H, W);
    else if (upsample)
        CNUDA_LAUNCH((dwconvt_fwd_f_kernel<4, 1>), fgrid, dim3(kT), 0, (hipStream_t)stream, x, w, skip, y, C, H, W);
    else
        CNUDA_LAUNCH(dwconvt_fwd_kernel, dim3(ceil_div((long long)Ho * Wo, kT), B * C), dim3(kT), 0,
                           (hipStream_t)stream, x, w, skip, y, C, H, W, Ho, Wo, k, s, p);
    return check_launch("cnuda_dwconvt2d_forward");
}
extern "C" int cnuda_dwconvt2d_forward(const float* x, const float* w, float* y, int B, int C, int H, int W, int k,
                                       int s, int p, cnuda_stream_t stream) {
    return cnuda_dwconvt2d_add_forward(x, w, nullptr, y, B, C, H, W, k, s, p, stream);
}
extern "C" size_t cnuda_dwconv2d_workspace_bytes(int B, int C, int k) {
    return (size_t)(B > 0 ? B : 0) * (C > 0 ? C : 0) * k * k * sizeof(float) + 256;
}
static int dwconv_geom(int B, int C, int H, int W, int k, int s, int p, int& Ho, int& Wo, const char* who) {
    CNUDA_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && s > 0 && p >= 0, "%s: bad arguments", who);
    CNUDA_REQUIRE(k == 3 || k == 5, "%s: kernel size %d (3 and 5 are built)", who, k);
    CNUDA_REQUIRE((long long)B * C <= 2147483647ll, "%s: too many planes", who);
    Ho = (H + 2 * p - k) / s + 1;
    Wo = (W + 2 * p - k) / s + 1;
    CNUDA_REQUIRE(Ho > 0 && Wo > 0, "%s: empty output", who);
    return 0;
}
extern "C" int cnuda_dwconv2d_forward(const float* x, const float* w, float* y, int B, int C, int H, int W, int k,
                                      int s, int p, cnuda_stream_t stream) {
    int Ho, Wo;
    if (int rc = dwconv_geom(B, C, H, W, k, s, p, Ho, Wo, "cnuda_dwconv2d_forward")) return rc;
    CNUDA_REQUIRE(x && w && y, "cnuda_dwconv2d_forward: null pointer");
    const dim3 grid(B * C, ceil_div(ceil_div(Ho * Wo, kT), 4) > 0 ? ceil_div(ceil_div(Ho * Wo, kT), 4) : 1);
    if (k == 3) CNUDA_LAUNCH(dwconv_fwd_kernel<3>, grid, dim3(kT), 0, (hipStream_t)stream, x, w, y, C, H, W, Ho, Wo, s, p);
    else CNUDA_LAUNCH(dwconv_fwd_kernel<5>, grid, dim3(kT), 0, (hipStream_t)stream, x, w, y, C, H, W, Ho, Wo, s, p);
    return check_launch("cnuda_dwconv2d_forward");
}
extern "C" int cnuda_dwconv2d_backward(const float* x, const float* w, const float* grad_y, float* grad_x, float* grad_w,
                                       int B, int C, int H, int W, int k, int s, int p, void* workspace,
                                       size_t workspace_bytes, cnuda_stream_t stream) {
    int Ho, Wo;
    if (int rc = dwconv_geom(B, C, H, W, k, s, p, Ho, Wo, "cnuda_dwconv2d_backward")) return rc;
    CNUDA_REQUIRE(x && w && grad_y, "cnuda_dwconv2d_backward: null pointer");
    hipStream_t st = (hipStream_t)stream;
    if (grad_x) {
        const dim3 grid(B * C, ceil_div(ceil_div(H * W, kT), 4) > 0 ? ceil_div(ceil_div(H * W, kT), 4) : 1);
        if (k == 3) CNUDA_LAUNCH(dwconv_bwd_data_kernel<3>, grid, dim3(kT), 0, st, grad_y, w, grad_x, C, H, W, Ho, Wo, s, p);
        else CNUDA_LAUNCH(dwconv_bwd_data_kernel<5>, grid, dim3(kT), 0, st, grad_y, w, grad_x, C, H, W, Ho, Wo, s, p);
    }
    if (grad_w) {
        CNUDA_REQUIRE(workspace && workspace_bytes >= cnuda_dwconv2d_workspace_bytes(B, C, k), "cnuda_dwconv2d_backward: workspace too small");
        CNUDA_REQUIRE(C <= 2147483647 && B <= 65535, "cnuda_dwconv2d_backward: batch > 65535");
        float* part = (float*)workspace;
        if (k == 3) CNUDA_LAUNCH(dwconv_bwd_weight_kernel<3>, dim3(C, B), dim3(kT), 0, st, x, grad_y, part, B, C, H, W, Ho, Wo, s, p);
        else CNUDA_LAUNCH(dwconv_bwd_weight_kernel<5>, dim3(C, B), dim3(kT), 0, st, x, grad_y, part, B, C, H, W, Ho, Wo, s, p);
        CNUDA_LAUNCH(dwconvt_wsum_kernel, dim3(ceil_div((long long)C * k * k, 256)), dim3(256), 0, st, part, grad_w, B, C, k * k);
    }
    return check_launch("cnuda_dwconv2d_backward");
}
extern "C" size_t cnuda_dwconvt2d_workspace_bytes(int B, int C, int k) {
    return (size_t)(B > 0 ? B : 0) * (C > 0 ? C : 0) * k * k * sizeof(float) + 256;
}
extern "C" int cnuda_dwconvt2d_backward(const float* x, const float* w, const float* grad_y, float* grad_x,
                                        float* grad_w, int B, int C, int H, int W, int k, int s, int p,
                                        void* workspace, size_t workspace_bytes, cnuda_stream_t stream) {
    CNUDA_REQUIRE(x && w && grad_y && B > 0 && C > 0 && H > 0 && W > 0 && k > 0 && s > 0 && p >= 0,
                  "cnuda_dwconvt2d_backward: bad arguments");
    const int Ho = (H - 1) * s - 2 * p + k, Wo = (W - 1) * s - 2 * p + k;
    hipStream_t st = (hipStream_t)stream;
    if ((k == 4 || k == 8) && C <= 65535 && B <= 65535) {
        float* part = nullptr;
        if (grad_w) {
            CNUDA_REQUIRE(workspace && workspace_bytes >= cnuda_dwconvt2d_workspace_bytes(B, C, k),
                          "cnuda_dwconvt2d_backward: workspace too small");
            part = (float*)workspace;
        }
        if (k == 4 && s == 2 && p == 1 && (W & 1) == 0 &&
            (((uintptr_t)x | (uintptr_t)grad_y | (uintptr_t)grad_x) & 15) == 0)
            CNUDA_LAUNCH(dwconvt_bwd_k4s2_kernel, dim3(C, B), dim3(kT), 0, st, x, w, grad_y, grad_x, part, B, C, H, W);
        else if (k == 4)
            CNUDA_LAUNCH(dwconvt_bwd_kernel<4>, dim3(C, B), dim3(kT), 0, st, x, w, grad_y, grad_x, part, B, C, H, W,
                               Ho, Wo, s, p);
        else
            CNUDA_LAUNCH(dwconvt_bwd_kernel<8>, dim3(C, B), dim3(kT), 0, st, x, w, grad_y, grad_x, part, B, C, H, W,
                               Ho, Wo, s, p);
        if (grad_w)
            CNUDA_LAUNCH(dwconvt_wsum_kernel, dim3(ceil_div((long long)C * k * k, 256)), dim3(256), 0, st, part,
                               grad_w, B, C, k * k);
        return check_launch("cnuda_dwconvt2d_backward");
    }
    if (grad_x)
        CNUDA_LAUNCH(dwconvt_bwd_data_kernel, dim3(stream_grid((long long)B * C * H * W, kT)), dim3(kT), 0, st,
                           grad_y, w, grad_x, B, C, H, W, Ho, Wo, k, s, p);
    if (grad_w)
        CNUDA_LAUNCH(dwconvt_bwd_weight_kernel, dim3(C, k * k), dim3(kT), 0, st, x, grad_y, grad_w, B, C, H, W, Ho,
                           Wo, k, s, p);
    return check_launch("cnuda_dwconvt2d_backward");
}

extern "C" int cnuda_add(const float* a, const float* b, float* out, long long n, cnuda_stream_t stream) {
    CNUDA_REQUIRE(a && b && out && n > 0, "cnuda_add: bad arguments");
    const bool aligned = (((uintptr_t)a | (uintptr_t)b | (uintptr_t)out) & 15) == 0;
    const long long n4 = aligned ? n / 4 : 0;
    CNUDA_LAUNCH(add_kernel, dim3(stream_grid(n4 > 0 ? n4 : n, kT)), dim3(kT), 0, (hipStream_t)stream, a, b, out,
                       n4, n);
    return check_launch("cnuda_add");
}
extern "C" int cnuda_act_backward(const float* grad_y, const float* y, float* grad_x, long long n, float slope,
                                  cnuda_stream_t stream) {
    CNUDA_REQUIRE(grad_y && y && grad_x && n > 0, "cnuda_act_backward: bad arguments");
    CNUDA_LAUNCH(act_bwd_kernel, dim3(stream_grid(n, kT)), dim3(kT), 0, (hipStream_t)stream, grad_y, y, grad_x, n,
                       slope);
    return check_launch("cnuda_act_backward");
}
extern "C" int cnuda_conv1x1_backward_data_act(const float* grad_y, const float* weight, const float* hidden,
                                               float* grad_hidden, int B, int Co, int Ch, long long HW, float slope,
                                               cnuda_stream_t stream) {
    CNUDA_REQUIRE(grad_y && weight && hidden && grad_hidden && B > 0 && Ch > 0 && HW > 0,
                  "cnuda_conv1x1_backward_data_act: bad arguments");
    CNUDA_REQUIRE(Co >= 1 && Co <= 8, "cnuda_conv1x1_backward_data_act: %d output channels (1..8 are built)", Co);
    CNUDA_REQUIRE((HW & 3) == 0 && B <= 65535, "cnuda_conv1x1_backward_data_act: plane size %lld not a multiple of 4",
                  HW);
    CNUDA_REQUIRE((((uintptr_t)grad_y | (uintptr_t)hidden | (uintptr_t)grad_hidden) & 15) == 0,
                  "cnuda_conv1x1_backward_data_act: tensors must be 16-byte aligned");
    // enough workgroups for the chip: channel chunks of >= 32 until there are ~2048
    const long long quads = ceil_div(HW / 4, kT);
    int chunks = 1;
    while (chunks * 2 * 32 <= Ch && quads * B * chunks < 2048) chunks *= 2;
    const int ch_chunk = (int)ceil_div(Ch, chunks);
    const dim3 grid((unsigned)quads, (unsigned)ceil_div(Ch, ch_chunk), (unsigned)B);
    hipStream_t st = (hipStream_t)stream;
#define CNUDA_C1(CO) case CO: CNUDA_LAUNCH(conv1x1_dgrad_act_kernel<CO>, grid, dim3(kT), 0, st, grad_y, weight, \
                                                  hidden, grad_hidden, Ch, ch_chunk, HW, slope); break
    switch (Co) {
        CNUDA_C1(1); CNUDA_C1(2); CNUDA_C1(3); CNUDA_C1(4); CNUDA_C1(5); CNUDA_C1(6); CNUDA_C1(7); CNUDA_C1(8);
    }
#undef CNUDA_C1
    return check_launch("cnuda_conv1x1_backward_data_act");
}
extern "C" int cnuda_copy_channels(const float* src, float* dst, int B, int Cn, long long HW, int Csrc, int src_off,
                                   int Cdst, int dst_off, cnuda_stream_t stream) {
    CNUDA_REQUIRE(src && dst && B > 0 && Cn > 0 && HW > 0, "cnuda_copy_channels: bad arguments");
    CNUDA_REQUIRE(src_off >= 0 && dst_off >= 0 && src_off + Cn <= Csrc && dst_off + Cn <= Cdst,
                  "cnuda_copy_channels: channel range out of bounds");
    const int chunks = (int)((HW / 4 + kT * 4 - 1) / (kT * 4)) > 0 ? (int)((HW / 4 + kT * 4 - 1) / (kT * 4)) : 1;
    const long long rows = (long long)B * Cn;
    CNUDA_REQUIRE(rows < (1ll << 31), "cnuda_copy_channels: too many planes");
    CNUDA_LAUNCH(copy_channels_kernel, dim3(chunks, (unsigned)(rows < 65535 ? rows : 65535)), dim3(kT), 0,
                       (hipStream_t)stream, src, dst, (int)rows, Cn, HW, Csrc, src_off, Cdst, dst_off);
    return check_launch("cnuda_copy_channels");
}
extern "C" int cnuda_split_offset_mask(const float* om, float* offset, float* mask, int B, int taps, long long HW,
                                       cnuda_stream_t stream) {
    CNUDA_REQUIRE(om && offset && mask && B > 0 && taps > 0 && HW > 0, "cnuda_split_offset_mask: bad arguments");
    const int chunks = (int)((HW / 4 + kT * 4 - 1) / (kT * 4)) > 0 ? (int)((HW / 4 + kT * 4 - 1) / (kT * 4)) : 1;
    const long long rows = (long long)B * 3 * taps;
    CNUDA_REQUIRE(rows < (1ll << 31), "cnuda_split_offset_mask: too many planes");
    CNUDA_LAUNCH(split_offset_mask_kernel, dim3(chunks, (unsigned)(rows < 65535 ? rows : 65535)), dim3(kT), 0,
                       (hipStream_t)stream, om, offset, mask, (int)rows, taps, HW);
    return check_launch("cnuda_split_offset_mask");
}
extern "C" int cnuda_split_offset_mask_backward(const float* grad_offset, const float* grad_mask, const float* mask,
                                                float* grad_om, int B, int taps, long long HW, cnuda_stream_t stream) {
    CNUDA_REQUIRE(grad_offset && grad_mask && mask && grad_om && B > 0 && taps > 0 && HW > 0,
                  "cnuda_split_offset_mask_backward: bad arguments");
    const int chunks = (int)((HW / 4 + kT * 4 - 1) / (kT * 4)) > 0 ? (int)((HW / 4 + kT * 4 - 1) / (kT * 4)) : 1;
    const long long rows = (long long)B * 3 * taps;
    CNUDA_REQUIRE(rows < (1ll << 31), "cnuda_split_offset_mask_backward: too many planes");
    CNUDA_LAUNCH(split_offset_mask_bwd_kernel, dim3(chunks, (unsigned)(rows < 65535 ? rows : 65535)), dim3(kT), 0,
                       (hipStream_t)stream, grad_offset, grad_mask, mask, grad_om, (int)rows, taps, HW);
    return check_launch("cnuda_split_offset_mask_backward");
}
