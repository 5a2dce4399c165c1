// Modulated deformable convolution (DCNv2) for gfx950 -- forward and backward
// without ever materialising the `columns` buffer of the reference
// (libs/DCNv2/src/cuda/dcn_v2_cuda.cu:89-102 allocates B*C*9*Ho*Wo floats and
// round-trips them through HBM twice per direction).
//
//   forward : out[b,o,p] = bias[o] + sum_{tap,c} W[o,c,tap] * mask[b,tap,p] * bilinear(in[b,c], p, tap)
//             one implicit GEMM on the fp32 MFMA whose B operand is sampled on the fly.
//   backward: (1) column gradient GEMM  dcol[(tap,c), p] = sum_o W[o,c,tap] * gout[b,o,p]
//                 fused with its three consumers in the accumulator registers:
//                 grad_mask, grad_offset (summed over c in-kernel, written once)
//                 and the bilinear scatter into grad_input (fp32 atomics, as the
//                 reference does, dcn_v2_im2col_cuda.cu:238-252);
//             (2) grad_weight = gout x sampled-columns^T as a split-K GEMM with
//                 fixed-order slab reduction; (3) grad_bias = channel sums.
//             The whole batch goes through each kernel once (the reference loops
//             over samples on the host, dcn_v2_cuda.cu:259: 6*B launches/layer).
//
// Sampling rule (dcn_v2_im2col_cuda.cu:25-54,180): a tap contributes iff
// -1 < y < H and -1 < x < W; corners outside the plane read as 0.
#include "igemm.cuh"
#include "igemm_host.h"

namespace cnuda {
namespace {

struct DcnGeom {
    int B, C, H, W, Co, kh, kw, sh, sw, ph, pw, dh, dw, dg, Ho, Wo;
};

// Per-(pixel, tap) sampling state: four corner offsets inside a plane, the
// bilinear fractions and which corners exist.
struct Tap {
    int o00, o01, o10, o11;
    float lh, lw, hh, hw;  // fractions; hh = 1-lh, hw = 1-lw
    float mask;
    bool inside, c00, c01, c10, c11;
};

__device__ __forceinline__ Tap make_tap(const DcnGeom& g, const float* __restrict__ off_b,
                                        const float* __restrict__ mask_b, int grp, int tap, int oy, int ox) {
    const int T = g.kh * g.kw, HoWo = g.Ho * g.Wo, p = oy * g.Wo + ox;
    const int i = tap / g.kw, j = tap - i * g.kw;
    const float dy = off_b[((size_t)grp * 2 * T + 2 * tap) * HoWo + p];
    const float dx = off_b[((size_t)grp * 2 * T + 2 * tap + 1) * HoWo + p];
    Tap t;
    t.mask = mask_b[((size_t)grp * T + tap) * HoWo + p];
    const float h = (float)(oy * g.sh - g.ph + i * g.dh) + dy;
    const float w = (float)(ox * g.sw - g.pw + j * g.dw) + dx;
    t.inside = (h > -1.0f) && (w > -1.0f) && (h < (float)g.H) && (w < (float)g.W);
    const float hf = floorf(h), wf = floorf(w);
    const int h0 = (int)hf, w0 = (int)wf;
    t.lh = h - hf;
    t.lw = w - wf;
    t.hh = 1.0f - t.lh;
    t.hw = 1.0f - t.lw;
    const bool top = h0 >= 0, bot = h0 + 1 <= g.H - 1, lef = w0 >= 0, rig = w0 + 1 <= g.W - 1;
    t.c00 = t.inside && top && lef;
    t.c01 = t.inside && top && rig;
    t.c10 = t.inside && bot && lef;
    t.c11 = t.inside && bot && rig;
    const int base = h0 * g.W + w0;
    t.o00 = t.c00 ? base : 0;
    t.o01 = t.c01 ? base + 1 : 0;
    t.o10 = t.c10 ? base + g.W : 0;
    t.o11 = t.c11 ? base + g.W + 1 : 0;
    return t;
}

__device__ __forceinline__ void tap_corners(const Tap& t, const float* __restrict__ plane, float& v00, float& v01,
                                            float& v10, float& v11) {
    v00 = t.c00 ? plane[t.o00] : 0.0f;
    v01 = t.c01 ? plane[t.o01] : 0.0f;
    v10 = t.c10 ? plane[t.o10] : 0.0f;
    v11 = t.c11 ? plane[t.o11] : 0.0f;
}
// same association as dmcn_im2col_bilinear (im2col_cuda.cu:50-53)
__device__ __forceinline__ float tap_sample(const Tap& t, float v00, float v01, float v10, float v11) {
    return t.hh * t.hw * v00 + t.hh * t.lw * v01 + t.lh * t.hw * v10 + t.lh * t.lw * v11;
}

// ---------------------------------------------------------------------------
// forward: igemm_fwd_kernel loader (deformable_group == 1)
// ---------------------------------------------------------------------------
struct DcnFwdParams {
    DcnGeom g;
    const float *in, *off, *mask, *bias;
    float* out;
};

struct DcnFwdLoader {
    using Params = DcnFwdParams;
    const DcnGeom& g;
    const float *in_b, *off_b, *mask_b;
    int oy, ox, K;
    bool valid;
    __device__ DcnFwdLoader(const Params& p, long long n, bool n_valid) : g(p.g), valid(n_valid) {
        const int HoWo = g.Ho * g.Wo;
        const long long nn = n_valid ? n : 0;
        const int b = (int)(nn / HoWo), pp = (int)(nn - (long long)b * HoWo);
        oy = pp / g.Wo;
        ox = pp - oy * g.Wo;
        const int T = g.kh * g.kw;
        in_b = p.in + (size_t)b * g.C * g.H * g.W;
        off_b = p.off + (size_t)b * 2 * T * HoWo;
        mask_b = p.mask + (size_t)b * T * HoWo;
        K = T * g.C;
    }
    __device__ __forceinline__ void load(int k0, int ksub, float (&v)[8]) {
        int cur = -1;
        Tap t;
        const int HW = g.H * g.W;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = k0 + ksub + 2 * j;
            float r = 0.0f;
            if (valid && k < K) {
                const int tap = k / g.C, c = k - tap * g.C;
                if (tap != cur) { t = make_tap(g, off_b, mask_b, 0, tap, oy, ox); cur = tap; }
                if (t.inside) {
                    float v00, v01, v10, v11;
                    tap_corners(t, in_b + (size_t)c * HW, v00, v01, v10, v11);
                    r = tap_sample(t, v00, v01, v10, v11) * t.mask;
                }
            }
            v[j] = r;
        }
    }
    struct Out {
        float* base;
        int HoWo;
        __device__ Out(const Params& p, long long n) {
            HoWo = p.g.Ho * p.g.Wo;
            const int b = (int)(n / HoWo), pp = (int)(n - (long long)b * HoWo);
            base = p.out + (size_t)b * p.g.Co * HoWo + pp;
        }
        __device__ __forceinline__ void store(const Params& p, int m, float v) {
            base[(size_t)m * HoWo] = v + p.bias[m];
        }
    };
};

// ---------------------------------------------------------------------------
// backward (1): fused column-gradient GEMM + grad_offset / grad_mask / grad_input
// Workgroup = (pixel tile of 128, tap); loops over 64-channel tiles of C.
// ---------------------------------------------------------------------------
struct DcnBwdParams {
    DcnGeom g;
    const float *in, *off, *mask, *gout;
    float *gin, *goff, *gmask;
};

__global__ __launch_bounds__(IG_THREADS) void dcn_bwd_data_kernel(DcnBwdParams p, const float* __restrict__ A2,
                                                                 int Mp2, int Kp, int Cpad, long long N, int n_tiles) {
    constexpr int BM = 64;
    using TL = IgTile<BM>;
    __shared__ float As[IG_BK * BM];
    __shared__ float Bs[IG_BK * IG_BN];
    __shared__ float red[3][IG_BN];
    const DcnGeom& g = p.g;
    const int T = g.kh * g.kw, HoWo = g.Ho * g.Wo, HW = g.H * g.W;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wg = xcd_remap(blockIdx.x, n_tiles * T);
    const int tap = wg % T;
    const long long n0 = (long long)(wg / T) * IG_BN;
    const int wm = wid / TL::WN, wn = wid % TL::WN;
    const int wm_off = wm * 32, wn_off = wn * 64;

    // B-operand (gout) staging coordinates
    const int nl = tid & (IG_BN - 1), ksub = tid >> 7;
    const long long nb = n0 + nl;
    const bool nb_valid = nb < N;
    const int bb = nb_valid ? (int)(nb / HoWo) : 0;
    const int pb = nb_valid ? (int)(nb - (long long)bb * HoWo) : 0;
    const float* gout_b = p.gout + (size_t)bb * g.Co * HoWo + pb;

    // epilogue coordinates: this lane's two pixels
    Tap tp[2];
    bool pv[2];
    const float* in_b[2];
    float* gin_b[2];
    float gm[2] = {0.f, 0.f}, gh[2] = {0.f, 0.f}, gw[2] = {0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const long long n = n0 + wn_off + j * 32 + (lane & 31);
        pv[j] = n < N;
        const long long nn = pv[j] ? n : 0;
        const int b = (int)(nn / HoWo), pp = (int)(nn - (long long)b * HoWo);
        const int oy = pp / g.Wo, ox = pp - oy * g.Wo;
        tp[j] = make_tap(g, p.off + (size_t)b * 2 * T * HoWo, p.mask + (size_t)b * T * HoWo, 0, tap, oy, ox);
        in_b[j] = p.in + (size_t)b * g.C * HW;
        gin_b[j] = p.gin + (size_t)b * g.C * HW;
    }

    for (int c0 = 0; c0 < Cpad; c0 += BM) {
        f32x16 acc[1][2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[0][j][r] = 0.0f;
        const int mbase = tap * Cpad + c0;
        for (int k0 = 0; k0 < Kp; k0 += IG_BK) {
            float ra[BM / 16], rb[8];
            ig_load_a<BM>(A2, Mp2, k0, mbase, tid, ra);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int o = k0 + ksub + 2 * j;
                rb[j] = (nb_valid && o < g.Co) ? gout_b[(size_t)o * HoWo] : 0.0f;
            }
            __syncthreads();
            ig_store_a<BM>(As, tid, ra);
#pragma unroll
            for (int j = 0; j < 8; ++j) Bs[(ksub + 2 * j) * IG_BN + nl] = rb[j];
            __syncthreads();
            ig_mma_chunk<BM>(As, Bs, acc, wm_off, wn_off, lane);
        }
        // consume the dcol tile straight from the accumulators
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (!pv[j] || !tp[j].inside) continue;
            const Tap& t = tp[j];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = c0 + wm_off + mfma_row(r, lane);
                if (c >= g.C) continue;
                const float d = acc[0][j][r];
                const float* plane = in_b[j] + (size_t)c * HW;
                float v00, v01, v10, v11;
                tap_corners(t, plane, v00, v01, v10, v11);
                gm[j] += d * tap_sample(t, v00, v01, v10, v11);
                const float dm = d * t.mask;
                // d(sample)/dy and /dx (dmcn_get_coordinate_weight, im2col_cuda.cu:82-123)
                gh[j] += (-t.hw * v00 - t.lw * v01 + t.hw * v10 + t.lw * v11) * dm;
                gw[j] += (-t.hh * v00 + t.hh * v01 - t.lh * v10 + t.lh * v11) * dm;
                float* gplane = gin_b[j] + (size_t)c * HW;
                if (t.c00) atomicAdd(gplane + t.o00, t.hh * t.hw * dm);
                if (t.c01) atomicAdd(gplane + t.o01, t.hh * t.lw * dm);
                if (t.c10) atomicAdd(gplane + t.o10, t.lh * t.hw * dm);
                if (t.c11) atomicAdd(gplane + t.o11, t.lh * t.lw * dm);
            }
        }
    }
    // sum over the two row halves of the lane pair (l, l+32), then over the two
    // wave rows through LDS; one plain store per (pixel, tap).
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        gm[j] += __shfl_xor(gm[j], 32, 64);
        gh[j] += __shfl_xor(gh[j], 32, 64);
        gw[j] += __shfl_xor(gw[j], 32, 64);
    }
    __syncthreads();
    if (wm == 1 && lane < 32) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = wn_off + j * 32 + lane;
            red[0][col] = gm[j];
            red[1][col] = gh[j];
            red[2][col] = gw[j];
        }
    }
    __syncthreads();
    if (wm == 0 && lane < 32) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = wn_off + j * 32 + lane;
            const long long n = n0 + col;
            if (n >= N) continue;
            const int b = (int)(n / HoWo), pp = (int)(n - (long long)b * HoWo);
            p.gmask[((size_t)b * T + tap) * HoWo + pp] = gm[j] + red[0][col];
            p.goff[((size_t)b * 2 * T + 2 * tap) * HoWo + pp] = gh[j] + red[1][col];
            p.goff[((size_t)b * 2 * T + 2 * tap + 1) * HoWo + pp] = gw[j] + red[2][col];
        }
    }
}

// ---------------------------------------------------------------------------
// backward (2): grad_weight, igemm_wgrad_kernel loader
// ---------------------------------------------------------------------------
struct DcnWParams {
    DcnGeom g;
    const float *in, *off, *mask, *gout;
};
struct DcnWLoader {
    using Params = DcnWParams;
    const Params& p;
    __device__ DcnWLoader(const Params& pp) : p(pp) {}
    __device__ __forceinline__ void load_g(long long n, bool valid, int m0, int msub, float (&v)[16]) {
        const DcnGeom& g = p.g;
        const int HoWo = g.Ho * g.Wo;
        const long long nn = valid ? n : 0;
        const int b = (int)(nn / HoWo), pp = (int)(nn - (long long)b * HoWo);
        const float* base = p.gout + (size_t)b * g.Co * HoWo + pp;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int m = m0 + msub + 4 * i;
            v[i] = (valid && m < g.Co) ? base[(size_t)m * HoWo] : 0.0f;
        }
    }
    __device__ __forceinline__ void load_b(long long n, bool valid, int j0, int jsub, float (&v)[16]) {
        const DcnGeom& g = p.g;
        const int HoWo = g.Ho * g.Wo, HW = g.H * g.W, T = g.kh * g.kw, K = T * g.C;
        const long long nn = valid ? n : 0;
        const int b = (int)(nn / HoWo), pp = (int)(nn - (long long)b * HoWo);
        const int oy = pp / g.Wo, ox = pp - oy * g.Wo;
        const float* in_b = p.in + (size_t)b * g.C * HW;
        const float* off_b = p.off + (size_t)b * 2 * T * HoWo;
        const float* mask_b = p.mask + (size_t)b * T * HoWo;
        int cur = -1;
        Tap t;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int k = j0 + jsub + 4 * i;
            float r = 0.0f;
            if (valid && k < K) {
                const int tap = k / g.C, c = k - tap * g.C;
                if (tap != cur) { t = make_tap(g, off_b, mask_b, 0, tap, oy, ox); cur = tap; }
                if (t.inside) {
                    float v00, v01, v10, v11;
                    tap_corners(t, in_b + (size_t)c * HW, v00, v01, v10, v11);
                    r = tap_sample(t, v00, v01, v10, v11) * t.mask;
                }
            }
            v[i] = r;
        }
    }
};

// ---------------------------------------------------------------------------
// deformable_group > 1: straightforward kernels (no backend of the reference
// uses it -- dla.py:358-368 and mobilenetv2.py:147 pass deformable_groups=1 --
// but `_ext` accepts it, testcpu.py:169-180).
// ---------------------------------------------------------------------------
struct DcnNaiveParams {
    DcnGeom g;
    const float *in, *weight, *bias, *off, *mask, *gout;
    float *out, *gin, *goff, *gmask, *gw;
};

__global__ void dcn_naive_fwd_kernel(DcnNaiveParams p) {
    const DcnGeom& g = p.g;
    const int T = g.kh * g.kw, HoWo = g.Ho * g.Wo, HW = g.H * g.W, cpg = g.C / g.dg;
    const long long total = (long long)g.B * g.Co * HoWo;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int pp = (int)(i % HoWo), o = (int)((i / HoWo) % g.Co), b = (int)(i / ((long long)HoWo * g.Co));
        const int oy = pp / g.Wo, ox = pp - oy * g.Wo;
        float s = p.bias[o];
        for (int grp = 0; grp < g.dg; ++grp)
            for (int tap = 0; tap < T; ++tap) {
                const Tap t = make_tap(g, p.off + (size_t)b * g.dg * 2 * T * HoWo,
                                       p.mask + (size_t)b * g.dg * T * HoWo, grp, tap, oy, ox);
                if (!t.inside) continue;
                for (int cc = 0; cc < cpg; ++cc) {
                    const int c = grp * cpg + cc;
                    float v00, v01, v10, v11;
                    tap_corners(t, p.in + ((size_t)b * g.C + c) * HW, v00, v01, v10, v11);
                    s += p.weight[((size_t)o * g.C + c) * T + tap] * (tap_sample(t, v00, v01, v10, v11) * t.mask);
                }
            }
        p.out[i] = s;
    }
}

// one thread per column element (b, c, tap, p); every output is accumulated
// with atomics into zero-initialised buffers.
__global__ void dcn_naive_bwd_kernel(DcnNaiveParams p) {
    const DcnGeom& g = p.g;
    const int T = g.kh * g.kw, HoWo = g.Ho * g.Wo, HW = g.H * g.W, cpg = g.C / g.dg;
    const long long total = (long long)g.B * g.C * T * HoWo;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int pp = (int)(i % HoWo), tap = (int)((i / HoWo) % T), c = (int)((i / ((long long)HoWo * T)) % g.C);
        const int b = (int)(i / ((long long)HoWo * T * g.C));
        const int oy = pp / g.Wo, ox = pp - oy * g.Wo, grp = c / cpg;
        const Tap t = make_tap(g, p.off + (size_t)b * g.dg * 2 * T * HoWo, p.mask + (size_t)b * g.dg * T * HoWo, grp,
                               tap, oy, ox);
        if (!t.inside) continue;
        float v00, v01, v10, v11;
        tap_corners(t, p.in + ((size_t)b * g.C + c) * HW, v00, v01, v10, v11);
        const float smp = tap_sample(t, v00, v01, v10, v11);
        const float* go = p.gout + (size_t)b * g.Co * HoWo + pp;
        float d = 0.0f;
        for (int o = 0; o < g.Co; ++o) {
            const float gv = go[(size_t)o * HoWo];
            d += p.weight[((size_t)o * g.C + c) * T + tap] * gv;
            atomicAdd(p.gw + ((size_t)o * g.C + c) * T + tap, gv * (smp * t.mask));
        }
        const float dm = d * t.mask;
        atomicAdd(p.gmask + ((size_t)(b * g.dg + grp) * T + tap) * HoWo + pp, d * smp);
        atomicAdd(p.goff + ((size_t)(b * g.dg + grp) * 2 * T + 2 * tap) * HoWo + pp,
                  (-t.hw * v00 - t.lw * v01 + t.hw * v10 + t.lw * v11) * dm);
        atomicAdd(p.goff + ((size_t)(b * g.dg + grp) * 2 * T + 2 * tap + 1) * HoWo + pp,
                  (-t.hh * v00 + t.hh * v01 - t.lh * v10 + t.lh * v11) * dm);
        float* gplane = p.gin + ((size_t)b * g.C + c) * HW;
        if (t.c00) atomicAdd(gplane + t.o00, t.hh * t.hw * dm);
        if (t.c01) atomicAdd(gplane + t.o01, t.hh * t.lw * dm);
        if (t.c10) atomicAdd(gplane + t.o10, t.lh * t.hw * dm);
        if (t.c11) atomicAdd(gplane + t.o11, t.lh * t.lw * dm);
    }
}

int fill_geom(DcnGeom& g, int B, int C, int H, int W, int Co, int kh, int kw, int sh, int sw, int ph, int pw, int dh,
              int dw, int dg, const char* who) {
    CNUDA_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && Co > 0, "%s: empty tensor", who);
    CNUDA_REQUIRE(kh > 0 && kw > 0 && sh > 0 && sw > 0 && dh > 0 && dw > 0 && ph >= 0 && pw >= 0,
                  "%s: bad kernel geometry", who);
    CNUDA_REQUIRE(dg > 0 && C % dg == 0, "%s: channels (%d) not divisible by deformable_group (%d)", who, C, dg);
    g = DcnGeom{B, C, H, W, Co, kh, kw, sh, sw, ph, pw, dh, dw, dg,
                (H + 2 * ph - (dh * (kh - 1) + 1)) / sh + 1, (W + 2 * pw - (dw * (kw - 1) + 1)) / sw + 1};
    CNUDA_REQUIRE(g.Ho > 0 && g.Wo > 0, "%s: kernel larger than padded input", who);
    return 0;
}

int pick_bm(int M) { return M > 64 ? 128 : (M > 32 ? 64 : 32); }

struct DcnPlan {
    int T, K, Kp, bm, Mp;           // forward pack [Kp][Mp]
    int Cpad, Kp2, Mp2;             // dcol pack [Kp2 = Co padded][Mp2 = T*Cpad]
    int Mpw, Jp, Z;                 // wgrad slabs [Z][Mpw][Jp]
    long long N, pix_per_split;
    size_t fwd_bytes, bwd_bytes;
};
DcnPlan make_plan(const DcnGeom& g) {
    DcnPlan q;
    q.T = g.kh * g.kw;
    q.K = q.T * g.C;
    q.Kp = round_up(q.K, IG_BK);
    q.bm = pick_bm(g.Co);
    q.Mp = round_up(g.Co, q.bm);
    q.Cpad = round_up(g.C, 64);
    q.Kp2 = round_up(g.Co, IG_BK);
    q.Mp2 = q.T * q.Cpad;
    q.Mpw = round_up(g.Co, WG_BM);
    q.Jp = round_up(q.K, WG_BJ);
    q.N = (long long)g.B * g.Ho * g.Wo;
    // enough pixel splits to fill the chip (>= ~1024 workgroups), each a multiple of the chunk
    const long long tiles = (long long)(q.Mpw / WG_BM) * (q.Jp / WG_BJ);
    long long z = (1024 + tiles - 1) / tiles;
    const long long max_z = (q.N + WG_BP - 1) / WG_BP;
    if (z > max_z) z = max_z;
    if (z < 1) z = 1;
    q.pix_per_split = ((q.N + z - 1) / z + WG_BP - 1) / WG_BP * WG_BP;
    q.Z = (int)((q.N + q.pix_per_split - 1) / q.pix_per_split);
    q.fwd_bytes = carve_bytes((size_t)q.Kp * q.Mp, 4) + 256;
    q.bwd_bytes = carve_bytes((size_t)q.Kp2 * q.Mp2, 4) + carve_bytes((size_t)q.Z * q.Mpw * q.Jp, 4) + 256;
    return q;
}

}  // namespace
}  // namespace cnuda

using namespace cnuda;

extern "C" size_t cnuda_dcn_v2_workspace_bytes(int B, int C, int H, int W, int Cout, int kh, int kw, int sh, int sw,
                                               int ph, int pw, int dh, int dw, int dg) {
    DcnGeom g;
    if (fill_geom(g, B, C, H, W, Cout, kh, kw, sh, sw, ph, pw, dh, dw, dg, "cnuda_dcn_v2_workspace_bytes")) return 0;
    const DcnPlan q = make_plan(g);
    return q.fwd_bytes > q.bwd_bytes ? q.fwd_bytes : q.bwd_bytes;
}

extern "C" int cnuda_dcn_v2_forward(const float* input, const float* weight, const float* bias, const float* offset,
                                    const float* mask, float* output, int B, int C, int H, int W, int Cout, int kh,
                                    int kw, int sh, int sw, int ph, int pw, int dh, int dw, int dg, void* workspace,
                                    size_t workspace_bytes, cnuda_stream_t stream) {
    CNUDA_REQUIRE(input && weight && bias && offset && mask && output, "cnuda_dcn_v2_forward: null pointer");
    DcnGeom g;
    if (int rc = fill_geom(g, B, C, H, W, Cout, kh, kw, sh, sw, ph, pw, dh, dw, dg, "cnuda_dcn_v2_forward")) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (dg != 1) {
        DcnNaiveParams p{g, input, weight, bias, offset, mask, nullptr, output, nullptr, nullptr, nullptr, nullptr};
        hipLaunchKernelGGL(dcn_naive_fwd_kernel, dim3(stream_grid((long long)B * Cout * g.Ho * g.Wo, 256)), dim3(256),
                           0, st, p);
        return check_launch("cnuda_dcn_v2_forward(dg>1)");
    }
    const DcnPlan q = make_plan(g);
    CNUDA_REQUIRE(workspace && workspace_bytes >= q.fwd_bytes, "cnuda_dcn_v2_forward: workspace too small");
    Carver cv(workspace, workspace_bytes);
    float* A = cv.take<float>((size_t)q.Kp * q.Mp);
    launch_pack(weight, A, Cout, C, q.T, PACK_FWD, q.Kp, q.Mp, 0, st);
    DcnFwdParams p{g, input, offset, mask, bias, output};
    const int n_tiles = ceil_div(q.N, IG_BN), m_tiles = q.Mp / q.bm;
    const dim3 grid(n_tiles * m_tiles), block(IG_THREADS);
    ProfScope prof(st);
    if (q.bm == 128)
        hipLaunchKernelGGL((igemm_fwd_kernel<128, DcnFwdLoader>), grid, block, 0, st, p, A, q.Mp, q.Kp, Cout, q.N,
                           n_tiles, m_tiles);
    else if (q.bm == 64)
        hipLaunchKernelGGL((igemm_fwd_kernel<64, DcnFwdLoader>), grid, block, 0, st, p, A, q.Mp, q.Kp, Cout, q.N,
                           n_tiles, m_tiles);
    else
        hipLaunchKernelGGL((igemm_fwd_kernel<32, DcnFwdLoader>), grid, block, 0, st, p, A, q.Mp, q.Kp, Cout, q.N,
                           n_tiles, m_tiles);
    return check_launch("cnuda_dcn_v2_forward");
}

extern "C" int cnuda_dcn_v2_backward(const float* input, const float* weight, const float* bias, const float* offset,
                                     const float* mask, const float* grad_output, float* grad_input,
                                     float* grad_offset, float* grad_mask, float* grad_weight, float* grad_bias, int B,
                                     int C, int H, int W, int Cout, int kh, int kw, int sh, int sw, int ph, int pw,
                                     int dh, int dw, int dg, void* workspace, size_t workspace_bytes,
                                     cnuda_stream_t stream) {
    CNUDA_REQUIRE(input && weight && offset && mask && grad_output && grad_input && grad_offset && grad_mask &&
                      grad_weight && grad_bias,
                  "cnuda_dcn_v2_backward: null pointer");
    (void)bias;
    DcnGeom g;
    if (int rc = fill_geom(g, B, C, H, W, Cout, kh, kw, sh, sw, ph, pw, dh, dw, dg, "cnuda_dcn_v2_backward")) return rc;
    hipStream_t st = (hipStream_t)stream;
    const int T = kh * kw, HoWo = g.Ho * g.Wo;
    if (hipMemsetAsync(grad_input, 0, (size_t)B * C * H * W * sizeof(float), st) != hipSuccess)
        return check_launch("cnuda_dcn_v2_backward(memset)");
    launch_channel_sum(grad_output, grad_bias, B, Cout, HoWo, st);
    if (dg != 1) {
        (void)hipMemsetAsync(grad_offset, 0, (size_t)B * dg * 2 * T * HoWo * sizeof(float), st);
        (void)hipMemsetAsync(grad_mask, 0, (size_t)B * dg * T * HoWo * sizeof(float), st);
        (void)hipMemsetAsync(grad_weight, 0, (size_t)Cout * C * T * sizeof(float), st);
        DcnNaiveParams p{g, input, weight, nullptr, offset, mask, grad_output, nullptr,
                         grad_input, grad_offset, grad_mask, grad_weight};
        hipLaunchKernelGGL(dcn_naive_bwd_kernel, dim3(stream_grid((long long)B * C * T * HoWo, 256)), dim3(256), 0, st,
                           p);
        return check_launch("cnuda_dcn_v2_backward(dg>1)");
    }
    const DcnPlan q = make_plan(g);
    CNUDA_REQUIRE(workspace && workspace_bytes >= q.bwd_bytes, "cnuda_dcn_v2_backward: workspace too small");
    Carver cv(workspace, workspace_bytes);
    float* A2 = cv.take<float>((size_t)q.Kp2 * q.Mp2);
    float* slabs = cv.take<float>((size_t)q.Z * q.Mpw * q.Jp);
    // (1) column gradient + offset / mask / input gradients
    launch_pack(weight, A2, Cout, C, q.T, PACK_DCOL, q.Kp2, q.Mp2, q.Cpad, st);
    {
        DcnBwdParams p{g, input, offset, mask, grad_output, grad_input, grad_offset, grad_mask};
        const int n_tiles = ceil_div(q.N, IG_BN);
        ProfScope prof(st);
        hipLaunchKernelGGL(dcn_bwd_data_kernel, dim3(n_tiles * q.T), dim3(IG_THREADS), 0, st, p, A2, q.Mp2, q.Kp2,
                           q.Cpad, q.N, n_tiles);
        if (int rc = check_launch("cnuda_dcn_v2_backward(data)")) return rc;
    }
    // (2) weight gradient
    {
        DcnWParams p{g, input, offset, mask, grad_output};
        hipLaunchKernelGGL((igemm_wgrad_kernel<DcnWLoader>), dim3(q.Jp / WG_BJ, q.Mpw / WG_BM, q.Z), dim3(IG_THREADS),
                           0, st, p, slabs, q.Mpw, q.Jp, q.N, q.pix_per_split);
        if (int rc = check_launch("cnuda_dcn_v2_backward(weight)")) return rc;
        launch_slab_reduce(slabs, grad_weight, q.Z, q.Mpw, q.Jp, Cout, C, q.T, st);
    }
    return check_launch("cnuda_dcn_v2_backward");
}
