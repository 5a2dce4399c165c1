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
    int h0, w0;            // integer corner (floor of the sample position)
    float lh, lw, hh, hw;  // fractions; hh = 1-lh, hw = 1-lw
    float mask;
    bool inside, c00, c01, c10, c11;
};

// the three per-(pixel, tap) inputs of the sampling geometry; loaded one tap ahead where latency matters
struct TapRaw { float dy, dx, mask; };
__device__ __forceinline__ TapRaw load_tap_raw(const DcnGeom& g, const float* __restrict__ off_b,
                                               const float* __restrict__ mask_b, int grp, int tap, int p) {
    const int T = g.kh * g.kw, HoWo = g.Ho * g.Wo;
    TapRaw r;
    r.dy = off_b[((size_t)grp * 2 * T + 2 * tap) * HoWo + p];
    r.dx = off_b[((size_t)grp * 2 * T + 2 * tap + 1) * HoWo + p];
    r.mask = mask_b[((size_t)grp * T + tap) * HoWo + p];
    return r;
}
__device__ __forceinline__ Tap tap_from_raw(const DcnGeom& g, const TapRaw& raw, int tap, int oy, int ox);
__device__ __forceinline__ Tap make_tap(const DcnGeom& g, const float* __restrict__ off_b,
                                        const float* __restrict__ mask_b, int grp, int tap, int oy, int ox) {
    return tap_from_raw(g, load_tap_raw(g, off_b, mask_b, grp, tap, oy * g.Wo + ox), tap, oy, ox);
}
__device__ __forceinline__ Tap tap_from_raw(const DcnGeom& g, const TapRaw& raw, int tap, int oy, int ox) {
    const int i = tap / g.kw, j = tap - i * g.kw;
    const float dy = raw.dy, dx = raw.dx;
    Tap t;
    t.mask = raw.mask;
    const float h = (float)(oy * g.sh - g.ph + i * g.dh) + dy;
    const float w = (float)(ox * g.sw - g.pw + j * g.dw) + dx;
    t.inside = (h > -1.0f) && (w > -1.0f) && (h < (float)g.H) && (w < (float)g.W);
    const float hf = floorf(h), wf = floorf(w);
    const int h0 = (int)hf, w0 = (int)wf;
    t.h0 = h0;
    t.w0 = w0;
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
    float* col;   // optional [B][T*C][Ho*Wo] side output (rows in (tap, channel) order) for the weight gradient
};

struct DcnFwdLoader {
    using Params = DcnFwdParams;
    static constexpr bool kHasSideOutput = true;
    const DcnGeom& g;
    const float *in_b, *off_b, *mask_b;
    int oy, ox, K;
    bool valid;
    int cur;   // tap whose sampling state currently sits in the registers below (K is tap-major:
               // consecutive chunks share it)
    // per (pixel, tap): clamped corner offsets and corner weights with validity and mask folded in, so
    // one sampled value is 4 unconditional loads + 4 multiply-adds
    int q00, q01, q10, q11;
    float m00, m01, m10, m11;
    float* col_n;     // this pixel's column in the side output (nullptr: not requested / not the first M tile)
    int col_stride;
    __device__ __forceinline__ void disable_col() { col_n = nullptr; }
    __device__ DcnFwdLoader(const Params& p, long long n, bool n_valid) : g(p.g), valid(n_valid), cur(-1) {
        const int HoWo = g.Ho * g.Wo;
        const int nn = n_valid ? (int)n : 0;   // N < 2^31 is checked on the host: 32-bit index math
        const int b = nn / HoWo, pp = nn - b * HoWo;
        oy = pp / g.Wo;
        ox = pp - oy * g.Wo;
        const int T = g.kh * g.kw;
        in_b = p.in + (size_t)b * g.C * g.H * g.W;
        off_b = p.off + (size_t)b * 2 * T * HoWo;
        mask_b = p.mask + (size_t)b * T * HoWo;
        K = T * g.C;
        col_n = (p.col && n_valid) ? p.col + (size_t)b * K * HoWo + pp : nullptr;
        col_stride = HoWo;
    }
    __device__ __forceinline__ void set_tap(int tap) {
        const Tap t = make_tap(g, off_b, mask_b, 0, tap, oy, ox);
        q00 = t.o00; q01 = t.o01; q10 = t.o10; q11 = t.o11;            // already 0 for missing corners
        const float mk = (valid && t.inside) ? t.mask : 0.0f;
        m00 = t.c00 ? t.hh * t.hw * mk : 0.0f;
        m01 = t.c01 ? t.hh * t.lw * mk : 0.0f;
        m10 = t.c10 ? t.lh * t.hw * mk : 0.0f;
        m11 = t.c11 ? t.lh * t.lw * mk : 0.0f;
        cur = tap;
    }
    __device__ __forceinline__ void load(int k0, int ksub, float (&v)[8]) {
        const int HW = g.H * g.W;
        if (g.C % IG_BK == 0) {
            // one tap per 16-deep chunk: no per-element index math
            const int tap = k0 / g.C, c0 = k0 - tap * g.C + ksub;
            if (k0 >= K) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = 0.0f;
                return;
            }
            if (tap != cur) set_tap(tap);
            const float* plane = in_b + (size_t)c0 * HW;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float* pl = plane + (size_t)(2 * j) * HW;
                const float r = m00 * pl[q00] + m01 * pl[q01] + m10 * pl[q10] + m11 * pl[q11];
                if (col_n) col_n[(size_t)(k0 + ksub + 2 * j) * col_stride] = r;
                v[j] = r;
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = k0 + ksub + 2 * j;
            float r = 0.0f;
            if (k < K) {
                const int tap = k / g.C, c = k - tap * g.C;
                if (tap != cur) set_tap(tap);
                const float* pl = in_b + (size_t)c * HW;
                r = m00 * pl[q00] + m01 * pl[q01] + m10 * pl[q10] + m11 * pl[q11];
                if (col_n) col_n[(size_t)k * col_stride] = r;
            }
            v[j] = r;
        }
    }
    struct Out {
        float* base;
        int HoWo;
        __device__ Out(const Params& p, long long n) {
            HoWo = p.g.Ho * p.g.Wo;
            const int ni = (int)n, b = ni / HoWo, pp = ni - b * HoWo;
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
    int dbg;   // ablation bits for scratch experiments (0 in production)
};
int g_dcn_dbg = 0;

// Structure: workgroup = one tile of 64 output pixels; it loops over 64-channel
// tiles of C and, inside, over all taps.  Each of the 4 waves owns 16 channels x
// 64 pixels (four 16x16 MFMA tiles, v_mfma_f32_16x16x4_f32), so the waves write
// disjoint channel planes of an LDS window that covers the input rows / columns
// the pixel tile can reach (|dy| < 2, |dx| < 1 beyond the 3x3 footprint).  The
// bilinear scatter is a plain LDS read-add-write: within a wave LDS operations
// execute in order, different waves touch different planes, and the only
// remaining hazard -- two pixels of one 16-lane group landing on the same cell
// in ONE instruction -- is detected per tap with a claim map and handled by LDS
// atomics (measured: ds_add_f32 costs ~200 cycles per wave instruction on
// gfx950, 50x a plain read+write, so it must stay off the common path).  The
// window is flushed once per channel tile with coalesced global atomics (~7x
// fewer HBM-side atomics than scattering every corner, and contiguous); corners
// outside the window (large offsets, odd shapes) go to global memory directly.
constexpr int DB_BM = 64, DB_BN = 32, DB_WIN = 256;   // window cells per channel (64*256*4 B = 64 KiB -> 2 workgroups / CU)
constexpr int DB_NT = DB_BN / 16;                      // 16-pixel MFMA tiles per wave

__global__ __launch_bounds__(IG_THREADS, 2) void dcn_bwd_data_kernel(DcnBwdParams p, const float* __restrict__ A2,
                                                                 int Mp2, int Kp, int Cpad, long long N, int n_tiles) {
    using f32x4 = __attribute__((ext_vector_type(4))) float;
    extern __shared__ __align__(16) float smem[];
    float* As = smem;                          // [16][64]  (k, channel)
    float* Bs = As + IG_BK * DB_BM;            // [16][DB_BN]  (k, pixel)
    float* win = Bs + IG_BK * DB_BN;           // [64][WSZ]
    float* dump = win + DB_BM * DB_WIN;        // [256] one cell per thread
    int* claim = reinterpret_cast<int*>(dump + IG_THREADS);   // [4 waves][DB_WIN]
    const DcnGeom& g = p.g;
    const int T = g.kh * g.kw, HoWo = g.Ho * g.Wo, HW = g.H * g.W;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wg = xcd_remap(blockIdx.x, n_tiles);
    const long long n0 = (long long)wg * DB_BN;
    const int kq = lane >> 4, il = lane & 15;   // MFMA k index / row-col index

    // ---- window geometry (workgroup-uniform) ----
    int wy0 = 0, wx0 = 0, WR = 0, WC = 0, tile_b = 0;
    bool use_win = false;
    if (HoWo % DB_BN == 0) {
        tile_b = (int)(n0 / HoWo);
        const int p0 = (int)(n0 - (long long)tile_b * HoWo), p1 = p0 + DB_BN - 1;
        const int y0 = p0 / g.Wo, y1 = p1 / g.Wo;
        int x0 = p0 - y0 * g.Wo, x1 = p1 - y1 * g.Wo;
        if (y1 != y0) { x0 = 0; x1 = g.Wo - 1; }
        wy0 = y0 * g.sh - g.ph - 2;
        WR = (y1 - y0) * g.sh + (g.kh - 1) * g.dh + 5;
        wx0 = x0 * g.sw - g.pw - 1;
        WC = (x1 - x0) * g.sw + (g.kw - 1) * g.dw + 3;
        if (wy0 < 0) { WR += wy0; wy0 = 0; }
        if (wx0 < 0) { WC += wx0; wx0 = 0; }
        if (wy0 + WR > g.H) WR = g.H - wy0;
        if (wx0 + WC > g.W) WC = g.W - wx0;
        use_win = WR > 0 && WC > 0 && WR * WC <= DB_WIN && (WR + 1) * (WC + 1) <= DB_WIN + 128;
    }
    const int WSZ = use_win ? WR * WC : 0;

    // ---- B-operand (gout) staging: pixel = tid % DB_BN, k phase = tid / DB_BN ----
    constexpr int KPH = IG_THREADS / DB_BN;     // k rows covered per pass
    constexpr int BPT = IG_BK / KPH;            // gout elements per thread per chunk
    const int nl = tid & (DB_BN - 1), ksub = tid / DB_BN;
    const long long nb = n0 + nl;
    const bool nb_valid = nb < N;
    const int bb = nb_valid ? (int)(nb / HoWo) : 0;
    const int pb = nb_valid ? (int)(nb - (long long)bb * HoWo) : 0;
    const float* gout_b = p.gout + (size_t)bb * g.Co * HoWo + pb;

    // ---- epilogue coordinates: lane owns pixel (j*16 + il) of each 16-pixel tile ----
    bool pv[DB_NT];
    int eb[DB_NT], ep[DB_NT], eoy[DB_NT], eox[DB_NT];
#pragma unroll
    for (int j = 0; j < DB_NT; ++j) {
        const long long ne = n0 + j * 16 + il;
        pv[j] = ne < N;
        eb[j] = pv[j] ? (int)(ne / HoWo) : 0;
        ep[j] = pv[j] ? (int)(ne - (long long)eb[j] * HoWo) : 0;
        eoy[j] = ep[j] / g.Wo;
        eox[j] = ep[j] - eoy[j] * g.Wo;
    }
    int* my_claim = claim + wid * (DB_WIN + 128);
    const int dcell = (int)(dump - win) + tid;

    // every wave zeroes the 16 channel planes it owns (only that wave ever touches them)
    for (int i = lane; i < 16 * WSZ; i += 64) win[wid * 16 * WSZ + i] = 0.0f;

    // sampling inputs of tap 0 (then always one tap ahead: their latency hides under the previous tap's work)
    TapRaw raw_next[DB_NT];
#pragma unroll
    for (int j = 0; j < DB_NT; ++j)
        raw_next[j] = load_tap_raw(g, p.off + (size_t)eb[j] * 2 * T * HoWo, p.mask + (size_t)eb[j] * T * HoWo, 0, 0,
                                   ep[j]);

    for (int c0 = 0; c0 < Cpad; c0 += DB_BM) {
#pragma unroll 1
        for (int tap = 0; tap < T; ++tap) {
            // ---- geometry of this tap for the lane's pixels, corner loads issued BEFORE the GEMM loop ----
            Tap tp[DB_NT];
            float v00[DB_NT][4], v01[DB_NT][4], v10[DB_NT][4], v11[DB_NT][4];
#pragma unroll
            for (int j = 0; j < DB_NT; ++j) {
                tp[j] = tap_from_raw(g, raw_next[j], tap, eoy[j], eox[j]);
                const bool live = pv[j] && tp[j].inside;
                const int o00 = (live && tp[j].c00) ? tp[j].o00 : 0, o01 = (live && tp[j].c01) ? tp[j].o01 : 0;
                const int o10 = (live && tp[j].c10) ? tp[j].o10 : 0, o11 = (live && tp[j].c11) ? tp[j].o11 : 0;
                const float* in_b = p.in + (size_t)eb[j] * g.C * HW;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    int c = c0 + wid * 16 + kq * 4 + r;
                    c = c < g.C ? c : g.C - 1;                       // padded rows carry dcol == 0
                    const float* plane = in_b + (size_t)c * HW;
                    v00[j][r] = plane[o00]; v01[j][r] = plane[o01]; v10[j][r] = plane[o10]; v11[j][r] = plane[o11];
                }
            }
            {   // next tap's sampling inputs (wraps to tap 0 of the next channel tile)
                const int nt = tap + 1 < T ? tap + 1 : 0;
#pragma unroll
                for (int j = 0; j < DB_NT; ++j)
                    raw_next[j] = load_tap_raw(g, p.off + (size_t)eb[j] * 2 * T * HoWo,
                                               p.mask + (size_t)eb[j] * T * HoWo, 0, nt, ep[j]);
            }
            f32x4 acc[DB_NT];
#pragma unroll
            for (int j = 0; j < DB_NT; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[j][r] = 0.0f;
            const int mbase = tap * Cpad + c0;
            float ra[4], rb[BPT];
            auto stage_load = [&](int k0) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int e = tid + i * IG_THREADS;
                    ra[i] = A2[(size_t)(k0 + (e >> 6)) * Mp2 + mbase + (e & 63)];
                }
#pragma unroll
                for (int i = 0; i < BPT; ++i) {
                    const int o = k0 + ksub + KPH * i;
                    rb[i] = (nb_valid && o < g.Co) ? gout_b[(size_t)o * HoWo] : 0.0f;
                }
            };
            stage_load(0);
            for (int k0 = 0; k0 < Kp; k0 += IG_BK) {
                __syncthreads();
#pragma unroll
                for (int i = 0; i < 4; ++i) As[tid + i * IG_THREADS] = ra[i];
#pragma unroll
                for (int i = 0; i < BPT; ++i) Bs[(ksub + KPH * i) * DB_BN + nl] = rb[i];
                __syncthreads();
                if (k0 + IG_BK < Kp) stage_load(k0 + IG_BK);
#pragma unroll
                for (int kk = 0; kk < IG_BK; kk += 4) {
                    const float a = As[(kk + kq) * DB_BM + wid * 16 + il];
#pragma unroll
                    for (int j = 0; j < DB_NT; ++j) {
                        const float b = Bs[(kk + kq) * DB_BN + j * 16 + il];
                        acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j], 0, 0, 0);
                    }
                }
            }
            // ---- consume the dcol tile: lane holds channels cl = wid*16 + kq*4 + r (r<4) of pixel j*16+il ----
#pragma unroll
            for (int j = 0; j < DB_NT; ++j) {
                const Tap& t = tp[j];
                const bool live = pv[j] && t.inside;
                const int h0 = t.h0 - wy0, w0 = t.w0 - wx0;
                const bool r0 = use_win && h0 >= 0 && h0 < WR, r1 = use_win && h0 + 1 >= 0 && h0 + 1 < WR;
                const bool q0 = w0 >= 0 && w0 < WC, q1 = w0 + 1 >= 0 && w0 + 1 < WC;
                const bool a00 = live && t.c00, a01 = live && t.c01, a10 = live && t.c10, a11 = live && t.c11;
                const bool i00 = a00 && r0 && q0, i01 = a01 && r0 && q1, i10 = a10 && r1 && q0, i11 = a11 && r1 && q1;
                const int l00 = i00 ? h0 * WC + w0 : -1, l01 = i01 ? h0 * WC + w0 + 1 : -1;
                const int l10 = i10 ? (h0 + 1) * WC + w0 : -1, l11 = i11 ? (h0 + 1) * WC + w0 + 1 : -1;
                const float k00 = a00 ? t.hh * t.hw : 0.f, k01 = a01 ? t.hh * t.lw : 0.f;
                const float k10 = a10 ? t.lh * t.hw : 0.f, k11 = a11 ? t.lh * t.lw : 0.f;
                const float mk = live ? t.mask : 0.f;
                const int o00 = a00 ? t.o00 : 0, o01 = a01 ? t.o01 : 0, o10 = a10 ? t.o10 : 0, o11 = a11 ? t.o11 : 0;
                // collision check: do two pixels of this 16-lane group share a window cell?  The four
                // corner cells of a pixel are a fixed pattern around (h0, w0), so comparing the anchor
                // cell of every in-window pixel is enough.  kq == 0 lanes vote (all kq see the same pixels).
                const int anchor = (live && use_win && h0 >= -1 && h0 < WR && w0 >= -1 && w0 < WC)
                                       ? (h0 + 1) * (WC + 1) + (w0 + 1) : -1;   // (WR+1) x (WC+1) grid incl. the -1 row/col
                float sm = 0.f, sh_ = 0.f, sw_ = 0.f;
                float dmv[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int cl = wid * 16 + kq * 4 + r;
                    const float d = (c0 + cl < g.C) ? acc[j][r] : 0.0f;
                    const float e00 = a00 ? v00[j][r] : 0.f, e01 = a01 ? v01[j][r] : 0.f;
                    const float e10 = a10 ? v10[j][r] : 0.f, e11 = a11 ? v11[j][r] : 0.f;
                    sm += d * (t.hh * t.hw * e00 + t.hh * t.lw * e01 + t.lh * t.hw * e10 + t.lh * t.lw * e11);
                    const float dm = d * mk;
                    dmv[r] = dm;
                    sh_ += (-t.hw * e00 - t.lw * e01 + t.hw * e10 + t.lw * e11) * dm;
                    sw_ += (-t.hh * e00 + t.hh * e01 - t.lh * e10 + t.lh * e11) * dm;
                }
                // scatter: one corner at a time; the four channels of a corner hit four different planes
                // (independent); consecutive corners may hit a neighbouring lane's previous cell, so the
                // compiler must keep LDS program order between them (the hardware does, per wave).
                // Pixels of this 16-lane group that share an anchor cell would collide inside ONE
                // instruction: they are serialised by rounds -- every pending pixel claims its anchor,
                // the pixel whose id survives in the claim map scatters, the others retry.
                const int wb0 = (wid * 16 + kq * 4) * WSZ;
                bool pending = true;
                volatile int* vc = my_claim;
                do {
                    bool won = pending;
                    if (kq == 0 && pending && anchor >= 0) vc[anchor] = il;
                    if (kq == 0 && pending && anchor >= 0) won = vc[anchor] == il;
                    won = __shfl((int)won, il, 64) != 0;          // the kq == 0 lane of this pixel decides
                    won = won && pending;
                    auto scatter = [&](int l, float k) {
                        const bool on = won && l >= 0;
                        float* base = win + (on ? wb0 + l : dcell);
                        const int stride = on ? WSZ : 0;
                        float cur[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) cur[r] = base[r * stride];
#pragma unroll
                        for (int r = 0; r < 4; ++r) base[r * stride] = cur[r] + (on ? k * dmv[r] : 0.f);
                        asm volatile("" ::: "memory");
                    };
                    asm volatile("" ::: "memory");
                    scatter(l00, k00);
                    scatter(l01, k01);
                    scatter(l10, k10);
                    scatter(l11, k11);
                    pending = pending && !won;
                } while (__any(pending));
                // rare: an existing corner outside the window -> global atomics
                const bool spill = (a00 && !i00) || (a01 && !i01) || (a10 && !i10) || (a11 && !i11);
                if (__any(spill)) {
                    if (spill) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int c = c0 + wid * 16 + kq * 4 + r;
                            if (c >= g.C) continue;
                            const float dm = acc[j][r] * mk;
                            float* gplane = p.gin + ((size_t)eb[j] * g.C + c) * HW;
                            if (a00 && !i00) atomicAdd(gplane + o00, k00 * dm);
                            if (a01 && !i01) atomicAdd(gplane + o01, k01 * dm);
                            if (a10 && !i10) atomicAdd(gplane + o10, k10 * dm);
                            if (a11 && !i11) atomicAdd(gplane + o11, k11 * dm);
                        }
                    }
                }
                // sums over this wave's 16 channels: 4 in-lane + the four kq groups
                sm += __shfl_xor(sm, 16, 64);  sm += __shfl_xor(sm, 32, 64);
                sh_ += __shfl_xor(sh_, 16, 64); sh_ += __shfl_xor(sh_, 32, 64);
                sw_ += __shfl_xor(sw_, 16, 64); sw_ += __shfl_xor(sw_, 32, 64);
                if (kq == 0 && live) {
                    atomicAdd(p.gmask + ((size_t)eb[j] * T + tap) * HoWo + ep[j], sm);
                    atomicAdd(p.goff + ((size_t)eb[j] * 2 * T + 2 * tap) * HoWo + ep[j], sh_);
                    atomicAdd(p.goff + ((size_t)eb[j] * 2 * T + 2 * tap + 1) * HoWo + ep[j], sw_);
                }
            }
        }
        // ---- flush this channel tile's window: every wave flushes (and re-zeroes) the 16 planes it owns,
        //      so no workgroup barrier is needed; lanes walk the cells, coalesced global atomics ----
        if (use_win) {
            asm volatile("" ::: "memory");
#pragma unroll 1
            for (int q = 0; q < (DB_WIN + 63) / 64; ++q) {
                const int pos = lane + 64 * q;
                if (pos < WSZ) {
                    const int yy = pos / WC, xx = pos - yy * WC;
                    float* gbase = p.gin + (size_t)tile_b * g.C * HW + (size_t)(wy0 + yy) * g.W + wx0 + xx;
#pragma unroll 4
                    for (int cc = 0; cc < 16; ++cc) {
                        const int cl = wid * 16 + cc;
                        const float v = win[cl * WSZ + pos];
                        if (v != 0.0f) {
                            if (c0 + cl < g.C) atomicAdd(gbase + (size_t)(c0 + cl) * HW, v);
                            win[cl * WSZ + pos] = 0.0f;
                        }
                    }
                }
            }
            asm volatile("" ::: "memory");
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
    // pixel cursor: image index, pixel index inside the image, output row / column
    long long n_, n_end_;
    int b_, pp_, oy_, ox_;
    bool valid_;
    __device__ __forceinline__ void cursor_init(long long n, long long n_end, int HoWo, int Wo) {
        n_ = n;
        n_end_ = n_end;
        valid_ = n < n_end;
        const long long nn = valid_ ? n : 0;
        b_ = (int)(nn / HoWo);
        pp_ = (int)(nn - (long long)b_ * HoWo);
        oy_ = pp_ / Wo;
        ox_ = pp_ - oy_ * Wo;
    }
    __device__ __forceinline__ void cursor_advance(int HoWo, int Wo) {
        n_ += WG_BP;
        valid_ = n_ < n_end_;
        pp_ += WG_BP;
        ox_ += WG_BP;
        while (ox_ >= Wo) { ox_ -= Wo; ++oy_; }
        while (pp_ >= HoWo) { pp_ -= HoWo; ++b_; oy_ = pp_ / Wo; ox_ = pp_ - oy_ * Wo; }
    }
    __device__ DcnWLoader(const Params& pp, long long n, long long n_end) : p(pp) {
        cursor_init(n, n_end, p.g.Ho * p.g.Wo, p.g.Wo);
    }
    __device__ __forceinline__ void advance() { cursor_advance(p.g.Ho * p.g.Wo, p.g.Wo); }
    template <int NV, int STEP>
    __device__ __forceinline__ void load_g(int m0, int msub, float (&v)[NV]) {
        const DcnGeom& g = p.g;
        const int HoWo = g.Ho * g.Wo;
        const float* base = p.gout + (size_t)b_ * g.Co * HoWo + pp_;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int m = m0 + msub + STEP * i;
            v[i] = (valid_ && m < g.Co) ? base[(size_t)m * HoWo] : 0.0f;
        }
    }
    template <int NV, int STEP>
    __device__ __forceinline__ void load_b(int j0, int jsub, float (&v)[NV]) {
        const DcnGeom& g = p.g;
        const int HoWo = g.Ho * g.Wo, HW = g.H * g.W, T = g.kh * g.kw, K = T * g.C;
        const float* in_b = p.in + (size_t)b_ * g.C * HW;
        const float* off_b = p.off + (size_t)b_ * 2 * T * HoWo;
        const float* mask_b = p.mask + (size_t)b_ * T * HoWo;
        int cur = -1;
        Tap t;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int k = j0 + jsub + STEP * i;
            float r = 0.0f;
            if (valid_ && k < K) {
                const int tap = k / g.C, c = k - tap * g.C;
                if (tap != cur) { t = make_tap(g, off_b, mask_b, 0, tap, oy_, ox_); cur = tap; }
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

// weight gradient from the column buffer saved by the forward pass: a plain GEMM
// gout[Co x px] * col[(tap,c) x px]^T with coalesced row reads (no resampling)
struct DcnColWParams {
    DcnGeom g;
    const float *col, *gout;
};
struct DcnColWLoader {
    using Params = DcnColWParams;
    const Params& p;
    // pixel cursor: image index, pixel index inside the image, output row / column
    long long n_, n_end_;
    int b_, pp_, oy_, ox_;
    bool valid_;
    __device__ __forceinline__ void cursor_init(long long n, long long n_end, int HoWo, int Wo) {
        n_ = n;
        n_end_ = n_end;
        valid_ = n < n_end;
        const long long nn = valid_ ? n : 0;
        b_ = (int)(nn / HoWo);
        pp_ = (int)(nn - (long long)b_ * HoWo);
        oy_ = pp_ / Wo;
        ox_ = pp_ - oy_ * Wo;
    }
    __device__ __forceinline__ void cursor_advance(int HoWo, int Wo) {
        n_ += WG_BP;
        valid_ = n_ < n_end_;
        pp_ += WG_BP;
        ox_ += WG_BP;
        while (ox_ >= Wo) { ox_ -= Wo; ++oy_; }
        while (pp_ >= HoWo) { pp_ -= HoWo; ++b_; oy_ = pp_ / Wo; ox_ = pp_ - oy_ * Wo; }
    }
    __device__ DcnColWLoader(const Params& pp, long long n, long long n_end) : p(pp) {
        cursor_init(n, n_end, p.g.Ho * p.g.Wo, p.g.Wo);
    }
    __device__ __forceinline__ void advance() { cursor_advance(p.g.Ho * p.g.Wo, p.g.Wo); }
    template <int NV, int STEP>
    __device__ __forceinline__ void load_g(int m0, int msub, float (&v)[NV]) {
        const DcnGeom& g = p.g;
        const int HoWo = g.Ho * g.Wo;
        const float* base = p.gout + (size_t)b_ * g.Co * HoWo + pp_;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int m = m0 + msub + STEP * i;
            v[i] = (valid_ && m < g.Co) ? base[(size_t)m * HoWo] : 0.0f;
        }
    }
    template <int NV, int STEP>
    __device__ __forceinline__ void load_b(int j0, int jsub, float (&v)[NV]) {
        const DcnGeom& g = p.g;
        const int HoWo = g.Ho * g.Wo, K = g.kh * g.kw * g.C;
        const float* base = p.col + (size_t)b_ * K * HoWo + pp_;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int k = j0 + jsub + STEP * i;
            v[i] = (valid_ && k < K) ? base[(size_t)k * HoWo] : 0.0f;
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

int pick_bm(int M, long long N) {
    int bm = M > 64 ? 128 : (M > 32 ? 64 : 32);
    const long long n_tiles = (N + IG_BN - 1) / IG_BN;
    while (bm > 32 && n_tiles * ((M + bm - 1) / bm) < 512) bm >>= 1;
    return bm;
}

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
    q.Kp = round_up(q.K, IG_KC);
    q.N = (long long)g.B * g.Ho * g.Wo;
    q.bm = pick_bm(g.Co, q.N);
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
    q.bwd_bytes = carve_bytes((size_t)q.Kp2 * q.Mp2, 4) + carve_bytes((size_t)q.Z * q.Mpw * q.Jp, 4) +
                  carve_bytes((size_t)g.Co * g.B, 4) + 256;
    return q;
}

}  // namespace
}  // namespace cnuda

using namespace cnuda;

extern "C" int cnuda_debug_dcn(int v) { g_dcn_dbg = v; return 0; }

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
    return cnuda_dcn_v2_forward_cols(input, weight, bias, offset, mask, output, nullptr, B, C, H, W, Cout, kh, kw, sh,
                                     sw, ph, pw, dh, dw, dg, workspace, workspace_bytes, stream);
}

extern "C" int cnuda_dcn_v2_forward_cols(const float* input, const float* weight, const float* bias,
                                         const float* offset, const float* mask, float* output, float* columns, int B,
                                         int C, int H, int W, int Cout, int kh, int kw, int sh, int sw, int ph, int pw,
                                         int dh, int dw, int dg, void* workspace, size_t workspace_bytes,
                                         cnuda_stream_t stream) {
    CNUDA_REQUIRE(input && weight && bias && offset && mask && output, "cnuda_dcn_v2_forward: null pointer");
    CNUDA_REQUIRE(!columns || dg == 1, "cnuda_dcn_v2_forward_cols: columns output needs deformable_group == 1");
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
    CNUDA_REQUIRE(q.N < (1ll << 31) - IG_BN, "cnuda_dcn_v2_forward: more than 2^31 pixels per call");
    CNUDA_REQUIRE(workspace && workspace_bytes >= q.fwd_bytes, "cnuda_dcn_v2_forward: workspace too small");
    Carver cv(workspace, workspace_bytes);
    float* A = cv.take<float>((size_t)q.Kp * q.Mp);
    launch_pack(weight, A, Cout, C, q.T, PACK_FWD, q.Kp, q.Mp, 0, st);
    DcnFwdParams p{g, input, offset, mask, bias, output, columns};
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
    return cnuda_dcn_v2_backward_cols(input, weight, bias, offset, mask, grad_output, nullptr, grad_input, grad_offset,
                                      grad_mask, grad_weight, grad_bias, B, C, H, W, Cout, kh, kw, sh, sw, ph, pw, dh,
                                      dw, dg, workspace, workspace_bytes, stream);
}

extern "C" int cnuda_dcn_v2_backward_cols(const float* input, const float* weight, const float* bias,
                                          const float* offset, const float* mask, const float* grad_output,
                                          const float* columns, float* grad_input, float* grad_offset,
                                          float* grad_mask, float* grad_weight, float* grad_bias, int B, int C, int H,
                                          int W, int Cout, int kh, int kw, int sh, int sw, int ph, int pw, int dh,
                                          int dw, int dg, void* workspace, size_t workspace_bytes,
                                          cnuda_stream_t stream) {
    CNUDA_REQUIRE(!columns || dg == 1, "cnuda_dcn_v2_backward_cols: columns input needs deformable_group == 1");
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
    if (dg != 1) {
        launch_channel_sum(grad_output, grad_bias, B, Cout, HoWo, st);
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
    float* bsum = cv.take<float>((size_t)Cout * B);
    launch_channel_sum(grad_output, grad_bias, B, Cout, HoWo, st, bsum);
    // (1) column gradient + offset / mask / input gradients
    launch_pack(weight, A2, Cout, C, q.T, PACK_DCOL, q.Kp2, q.Mp2, q.Cpad, st);
    {
        DcnBwdParams p{g, input, offset, mask, grad_output, grad_input, grad_offset, grad_mask, g_dcn_dbg};
        const int n_tiles = ceil_div(q.N, DB_BN);
        // grad_offset / grad_mask are summed over channel tiles and waves with a few atomics per pixel
        (void)hipMemsetAsync(grad_offset, 0, (size_t)B * 2 * T * HoWo * sizeof(float), st);
        (void)hipMemsetAsync(grad_mask, 0, (size_t)B * T * HoWo * sizeof(float), st);
        // As + Bs + window + dump cells + claim grids (4 waves x (WR+1)*(WC+1) <= DB_WIN + 128 ints)
        const size_t lds = (size_t)(IG_BK * DB_BM + IG_BK * DB_BN + DB_BM * DB_WIN + IG_THREADS + 4 * (DB_WIN + 128)) *
                           sizeof(float);
        static bool attr_set = false;
        if (!attr_set) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dcn_bwd_data_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            attr_set = true;
        }
        ProfScope prof(st);
        hipLaunchKernelGGL(dcn_bwd_data_kernel, dim3(n_tiles), dim3(IG_THREADS), lds, st, p, A2, q.Mp2, q.Kp2,
                           q.Cpad, q.N, n_tiles);
        if (int rc = check_launch("cnuda_dcn_v2_backward(data)")) return rc;
    }
    // (2) weight gradient
    {
        if (columns) {
            DcnColWParams p{g, columns, grad_output};
            hipLaunchKernelGGL((igemm_wgrad_kernel<DcnColWLoader, 64, 64>), dim3(q.Jp / WG_BJ, q.Mpw / WG_BM, q.Z),
                               dim3(IG_THREADS), 0, st, p, slabs, q.Mpw, q.Jp, q.N, q.pix_per_split);
        } else {
            DcnWParams p{g, input, offset, mask, grad_output};
            hipLaunchKernelGGL((igemm_wgrad_kernel<DcnWLoader, 64, 64>), dim3(q.Jp / WG_BJ, q.Mpw / WG_BM, q.Z),
                               dim3(IG_THREADS), 0, st, p, slabs, q.Mpw, q.Jp, q.N, q.pix_per_split);
        }
        if (int rc = check_launch("cnuda_dcn_v2_backward(weight)")) return rc;
        launch_slab_reduce(slabs, grad_weight, q.Z, q.Mpw, q.Jp, Cout, C, q.T, st);
    }
    return check_launch("cnuda_dcn_v2_backward");
}
