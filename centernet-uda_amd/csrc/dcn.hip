// Modulated deformable convolution (DCNv2) for gfx950.  The forward samples inside the implicit GEMM's
// loader (the reference materialises `columns`, libs/DCNv2/src/cuda/dcn_v2_cuda.cu:89-102, B*C*9*Ho*Wo
// floats, and round-trips them through HBM twice per direction); the sampled columns are an optional
// side output that the weight gradient reuses.
//
//   forward : out[b,o,p] = bias[o] + sum_{tap,c} W[o,c,tap] * mask[b,tap,p] * bilinear(in[b,c], p, tap)
//             one implicit GEMM on the fp32 MFMA whose B operand is sampled on the fly.
//   backward: (1) column gradient  dcol[(tap,c), p] = sum_o W[o,c,tap] * gout[b,o,p]  as a plain 1x1
//                 implicit GEMM on the MFMA, written once to workspace and streamed by two consumers:
//                 dcn_coord_grad_kernel (grad_offset / grad_mask: one thread per (pixel, tap), channels
//                 serial, plain stores) and dcn_col2im_kernel (bilinear scatter into grad_input through
//                 an LDS window; fp32 global atomics only for the window flush, collisions and strays
//                 -- the reference uses one atomic per corner, dcn_v2_im2col_cuda.cu:238-252);
//             (2) grad_weight = gout x sampled-columns^T as a split-K GEMM with
//                 fixed-order slab reduction; (3) grad_bias = channel sums.
//             The whole batch goes through each kernel once (the reference loops
//             over samples on the host, dcn_v2_cuda.cu:259: 6*B launches/layer).
//
// Sampling rule (dcn_v2_im2col_cuda.cu:25-54,180): a tap contributes iff
// -1 < y < H and -1 < x < W; corners outside the plane read as 0.
#include <algorithm>
#include "igemm.cuh"
#include "igemm_host.h"

namespace cnuda {
namespace {

struct DcnGeom {
    int B, C, H, W, Co, kh, kw, sh, sw, ph, pw, dh, dw, dg, Ho, Wo;
    // elements per image of the offset / mask tensors and of their gradients: dg * 2T * HoWo and dg * T * HoWo for the
    // reference's separate tensors; 3T * HoWo for all four when offsets and mask are read straight out of the 27-channel
    // output `om` of DCN's own offset convolution (rows 0 .. 2T-1 offsets, 2T .. 3T-1 the mask, already sigmoid:
    // cnuda_dcn_v2_forward_om / _backward_om, round 6).  gmask_logit: the mask gradient is written as the gradient of the
    // mask's LOGIT, g * m * (1 - m) -- what the offset convolution's backward wants (libs/DCNv2/dcn_v2.py:120-122).
    int off_bs, mask_bs, goff_bs, gmask_bs, gmask_logit;
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
    float act_slope;   // < 0: none; 0: ReLU fused into the epilogue (BatchNorm-folded inference, DeformConv dla.py:369-372)
    float* out;
    float* col;   // optional [B][T*C][Ho*Wo] side output (rows in (tap, channel) order) for the weight gradient
    float* stats; // optional BatchNorm statistics of `out` per (pixel block, output channel): igemm.cuh, "BatchNorm statistics"
    int stats_mp; // rows per pixel block of `stats`
};

// BUF: the corner loads and the column stores go through buffer descriptors (igemm.cuh, "Buffer addressing"): the
// per-lane offset is the corner pair / the pixel's column, the channel / row term a scalar offset -- the 24 address
// computations per chunk and thread that the pointer form needs would otherwise come out of the matrix pipe's cycles.
template <bool BUF>
struct DcnFwdLoaderT {
    using Params = DcnFwdParams;
    static const char* name() { return "DcnFwdLoader"; }
    static constexpr bool kHasSideOutput = true;
    const DcnGeom& g;
    const float *in_b, *off_b, *mask_b;
    buf_rsrc rin;
    float* col_base = nullptr;      // BUF: the column buffer (uniform)
    unsigned in_boff = 0, col_voff = IG_BUF_OOB;    // BUF: byte offset of the image in `in`, of the pixel's column in `col`
    bool col_on = false;
    int oy, ox, K;
    bool valid;
    int cur;   // tap whose sampling state currently sits in the registers below (K is tap-major:
               // consecutive chunks share it)
    int ck0 = 0, ctap = 0, cc0 = 0;     // chunk cursor (k0, its tap, its first channel)
    // per (pixel, tap): the two corner ROWS as clamped offsets of a horizontally adjacent pair, and the four
    // corner weights with validity and mask folded in: one sampled value is 2 unconditional 8-byte loads
    // (4-byte aligned; the buffer path takes them) + 4 multiply-adds
    int qT, qB;
    float m00, m01, m10, m11;     // weights of (top-left, top-right, bottom-left, bottom-right) of the loaded pairs
    float* col_n;     // this pixel's column in the side output (nullptr: not requested / not the first M tile)
    int col_stride;
    __device__ __forceinline__ void disable_col() { col_n = nullptr; col_on = false; }
    __device__ DcnFwdLoaderT(const Params& p, long long n, bool n_valid) : g(p.g), valid(n_valid), cur(-1) {
        const int HoWo = g.Ho * g.Wo;
        const int nn = n_valid ? (int)n : 0;   // N < 2^31 is checked on the host: 32-bit index math
        const int b = nn / HoWo, pp = nn - b * HoWo;
        oy = pp / g.Wo;
        ox = pp - oy * g.Wo;
        const int T = g.kh * g.kw;
        in_b = p.in + (size_t)b * g.C * g.H * g.W;
        off_b = p.off + (size_t)b * g.off_bs;
        mask_b = p.mask + (size_t)b * g.mask_bs;
        K = T * g.C;
        col_n = (p.col && n_valid) ? p.col + (size_t)b * K * HoWo + pp : nullptr;
        col_stride = HoWo;
        if constexpr (BUF) {
            rin = ig_make_rsrc(p.in, (unsigned)((size_t)g.B * g.C * g.H * g.W * sizeof(float)));
            col_base = p.col;
            in_boff = (unsigned)(b * g.C * g.H * g.W) * (unsigned)sizeof(float);
            col_on = p.col != nullptr;
            col_voff = (p.col && n_valid) ? (unsigned)(b * K * HoWo + pp) * (unsigned)sizeof(float) : IG_BUF_OOB;
        }
    }
    struct __attribute__((packed, aligned(4))) Pair { float l, r; };
    __device__ __forceinline__ void set_tap(int tap) {
        const Tap t = make_tap(g, off_b, mask_b, 0, tap, oy, ox);
        const float mk = (valid && t.inside) ? t.mask : 0.0f;
        const float w00 = t.c00 ? t.hh * t.hw * mk : 0.0f, w01 = t.c01 ? t.hh * t.lw * mk : 0.0f;
        const float w10 = t.c10 ? t.lh * t.hw * mk : 0.0f, w11 = t.c11 ? t.lh * t.lw * mk : 0.0f;
        // the pair starts at column clamp(w0, 0, W-2) (host guarantees W >= 2): at the left edge (w0 == -1) the
        // existing right corner is the pair's LEFT element, at the right edge (w0 == W-1) the existing left
        // corner is the pair's RIGHT element; missing corners carry zero weight
        const bool inside = valid && t.inside;
        const int w0 = inside ? t.w0 : 0, h0 = inside ? t.h0 : 0;
        const int wa = w0 < 0 ? 0 : (w0 > g.W - 2 ? g.W - 2 : w0);
        const bool ledge = w0 < 0, redge = w0 > g.W - 2;
        m00 = ledge ? w01 : (redge ? 0.0f : w00);
        m01 = ledge ? 0.0f : (redge ? w00 : w01);
        m10 = ledge ? w11 : (redge ? 0.0f : w10);
        m11 = ledge ? 0.0f : (redge ? w10 : w11);
        const int ht = h0 < 0 ? 0 : h0, hb = h0 + 1 > g.H - 1 ? g.H - 1 : h0 + 1;
        qT = ht * g.W + wa;
        qB = hb * g.W + wa;
        cur = tap;
    }
    __device__ __forceinline__ float sample(const float* __restrict__ pl) const {
        const Pair a = *reinterpret_cast<const Pair*>(pl + qT), b = *reinterpret_cast<const Pair*>(pl + qB);
        return m00 * a.l + m01 * a.r + m10 * b.l + m11 * b.r;
    }
    // Two-phase loading (kHasSideOutput loaders): load_raw() only ISSUES the corner loads of a chunk, finish()
    // turns them into samples (and writes the column side output) when the kernel stores the chunk to LDS, one
    // chunk of MFMAs later.  Sampling inside load() consumed every load on the spot -- and its column stores sat
    // in front of the next chunk's loads in the in-order vmcnt queue -- so the gather latency was exposed in full
    // (PMC: 65 % of wave cycles parked, matrix pipe 26 % busy).
    struct Raw {
        Pair t[8], b[8];
        int k0;
        bool live, eager;      // eager: channel count not a multiple of 16 -> values were sampled in load_raw (t[j].l)
    };
    __device__ __forceinline__ void load_raw(int k0, int ksub, Raw& r) {
        const int HW = g.H * g.W;
        r.k0 = k0 + ksub;
        r.live = k0 < K;
        r.eager = g.C % IG_BK != 0;
        if (!r.eager) {
            if (!r.live) return;
            // one tap per 16-deep chunk: no per-element index math, and (chunks arrive in increasing k0) the
            // tap / first channel advance by additions
            while (ck0 < k0) {
                ck0 += IG_BK;
                cc0 += IG_BK;
                if (cc0 >= g.C) { cc0 -= g.C; ++ctap; }
            }
            const int tap = ctap, c0 = cc0 + ksub;
            if (tap != cur) set_tap(tap);
            if constexpr (BUF) {
                const unsigned vT = in_boff + (unsigned)qT * 4u, vB = in_boff + (unsigned)qB * 4u;
                const int c0s = cc0 + __builtin_amdgcn_readfirstlane(ksub);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const unsigned soff = (unsigned)((c0s + 2 * j) * HW) * 4u;
                    r.t[j] = __builtin_bit_cast(Pair, __builtin_amdgcn_raw_buffer_load_b64(rin, (int)vT, (int)soff, 0));
                    r.b[j] = __builtin_bit_cast(Pair, __builtin_amdgcn_raw_buffer_load_b64(rin, (int)vB, (int)soff, 0));
                }
                return;
            }
            const float* plane = in_b + (size_t)c0 * HW;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float* pl = plane + (size_t)(2 * j) * HW;
                r.t[j] = *reinterpret_cast<const Pair*>(pl + qT);
                r.b[j] = *reinterpret_cast<const Pair*>(pl + qB);
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = k0 + ksub + 2 * j;
            float v = 0.0f;
            if (k < K) {
                const int tap = k / g.C, c = k - tap * g.C;
                if (tap != cur) set_tap(tap);
                v = sample(in_b + (size_t)c * HW);
            }
            r.t[j].l = v;
        }
    }
    __device__ __forceinline__ void finish(const Raw& r, float (&v)[8]) {
        if (!r.eager) {
            // fast path: the whole chunk is one tap and (K % 16 == 0) entirely inside or outside K
            if (!r.live) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = 0.0f;
                return;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = m00 * r.t[j].l + m01 * r.t[j].r + m10 * r.b[j].l + m11 * r.b[j].r;
            if constexpr (BUF) {
                // Column side output: GLOBAL stores in the scalar-base form (`global_store_dword v_off, v_data, s[base]`):
                // the row (k0 + 2j) * HoWo is a wave-uniform 64-bit base built with scalar adds, the lane's part is its
                // 32-bit column offset -- as free of vector address arithmetic as a buffer store.  NOT buffer stores: a
                // MUBUF store reads its data registers late, nothing orders that read against a later LDS return into
                // the same registers (dropped stores under CU sharing), and `v` dies right after
                // this function -- the register allocator is free to hand it to the next chunk's fragment reads.
                if (col_on && col_voff != IG_BUF_OOB) {
                    const int k0s = __builtin_amdgcn_readfirstlane(r.k0);
                    const char* cbase = reinterpret_cast<const char*>(col_base);
                    unsigned voff = col_voff;
                    asm volatile("" : "+v"(voff));      // (instruction selection is per basic block: keep the zero-extension here)
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const size_t row = (size_t)(unsigned)((k0s + 2 * j) * col_stride) * 4u;      // uniform
                        unsigned long long rbase = reinterpret_cast<unsigned long long>(cbase) + row;
                        asm volatile("" : "+s"(rbase));     // keep base and lane offset apart: scalar-base store form
                        *reinterpret_cast<__attribute__((address_space(1))) float*>(rbase + (unsigned long long)voff) = v[j];
                    }
                }
                return;
            }
            if (col_n) {
                float* cp = col_n + (size_t)r.k0 * col_stride;
                const size_t step = (size_t)2 * col_stride;
#pragma unroll
                for (int j = 0; j < 8; ++j) cp[j * step] = v[j];
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = r.k0 + 2 * j;
            v[j] = r.t[j].l;
            if (col_n && k < K) col_n[(size_t)k * col_stride] = v[j];
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
            v += p.bias[m];
            if (p.act_slope >= 0.0f && v < 0.0f) v *= p.act_slope;
            base[(size_t)m * HoWo] = v;
        }
        static constexpr bool kVec4 = true;      // 16-byte epilogue (igemm.cuh)
        __device__ static bool vec4_ok(const Params& p) { return ((p.g.Ho * p.g.Wo) & 3) == 0; }
        __device__ __forceinline__ void store4(const Params& p, int m, f32x4 v) {
            v += p.bias[m];
            if (p.act_slope >= 0.0f) {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (v[e] < 0.0f) v[e] *= p.act_slope;
            }
            *reinterpret_cast<f32x4*>(base + (size_t)m * HoWo) = v;
        }
    };
};

using DcnFwdLoader = DcnFwdLoaderT<false>;
using DcnFwdBufLoader = DcnFwdLoaderT<true>;

// ---------------------------------------------------------------------------
// forward from an LDS input window (round 4): 3x3, stride 1, padding 1, dilation 1, deformable_group 1, C % 16 == 0,
// row width a multiple of 16 (tiles of 4 x 32 or 8 x 16 pixels; H a multiple of the tile's rows), one M tile (Co <= 128).
//
// The loader above gathers two unaligned 8-byte corner pairs per (pixel, tap, channel) from global memory: the
// texture-address unit is busy 0.77 of the kernel's cycles and the matrix pipe 0.38 (profiles/r3_pmc_dcn.md).  Here --
// as in hconv.cuh -- the K axis is ordered (16-channel group, tap, channel) and a workgroup stages the INPUT window of
// a channel group once per nine chunks: the 128 pixels' rows plus (1 + DW_MARGIN) rows above and below and the whole
// row width plus 4 zero columns left and right, zero wherever the image ends, with 16-byte loads.  Every wave owns 32
// pixels and ALL output rows: lane l of k-step s samples channel 2s + (l >> 5) of its pixel -- two ds_read2_b32 of the
// corner pairs at the pixel's window address, one multiply and three fused multiply-adds with the tap's four corner
// weights (mask folded in), which live in registers for all nine taps of the tile -- and the sample IS the MFMA's B
// fragment: it never goes through LDS.  Out-of-image corners read the window's zeros, which reproduces the reference's
// validity rule (dcn_v2_im2col_cuda.cu:37-48,180) without a predicate.  A sample whose corners leave the window
// (|offset| >= DW_MARGIN = 3 pixels) is a STRAY: its lane takes the four corners from global memory
// (exec-masked, the slow path; correct for any offset).  The column side output for the weight gradient is one
// 4-byte store per sample, rows in (tap, channel) order as before.
// ---------------------------------------------------------------------------
constexpr int DW_MARGIN = 3;         // window rows beyond the undeformed 3x3 footprint, above and below
// Pixel tile = TR rows x TC columns = 128 pixels, TC = min(W, 32): four rows of 32 (eight of 16 on the 16-wide maps).
// Window of a 16-channel group: NR = TR + 2 + 2 * DW_MARGIN rows x (TC + 8) columns -- tile column x sits at index
// x - x0 + 4, four real (the neighbouring tile's) or zero columns on either side -- staged as 16-byte cells, TC / 4 + 2
// per row, zero by the buffer range check wherever the image ends.  The squarer the tile, the smaller the window per
// pixel: 12 x 40 cells for 4 x 32 pixels (31 KB per group, offsets up to +-3 pixels stay inside) against 10 x 72 for
// 2 x 64; a wave's 32 pixels are still one row segment (128-byte stores).
typedef __attribute__((address_space(3))) volatile unsigned char lds_vu8;   // volatile through a GENERIC pointer compiles to flat_load / flat_store
template <int TC> struct DwTile {
    static constexpr int TR = IG_BN / TC, NR = TR + 2 + 2 * DW_MARGIN, RS = TC + 8, PL = NR * RS;
    static constexpr int CPR = TC / 4 + 2, CPP = NR * CPR, CELLS = 16 * CPP;
    static constexpr int NCELL = (CELLS + IG_THREADS - 1) / IG_THREADS;       // per thread: 8 (TC 32), 6 (16)
    static constexpr int SHIFT = TC == 32 ? 5 : 4;
};
template <int TC> constexpr size_t dcnw_lds_floats(int bm) { return (size_t)16 * DwTile<TC>::PL + 2 * IG_KC * bm; }

template <int BM, int TC>
__global__ __launch_bounds__(IG_THREADS, 3) void dcnw_fwd_kernel(DcnFwdParams p, const float* __restrict__ A, int Mp, int Kp,
                                                                 int n_tiles, int tiles_x) {
    using Q = DwTile<TC>;
    constexpr int TM = BM / 32, PL = Q::PL, RS = Q::RS;                // every wave: all BM rows x 32 pixels
    extern __shared__ __attribute__((aligned(16))) float smem[];      // Hs[16 * PL] | As[2][16 * BM]; reused by the epilogue
    const DcnGeom& g = p.g;
    const int W = g.W, H = g.H, HW = g.H * g.W;
    float* const Hs = smem;
    float* const Asb = smem + 16 * PL;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    // tile -> (image, tile row, tile column)
    int tile = xcd_remap(blockIdx.x, n_tiles);
    const int tile_id = tile;                                          // (image-major: the statistics blocks' order)
    const int tx = tile % tiles_x; tile /= tiles_x;
    const int tiles_y = H / Q::TR;
    const int ty = tile % tiles_y, b = tile / tiles_y;
    const int y0 = ty * Q::TR, x0 = tx * TC;
    const int yw0 = y0 - 1 - DW_MARGIN;                                // image row of window row 0
    const int pxl = wid * 32 + (lane & 31), kl = lane >> 5;
    const int py = y0 + (pxl >> Q::SHIFT), px = x0 + (pxl & (TC - 1)), pp = py * W + px;

    const buf_rsrc rs = ig_make_rsrc(p.in, (unsigned)((size_t)g.B * g.C * HW * sizeof(float)));
    // window cell e = tid + 256 i of a channel group -> (channel, row, 16-byte column cell); offsets recomputed per use
    // (constant divisors: a few multiply-shifts) instead of kept in 2 x NCELL registers
    auto cell = [&](int i, unsigned& voff, int& loff) {
        // (an opaque zero: the compiler would otherwise compute every cell's offsets once per group, keep them across
        // the nine taps for win_store and spill them -- and a scratch reload sits behind the window loads in vmcnt order)
        int salt = 0;
        asm volatile("" : "+v"(salt));
        const int e = tid + i * IG_THREADS + salt;
        const int c = e / Q::CPP, rem = e - c * Q::CPP;
        const int row = rem / Q::CPR, q4 = rem - row * Q::CPR;
        const int iy = yw0 + row, ix = x0 - 4 + 4 * q4;
        loff = e < Q::CELLS ? c * PL + row * RS + 4 * q4 : -1;
        voff = (e < Q::CELLS && iy >= 0 && iy < H && ix >= 0 && ix < W)
                   ? (unsigned)(((b * g.C + c) * HW + iy * W + ix) * (int)sizeof(float)) : IG_BUF_OOB;
    };

    // per-tap sampling state of this lane's pixel: window address of the top-left corner, the two fractions and the
    // mask (zero for a stray or an invalid sample); `stray`: bit t set -> tap t takes its corners from global memory
    int addr[9];
    float flh[9], flw[9], fmk[9];
    unsigned stray = 0;
    {
        const float* off_b = p.off + (size_t)b * g.off_bs + pp;
        const float* mask_b = p.mask + (size_t)b * g.mask_bs + pp;
        float dy[9], dx[9], mk[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            dy[t] = off_b[(size_t)(2 * t) * HW];
            dx[t] = off_b[(size_t)(2 * t + 1) * HW];
            mk[t] = mask_b[(size_t)t * HW];
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const float h = (float)(py - 1 + t / 3) + dy[t], w = (float)(px - 1 + t % 3) + dx[t];
            const float hf = floorf(h), wf = floorf(w);
            const bool valid = h > -1.0f && w > -1.0f && h < (float)H && w < (float)W;    // (false for NaN)
            const int h0 = valid ? (int)hf : 0, w0i = valid ? (int)wf : 0;
            const int wr = h0 - yw0, wc = w0i - x0 + 4;
            const bool inwin = valid && wr >= 0 && wr + 1 <= Q::NR - 1 && wc >= 0 && wc + 1 <= RS - 1;
            addr[t] = inwin ? wr * RS + wc : 0;
            flh[t] = h - hf;
            flw[t] = w - wf;
            fmk[t] = inwin ? mk[t] : 0.0f;
            if (valid && !inwin) stray |= 1u << t;
        }
    }
    const bool any_stray = __any(stray != 0);

    f32x16 acc[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;

    f32x4 hreg[Q::NCELL];
    auto win_load = [&](int grp) {
        const unsigned soff = (unsigned)(grp * 16 * HW) * (unsigned)sizeof(float);
#pragma unroll
        for (int i = 0; i < Q::NCELL; ++i) {
            unsigned voff; int loff;
            cell(i, voff, loff);
            hreg[i] = ig_buf_load4(rs, voff, soff);
        }
    };
    auto win_store = [&]() {
#pragma unroll
        for (int i = 0; i < Q::NCELL; ++i) {
            unsigned voff; int loff;
            cell(i, voff, loff);
            if (loff >= 0) *reinterpret_cast<f32x4*>(Hs + loff) = hreg[i];
        }
    };
    f32x4 ra[ig_a_per<BM>()];
    const IgABuf<BM> abuf(A, Mp, Kp, 0, tid);
    const int nchunk = Kp / IG_KC, G = g.C >> 4;
    // column side output: rows (tap, channel), this lane's part of the address is fixed (k parity row + pixel)
    const bool col_on = p.col != nullptr;
    float* const col_l = col_on ? p.col + (size_t)b * 9 * g.C * HW + (size_t)kl * HW + pp : nullptr;

    win_load(0);
    abuf.load(0, ra);
    win_store();
    ig_store_a<BM>(Asb, tid, ra);
    if (1 < nchunk) abuf.load(IG_KC, ra);
    __syncthreads();
    int grp = 0;
    for (int c0 = 0; c0 < nchunk; c0 += 9, ++grp) {
        const bool more = grp + 1 < G;
        if (more) win_load(grp + 1);
#pragma unroll
        for (int t = 0; t < 9; ++t) {                                  // (unrolled: the tap's registers are addressed statically)
            const int c = c0 + t;
            // chunk c + 2's A tile goes out FIRST: vmcnt retires in issue order, so behind this tap's eight column stores
            // the load's data would arrive only after every one of them has drained
            f32x4 ra2[ig_a_per<BM>()];
            if (c + 2 < nchunk) abuf.load((c + 2) * IG_KC, ra2);
            const float* As = Asb + (c & 1) * IG_KC * BM;
            const float* ap = As + kl * BM + (lane & 31);
            // (opaque copies: without them the compiler hoists the 72 per-(tap, k-step) window addresses and the 72
            // column row bases out of the group loop as loop invariants and spills ~300 registers)
            int a_t = addr[t];
            asm volatile("" : "+v"(a_t));
            const float* hb = Hs + kl * PL + a_t;
            const float lh = flh[t], lw = flw[t], hh = 1.0f - lh, hw = 1.0f - lw, mk = fmk[t];
            const float w00 = hh * hw * mk, w01 = hh * lw * mk, w10 = lh * hw * mk, w11 = lh * lw * mk;
            // The eight samples of this chunk (channels kl, kl + 2, ..) in two halves: the corner pairs of the second half
            // are read from the window while the first half's MFMAs run (sched_barrier pins the order; left alone the
            // compiler reads all sixteen pairs, waits, then issues sixteen MFMAs: PMC showed 38 % of the wave cycles waiting)
            float v[IG_KC / 2], cr[IG_KC / 4][4];
            float* cp = nullptr;
            if (col_on) {
                unsigned long long rowoff = (unsigned long long)(unsigned)((t * g.C + grp * 16) * HW) * sizeof(float);
                asm volatile("" : "+s"(rowoff));
                cp = reinterpret_cast<float*>(reinterpret_cast<char*>(col_l) + rowoff);
            }
            auto read_half = [&](int h) {
#pragma unroll
                for (int s = 0; s < IG_KC / 4; ++s) {
                    const float* q = hb + 2 * (h * (IG_KC / 4) + s) * PL;
                    cr[s][0] = q[0]; cr[s][1] = q[1]; cr[s][2] = q[RS]; cr[s][3] = q[RS + 1];
                }
            };
            auto blend_half = [&](int h) {
#pragma unroll
                for (int s = 0; s < IG_KC / 4; ++s) {
                    float x = w00 * cr[s][0];
                    x = fmaf(w01, cr[s][1], x);
                    x = fmaf(w10, cr[s][2], x);
                    v[h * (IG_KC / 4) + s] = fmaf(w11, cr[s][3], x);
                }
                if (col_on) {
#pragma unroll
                    for (int s = 0; s < IG_KC / 4; ++s) cp[(size_t)(2 * (h * (IG_KC / 4) + s)) * HW] = v[h * (IG_KC / 4) + s];
                }
            };
            auto mma_half = [&](int h) {
#pragma unroll
                for (int s = h * (IG_KC / 4); s < (h + 1) * (IG_KC / 4); ++s) {
#pragma unroll
                    for (int i = 0; i < TM; ++i)
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * s * BM + i * 32], v[s], acc[i], 0, 0, 0);
                }
            };
            read_half(0);
            blend_half(0);
            read_half(1);
            __builtin_amdgcn_sched_barrier(0);
            mma_half(0);
            __builtin_amdgcn_sched_barrier(0);
            blend_half(1);
            mma_half(1);
            if (c + 1 < nchunk) ig_store_a<BM>(Asb + ((c + 1) & 1) * IG_KC * BM, tid, ra);
#pragma unroll
            for (int i = 0; i < ig_a_per<BM>(); ++i) ra[i] = ra2[i];
            __syncthreads();
            __builtin_amdgcn_sched_barrier(0);                         // (nothing moves across taps: register pressure)
        }
        if (any_stray) {
            // Strays of this wave (rare): their window weights were zero above; here the tap's eight channels come from
            // global memory for the stray lanes (zero for the others), the chunk's A fragments from the packed matrix in
            // L2, and the same 16 MFMAs add the missing products.  A runtime loop over the taps, apart from the unrolled
            // hot path: inlined there, its address arithmetic was hoisted out of the group loop for all nine taps and
            // spilled ~200 registers.  The tap index goes through an opaque copy for the same reason.
#pragma unroll 1
            for (int t0 = 0; t0 < 9; ++t0) {
                int t = t0;
                asm volatile("" : "+s"(t));
                if (!__any((stray >> t) & 1u)) continue;
                float sv[IG_KC / 2];
#pragma unroll
                for (int s = 0; s < IG_KC / 2; ++s) sv[s] = 0.0f;
                if ((stray >> t) & 1u) {
                    // the tap's state again from the raw offsets, corner addresses clamped into the plane, weights zeroed
                    // where a corner lies outside it (dcn_v2_im2col_cuda.cu:37-48)
                    const float* off_b = p.off + (size_t)b * g.off_bs + pp;
                    const int tr = t / 3;
                    const float sh_ = (float)(py - 1 + tr) + off_b[(size_t)(2 * t) * HW];
                    const float sw_ = (float)(px - 1 + t - 3 * tr) + off_b[(size_t)(2 * t + 1) * HW];
                    const float smk = p.mask[(size_t)b * g.mask_bs + (size_t)t * HW + pp];
                    const float shf = floorf(sh_), swf = floorf(sw_);
                    const int sh0 = (int)shf, sw0 = (int)swf;
                    const float slh = sh_ - shf, slw = sw_ - swf, shh = 1.0f - slh, shw = 1.0f - slw;
                    const bool top = sh0 >= 0, bot = sh0 + 1 <= H - 1, lef = sw0 >= 0, rig = sw0 + 1 <= W - 1;
                    const float g00 = (top && lef) ? shh * shw * smk : 0.0f, g01 = (top && rig) ? shh * slw * smk : 0.0f;
                    const float g10 = (bot && lef) ? slh * shw * smk : 0.0f, g11 = (bot && rig) ? slh * slw * smk : 0.0f;
                    const int cy0 = top ? sh0 : 0, cy1 = bot ? sh0 + 1 : H - 1, cx0 = lef ? sw0 : 0, cx1 = rig ? sw0 + 1 : W - 1;
                    const float* in_c = p.in + ((size_t)b * g.C + grp * 16 + kl) * HW;
                    const int o00 = cy0 * W + cx0, o01 = cy0 * W + cx1, o10 = cy1 * W + cx0, o11 = cy1 * W + cx1;
#pragma unroll
                    for (int s = 0; s < IG_KC / 2; ++s) {
                        const float* pl = in_c + (size_t)(2 * s) * HW;
                        float x = g00 * pl[o00];
                        x = fmaf(g01, pl[o01], x);
                        x = fmaf(g10, pl[o10], x);
                        sv[s] = fmaf(g11, pl[o11], x);
                    }
                    if (col_on) {
                        float* cp = col_l + (size_t)(t * g.C + grp * 16) * HW;
#pragma unroll
                        for (int s = 0; s < IG_KC / 2; ++s) cp[(size_t)(2 * s) * HW] = sv[s];
                    }
                }
                const float* ag = A + (size_t)((c0 + t) * IG_KC + kl) * Mp + (lane & 31);
#pragma unroll
                for (int s = 0; s < IG_KC / 2; ++s) {
#pragma unroll
                    for (int i = 0; i < TM; ++i)
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(ag[(size_t)(2 * s) * Mp + i * 32], sv[s], acc[i], 0, 0, 0);
                }
            }
        }
        if (more) {          // every wave has read the group's last corners: the next group's window moves in
            win_store();
            __syncthreads();
        }
    }
    // epilogue: bias (+ activation), 16 bytes per lane through LDS (every wave: TM tiles of 32 rows x its 32 pixels)
    float* stage = smem + wid * IG_EPI_WAVE;
    const int col = lane & 31, cg = lane & 7, rsub = lane >> 3;
    const int epl = wid * 32 + 4 * cg;                                  // four consecutive pixels of one tile row
    float* const obase = p.out + (size_t)b * g.Co * HW + (size_t)(y0 + (epl >> Q::SHIFT)) * W + x0 + (epl & (TC - 1));
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) stage[mfma_row(r, lane) * IG_EPI_LD + col] = acc[i][r];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int row = it * 8 + rsub;
            f32x4 v4 = *reinterpret_cast<const f32x4*>(stage + row * IG_EPI_LD + 4 * cg);
            const int m = i * 32 + row;
            if (m < g.Co) {
                v4 += p.bias[m];
                if (p.act_slope >= 0.0f) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (v4[e] < 0.0f) v4[e] *= p.act_slope;
                }
                *reinterpret_cast<f32x4*>(obase + (size_t)m * HW) = v4;
            }
            if (p.stats) {
                // BatchNorm statistics of what was stored (igemm.cuh, "BatchNorm statistics"): this wave's 32 pixels are one
                // block; the row's eight lanes meet through two quad permutes and a half-row mirror
                float s1 = (v4[0] + v4[1]) + (v4[2] + v4[3]);
                float s2 = (v4[0] * v4[0] + v4[1] * v4[1]) + (v4[2] * v4[2] + v4[3] * v4[3]);
                auto row8 = [](float v) {
                    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));
                    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false));
                    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, false));
                    return v;
                };
                s1 = row8(s1);
                s2 = row8(s2);
                if (cg == 0)
                    reinterpret_cast<float2*>(p.stats)[((size_t)tile_id * 4 + wid) * p.stats_mp + m] = make_float2(s1, s2);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

// ---------------------------------------------------------------------------
// forward, two-kernel form for layers whose output channels span several M tiles (small feature maps run
// 32- or 64-row tiles to fill the chip): the fused loader would re-sample the columns once per M tile, so the
// columns are sampled ONCE by a streaming kernel (they are the weight gradient's side output anyway) and a
// plain implicit GEMM reads them back.
// ---------------------------------------------------------------------------
struct DcnSampleParams {
    DcnGeom g;
    const float *in, *off, *mask;
    float* col;     // [B][T*C][Ho*Wo]
};
// block = (64 pixels, TW tap slots); one thread per (pixel, tap), channels serial in batches of 8
__global__ __launch_bounds__(1024) void dcn_sample_kernel(DcnSampleParams p, int tiles_per_image) {
    const DcnGeom& g = p.g;
    const int T = g.kh * g.kw, HoWo = g.Ho * g.Wo, HW = g.H * g.W;
    const int b = blockIdx.x / tiles_per_image, tile = blockIdx.x - b * tiles_per_image;
    const int px = tile * 64 + threadIdx.x;
    if (px >= HoWo) return;
    const int oy = px / g.Wo, ox = px - oy * g.Wo;
    const float* in_b = p.in + (size_t)b * g.C * HW;
    for (int tap = threadIdx.y; tap < T; tap += blockDim.y) {
        const Tap t = make_tap(g, p.off + (size_t)b * g.off_bs, p.mask + (size_t)b * g.mask_bs, 0, tap, oy, ox);
        const float mk = t.inside ? t.mask : 0.0f;
        const float w00 = t.c00 ? t.hh * t.hw * mk : 0.0f, w01 = t.c01 ? t.hh * t.lw * mk : 0.0f;
        const float w10 = t.c10 ? t.lh * t.hw * mk : 0.0f, w11 = t.c11 ? t.lh * t.lw * mk : 0.0f;
        float* dst = p.col + ((size_t)b * T + tap) * g.C * HoWo + px;
        // the two corners of a row as ONE 8-byte load (as the coordinate-gradient walk does): the pair starts at column
        // w0, moved inside the row at the left / right edge, where the corner that exists sits in the pair's other half --
        // the same four values and weights as four single loads (bit-identical), half the gather instructions
        const bool paired = g.W >= 2;                           // (uniform)
        const bool ledge = t.w0 < 0, redge = t.w0 > g.W - 2;
        const int wa = ledge ? 0 : (redge ? g.W - 2 : t.w0);
        const int ht = t.h0 < 0 ? 0 : t.h0, hb = t.h0 + 1 > g.H - 1 ? g.H - 1 : t.h0 + 1;
        const int qT = t.inside ? ht * g.W + wa : 0, qB = t.inside ? hb * g.W + wa : 0;
        struct __attribute__((packed, aligned(4))) Pair { float l, r; };
        for (int c0 = 0; c0 < g.C; c0 += 8) {
            float e00[8], e01[8], e10[8], e11[8];
            if (paired) {
                Pair pt[8], pb[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const float* plane = in_b + (size_t)(c0 + u < g.C ? c0 + u : g.C - 1) * HW;
                    pt[u] = *reinterpret_cast<const Pair*>(plane + qT);
                    pb[u] = *reinterpret_cast<const Pair*>(plane + qB);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    e00[u] = redge ? pt[u].r : pt[u].l; e01[u] = ledge ? pt[u].l : pt[u].r;
                    e10[u] = redge ? pb[u].r : pb[u].l; e11[u] = ledge ? pb[u].l : pb[u].r;
                }
            } else {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const float* plane = in_b + (size_t)(c0 + u < g.C ? c0 + u : g.C - 1) * HW;
                    e00[u] = plane[t.o00]; e01[u] = plane[t.o01]; e10[u] = plane[t.o10]; e11[u] = plane[t.o11];
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (c0 + u < g.C)
                    dst[(size_t)(c0 + u) * HoWo] = w00 * e00[u] + w01 * e01[u] + w10 * e10[u] + w11 * e11[u];
        }
    }
}

struct DcnColsParams {
    DcnGeom g;
    const float *col, *bias;
    float act_slope;
    float* out;
    float* stats;  // (as DcnFwdParams)
    int stats_mp;
};
struct DcnColsLoader {
    using Params = DcnColsParams;
    static const char* name() { return "DcnColsLoader"; }
    static constexpr bool kHasSideOutput = false;
    const float* base;
    int K, HoWo;
    bool valid;
    __device__ DcnColsLoader(const Params& p, long long n, bool n_valid) : valid(n_valid) {
        HoWo = p.g.Ho * p.g.Wo;
        K = p.g.kh * p.g.kw * p.g.C;
        const int nn = n_valid ? (int)n : 0;
        const int b = nn / HoWo, pp = nn - b * HoWo;
        base = p.col + (size_t)b * K * HoWo + pp;
    }
    __device__ __forceinline__ void load(int k0, int ksub, float (&v)[8]) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = k0 + ksub + 2 * j;
            v[j] = (valid && k < K) ? base[(size_t)k * HoWo] : 0.0f;
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
            v += p.bias[m];
            if (p.act_slope >= 0.0f && v < 0.0f) v *= p.act_slope;
            base[(size_t)m * HoWo] = v;
        }
        static constexpr bool kVec4 = true;      // 16-byte epilogue (igemm.cuh)
        __device__ static bool vec4_ok(const Params& p) { return ((p.g.Ho * p.g.Wo) & 3) == 0; }
        __device__ __forceinline__ void store4(const Params& p, int m, f32x4 v) {
            v += p.bias[m];
            if (p.act_slope >= 0.0f) {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (v[e] < 0.0f) v[e] *= p.act_slope;
            }
            *reinterpret_cast<f32x4*>(base + (size_t)m * HoWo) = v;
        }
    };
};

// the same row reads with buffer addressing: per-lane (image, pixel) offset, the row as a scalar offset
struct DcnColsBufLoader {
    using Params = DcnColsParams;
    static const char* name() { return "DcnColsBufLoader"; }
    static constexpr bool kHasSideOutput = false;
    buf_rsrc rs;
    unsigned voff;
    int K, HoWo;
    __device__ DcnColsBufLoader(const Params& p, long long n, bool n_valid) {
        HoWo = p.g.Ho * p.g.Wo;
        K = p.g.kh * p.g.kw * p.g.C;
        const int nn = n_valid ? (int)n : 0;
        const int b = nn / HoWo, pp = nn - b * HoWo;
        rs = ig_make_rsrc(p.col, (unsigned)((size_t)p.g.B * K * HoWo * sizeof(float)));
        voff = n_valid ? (unsigned)(b * K * HoWo + pp) * (unsigned)sizeof(float) : IG_BUF_OOB;
    }
    __device__ __forceinline__ void load(int k0, int ksub, float (&v)[8]) {
        const int ks = k0 + __builtin_amdgcn_readfirstlane(ksub);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = ks + 2 * j;
            // rows past K (K padding): the scalar offset of row K - 1 and a weight of zero
            v[j] = ig_buf_load(rs, voff, (unsigned)((k < K ? k : K - 1) * HoWo) * (unsigned)sizeof(float));
        }
    }
    using Out = DcnColsLoader::Out;
};

// ---------------------------------------------------------------------------
// backward (1): the column gradient dcol[b][(tap,c)][p] is produced by a plain
// 1x1 implicit GEMM over grad_output (weights transposed by dcn_prep_kernel) and consumed by
// two HBM-streaming kernels that need no MFMA, no workgroup barriers and few registers:
//   dcn_coord_grad_kernel : one thread per (pixel, tap), channels serial -> grad_offset /
//                           grad_mask written once (no atomics, no memset) + a 16-byte
//                           geometry record per (pixel, tap) for the scatter kernel
//   dcn_col2im_kernel     : bilinear scatter of dcol*mask into grad_input through an LDS
//                           window; every wave owns four channel planes.
// ---------------------------------------------------------------------------
struct DcnGeo { int cell; float lh, lw, mask; };   // cell = h0 << 16 | (w0 & 0xffff); h0 = -32768: tap outside
static_assert(sizeof(DcnGeo) == 16, "geometry record is one dwordx4");

// channels c0 .. c0+7 of one (pixel, tap) out of the quad-interleaved column gradient: dq = its pixel's cell in row quad 0 of
// the tap, row quads HoWo cells apart.  C % 4 == 0; a second quad past C is read clamped and the caller drops it.
__device__ __forceinline__ void dcn_load_dcol_quads(const float* __restrict__ dq, int c0, int C, int HoWo, float (&d)[8]) {
    const int q0 = c0 >> 2, q1 = c0 + 4 < C ? q0 + 1 : q0;
    const float4 a = *reinterpret_cast<const float4*>(dq + (size_t)q0 * HoWo * 4);
    const float4 b = *reinterpret_cast<const float4*>(dq + (size_t)q1 * HoWo * 4);
    d[0] = a.x; d[1] = a.y; d[2] = a.z; d[3] = a.w;
    d[4] = b.x; d[5] = b.y; d[6] = b.z; d[7] = b.w;
}

struct DcnCoordParams {
    DcnGeom g;
    const float *in, *off, *mask, *dcol;
    float *goff, *gmask;
    DcnGeo* geo;
};
// block = (64 pixels, TW tap slots); no barriers.  QUADS: dcol's rows are interleaved in quads, [T * C / 4][HoWo][4] (C % 4 == 0;
// cnuda_conv2d_forward_rowquads) -- the four channels of a (pixel, tap) are one 16-byte load.
template <bool QUADS>
__global__ __launch_bounds__(1024) void dcn_coord_grad_kernel(DcnCoordParams p, int tiles_per_image) {
    const DcnGeom& g = p.g;
    const int T = g.kh * g.kw, HoWo = g.Ho * g.Wo, HW = g.H * g.W;
    const int b = blockIdx.x / tiles_per_image, tile = blockIdx.x - b * tiles_per_image;
    const int px = tile * 64 + threadIdx.x;
    if (px >= HoWo) return;
    const int oy = px / g.Wo, ox = px - oy * g.Wo;
    const float* in_b = p.in + (size_t)b * g.C * HW;
    for (int tap = threadIdx.y; tap < T; tap += blockDim.y) {
        const Tap t = make_tap(g, p.off + (size_t)b * g.off_bs, p.mask + (size_t)b * g.mask_bs, 0, tap, oy, ox);
        const float* dc = p.dcol + ((size_t)b * T + tap) * g.C * HoWo + (QUADS ? (size_t)px * 4 : (size_t)px);
        float sm = 0.f, sh_ = 0.f, sw_ = 0.f;
        if (t.inside && g.W >= 2) {
            // The two corners of a row are ONE 8-byte load starting at column clamp(w0, 0, W-2) (halves the L1/TA
            // requests, which bound this kernel); s*l / s*r say which element of the pair each corner is -- at the
            // left edge the existing right corner is the pair's left element, at the right edge the existing left
            // corner is its right element.  The three sums are linear in the corners, so the per-tap coefficients
            // of (top.l, top.r, bottom.l, bottom.r) are folded once per tap.
            struct __attribute__((packed, aligned(4))) Pair { float l, r; };
            const bool ledge = t.w0 < 0, redge = t.w0 > g.W - 2;
            const int wa = ledge ? 0 : (redge ? g.W - 2 : t.w0);
            const int ht = t.h0 < 0 ? 0 : t.h0, hb = t.h0 + 1 > g.H - 1 ? g.H - 1 : t.h0 + 1;
            const int qT = ht * g.W + wa, qB = hb * g.W + wa;
            // corner value e_xy = pair.l * L + pair.r * R with (L, R) in {(1,0), (0,1), (0,0)}
            const float l0 = (!ledge && !redge) ? 1.f : 0.f, r0 = redge ? 1.f : 0.f;     // left corner  (column w0)
            const float l1 = ledge ? 1.f : 0.f, r1 = (!ledge && !redge) ? 1.f : 0.f;     // right corner (column w0+1)
            const float f00 = t.c00 ? 1.f : 0.f, f01 = t.c01 ? 1.f : 0.f, f10 = t.c10 ? 1.f : 0.f,
                        f11 = t.c11 ? 1.f : 0.f;
            // coefficient of e00, e01, e10, e11 in: sample (A), d/dh (Bh), d/dw (Bw)
            const float A00 = t.hh * t.hw * f00, A01 = t.hh * t.lw * f01, A10 = t.lh * t.hw * f10, A11 = t.lh * t.lw * f11;
            const float H00 = -t.hw * f00, H01 = -t.lw * f01, H10 = t.hw * f10, H11 = t.lw * f11;
            const float W00 = -t.hh * f00, W01 = t.hh * f01, W10 = -t.lh * f10, W11 = t.lh * f11;
            const float aTl = A00 * l0 + A01 * l1, aTr = A00 * r0 + A01 * r1, aBl = A10 * l0 + A11 * l1, aBr = A10 * r0 + A11 * r1;
            const float hTl = H00 * l0 + H01 * l1, hTr = H00 * r0 + H01 * r1, hBl = H10 * l0 + H11 * l1, hBr = H10 * r0 + H11 * r1;
            const float wTl = W00 * l0 + W01 * l1, wTr = W00 * r0 + W01 * r1, wBl = W10 * l0 + W11 * l1, wBr = W10 * r0 + W11 * r1;
            // All three sums are linear in u = sum_c dcol_c * (top.l, top.r, bottom.l, bottom.r)_c: the channel loop
            // accumulates the four components of u only (4 multiply-adds per channel instead of 14) and the per-tap
            // coefficient rows are applied once at the end.  (Measured: no change in run time -- the kernel is bound
            // by its corner gathers, not by the VALU; kept because it is the simpler arithmetic.)
            float uTl = 0.f, uTr = 0.f, uBl = 0.f, uBr = 0.f;
            for (int c0 = 0; c0 < g.C; c0 += 8) {
                float d[8];
                Pair pt[8], pb[8];
                if constexpr (QUADS) dcn_load_dcol_quads(dc, c0, g.C, HoWo, d);
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int c = c0 + u < g.C ? c0 + u : g.C - 1;
                    const float* plane = in_b + (size_t)c * HW;
                    if constexpr (!QUADS) d[u] = dc[(size_t)c * HoWo];
                    pt[u] = *reinterpret_cast<const Pair*>(plane + qT);
                    pb[u] = *reinterpret_cast<const Pair*>(plane + qB);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const float dd = c0 + u < g.C ? d[u] : 0.f;
                    uTl = fmaf(dd, pt[u].l, uTl); uTr = fmaf(dd, pt[u].r, uTr);      // (-ffp-contract=off: explicit)
                    uBl = fmaf(dd, pb[u].l, uBl); uBr = fmaf(dd, pb[u].r, uBr);
                }
            }
            sm = aTl * uTl + aTr * uTr + aBl * uBl + aBr * uBr;
            sh_ = (hTl * uTl + hTr * uTr + hBl * uBl + hBr * uBr) * t.mask;
            sw_ = (wTl * uTl + wTr * uTr + wBl * uBl + wBr * uBr) * t.mask;
        } else if (t.inside) {     // width 1: no horizontal neighbour to pair with
            const float f00 = t.c00 ? 1.f : 0.f, f01 = t.c01 ? 1.f : 0.f, f10 = t.c10 ? 1.f : 0.f,
                        f11 = t.c11 ? 1.f : 0.f;
            for (int c = 0; c < g.C; ++c) {
                const float* plane = in_b + (size_t)c * HW;
                const float dd = QUADS ? dc[(size_t)(c >> 2) * HoWo * 4 + (c & 3)] : dc[(size_t)c * HoWo];
                const float a00 = plane[t.o00] * f00, a01 = plane[t.o01] * f01, a10 = plane[t.o10] * f10,
                            a11 = plane[t.o11] * f11;
                sm += dd * tap_sample(t, a00, a01, a10, a11);
                const float dm = dd * t.mask;
                sh_ += (-t.hw * a00 - t.lw * a01 + t.hw * a10 + t.lw * a11) * dm;
                sw_ += (-t.hh * a00 + t.hh * a01 - t.lh * a10 + t.lh * a11) * dm;
            }
        }
        if (g.gmask_logit) sm = sm * t.mask * (1.0f - t.mask);        // (cnuda_split_offset_mask_backward's own expression)
        p.gmask[(size_t)b * g.gmask_bs + (size_t)tap * HoWo + px] = sm;
        p.goff[(size_t)b * g.goff_bs + (size_t)(2 * tap) * HoWo + px] = sh_;
        p.goff[(size_t)b * g.goff_bs + (size_t)(2 * tap + 1) * HoWo + px] = sw_;
        DcnGeo r;
        r.cell = t.inside ? (int)(((unsigned)t.h0 << 16) | ((unsigned)t.w0 & 0xffffu)) : (int)0x80000000u;
        r.lh = t.lh; r.lw = t.lw; r.mask = t.inside ? t.mask : 0.f;
        p.geo[((size_t)b * T + tap) * HoWo + px] = r;
    }
}

constexpr int CI_CG = 16;        // channels per workgroup (4 per wave)
constexpr int CI_MARGIN = 2;     // window rows / columns beyond the undeformed footprint (default; DcnPlan::margin is what runs)
struct DcnCol2imParams {
    DcnGeom g;
    const float* dcol;
    const DcnGeo* geo;
    float* gin;
    int TR, TC, tc_shift, tiles_y, tiles_x, ncg, WSZmax, claim_sz, margin;
};
// The col2im walk of ONE wave over ONE 16-channel group of a TR x TC pixel tile (256 pixels): the wave owns
// channel planes c_w..c_w+3 of the LDS window ([cell][4 channels]: the four channels of a cell are ONE 16-byte
// access), zeroes them, walks all 256 pixels x taps (64 pixels per step, lanes along x) doing plain LDS
// read-add-write one corner at a time, and flushes every touched in-image cell with one coalesced atomic.  No
// workgroup synchronisation.
// Written for INSTRUCTION COUNT (round 3, in-kernel stamps: an item of this walk cost ~5,000 cycles, ~650 vector +
// scalar instructions -- the walk was issue-bound, not LDS- or memory-bound):
//   * the window is the UNCLIPPED rectangle around the tile (cells outside the image exist in LDS and are simply
//     never flushed), so a sample is either wholly inside the window (two unsigned compares) or a stray -- no
//     per-corner image / window predicates, no zeroed weights: what lands outside the image is dropped at the flush;
//   * row bases of the dcol / geometry loads are scalar (the wave id comes through readfirstlane), the lane adds a
//     32-bit byte offset (`global_load ... v_off, s[base]`); (group, tap) run on counters, products on 24-bit mads;
//   * the loads of an item are issued two items ahead.
// Two lanes hit the same cell in one instruction only if their anchors (h0, w0) are equal:
//   * fast path (a step whose 64 lanes lie on one output row and whose samples all sit within [-2, 2) cell columns
//     of their undeformed position): only lanes at most 3 apart can share an anchor, so three DPP shifts rank
//     every lane among its equals -- no LDS round trip; round r scatters the lanes of rank r;
//   * otherwise every lane writes its id into a per-wave claim map at its anchor and reads it back; the
//     survivor scatters, losers retry (<= 3 rounds), leftovers and strays use global atomics.
struct DcnScatterCtx {
    int y0, x0, TC, tc_shift;                 // tile origin, tile width
    int wy0, wx0, WR, WC, WSZ;                // window (unclipped; WR = WC = 0: no window, global atomics only)
    float4* wp;                               // this wave's window planes
    float4* dump;                             // a private cell for inactive lanes
    lds_vu8* claim;                           // this wave's claim map [WSZ]
};
__device__ __forceinline__ void dcn_scatter_window(DcnScatterCtx& x, const DcnGeom& g, int TR, int WSZmax, int margin) {
    x.wy0 = x.y0 * g.sh - g.ph - margin;
    x.wx0 = x.x0 * g.sw - g.pw - margin;
    x.WR = WSZmax ? (TR - 1) * g.sh + (g.kh - 1) * g.dh + 2 * margin + 1 : 0;
    x.WC = WSZmax ? (x.TC - 1) * g.sw + (g.kw - 1) * g.dw + 2 * margin + 1 : 0;
    x.WSZ = x.WR * x.WC;                       // == WSZmax (or 0: the window does not fit the LDS)
}
template <bool QUADS>
__device__ __forceinline__ void dcn_scatter_group(const DcnScatterCtx& x, const DcnGeom& g, const DcnGeo* __restrict__ geo_b,
                                                  const float* __restrict__ dcol_b, float* __restrict__ gin_b, int c_w,
                                                  int lane) {
    const int T = g.kh * g.kw, HoWo = g.Ho * g.Wo, HW = g.H * g.W;
    const int WR = x.WR, WC = x.WC, WSZ = x.WSZ, wy0 = x.wy0, wx0 = x.wx0;
    float4* const wp = x.wp;
    for (int i = lane; i < WSZ; i += 64) wp[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int items = 4 * T;                          // (pixel group, tap)
    int cc[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) cc[r] = c_w + r < g.C ? c_w + r : g.C - 1;   // clamped loads; results dropped
    if constexpr (QUADS) cc[0] = c_w < g.C ? c_w : g.C - 4;                  // (the quad's first channel: C % 4 == 0)
    const float cvf[4] = {c_w < g.C ? 1.f : 0.f, c_w + 1 < g.C ? 1.f : 0.f, c_w + 2 < g.C ? 1.f : 0.f,
                          c_w + 3 < g.C ? 1.f : 0.f};
    const bool one_row = x.TC == 64;                  // a step's 64 lanes are 64 consecutive pixels of one row
    struct Item { DcnGeo rec; float d[4]; bool valid; int ox; };
    int f_grp = 0, f_tap = 0;                         // (group, tap) of the next fetch: items run group-major
    auto fetch = [&](Item& o) {
        const int t = f_grp * 64 + lane;
        const int oy = x.y0 + (t >> x.tc_shift);
        o.ox = x.x0 + (t & (x.TC - 1));
        o.valid = oy < g.Ho && o.ox < g.Wo;
        unsigned px = o.valid ? (unsigned)(__mul24(oy, g.Wo) + o.ox) : 0u;
        asm volatile("" : "+v"(px));                  // (keeps the zero-extension next to the loads: scalar-base form)
        const unsigned long long gb = reinterpret_cast<unsigned long long>(geo_b + (size_t)f_tap * HoWo);
        o.rec = __builtin_bit_cast(DcnGeo, *reinterpret_cast<__attribute__((address_space(1))) const u32x4*>(gb + (unsigned long long)(px * 16u)));
        if constexpr (QUADS) {       // (c_w is a multiple of 4 below C: the wave's four channels are one cell)
            const unsigned long long db = reinterpret_cast<unsigned long long>(dcol_b + (size_t)(f_tap * g.C + cc[0]) * HoWo);
            const f32x4 t4 = *reinterpret_cast<__attribute__((address_space(1))) const f32x4*>(db + (unsigned long long)(px * 16u));
            o.d[0] = t4[0]; o.d[1] = t4[1]; o.d[2] = t4[2]; o.d[3] = t4[3];
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const unsigned long long db = reinterpret_cast<unsigned long long>(dcol_b + (size_t)(f_tap * g.C + cc[r]) * HoWo);
                o.d[r] = *reinterpret_cast<__attribute__((address_space(1))) const float*>(db + (unsigned long long)(px * 4u));
            }
        }
        if (++f_tap == T) { f_tap = 0; ++f_grp; }
    };
    Item q0i, q1i;                                    // even / odd items in flight
    fetch(q0i);
    if (1 < items) fetch(q1i);
    int c_tapx = 0;                                   // kernel column of the current item's tap
#pragma unroll 1
    for (int it0 = 0; it0 < items; it0 += 2)
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int it = it0 + half;
        if (it >= items) break;
        Item& slot = half == 0 ? q0i : q1i;
        const DcnGeo rec = slot.rec;
        float d[4] = {slot.d[0], slot.d[1], slot.d[2], slot.d[3]};
        const bool valid = slot.valid;
        const int ox = slot.ox;
        if (it + 2 < items) fetch(slot);
        const int tapx = c_tapx;
        if (++c_tapx == g.kw) c_tapx = 0;             // (T = kh * kw taps per group: the column counter wraps with it)
        const int h0 = rec.cell >> 16, w0 = (int)(short)(rec.cell & 0xffff);
        const bool live = valid && rec.cell != (int)0x80000000u;
        const float lh = rec.lh, lw = rec.lw, hh = 1.0f - lh, hw = 1.0f - lw;
        const float k00 = hh * hw, k01 = hh * lw, k10 = lh * hw, k11 = lh * lw;
#pragma unroll
        for (int r = 0; r < 4; ++r) d[r] *= rec.mask * cvf[r];
        const int rh = h0 - wy0, rw = w0 - wx0;
        // wholly inside the window (both rows, both columns)?  else a stray
        const bool near = WSZ > 0 && live && (unsigned)rh < (unsigned)(WR - 1) && (unsigned)rw < (unsigned)(WC - 1);
        const int base = __mul24(rh, WC) + rw;
        const int anchor = near ? base : -1;
        auto scatter = [&](bool on, int cell, float k) {
            float4* a = on ? wp + cell : x.dump;
            const float kk = on ? k : 0.f;
            float4 cur = *a;
            cur.x += kk * d[0]; cur.y += kk * d[1]; cur.z += kk * d[2]; cur.w += kk * d[3];
            *a = cur;
            asm volatile("" ::: "memory");     // LDS program order between corners (neighbouring lanes' cells)
        };
        bool done_lds = false;
        // displacement of the sample's cell column from the undeformed one: with every near lane in {-2 .. 1}
        // (horizontal offsets in [-2, 2)) two lanes can share an anchor only if they are at most 3 lanes apart
        const int disp = w0 - (__mul24(ox, g.sw) - g.pw + __mul24(tapx, g.dw));
        if (one_row && !__any(near && (unsigned)(disp + 2) > 3u)) {
            // rank = number of EARLIER lanes (distance 1..3) with the same anchor, found with three DPP shifts -- no
            // LDS round trip; round r scatters the lanes of rank r (LDS instructions of one wave execute in order)
            const int p1 = __builtin_amdgcn_update_dpp(-2, anchor, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
            const int p2 = __builtin_amdgcn_update_dpp(-2, p1, 0x138, 0xf, 0xf, false);
            const int p3 = __builtin_amdgcn_update_dpp(-2, p2, 0x138, 0xf, 0xf, false);
            const int rank = near ? (int)(p1 == anchor) + (int)(p2 == anchor) + (int)(p3 == anchor) : -1;
#pragma unroll 1
            for (int round = 0; round < 4; ++round) {
                const bool mine = rank == round;
                if (round > 0 && !__any(mine)) break;          // ranks are dense: none of rank r -> none above
                scatter(mine, base, k00);
                scatter(mine, base + 1, k01);
                scatter(mine, base + WC, k10);
                scatter(mine, base + WC + 1, k11);
            }
            done_lds = near;
        } else {
            // rounds: lanes whose anchor is claimed by another lane of this instruction wait for the next round
            // (one round unless two pixels of the wave sample the same cell); after three rounds the rest spills
            bool pending = near;
#pragma unroll 1
            for (int round = 0; round < 3 && __any(pending); ++round) {
                bool won = false;
                if (pending) {
                    x.claim[anchor] = (unsigned char)lane;
                    won = x.claim[anchor] == (unsigned char)lane;     // same wave, LDS executes in order
                }
                asm volatile("" ::: "memory");
                scatter(won, base, k00);
                scatter(won, base + 1, k01);
                scatter(won, base + WC, k10);
                scatter(won, base + WC + 1, k11);
                done_lds = done_lds || won;
                pending = pending && !won;
            }
        }
        // strays (a corner outside the window) and claim-round leftovers: global atomics, corners checked against the image
        if (__any(live && !done_lds)) {
            if (live && !done_lds) {
                const bool top = h0 >= 0, bot = h0 + 1 <= g.H - 1, lef = w0 >= 0, rig = w0 + 1 <= g.W - 1;
                const int o = h0 * g.W + w0;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (c_w + r >= g.C) continue;
                    float* plane = gin_b + (size_t)(c_w + r) * HW;
                    if (top && lef) atomicAdd(plane + o, k00 * d[r]);
                    if (top && rig) atomicAdd(plane + o + 1, k01 * d[r]);
                    if (bot && lef) atomicAdd(plane + o + g.W, k10 * d[r]);
                    if (bot && rig) atomicAdd(plane + o + g.W + 1, k11 * d[r]);
                }
            }
        }
    }
    // flush: a wave walks the window row by row (lanes along the row: coalesced), four planes each; cells outside the
    // image were scratch
    asm volatile("" ::: "memory");
    for (int yy = 0; yy < WR; ++yy) {
        const int gy = wy0 + yy;
        if (gy < 0 || gy >= g.H) continue;            // (uniform)
        for (int xx = lane; xx < WC; xx += 64) {
            const int gx = wx0 + xx;
            if (gx < 0 || gx >= g.W) continue;
            const float4 v4 = wp[yy * WC + xx];
            const float v[4] = {v4.x, v4.y, v4.z, v4.w};
            float* gcell = gin_b + (size_t)gy * g.W + gx;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (v[r] != 0.0f && c_w + r < g.C) atomicAdd(gcell + (size_t)(c_w + r) * HW, v[r]);
        }
    }
    asm volatile("" ::: "memory");
}

// Workgroup = (image, TR x TC tile of output pixels (256), 16 channels); the four waves never synchronise.
template <bool QUADS>
__global__ __launch_bounds__(256) void dcn_col2im_kernel(DcnCol2imParams p, int n_wg) {
    extern __shared__ __align__(16) float win[];     // 4 waves x [WSZ cells][4 channels] + 256 dump cells x 4 + claim maps
    const DcnGeom& g = p.g;
    const int T = g.kh * g.kw, HoWo = g.Ho * g.Wo, HW = g.H * g.W;
    // (wid through readfirstlane: the compiler then KNOWS it is wave-uniform, and everything derived from it -- channel
    // planes, row bases of the dcol loads, the window base -- is scalar arithmetic instead of per-lane 64-bit math)
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    int id = xcd_remap(blockIdx.x, n_wg);
    const int cg = id % p.ncg; id /= p.ncg;
    const int tx = id % p.tiles_x; id /= p.tiles_x;
    const int ty = id % p.tiles_y;
    const int b = id / p.tiles_y;
    DcnScatterCtx x;
    x.y0 = ty * p.TR; x.x0 = tx * p.TC; x.TC = p.TC; x.tc_shift = p.tc_shift;
    dcn_scatter_window(x, g, p.TR, p.WSZmax, p.margin);
    x.wp = reinterpret_cast<float4*>(win) + (size_t)wid * x.WSZ;
    x.dump = reinterpret_cast<float4*>(win + CI_CG * p.WSZmax) + tid;
    x.claim = (lds_vu8*)(reinterpret_cast<unsigned char*>(win + CI_CG * p.WSZmax + 4 * 256) + wid * p.claim_sz);
    *x.dump = make_float4(0.f, 0.f, 0.f, 0.f);
    dcn_scatter_group<QUADS>(x, g, p.geo + (size_t)b * T * HoWo, p.dcol + (size_t)b * T * g.C * HoWo,
                             p.gin + (size_t)b * g.C * HW, cg * CI_CG + wid * 4, lane);
}

// ---------------------------------------------------------------------------
// backward (1b): the two consumers of dcol as ONE launch for the large feature maps (round 3).
// dcn_coord_grad_kernel is bound by its corner gathers (L1 / texture-address cycles; no LDS), dcn_col2im_kernel by
// dependent LDS read-add-write chains and its atomic window flush (no gathers): run one after the other each
// leaves the other's unit idle.  Here a workgroup of 8 waves owns one (image, TR x TC pixel tile) for ALL
// channels, in two roles that never synchronise:
//   waves 0-3 (scatter): the col2im walk -- wave w owns channel planes 4w..4w+3 of a 16-channel LDS window, and
//                        loops over the channel groups (zero, walk 4 x T items, flush) on its own;
//   waves 4-7 (gather) : the coordinate-gradient walk -- wave v owns pixel group v (64 pixels), taps and
//                        channels serial, plain stores of grad_offset / grad_mask.
// The per-(pixel, tap) geometry records both roles read are written by dcn_prep_kernel (with the transposed
// weights of the column-gradient GEMM: one small launch in front of it).  Small maps (fewer tiles than CUs)
// keep the two-kernel form, whose grid also spans the channel groups.
// ---------------------------------------------------------------------------
struct DcnPrepParams {
    DcnGeom g;
    const float *w, *off, *mask;
    float* wt;
    DcnGeo* geo;
    int wt_blocks;
    float* zero;              // nullable: grad_input, cleared by the blocks behind the other two kinds (round 6: was a memset launch)
    long long zero_quads;     // its size in 16-byte quads (the tensor is 16-byte aligned and a multiple of 4 elements, else nullptr)
    int zero_first;           // first block of that kind
};
__global__ __launch_bounds__(256) void dcn_prep_kernel(DcnPrepParams p) {
    const DcnGeom& g = p.g;
    if (p.zero && (int)blockIdx.x >= p.zero_first) {
        const long long nb = (long long)gridDim.x - p.zero_first;
        float4* z = reinterpret_cast<float4*>(p.zero);
        for (long long i = ((long long)blockIdx.x - p.zero_first) * 256 + threadIdx.x; i < p.zero_quads; i += nb * 256)
            z[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        return;
    }
    const int T = g.kh * g.kw;
    if ((int)blockIdx.x < p.wt_blocks) {          // wt[(tap*C + c)][o] = w[o][c][tap]
        const long long total = (long long)g.Co * g.C * T;
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)p.wt_blocks * 256) {
            const int o = (int)(i % g.Co);
            const long long r = i / g.Co;
            const int c = (int)(r % g.C), tap = (int)(r / g.C);
            p.wt[i] = p.w[((size_t)o * g.C + c) * T + tap];
        }
        return;
    }
    if (!p.geo) return;
    const int HoWo = g.Ho * g.Wo;
    const long long total = (long long)g.B * T * HoWo;
    const long long nb = (long long)(p.zero ? p.zero_first : (int)gridDim.x) - p.wt_blocks;
    for (long long i = ((long long)blockIdx.x - p.wt_blocks) * 256 + threadIdx.x; i < total; i += nb * 256) {
        const int px = (int)(i % HoWo);
        const long long r = i / HoWo;
        const int tap = (int)(r % T), b = (int)(r / T);
        const int oy = px / g.Wo, ox = px - oy * g.Wo;
        const Tap t = make_tap(g, p.off + (size_t)b * g.off_bs, p.mask + (size_t)b * g.mask_bs, 0, tap, oy, ox);
        DcnGeo rec;
        rec.cell = t.inside ? (int)(((unsigned)t.h0 << 16) | ((unsigned)t.w0 & 0xffffu)) : (int)0x80000000u;
        rec.lh = t.lh; rec.lw = t.lw; rec.mask = t.inside ? t.mask : 0.f;
        p.geo[i] = rec;
    }
}

struct DcnBwdDataParams {
    DcnGeom g;
    const float *in, *dcol;
    const DcnGeo* geo;
    float *gin, *goff, *gmask;
    int TR, TC, tc_shift, tiles_y, tiles_x, ncg, WSZmax, claim_sz, margin;
    int nsplit;      // workgroups per tile: workgroup s takes the channel groups [s*ncg/nsplit, (s+1)*ncg/nsplit) and the taps = s (mod nsplit)
};
template <bool QUADS>
__global__ __launch_bounds__(512, 6) void dcn_bwd_data_kernel(DcnBwdDataParams p, int n_wg) {
    extern __shared__ __align__(16) float win[];     // as dcn_col2im_kernel: 4 waves x [WSZ][4] + dump cells + claim maps
    const DcnGeom& g = p.g;
    const int T = g.kh * g.kw, HoWo = g.Ho * g.Wo, HW = g.H * g.W;
    const int tid = threadIdx.x & 255, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);   // uniform: see col2im
    int id = xcd_remap(blockIdx.x, n_wg);
    const int part = id % p.nsplit; id /= p.nsplit;
    const int tx = id % p.tiles_x; id /= p.tiles_x;
    const int ty = id % p.tiles_y;
    const int b = id / p.tiles_y;
    const int y0 = ty * p.TR, x0 = tx * p.TC;
    const DcnGeo* geo_b = p.geo + (size_t)b * T * HoWo;
    const float* dcol_b = p.dcol + (size_t)b * T * g.C * HoWo;
    auto pixel_of = [&](int grp, bool& valid) {
        const int t = grp * 64 + lane;
        const int oy = y0 + (t >> p.tc_shift), ox = x0 + (t & (p.TC - 1));
        valid = oy < g.Ho && ox < g.Wo;
        return valid ? oy * g.Wo + ox : 0;
    };
#ifdef DCN_ABLATE          // A/B builds only (profiles/microbench): 1 = no gather role, 2 = no scatter role
    if ((DCN_ABLATE == 1) == (threadIdx.x >= 256)) return;
#endif
    if (threadIdx.x >= 256) {
        // ---- gather role: grad_offset / grad_mask of pixel group `wid` (dcn_coord_grad_kernel's arithmetic) ----
        bool valid;
        const int px = pixel_of(wid, valid);
        if (!valid) return;
        const float* in_b = p.in + (size_t)b * g.C * HW;
        struct __attribute__((packed, aligned(4))) Pair { float l, r; };
#pragma unroll 1
        for (int tap = part; tap < T; tap += p.nsplit) {
            const DcnGeo rec = geo_b[(size_t)tap * HoWo + px];
            const float* dc = dcol_b + (size_t)tap * g.C * HoWo + (QUADS ? (size_t)px * 4 : (size_t)px);
            float sm = 0.f, sh_ = 0.f, sw_ = 0.f;
            if (rec.cell != (int)0x80000000u) {
                const int h0 = rec.cell >> 16, w0 = (int)(short)(rec.cell & 0xffff);
                const float lh = rec.lh, lw = rec.lw, hh = 1.0f - lh, hw = 1.0f - lw;
                const bool ledge = w0 < 0, redge = w0 > g.W - 2;
                const int wa = ledge ? 0 : (redge ? g.W - 2 : w0);
                const int ht = h0 < 0 ? 0 : h0, hb = h0 + 1 > g.H - 1 ? g.H - 1 : h0 + 1;
                const int qT = ht * g.W + wa, qB = hb * g.W + wa;
                const float l0 = (!ledge && !redge) ? 1.f : 0.f, r0 = redge ? 1.f : 0.f;
                const float l1 = ledge ? 1.f : 0.f, r1 = (!ledge && !redge) ? 1.f : 0.f;
                const bool top = h0 >= 0, bot = h0 + 1 <= g.H - 1, lef = w0 >= 0, rig = w0 + 1 <= g.W - 1;
                const float f00 = (top && lef) ? 1.f : 0.f, f01 = (top && rig) ? 1.f : 0.f, f10 = (bot && lef) ? 1.f : 0.f,
                            f11 = (bot && rig) ? 1.f : 0.f;
                const float A00 = hh * hw * f00, A01 = hh * lw * f01, A10 = lh * hw * f10, A11 = lh * lw * f11;
                const float H00 = -hw * f00, H01 = -lw * f01, H10 = hw * f10, H11 = lw * f11;
                const float W00 = -hh * f00, W01 = hh * f01, W10 = -lh * f10, W11 = lh * f11;
                const float aTl = A00 * l0 + A01 * l1, aTr = A00 * r0 + A01 * r1, aBl = A10 * l0 + A11 * l1, aBr = A10 * r0 + A11 * r1;
                const float hTl = H00 * l0 + H01 * l1, hTr = H00 * r0 + H01 * r1, hBl = H10 * l0 + H11 * l1, hBr = H10 * r0 + H11 * r1;
                const float wTl = W00 * l0 + W01 * l1, wTr = W00 * r0 + W01 * r1, wBl = W10 * l0 + W11 * l1, wBr = W10 * r0 + W11 * r1;
                float uTl = 0.f, uTr = 0.f, uBl = 0.f, uBr = 0.f;
                for (int c0 = 0; c0 < g.C; c0 += 8) {
                    float d[8];
                    Pair pt[8], pb[8];
                    if constexpr (QUADS) dcn_load_dcol_quads(dc, c0, g.C, HoWo, d);
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int c = c0 + u < g.C ? c0 + u : g.C - 1;
                        const float* plane = in_b + (size_t)c * HW;
                        if constexpr (!QUADS) d[u] = dc[(size_t)c * HoWo];
                        pt[u] = *reinterpret_cast<const Pair*>(plane + qT);
                        pb[u] = *reinterpret_cast<const Pair*>(plane + qB);
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const float dd = c0 + u < g.C ? d[u] : 0.f;
                        uTl = fmaf(dd, pt[u].l, uTl); uTr = fmaf(dd, pt[u].r, uTr);
                        uBl = fmaf(dd, pb[u].l, uBl); uBr = fmaf(dd, pb[u].r, uBr);
                    }
                }
                sm = aTl * uTl + aTr * uTr + aBl * uBl + aBr * uBr;
                sh_ = (hTl * uTl + hTr * uTr + hBl * uBl + hBr * uBr) * rec.mask;
                sw_ = (wTl * uTl + wTr * uTr + wBl * uBl + wBr * uBr) * rec.mask;
            }
            if (g.gmask_logit) sm = sm * rec.mask * (1.0f - rec.mask);    // (sm == 0 where the record's mask was zeroed: outside taps)
            p.gmask[(size_t)b * g.gmask_bs + (size_t)tap * HoWo + px] = sm;
            p.goff[(size_t)b * g.goff_bs + (size_t)(2 * tap) * HoWo + px] = sh_;
            p.goff[(size_t)b * g.goff_bs + (size_t)(2 * tap + 1) * HoWo + px] = sw_;
        }
        return;
    }
    // ---- scatter role: dcn_col2im_kernel's walk, channel groups serial ----
    DcnScatterCtx x;
    x.y0 = y0; x.x0 = x0; x.TC = p.TC; x.tc_shift = p.tc_shift;
    dcn_scatter_window(x, g, p.TR, p.WSZmax, p.margin);
    x.wp = reinterpret_cast<float4*>(win) + (size_t)wid * x.WSZ;
    x.dump = reinterpret_cast<float4*>(win + CI_CG * p.WSZmax) + tid;
    x.claim = (lds_vu8*)(reinterpret_cast<unsigned char*>(win + CI_CG * p.WSZmax + 4 * 256) + wid * p.claim_sz);
    *x.dump = make_float4(0.f, 0.f, 0.f, 0.f);
    float* gin_b = p.gin + (size_t)b * g.C * HW;
    const int cg_end = (part + 1) * p.ncg / p.nsplit;
#pragma unroll 1
    for (int cg = part * p.ncg / p.nsplit; cg < cg_end; ++cg)
        dcn_scatter_group<QUADS>(x, g, geo_b, dcol_b, gin_b, cg * CI_CG + wid * 4, lane);
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
        const int HW = g.H * g.W, T = g.kh * g.kw, K = T * g.C;
        const float* in_b = p.in + (size_t)b_ * g.C * HW;
        const float* off_b = p.off + (size_t)b_ * g.off_bs;
        const float* mask_b = p.mask + (size_t)b_ * g.mask_bs;
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

// the same two row reads with buffer addressing (igemm.cuh: ig_buf_rows)
struct DcnColWBufLoader {
    using Params = DcnColWParams;
    const Params& p;
    buf_rsrc rg, rc;
    IgPixelCursor c;
    unsigned gimg, cimg;          // byte offsets of the cursor's image in grad_output / the columns
    __device__ DcnColWBufLoader(const Params& pp, long long n, long long n_end) : p(pp) {
        const DcnGeom& g = p.g;
        const size_t px = (size_t)g.B * g.Ho * g.Wo;
        rg = ig_make_rsrc(p.gout, (unsigned)(px * g.Co * sizeof(float)));
        rc = ig_make_rsrc(p.col, (unsigned)(px * g.kh * g.kw * g.C * sizeof(float)));
        c.init(n, n_end, g.Ho * g.Wo, g.Wo);
        gimg = (unsigned)(c.b_ * g.Co * g.Ho * g.Wo) * 4u;
        cimg = (unsigned)(c.b_ * g.kh * g.kw * g.C * g.Ho * g.Wo) * 4u;
    }
    __device__ __forceinline__ void advance() {
        const DcnGeom& g = p.g;
        c.advance(g.Ho * g.Wo, g.Wo);
        if (c.crossed_) {
            gimg += (unsigned)(c.crossed_ * g.Co * g.Ho * g.Wo) * 4u;
            cimg += (unsigned)(c.crossed_ * g.kh * g.kw * g.C * g.Ho * g.Wo) * 4u;
        }
    }
    template <int NV, int STEP>
    __device__ __forceinline__ void load_g(int m0, int msub, float (&v)[NV]) {
        ig_buf_rows<NV, STEP>(rg, c, gimg, p.g.Ho * p.g.Wo, m0, msub, v);
    }
    template <int NV, int STEP>
    __device__ __forceinline__ void load_b(int j0, int jsub, float (&v)[NV]) {
        ig_buf_rows<NV, STEP>(rc, c, cimg, p.g.Ho * p.g.Wo, j0, jsub, v);
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
                const Tap t = make_tap(g, p.off + (size_t)b * g.off_bs,
                                       p.mask + (size_t)b * g.mask_bs, grp, tap, oy, ox);
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
        const Tap t = make_tap(g, p.off + (size_t)b * g.off_bs, p.mask + (size_t)b * g.mask_bs, grp,
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
                (H + 2 * ph - (dh * (kh - 1) + 1)) / sh + 1, (W + 2 * pw - (dw * (kw - 1) + 1)) / sw + 1, 0, 0, 0, 0, 0};
    CNUDA_REQUIRE(g.Ho > 0 && g.Wo > 0, "%s: kernel larger than padded input", who);
    CNUDA_REQUIRE((long long)dg * 3 * kh * kw * g.Ho * g.Wo < (1ll << 31), "%s: offset planes of an image exceed 2^31 elements", who);
    g.off_bs = g.goff_bs = dg * 2 * kh * kw * g.Ho * g.Wo;       // the reference's separate offset / mask tensors
    g.mask_bs = g.gmask_bs = dg * kh * kw * g.Ho * g.Wo;
    return 0;
}

int pick_bm(int M, long long N) {
    int bm = M > 64 ? 128 : (M > 32 ? 64 : 32);
    const long long n_tiles = (N + IG_BN - 1) / IG_BN;
    while (bm > 32 && n_tiles * ((M + bm - 1) / bm) < 512) bm >>= 1;
    return bm;
}

constexpr int kFusedMinTilesDefault = 512;
int g_fused_min_tiles = kFusedMinTilesDefault;      // cnuda_dcn_set_fused_min_tiles (tests)
int g_scatter_margin = 0;                           // cnuda_dcn_set_scatter_margin (measurements, tests); 0: by regime
int g_walk_tc = 0;                                  // cnuda_dcn_set_walk_tile (measurements); 0: by the map width
// Offset regime of the NEXT forward / backward call (cnuda_dcn_set_offset_regime; the host layer sets it per call from a
// census of the layer's own offsets, libs/DCNv2/dcn_v2.py).  A freshly initialised model samples within a fraction of a
// pixel of the regular grid; a trained CenterNet has learned offsets of pixels.  The kernels that keep a window on chip
// are sized for the first case and pay per sample that leaves the window (round 5, B = 32, 64 -> 64 at 128 x 128, us per
// launch at offsets of sigma 0.5 / 1 / 1.4 / 2 px -- data-gradient walk with a window margin of 2 cells: 1085 / 1322 /
// 1727 / 3730, of 4 cells: 1106 / 1235 / 1336 / 1600; profiles/r5_dcn_margin_sweep.txt):
//   bit 0: many samples beyond +-2 px -> the walk's window gets a margin of 4 cells (2 workgroups per CU instead of 3)
//   bit 1: many samples beyond +-3 px -> the forward takes the gathering loader instead of the LDS-window kernel
int g_offset_regime = 0;


struct DcnPlan {
    int T, K, Kp, bm, Mp;           // forward pack [Kp][Mp]
    int Mpw, Jp, Z;                 // wgrad slabs [Z][Mpw][Jp]
    long long N, pix_per_split;
    size_t fwd_bytes, bwd_bytes;
    bool fwd_two_kernels;           // several M tiles: sample the columns once, then a plain GEMM
    // split backward: 1x1 GEMM workspace, transposed weights, dcol, geometry records; col2im tiling
    size_t gemm_bytes;
    int TR, TC, tc_shift, tiles_y, tiles_x, ncg, WSZmax, claim_sz, margin;
    size_t col2im_lds;
    bool fused_consumers;           // dcn_bwd_data_kernel (large maps) instead of coord_grad + col2im
    int fused_split;                // its workgroups per tile
};
DcnPlan make_plan(const DcnGeom& g) {
    DcnPlan q;
    q.T = g.kh * g.kw;
    q.K = q.T * g.C;
    q.Kp = round_up(q.K, IG_KC);
    q.N = (long long)g.B * g.Ho * g.Wo;
    q.bm = pick_bm(g.Co, q.N);
    q.Mp = round_up(g.Co, q.bm);
    q.Mpw = round_up(g.Co, WG_BM);
    q.Jp = round_up(q.K, WG_BJ);
    q.N = (long long)g.B * g.Ho * g.Wo;
    // enough pixel splits to fill the chip (>= ~1024 workgroups), each a multiple of the chunk
    const int wbj = q.Jp % 128 == 0 ? 128 : WG_BJ;     // (the launch below picks the 64 x 128 tile the same way)
    const long long tiles = (long long)(q.Mpw / WG_BM) * (q.Jp / wbj);
    const long long z = wgrad_splits(tiles, WG_BM, wbj, (q.N + WG_BP - 1) / WG_BP);
    q.pix_per_split = ((q.N + z - 1) / z + WG_BP - 1) / WG_BP * WG_BP;
    q.Z = (int)((q.N + q.pix_per_split - 1) / q.pix_per_split);
    q.fwd_two_kernels = q.Mp / q.bm > 1;
    q.fwd_bytes = carve_bytes((size_t)q.Kp * q.Mp, 4) + 256 +
                  (q.fwd_two_kernels ? carve_bytes((size_t)g.B * q.K * g.Ho * g.Wo, 4) : 0);
    if (g.Co <= 64) {       // the window kernel (dcnw_fwd_kernel) packs [K][64]
        const size_t wb = carve_bytes((size_t)q.Kp * 64, 4) + 256;
        if (wb > q.fwd_bytes) q.fwd_bytes = wb;
    }
    q.gemm_bytes = cnuda_conv2d_workspace_bytes(g.B, g.Co, g.Ho, g.Wo, q.T * g.C, 1, 1, 1, 1, 0, 0);
    q.bwd_bytes = carve_bytes((size_t)q.Z * q.Mpw * q.Jp, 4) +
                  carve_bytes(std::max((size_t)g.Co * g.B, (size_t)q.Z * q.Mpw), 4) +
                  carve_bytes((size_t)12 * g.C * (g.Co < 64 ? 64 : g.Co), 4) +
                  carve_bytes((size_t)g.B * q.T * g.C * g.Ho * g.Wo, 4) +
                  carve_bytes((size_t)g.B * q.T * g.Ho * g.Wo, sizeof(DcnGeo)) + carve_bytes(q.gemm_bytes, 1) + 256;
    // col2im tile: 256 output pixels, lanes along x
    q.TC = 64;
    while (q.TC > 16 && q.TC / 2 >= g.Wo) q.TC >>= 1;
    // (round 6) a row width that 64-column tiles cover with a half-empty last tile -- 160 = 2.5 x 64, 80 = 1.25 x 64: the maps of
    // a 640 x 640 input -- takes 32-column tiles when those pad less: idle lanes cost the walk whole steps.  Measured, B = 32:
    // 64 -> 64 at 160 x 160 1861 -> 1702 us, 128 -> 64 at 80 x 80 1177 -> 1071 us (profiles/r6_dcn_walk_tile.txt), although the
    // 32-column tile ranks colliding lanes through the claim map instead of three DPP shifts (two rows per wave step).
    if (q.TC == 64 && round_up(g.Wo, 32) < round_up(g.Wo, 64)) q.TC = 32;
    if (g_walk_tc == 16 || g_walk_tc == 32 || g_walk_tc == 64) q.TC = g_walk_tc;     // (measurements: cnuda_dcn_set_walk_tile)
    q.TR = 256 / q.TC;
    q.tc_shift = q.TC == 64 ? 6 : (q.TC == 32 ? 5 : 4);
    q.tiles_y = ceil_div(g.Ho, q.TR);
    q.tiles_x = ceil_div(g.Wo, q.TC);
    q.ncg = ceil_div(g.C, CI_CG);
    q.margin = g_scatter_margin > 0 ? g_scatter_margin : ((g_offset_regime & 1) ? 4 : CI_MARGIN);
    const int wr = (q.TR - 1) * g.sh + (g.kh - 1) * g.dh + 2 * q.margin + 1;
    const int wc = (q.TC - 1) * g.sw + (g.kw - 1) * g.dw + 2 * q.margin + 1;
    q.WSZmax = wr * wc;
    q.claim_sz = ((wr + 1) * (wc + 1) + 15) / 16 * 16;
    // three workgroups per CU need <= 53 KiB each; larger windows (strides, dilations, big kernels) run windowless
    if ((size_t)CI_CG * q.WSZmax * 4 + 4096 + 4 * (size_t)q.claim_sz > (q.margin > CI_MARGIN ? 80 : 53) * 1024) { q.WSZmax = 0; q.claim_sz = 16; }
    q.col2im_lds = ((size_t)CI_CG * q.WSZmax + 4 * 256) * sizeof(float) + 4 * (size_t)q.claim_sz;
    // one workgroup per (image, tile) must still fill the chip (three resident per CU): the 128 x 128 and 64 x 64 maps
    q.fused_consumers = q.WSZmax > 0 && g.W >= 2 && (long long)g.B * q.tiles_y * q.tiles_x >= g_fused_min_tiles;
    q.fused_split = 4;     // measured 1..4 on the 128 x 128 / 64 x 64 layers: 1116/1077/1093/1064 and 716/645/611/584 us
    if (q.fused_split > q.ncg) q.fused_split = q.ncg;
    return q;
}

}  // namespace
}  // namespace cnuda

using namespace cnuda;

extern "C" int cnuda_dcn_set_scatter_margin(int margin) {
    const int prev = g_scatter_margin;
    g_scatter_margin = margin < 1 ? 0 : (margin > 8 ? 8 : margin);
    return prev;
}
extern "C" int cnuda_dcn_set_walk_tile(int tile_cols) {
    const int prev = g_walk_tc;
    g_walk_tc = tile_cols;
    return prev;
}
extern "C" int cnuda_dcn_set_offset_regime(int regime) {
    const int prev = g_offset_regime;
    g_offset_regime = regime & 3;
    return prev;
}

namespace cnuda {
namespace {
// counts[0] / counts[1] += samples whose offset leaves +-2 px / +-3 px in either direction (offset: [B][2T][HW], channel
// 2t = dy, 2t + 1 = dx of tap t).  A statistic for the host's choice of kernels, launched once in a while.
__global__ __launch_bounds__(256) void dcn_offset_census_kernel(const float* __restrict__ offset, long long n_pairs, long long HW,
                                                                unsigned* __restrict__ counts) {
    unsigned c2 = 0, c3 = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n_pairs; i += (long long)gridDim.x * 256) {
        const long long plane = i / HW, px = i - plane * HW;            // plane = b * T + t
        const float dy = fabsf(offset[(2 * plane) * HW + px]), dx = fabsf(offset[(2 * plane + 1) * HW + px]);
        const float d = dy > dx ? dy : dx;
        c2 += d >= 2.0f;
        c3 += d >= 3.0f;
    }
    __shared__ unsigned s2, s3;
    if (threadIdx.x == 0) { s2 = 0; s3 = 0; }
    __syncthreads();
    if (c2) atomicAdd(&s2, c2);
    if (c3) atomicAdd(&s3, c3);
    __syncthreads();
    if (threadIdx.x == 0) {
        if (s2) atomicAdd(counts, s2);
        if (s3) atomicAdd(counts + 1, s3);
    }
}
}  // namespace
}  // namespace cnuda
extern "C" int cnuda_dcn_offset_census(const float* offset, int B, int taps, long long HW, unsigned* counts,
                                       cnuda_stream_t stream) {
    CNUDA_REQUIRE(offset && counts && B > 0 && taps > 0 && HW > 0, "cnuda_dcn_offset_census: bad arguments");
    const long long n = (long long)B * taps * HW;
    CNUDA_LAUNCH(cnuda::dcn_offset_census_kernel, dim3(stream_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, offset, n, HW,
                 counts);
    return check_launch("cnuda_dcn_offset_census");
}
extern "C" int cnuda_dcn_set_fused_min_tiles(int min_tiles) {
    const int prev = g_fused_min_tiles;
    g_fused_min_tiles = min_tiles < 1 ? kFusedMinTilesDefault : min_tiles;
    return prev;
}

extern "C" size_t cnuda_dcn_v2_workspace_bytes(int B, int C, int H, int W, int Cout, int kh, int kw, int sh, int sw,
                                               int ph, int pw, int dh, int dw, int dg) {
    DcnGeom g;
    if (fill_geom(g, B, C, H, W, Cout, kh, kw, sh, sw, ph, pw, dh, dw, dg, "cnuda_dcn_v2_workspace_bytes")) return 0;
    if (dg > 1 && W >= 2) {     // the composed path: group copies of input / weights and their gradients, one output, the inner call's
        const int Cg = C / dg, T = kh * kw;
        const size_t inner = cnuda_dcn_v2_workspace_bytes(B, Cg, H, W, Cout, kh, kw, sh, sw, ph, pw, dh, dw, 1);
        return 2 * carve_bytes((size_t)B * Cg * H * W, 4) + 2 * carve_bytes((size_t)Cout * Cg * T, 4) +
               carve_bytes((size_t)B * Cout * g.Ho * g.Wo, 4) + 2 * carve_bytes((size_t)Cout, 4) + carve_bytes(inner, 1) + 512;
    }
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
    return cnuda_dcn_v2_forward_act(input, weight, bias, offset, mask, output, columns, -1.0f, B, C, H, W, Cout, kh, kw,
                                    sh, sw, ph, pw, dh, dw, dg, workspace, workspace_bytes, stream);
}

namespace {
// the LDS-window kernel's layers: 3x3 / stride 1 / padding 1 / dilation 1, row width a multiple of 16, one M tile.  CNUDA_DCNW=0
// keeps the gathering loader (A/B measurements; tests/test_gpu_kernel_switches.py).
// (round 6: any row width that is a multiple of 16 -- 160 / 80 / 96 -- in whole tiles of 4 x 32 or 8 x 16 pixels; rounds 4-5:
// 16 / 32 / 64 / 128 only)
int dcnw_tile_cols(int W) { return W % 32 == 0 ? 32 : 16; }
bool dcnw_takes(const DcnGeom& g) {
    static const bool dcnw_on = !(getenv("CNUDA_DCNW") && getenv("CNUDA_DCNW")[0] == '0');
    return dcnw_on && (g_offset_regime & 2) == 0 && matrix_mode() == 0 && g.kh == 3 && g.kw == 3 && g.sh == 1 && g.sw == 1 && g.ph == 1 && g.pw == 1 &&
           g.dh == 1 && g.dw == 1 && g.dg == 1 && g.C % 16 == 0 && g.Co <= 64 &&
           g.W % 16 == 0 && g.H % (IG_BN / dcnw_tile_cols(g.W)) == 0 &&
           (size_t)g.B * g.C * g.H * g.W * sizeof(float) < IG_BUF_OOB;
}
}  // namespace

// Pixel blocks of the BatchNorm statistics a forward call of this geometry can leave (cnuda_dcn_v2_forward_stats): 0 none
// (deformable_group > 1, width 1, planes that are no multiple of four pixels), else pixels per block; *rows = rows per block.
extern "C" int cnuda_dcn_v2_stats_block(int B, int C, int H, int W, int Cout, int kh, int kw, int sh, int sw, int ph, int pw,
                                        int dh, int dw, int dg, int* rows) {
    DcnGeom g;
    if (fill_geom(g, B, C, H, W, Cout, kh, kw, sh, sw, ph, pw, dh, dw, dg, "cnuda_dcn_v2_stats_block")) return 0;
    if (dg != 1 || W < 2 || ((g.Ho * g.Wo) & 3) != 0) return 0;
    if (dcnw_takes(g)) { if (rows) *rows = 64; return 32; }
    const DcnPlan q = make_plan(g);
    if (rows) *rows = q.Mp;
    return q.bm == 32 ? 32 : 64;
}

// elements per image of the offset / mask tensors and of their gradients when they are NOT the reference's own contiguous
// tensors (0: default): rows of the offset convolution's 3T-channel output (cnuda_dcn_v2_*_om), or one deformable group's
// rows of the dg-group tensors (the composed deformable_group > 1 path)
struct DcnStrides { int off_bs = 0, mask_bs = 0, goff_bs = 0, gmask_bs = 0, gmask_logit = 0; };
static int dcn_forward_impl(const float* input, const float* weight, const float* bias, const float* offset,
                            const float* mask, float* output, float* columns, float* stats, float act_slope, int B, int C,
                            int H, int W, int Cout, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw, int dg,
                            void* workspace, size_t workspace_bytes, cnuda_stream_t stream,
                            const DcnStrides* strides = nullptr);

extern "C" int cnuda_dcn_v2_forward_act(const float* input, const float* weight, const float* bias,
                                        const float* offset, const float* mask, float* output, float* columns,
                                        float act_slope, int B, int C, int H, int W, int Cout, int kh, int kw, int sh,
                                        int sw, int ph, int pw, int dh, int dw, int dg, void* workspace,
                                        size_t workspace_bytes, cnuda_stream_t stream) {
    return dcn_forward_impl(input, weight, bias, offset, mask, output, columns, nullptr, act_slope, B, C, H, W, Cout, kh, kw,
                            sh, sw, ph, pw, dh, dw, dg, workspace, workspace_bytes, stream);
}

// forward (+ saved columns) that also leaves the BatchNorm statistics of the output: DeformConv = DCN + BatchNorm + ReLU
// (backends/dla.py:351-372).  stats as cnuda_conv2d_forward_stats; cnuda_dcn_v2_stats_block says block size and rows.
extern "C" int cnuda_dcn_v2_forward_stats(const float* input, const float* weight, const float* bias,
                                          const float* offset, const float* mask, float* output, float* columns,
                                          float* stats, int stats_block, int stats_rows, int B, int C, int H, int W,
                                          int Cout, int kh, int kw, int sh,
                                          int sw, int ph, int pw, int dh, int dw, int dg, void* workspace,
                                          size_t workspace_bytes, cnuda_stream_t stream) {
    if (stats) {
        // the layout THIS call will write (it follows the kernel choice, and that the offset regime in force now) against
        // the one the caller sized the buffer for
        int rows = 0;
        const int blk = cnuda_dcn_v2_stats_block(B, C, H, W, Cout, kh, kw, sh, sw, ph, pw, dh, dw, dg, &rows);
        CNUDA_REQUIRE(blk != 0, "cnuda_dcn_v2_forward_stats: no statistics for this call (cnuda_dcn_v2_stats_block says which)");
        CNUDA_REQUIRE(blk == stats_block && rows == stats_rows,
                      "cnuda_dcn_v2_forward_stats: the statistics buffer was sized for blocks of %d pixels x %d rows, this call "
                      "writes %d x %d (did the offset regime change between cnuda_dcn_v2_stats_block and the call?)",
                      stats_block, stats_rows, blk, rows);
    }
    return dcn_forward_impl(input, weight, bias, offset, mask, output, columns, stats, -1.0f, B, C, H, W, Cout, kh, kw, sh,
                            sw, ph, pw, dh, dw, dg, workspace, workspace_bytes, stream);
}

static int dcn_forward_impl(const float* input, const float* weight, const float* bias, const float* offset,
                            const float* mask, float* output, float* columns, float* stats, float act_slope, int B, int C,
                            int H, int W, int Cout, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw, int dg,
                            void* workspace, size_t workspace_bytes, cnuda_stream_t stream, const DcnStrides* strides) {
    CNUDA_REQUIRE(input && weight && bias && offset && mask && output, "cnuda_dcn_v2_forward: null pointer");
    // (deformable_group > 1: the groups' column buffers one behind the other, [dg][B][T * C / dg][Ho * Wo] -- the same bytes)
    CNUDA_REQUIRE(!columns || W >= 2, "cnuda_dcn_v2_forward_cols: columns output needs width >= 2");
    DcnGeom g;
    if (int rc = fill_geom(g, B, C, H, W, Cout, kh, kw, sh, sw, ph, pw, dh, dw, dg, "cnuda_dcn_v2_forward")) return rc;
    if (strides && strides->off_bs) g.off_bs = strides->off_bs;
    if (strides && strides->mask_bs) g.mask_bs = strides->mask_bs;
    hipStream_t st = (hipStream_t)stream;
    CNUDA_REQUIRE(act_slope < 0.0f || (dg == 1 && W >= 2), "cnuda_dcn_v2_forward_act: fused activation needs deformable_group == 1 and width >= 2");
    if (dg > 1 && W >= 2) {
        // deformable_group > 1 (libs/DCNv2/dcn_v2.py:54-94 accepts any; testcpu.py:169-180 uses 2), round 6: the output is the sum
        // over the groups of a deformable_group = 1 convolution of the group's C / dg input channels with its own offsets and
        // mask -- the fast kernels run per group on a contiguous copy of the group's channels and weights, offsets / mask are
        // read in place (batch stride of the dg-group tensors, DcnStrides); the group outputs are added in group order.
        // (rounds 1-5: one thread per output element with global atomics in the backward: 45 ms / 1.5 s for the layer that
        // takes 0.9 / 2.5 ms with one group, profiles/r6_dcn_dg2.txt)
        const int Cg = C / dg, T = kh * kw, HoWo = g.Ho * g.Wo;
        const long long HW = (long long)H * W;
        const size_t inner = cnuda_dcn_v2_workspace_bytes(B, Cg, H, W, Cout, kh, kw, sh, sw, ph, pw, dh, dw, 1);
        Carver cv(workspace, workspace_bytes);
        float* xg = cv.take<float>((size_t)B * Cg * HW);
        float* wg = cv.take<float>((size_t)Cout * Cg * T);
        float* tmp = cv.take<float>((size_t)B * Cout * HoWo);
        float* zero = cv.take<float>((size_t)Cout);
        void* iws = cv.take<char>(inner);
        CNUDA_REQUIRE(workspace && cv.cur <= cv.end, "cnuda_dcn_v2_forward: workspace too small");
        if (hipMemsetAsync(zero, 0, (size_t)Cout * sizeof(float), st) != hipSuccess) return check_launch("cnuda_dcn_v2_forward(dg>1)");
        const DcnStrides gs{dg * 2 * T * HoWo, dg * T * HoWo, 0, 0, 0};
        for (int grp = 0; grp < dg; ++grp) {
            if (int rc = cnuda_copy_channels(input, xg, B, Cg, HW, C, grp * Cg, Cg, 0, stream)) return rc;
            if (int rc = cnuda_copy_channels(weight, wg, Cout, Cg, T, C, grp * Cg, Cg, 0, stream)) return rc;
            if (int rc = dcn_forward_impl(xg, wg, grp == 0 ? bias : zero, offset + (size_t)grp * 2 * T * HoWo,
                                          mask + (size_t)grp * T * HoWo, grp == 0 ? output : tmp,
                                          columns ? columns + (size_t)grp * B * T * Cg * HoWo : nullptr, nullptr, -1.0f, B, Cg, H,
                                          W, Cout, kh, kw, sh, sw, ph, pw, dh, dw, 1, iws, inner, stream, &gs))
                return rc;
            if (grp > 0)
                if (int rc = cnuda_add(output, tmp, output, (long long)B * Cout * HoWo, stream)) return rc;
        }
        return check_launch("cnuda_dcn_v2_forward(dg>1)");
    }
    if (dg != 1 || W < 2) {   // the MFMA path samples horizontally adjacent pairs
        DcnNaiveParams p{g, input, weight, bias, offset, mask, nullptr, output, nullptr, nullptr, nullptr, nullptr};
        CNUDA_LAUNCH(dcn_naive_fwd_kernel, dim3(stream_grid((long long)B * Cout * g.Ho * g.Wo, 256)), dim3(256),
                           0, st, p);
        return check_launch("cnuda_dcn_v2_forward(dg>1)");
    }
    const DcnPlan q = make_plan(g);
    CNUDA_REQUIRE(q.N < (1ll << 31) - IG_BN, "cnuda_dcn_v2_forward: more than 2^31 pixels per call");
    CNUDA_REQUIRE(workspace && workspace_bytes >= q.fwd_bytes, "cnuda_dcn_v2_forward: workspace too small");
    Carver cv(workspace, workspace_bytes);
    if (dcnw_takes(g)) {
        const int bm = 64;
        const float* Aw = launch_pack(weight, cv.take<float>((size_t)q.Kp * bm), (size_t)q.Kp * bm * sizeof(float), Cout, C,
                                      q.T, PACK_HALO_FWD, q.Kp, bm, 0, st);
        DcnFwdParams p{g, input, offset, mask, bias, act_slope, output, columns, stats, 64};
        ProfScope prof(st);
        prof.name("dcnw_fwd_kernel<%d>%s", bm, columns ? " (+ column side output)" : "");
        const int tc = dcnw_tile_cols(W), tiles_x = W / tc, n_tiles = (int)(q.N / IG_BN);
#define CNUDA_DCNW_LAUNCH(BMV, TCV)                                                                                    \
    do {                                                                                                               \
        const size_t fl = dcnw_lds_floats<TCV>(BMV);                                                                   \
        const size_t lds = (fl < (size_t)4 * IG_EPI_WAVE ? (size_t)4 * IG_EPI_WAVE : fl) * sizeof(float);             \
        CNUDA_REQUIRE(raise_dynamic_lds(reinterpret_cast<const void*>(&dcnw_fwd_kernel<BMV, TCV>), lds),               \
                      "cnuda_dcn_v2_forward: dynamic LDS");                                                            \
        CNUDA_LAUNCH((dcnw_fwd_kernel<BMV, TCV>), dim3(n_tiles), dim3(IG_THREADS), lds, st, p, Aw, bm, q.Kp, n_tiles,  \
                     tiles_x);                                                                                         \
    } while (0)
        // (a 128-row variant -- four accumulator tiles per wave -- spills and measured slower than the gathering
        // loader's 128-row tile, 571 vs 452 us at 128 -> 128, 64 x 64, B = 32: layers with more than 64 outputs stay there)
        if (tc == 32) CNUDA_DCNW_LAUNCH(64, 32); else CNUDA_DCNW_LAUNCH(64, 16);
#undef CNUDA_DCNW_LAUNCH
        return check_launch("cnuda_dcn_v2_forward(window)");
    }
    const float* A = launch_pack(weight, cv.take<float>((size_t)q.Kp * q.Mp), (size_t)q.Kp * q.Mp * sizeof(float), Cout,
                                 C, q.T, PACK_FWD, q.Kp, q.Mp, 0, st);
    const int n_tiles = ceil_div(q.N, IG_BN), m_tiles = q.Mp / q.bm;
    const dim3 grid(n_tiles * m_tiles), block(IG_THREADS);
    static const bool buf_on = !(getenv("CNUDA_BUF") && getenv("CNUDA_BUF")[0] == '0');
    const bool buf = buf_on && C % IG_BK == 0 && (size_t)B * C * H * W * sizeof(float) < IG_BUF_OOB &&
                     (size_t)B * q.K * g.Ho * g.Wo * sizeof(float) < IG_BUF_OOB;
    if (q.fwd_two_kernels) {
        float* cols = columns ? columns : cv.take<float>((size_t)B * q.K * g.Ho * g.Wo);
        ProfScope prof(st);   // brackets both kernels
        prof.name("dcn_sample_kernel + igemm_fwd_kernel<%d, DcnColsLoader>", q.bm);
        {
            DcnSampleParams sp{g, input, offset, mask, cols};
            const int tiles = ceil_div(g.Ho * g.Wo, 64), tw = q.T < 16 ? q.T : 16;
            CNUDA_LAUNCH(dcn_sample_kernel, dim3(B * tiles), dim3(64, tw), 0, st, sp, tiles);
        }
        DcnColsParams p{g, cols, bias, act_slope, output, stats, q.Mp};
#ifndef DCN_COLS_WS
#define DCN_COLS_WS 1
#endif
        // (a plain GEMM over coalesced rows: the 8-wave producer / consumer kernel of the dense convolutions takes it)
        if (DCN_COLS_WS && buf && q.bm == 128 && wave_specialised() && matrix_mode() == 0)
            CNUDA_LAUNCH((igemm_fwd_ws_kernel<128, DcnColsBufLoader>), grid, dim3(2 * IG_THREADS), 0, st, p, A, q.Mp, q.Kp, Cout,
                               q.N, n_tiles, m_tiles);
        else if (DCN_COLS_WS && buf && q.bm == 64 && wave_specialised() && matrix_mode() == 0)
            CNUDA_LAUNCH((igemm_fwd_ws_kernel<64, DcnColsBufLoader>), grid, dim3(2 * IG_THREADS), 0, st, p, A, q.Mp, q.Kp, Cout,
                               q.N, n_tiles, m_tiles);
        else if (buf && q.bm == 128)
            CNUDA_LAUNCH((igemm_fwd_kernel<128, DcnColsBufLoader>), grid, block, 0, st, p, A, q.Mp, q.Kp, Cout, q.N,
                               n_tiles, m_tiles);
        else if (buf && q.bm == 64)
            CNUDA_LAUNCH((igemm_fwd_kernel<64, DcnColsBufLoader>), grid, block, 0, st, p, A, q.Mp, q.Kp, Cout, q.N,
                               n_tiles, m_tiles);
        else if (buf)
            CNUDA_LAUNCH((igemm_fwd_kernel<32, DcnColsBufLoader>), grid, block, 0, st, p, A, q.Mp, q.Kp, Cout, q.N,
                               n_tiles, m_tiles);
        else if (q.bm == 128)
            CNUDA_LAUNCH((igemm_fwd_kernel<128, DcnColsLoader>), grid, block, 0, st, p, A, q.Mp, q.Kp, Cout, q.N,
                               n_tiles, m_tiles);
        else if (q.bm == 64)
            CNUDA_LAUNCH((igemm_fwd_kernel<64, DcnColsLoader>), grid, block, 0, st, p, A, q.Mp, q.Kp, Cout, q.N,
                               n_tiles, m_tiles);
        else
            CNUDA_LAUNCH((igemm_fwd_kernel<32, DcnColsLoader>), grid, block, 0, st, p, A, q.Mp, q.Kp, Cout, q.N,
                               n_tiles, m_tiles);
        return check_launch("cnuda_dcn_v2_forward(columns + GEMM)");
    }
    DcnFwdParams p{g, input, offset, mask, bias, act_slope, output, columns, stats, q.Mp};
    ProfScope prof(st);
    prof.name("igemm_fwd_kernel<%d, DcnFwdLoader>%s", q.bm, columns ? " (+ column side output)" : "");
    if (buf && q.bm == 128)
        CNUDA_LAUNCH((igemm_fwd_kernel<128, DcnFwdBufLoader>), grid, block, 0, st, p, A, q.Mp, q.Kp, Cout, q.N,
                           n_tiles, m_tiles);
    else if (buf && q.bm == 64)
        CNUDA_LAUNCH((igemm_fwd_kernel<64, DcnFwdBufLoader>), grid, block, 0, st, p, A, q.Mp, q.Kp, Cout, q.N,
                           n_tiles, m_tiles);
    else if (buf)
        CNUDA_LAUNCH((igemm_fwd_kernel<32, DcnFwdBufLoader>), grid, block, 0, st, p, A, q.Mp, q.Kp, Cout, q.N,
                           n_tiles, m_tiles);
    else if (q.bm == 128)
        CNUDA_LAUNCH((igemm_fwd_kernel<128, DcnFwdLoader>), grid, block, 0, st, p, A, q.Mp, q.Kp, Cout, q.N,
                           n_tiles, m_tiles);
    else if (q.bm == 64)
        CNUDA_LAUNCH((igemm_fwd_kernel<64, DcnFwdLoader>), grid, block, 0, st, p, A, q.Mp, q.Kp, Cout, q.N,
                           n_tiles, m_tiles);
    else
        CNUDA_LAUNCH((igemm_fwd_kernel<32, DcnFwdLoader>), grid, block, 0, st, p, A, q.Mp, q.Kp, Cout, q.N,
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
    return cnuda_dcn_v2_backward_acc(input, weight, bias, offset, mask, grad_output, columns, grad_input, 0, grad_offset,
                                     grad_mask, grad_weight, grad_bias, B, C, H, W, Cout, kh, kw, sh, sw, ph, pw, dh, dw, dg,
                                     workspace, workspace_bytes, stream);
}

static int dcn_backward_impl(const float* input, const float* weight, const float* bias,
                             const float* offset, const float* mask, const float* grad_output,
                             const float* columns, float* grad_input, int accumulate_input,
                             float* grad_offset, float* grad_mask, float* grad_weight, float* grad_bias, int B,
                             int C, int H, int W, int Cout, int kh, int kw, int sh, int sw, int ph, int pw,
                             int dh, int dw, int dg, void* workspace, size_t workspace_bytes,
                             cnuda_stream_t stream, const DcnStrides* strides);
extern "C" int cnuda_dcn_v2_backward_acc(const float* input, const float* weight, const float* bias,
                                         const float* offset, const float* mask, const float* grad_output,
                                         const float* columns, float* grad_input, int accumulate_input,
                                         float* grad_offset, float* grad_mask, float* grad_weight, float* grad_bias, int B,
                                         int C, int H, int W, int Cout, int kh, int kw, int sh, int sw, int ph, int pw,
                                         int dh, int dw, int dg, void* workspace, size_t workspace_bytes,
                                         cnuda_stream_t stream) {
    return dcn_backward_impl(input, weight, bias, offset, mask, grad_output, columns, grad_input, accumulate_input, grad_offset,
                             grad_mask, grad_weight, grad_bias, B, C, H, W, Cout, kh, kw, sh, sw, ph, pw, dh, dw, dg, workspace,
                             workspace_bytes, stream, nullptr);
}

// Offsets and mask read straight out of `om`, the 3T-channel output of DCN's own offset convolution
// (libs/DCNv2/dcn_v2.py:118-122: o1, o2, mask = chunk(out, 3); offset = cat(o1, o2); mask = sigmoid(mask)): rows 0 .. 2T-1
// ARE the offsets, rows 2T .. 3T-1 the mask -- ALREADY sigmoid (cnuda_conv2d_forward_rowsig applies it in the
// convolution's epilogue) -- and the backward writes one tensor `gom` of the same shape: the offsets' gradient and the
// gradient of the mask's LOGIT (the walk multiplies by m (1 - m) where it stores).  No split / concatenate / sigmoid passes:
// 32 launches of a benched step and two tensors per layer less (round 6).  deformable_group == 1.
extern "C" int cnuda_dcn_v2_forward_om(const float* input, const float* weight, const float* bias, const float* om,
                                       float* output, float* columns, float* stats, int stats_block, int stats_rows,
                                       float act_slope, int B,
                                       int C, int H, int W, int Cout, int kh, int kw, int sh, int sw, int ph, int pw, int dh,
                                       int dw, int dg, void* workspace, size_t workspace_bytes, cnuda_stream_t stream) {
    CNUDA_REQUIRE(om && dg == 1 && W >= 2, "cnuda_dcn_v2_forward_om: needs deformable_group == 1 and width >= 2");
    DcnGeom g;
    if (int rc = fill_geom(g, B, C, H, W, Cout, kh, kw, sh, sw, ph, pw, dh, dw, dg, "cnuda_dcn_v2_forward_om")) return rc;
    if (stats) {
        int rows = 0;
        const int blk = cnuda_dcn_v2_stats_block(B, C, H, W, Cout, kh, kw, sh, sw, ph, pw, dh, dw, dg, &rows);
        CNUDA_REQUIRE(blk != 0 && blk == stats_block && rows == stats_rows,
                      "cnuda_dcn_v2_forward_om: the statistics buffer was sized for blocks of %d pixels x %d rows, this call "
                      "writes %d x %d", stats_block, stats_rows, blk, rows);
    }
    CNUDA_REQUIRE(!stats || act_slope < 0.0f, "cnuda_dcn_v2_forward_om: statistics are those of the output before an activation");
    const int s3 = 3 * kh * kw * g.Ho * g.Wo;      // offsets and mask are rows of one 3T-channel tensor
    const DcnStrides ss{s3, s3, 0, 0, 0};
    return dcn_forward_impl(input, weight, bias, om, om + (size_t)2 * kh * kw * g.Ho * g.Wo, output, columns, stats, act_slope, B, C,
                            H, W, Cout, kh, kw, sh, sw, ph, pw, dh, dw, dg, workspace, workspace_bytes, stream, &ss);
}
extern "C" int cnuda_dcn_v2_backward_om(const float* input, const float* weight, const float* bias, const float* om,
                                        const float* grad_output, const float* columns, float* grad_input,
                                        int accumulate_input, float* grad_om, float* grad_weight, float* grad_bias, int B,
                                        int C, int H, int W, int Cout, int kh, int kw, int sh, int sw, int ph, int pw, int dh,
                                        int dw, int dg, void* workspace, size_t workspace_bytes, cnuda_stream_t stream) {
    CNUDA_REQUIRE(om && grad_om && dg == 1 && W >= 2, "cnuda_dcn_v2_backward_om: needs deformable_group == 1 and width >= 2");
    DcnGeom g;
    if (int rc = fill_geom(g, B, C, H, W, Cout, kh, kw, sh, sw, ph, pw, dh, dw, dg, "cnuda_dcn_v2_backward_om")) return rc;
    const size_t mo = (size_t)2 * kh * kw * g.Ho * g.Wo;
    const int s3 = 3 * kh * kw * g.Ho * g.Wo;      // offsets / mask and their gradients as rows of 3T-channel tensors; the mask's as its logit's
    const DcnStrides ss{s3, s3, s3, s3, 1};
    return dcn_backward_impl(input, weight, bias, om, om + mo, grad_output, columns, grad_input, accumulate_input, grad_om,
                             grad_om + mo, grad_weight, grad_bias, B, C, H, W, Cout, kh, kw, sh, sw, ph, pw, dh, dw, dg, workspace,
                             workspace_bytes, stream, &ss);
}

static int dcn_backward_impl(const float* input, const float* weight, const float* bias,
                             const float* offset, const float* mask, const float* grad_output,
                             const float* columns, float* grad_input, int accumulate_input,
                             float* grad_offset, float* grad_mask, float* grad_weight, float* grad_bias, int B,
                             int C, int H, int W, int Cout, int kh, int kw, int sh, int sw, int ph, int pw,
                             int dh, int dw, int dg, void* workspace, size_t workspace_bytes,
                             cnuda_stream_t stream, const DcnStrides* strides) {
    CNUDA_REQUIRE(!columns || W >= 2, "cnuda_dcn_v2_backward_cols: columns input needs width >= 2");
    CNUDA_REQUIRE(input && weight && offset && mask && grad_output && grad_input && grad_offset && grad_mask &&
                      grad_weight && grad_bias,
                  "cnuda_dcn_v2_backward: null pointer");
    (void)bias;
    DcnGeom g;
    if (int rc = fill_geom(g, B, C, H, W, Cout, kh, kw, sh, sw, ph, pw, dh, dw, dg, "cnuda_dcn_v2_backward")) return rc;
    if (strides) {
        if (strides->off_bs) g.off_bs = strides->off_bs;
        if (strides->mask_bs) g.mask_bs = strides->mask_bs;
        if (strides->goff_bs) g.goff_bs = strides->goff_bs;
        if (strides->gmask_bs) g.gmask_bs = strides->gmask_bs;
        g.gmask_logit = strides->gmask_logit;
    }
    hipStream_t st = (hipStream_t)stream;
    const int T = kh * kw, HoWo = g.Ho * g.Wo;
    if (dg > 1 && W >= 2) {
        // deformable_group > 1, composed from the deformable_group = 1 kernels (see dcn_forward_impl): per group the data
        // gradient of its C / dg input channels, its rows of grad_offset / grad_mask (written in place through the batch strides
        // of the dg-group tensors) and its slice of grad_weight; grad_bias by the first group's call
        const int Cg = C / dg;
        const long long HW = (long long)H * W;
        const size_t inner = cnuda_dcn_v2_workspace_bytes(B, Cg, H, W, Cout, kh, kw, sh, sw, ph, pw, dh, dw, 1);
        Carver cv(workspace, workspace_bytes);
        float* xg = cv.take<float>((size_t)B * Cg * HW);
        float* wg = cv.take<float>((size_t)Cout * Cg * T);
        float* ging = cv.take<float>((size_t)B * Cg * HW);
        float* gwg = cv.take<float>((size_t)Cout * Cg * T);
        float* gbt = cv.take<float>((size_t)Cout);
        void* iws = cv.take<char>(inner);
        CNUDA_REQUIRE(workspace && cv.cur <= cv.end, "cnuda_dcn_v2_backward: workspace too small");
        const DcnStrides gs{dg * 2 * T * HoWo, dg * T * HoWo, dg * 2 * T * HoWo, dg * T * HoWo, 0};
        for (int grp = 0; grp < dg; ++grp) {
            if (int rc = cnuda_copy_channels(input, xg, B, Cg, HW, C, grp * Cg, Cg, 0, stream)) return rc;
            if (int rc = cnuda_copy_channels(weight, wg, Cout, Cg, T, C, grp * Cg, Cg, 0, stream)) return rc;
            // (a caller's running sum in grad_input: the group's slice goes in, is added to, and comes back)
            if (accumulate_input)
                if (int rc = cnuda_copy_channels(grad_input, ging, B, Cg, HW, C, grp * Cg, Cg, 0, stream)) return rc;
            if (int rc = dcn_backward_impl(xg, wg, bias, offset + (size_t)grp * 2 * T * HoWo, mask + (size_t)grp * T * HoWo,
                                           grad_output, columns ? columns + (size_t)grp * B * T * Cg * HoWo : nullptr, ging,
                                           accumulate_input, grad_offset + (size_t)grp * 2 * T * HoWo,
                                           grad_mask + (size_t)grp * T * HoWo, gwg, grp == 0 ? grad_bias : gbt, B, Cg, H, W, Cout,
                                           kh, kw, sh, sw, ph, pw, dh, dw, 1, iws, inner, stream, &gs))
                return rc;
            if (int rc = cnuda_copy_channels(ging, grad_input, B, Cg, HW, Cg, 0, C, grp * Cg, stream)) return rc;
            if (int rc = cnuda_copy_channels(gwg, grad_weight, Cout, Cg, T, Cg, 0, C, grp * Cg, stream)) return rc;
        }
        return check_launch("cnuda_dcn_v2_backward(dg>1)");
    }
    // (every data-gradient walk below ADDS into grad_input -- window flushes and strays are atomics -- so a caller that
    // already holds another consumer's share of the input's gradient there passes accumulate_input and saves the sum)
    // (deformable_group == 1: dcn_prep_kernel clears it beside its other work -- one launch less per layer; the walks run
    // behind it on the same stream)
    const long long gin_elems = (long long)B * C * H * W;
    const bool zero_in_prep = !accumulate_input && dg == 1 && (gin_elems & 3) == 0 && (reinterpret_cast<uintptr_t>(grad_input) & 15) == 0;
    if (!accumulate_input && !zero_in_prep &&
        hipMemsetAsync(grad_input, 0, (size_t)gin_elems * sizeof(float), st) != hipSuccess)
        return check_launch("cnuda_dcn_v2_backward(memset)");
    if (dg != 1) {
        launch_channel_sum(grad_output, grad_bias, B, Cout, HoWo, st);
        (void)hipMemsetAsync(grad_offset, 0, (size_t)B * dg * 2 * T * HoWo * sizeof(float), st);
        (void)hipMemsetAsync(grad_mask, 0, (size_t)B * dg * T * HoWo * sizeof(float), st);
        (void)hipMemsetAsync(grad_weight, 0, (size_t)Cout * C * T * sizeof(float), st);
        DcnNaiveParams p{g, input, weight, nullptr, offset, mask, grad_output, nullptr,
                         grad_input, grad_offset, grad_mask, grad_weight};
        CNUDA_LAUNCH(dcn_naive_bwd_kernel, dim3(stream_grid((long long)B * C * T * HoWo, 256)), dim3(256), 0, st,
                           p);
        return check_launch("cnuda_dcn_v2_backward(dg>1)");
    }
    const DcnPlan q = make_plan(g);
    CNUDA_REQUIRE(workspace && workspace_bytes >= q.bwd_bytes, "cnuda_dcn_v2_backward: workspace too small");
    Carver cv(workspace, workspace_bytes);
    float* slabs = cv.take<float>((size_t)q.Z * q.Mpw * q.Jp);
    float* bsl = cv.take<float>(std::max((size_t)Cout * B, (size_t)q.Z * q.Mpw));    // bias row sums per split: [Z][Mpw]
    float* wt = cv.take<float>((size_t)12 * C * (Cout < 64 ? 64 : Cout));   // (or the window kernels' packs: [64][10 C], [12 C][64])
    float* dcol = cv.take<float>((size_t)B * q.T * C * HoWo);
    DcnGeo* geo = cv.take<DcnGeo>((size_t)B * q.T * HoWo);
    void* gemm_ws = cv.take<char>(q.gemm_bytes);
    // (running the weight gradient on a second stream beside the data-gradient chain was tried in round 2: the
    // kernels do overlap but contend for the same LDS / issue slots -- nothing gained, removed)
    hipStream_t wst = st;
    // every timed scope below is recorded under the call's tag; sub 0: the column-gradient 1x1 GEMM (its own scope
    // inside cnuda_conv2d_forward), 1: coord_grad, 2: col2im, 3: both as one launch, 4: the weight-gradient GEMM
    ProfGroup prof;
    // (2) weight gradient (and, from the same staging registers, the bias gradient: bsl -> slab reduce)
    {
      {
        ProfScope wscope(st, 4);
        if (columns) {
            DcnColWParams p{g, columns, grad_output};
            static const bool buf_on = !(getenv("CNUDA_BUF") && getenv("CNUDA_BUF")[0] == '0');
            const bool buf = buf_on && (size_t)B * q.T * C * HoWo * sizeof(float) < IG_BUF_OOB &&
                             (size_t)B * Cout * HoWo * sizeof(float) < IG_BUF_OOB && HoWo < (1 << 23);
#ifndef DCN_COLW_WS
#define DCN_COLW_WS 1
#endif
            // (plain row reads of the saved columns: the 8-wave producer / consumer kernel of the dense convolutions takes them)
            const bool ws = DCN_COLW_WS && wave_specialised() && matrix_mode() == 0;
            wscope.name(ws && buf ? "igemm_wgrad_ws_kernel<%s, 64, %d>" : "igemm_wgrad_kernel<%s, 64, %d>",
                        buf ? "DcnColWBufLoader" : "DcnColWLoader", q.Jp % 128 == 0 ? 128 : 64);
            if (buf && q.Jp % 128 == 0) {
                if (ws)
                    CNUDA_LAUNCH((igemm_wgrad_ws_kernel<DcnColWBufLoader, 64, 128>), dim3(q.Jp / 128, q.Mpw / WG_BM, q.Z),
                                       dim3(2 * IG_THREADS), 0, wst, p, slabs, q.Mpw, q.Jp, q.N, q.pix_per_split, bsl);
                else
                    CNUDA_LAUNCH((igemm_wgrad_kernel<DcnColWBufLoader, 64, 128>), dim3(q.Jp / 128, q.Mpw / WG_BM, q.Z),
                                       dim3(IG_THREADS), 0, wst, p, slabs, q.Mpw, q.Jp, q.N, q.pix_per_split, bsl);
            } else if (buf) {
                if (ws)
                    CNUDA_LAUNCH((igemm_wgrad_ws_kernel<DcnColWBufLoader, 64, 64>), dim3(q.Jp / WG_BJ, q.Mpw / WG_BM, q.Z),
                                       dim3(2 * IG_THREADS), 0, wst, p, slabs, q.Mpw, q.Jp, q.N, q.pix_per_split, bsl);
                else
                    CNUDA_LAUNCH((igemm_wgrad_kernel<DcnColWBufLoader, 64, 64>), dim3(q.Jp / WG_BJ, q.Mpw / WG_BM, q.Z),
                                       dim3(IG_THREADS), 0, wst, p, slabs, q.Mpw, q.Jp, q.N, q.pix_per_split, bsl);
            }
            else if (q.Jp % 128 == 0)
                CNUDA_LAUNCH((igemm_wgrad_kernel<DcnColWLoader, 64, 128>), dim3(q.Jp / 128, q.Mpw / WG_BM, q.Z),
                                   dim3(IG_THREADS), 0, wst, p, slabs, q.Mpw, q.Jp, q.N, q.pix_per_split, bsl);
            else
                CNUDA_LAUNCH((igemm_wgrad_kernel<DcnColWLoader, 64, 64>), dim3(q.Jp / WG_BJ, q.Mpw / WG_BM, q.Z),
                                   dim3(IG_THREADS), 0, wst, p, slabs, q.Mpw, q.Jp, q.N, q.pix_per_split, bsl);
        } else {
            DcnWParams p{g, input, offset, mask, grad_output};
            wscope.name("igemm_wgrad_kernel<DcnWLoader, 64, 64>");
            CNUDA_LAUNCH((igemm_wgrad_kernel<DcnWLoader, 64, 64>), dim3(q.Jp / WG_BJ, q.Mpw / WG_BM, q.Z),
                               dim3(IG_THREADS), 0, wst, p, slabs, q.Mpw, q.Jp, q.N, q.pix_per_split, bsl);
        }
      }
        if (int rc = check_launch("cnuda_dcn_v2_backward(weight)")) return rc;
        launch_slab_reduce(slabs, grad_weight, q.Z, q.Mpw, q.Jp, Cout, C, q.T, wst, bsl, grad_bias);
    }
    {
        // (1) dcol = W^T x grad_output as a 1x1 implicit GEMM, then the two streaming consumers
        {
            // transposed weights of the 1x1 GEMM + (fused form) the geometry records, one launch
            const int wt_blocks = stream_grid((long long)q.T * C * Cout, 256);
            const int geo_blocks = q.fused_consumers ? stream_grid((long long)B * q.T * HoWo, 256) : 0;
            const int zero_blocks = zero_in_prep ? stream_grid(gin_elems / 4, 256) : 0;
            DcnPrepParams pp{g, weight, offset, mask, wt, q.fused_consumers ? geo : nullptr, wt_blocks,
                             zero_in_prep ? grad_input : nullptr, gin_elems / 4, wt_blocks + geo_blocks};
            CNUDA_LAUNCH(dcn_prep_kernel, dim3(wt_blocks + geo_blocks + zero_blocks), dim3(256), 0, st, pp);
        }
        // (round 6) rows interleaved in quads where the GEMM has that epilogue: one 16-byte load per (pixel, tap, 4 channels)
        // in both consumers
        static const bool quads_on = !(getenv("CNUDA_DCOL_QUADS") && getenv("CNUDA_DCOL_QUADS")[0] == '0');
        const bool quads = quads_on && C % 4 == 0 && HoWo < (1 << 26) &&
                           cnuda_conv2d_rowquads_supported(B, Cout, g.Ho, g.Wo, q.T * C, 1, 1, 1, 1, 0, 0);
        if (int rc = quads ? cnuda_conv2d_forward_rowquads(grad_output, wt, dcol, B, Cout, g.Ho, g.Wo, q.T * C, 1, 1, 1, 1, 0, 0,
                                                           gemm_ws, q.gemm_bytes, stream)
                           : cnuda_conv2d_forward(grad_output, wt, nullptr, dcol, B, Cout, g.Ho, g.Wo, q.T * C, 1, 1, 1, 1, 0,
                                                  0, -1.0f, gemm_ws, q.gemm_bytes, stream))
            return rc;
        if (q.fused_consumers) {
            DcnBwdDataParams p{g, input, dcol, geo, grad_input, grad_offset, grad_mask, q.TR, q.TC, q.tc_shift,
                               q.tiles_y, q.tiles_x, q.ncg, q.WSZmax, q.claim_sz, q.margin, q.fused_split};
            const int n_wg = B * q.tiles_y * q.tiles_x * q.fused_split;
            ProfScope scope(st, 3);
            scope.name("dcn_bwd_data_kernel");
            // (a wide-margin window: dynamic LDS beyond 64 KiB is opt-in)
            CNUDA_REQUIRE(raise_dynamic_lds(quads ? reinterpret_cast<const void*>(&dcn_bwd_data_kernel<true>)
                                                  : reinterpret_cast<const void*>(&dcn_bwd_data_kernel<false>), q.col2im_lds),
                          "cnuda_dcn_v2_backward: dynamic LDS");
            if (quads) CNUDA_LAUNCH((dcn_bwd_data_kernel<true>), dim3(n_wg), dim3(512), q.col2im_lds, st, p, n_wg);
            else CNUDA_LAUNCH((dcn_bwd_data_kernel<false>), dim3(n_wg), dim3(512), q.col2im_lds, st, p, n_wg);
        } else {
            {
                DcnCoordParams p{g, input, offset, mask, dcol, grad_offset, grad_mask, geo};
                const int tiles = ceil_div(HoWo, 64), tw = q.T < 16 ? q.T : 16;
                ProfScope scope(st, 1);
                scope.name("dcn_coord_grad_kernel");
                if (quads) CNUDA_LAUNCH((dcn_coord_grad_kernel<true>), dim3(B * tiles), dim3(64, tw), 0, st, p, tiles);
                else CNUDA_LAUNCH((dcn_coord_grad_kernel<false>), dim3(B * tiles), dim3(64, tw), 0, st, p, tiles);
            }
            {
                DcnCol2imParams p{g, dcol, geo, grad_input, q.TR, q.TC, q.tc_shift, q.tiles_y, q.tiles_x,
                                  q.ncg, q.WSZmax, q.claim_sz, q.margin};
                const int n_wg = B * q.tiles_y * q.tiles_x * q.ncg;
                ProfScope scope(st, 2);
                scope.name("dcn_col2im_kernel");
                CNUDA_REQUIRE(raise_dynamic_lds(quads ? reinterpret_cast<const void*>(&dcn_col2im_kernel<true>)
                                                      : reinterpret_cast<const void*>(&dcn_col2im_kernel<false>), q.col2im_lds),
                              "cnuda_dcn_v2_backward: dynamic LDS");
                if (quads) CNUDA_LAUNCH((dcn_col2im_kernel<true>), dim3(n_wg), dim3(256), q.col2im_lds, st, p, n_wg);
                else CNUDA_LAUNCH((dcn_col2im_kernel<false>), dim3(n_wg), dim3(256), q.col2im_lds, st, p, n_wg);
            }
        }
        if (int rc = check_launch("cnuda_dcn_v2_backward(data)")) return rc;
    }
    return check_launch("cnuda_dcn_v2_backward");
}
