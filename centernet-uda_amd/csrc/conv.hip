// Dense 2-D convolution (NCHW fp32, groups = 1, dilation = 1) as implicit GEMM on
// the fp32 MFMA: forward, input gradient (transposed-conv gather) and weight
// gradient (split-K with fixed-order slab reduction).  These replace the
// torch.nn.Conv2d -> cuDNN calls on the reference's hot path
// (backends/dla.py:37-44,153-155,234-235,281-283,478-483; the DCN
// offset/mask convolution libs/DCNv2/dcn_v2.py:104-110; the ADVENT
// discriminator uda/adversarial_entropy_minimization.py:51-68).
//
// K axis order is (tap, channel): within one 16-deep K chunk the tap is constant
// whenever C % 16 == 0 (every layer but the 3-channel stem), so the bounds test
// and address of a gathered element are computed once per chunk.
#include <stdlib.h>

#include <type_traits>
#include <algorithm>
#include "igemm.cuh"
#include "hconv.cuh"
#include "hwgrad.cuh"
#include "igemm_host.h"

namespace cnuda {
namespace {

struct ConvGeom {
    int B, C, H, W, Co, kh, kw, sh, sw, ph, pw, Ho, Wo;
};

struct ConvFwdParams {
    ConvGeom g;
    const float *x, *bias;
    float* y;
    float act_slope;   // < 0: none, 0: ReLU, 0.2: LeakyReLU(0.2)
    const float* residual;   // nullable, shaped like y: y = act(conv + bias + residual) (BatchNorm-folded inference)
};
// ... with the BatchNorm statistics of y as a second output (igemm.cuh, "BatchNorm statistics").  Its own parameter type, and
// with it its own kernel instances (ConvFwdBufStatsLoader below): compiled into every forward kernel, the statistics tail cost
// the launches that do not want it -- the detection heads' 64 -> 256 convolutions, 8 of the dominant kernel's 11 ms -- 5 %
// (12 more registers, 114.7 instead of 120.9 TFLOP/s; round 5).
struct ConvFwdStatsParams : ConvFwdParams {
    float* stats;            // per (pixel block, output channel) sum / sum of squares of y
    int stats_mp;            // rows per pixel block of `stats` (>= the padded row count of the launch)
};

// ... with a sigmoid on the output rows m >= sig_from (after the bias): DCN's offset convolution, whose rows 2T .. 3T-1 are the
// modulation mask's logits (libs/DCNv2/dcn_v2.py:118-122) -- the DCN kernels then read offsets and mask straight out of this
// one tensor (cnuda_dcn_v2_forward_om).  Its own parameter type and kernel instances, like the statistics variant.
struct ConvFwdSigParams : ConvFwdParams {
    int sig_from;
};
// (the expression of cnuda_split_offset_mask, which this replaces: bit-identical masks)
__device__ __forceinline__ float conv_row_sigmoid(float v) { return 1.0f / (1.0f + expf(-v)); }

// B[k = tap*C + c][n] = x[b][c][oy*sh - ph + r][ox*sw - pw + s]   (0 outside)
template <bool FAST>
struct ConvFwdLoader {
    using Params = ConvFwdParams;
    static const char* name() { return FAST ? "ConvFwdLoader<true>" : "ConvFwdLoader<false>"; }
    static constexpr bool kHasSideOutput = false;
    const ConvGeom& g;
    const float* x_b;
    int iy0, ix0;
    bool valid;
    // chunk cursor of the FAST path: chunks arrive in increasing k0, 16 at a time, so (tap, first channel,
    // kernel row / column) advance by additions -- two integer divisions per chunk and thread otherwise
    int ck0 = 0, ctap = 0, cc0 = 0, cr = 0, cs = 0;
    __device__ __forceinline__ void seek(int k0) {
        while (ck0 < k0) {
            ck0 += IG_BK;
            cc0 += IG_BK;
            if (cc0 >= g.C) { cc0 -= g.C; ++ctap; if (++cs == g.kw) { cs = 0; ++cr; } }
        }
    }
    __device__ ConvFwdLoader(const Params& p, long long n, bool n_valid) : g(p.g), valid(n_valid) {
        const int HoWo = g.Ho * g.Wo;
        const int nn = n_valid ? (int)n : 0;   // N < 2^31 is checked on the host: 32-bit index math
        const int b = nn / HoWo, pp = nn - b * HoWo;
        const int oy = pp / g.Wo, ox = pp - oy * g.Wo;
        iy0 = oy * g.sh - g.ph;
        ix0 = ox * g.sw - g.pw;
        x_b = p.x + (size_t)b * g.C * g.H * g.W;
    }
    __device__ __forceinline__ void load(int k0, int ksub, float (&v)[8]) {
        const int HW = g.H * g.W, K = g.kh * g.kw * g.C;
        if (FAST) {
            seek(k0);
            const int c0 = cc0 + ksub;
            const int iy = iy0 + cr, ix = ix0 + cs;
            const bool ok = valid && k0 < K && iy >= 0 && iy < g.H && ix >= 0 && ix < g.W;
            const float* ptr = x_b + (size_t)c0 * HW + (ok ? iy * g.W + ix : 0);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = ok ? ptr[(size_t)(2 * j) * HW] : 0.0f;
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = k0 + ksub + 2 * j;
                float val = 0.0f;
                if (valid && k < K) {
                    const int tap = k / g.C, c = k - tap * g.C;
                    const int r = tap / g.kw, s = tap - r * g.kw;
                    const int iy = iy0 + r, ix = ix0 + s;
                    if (iy >= 0 && iy < g.H && ix >= 0 && ix < g.W) val = x_b[(size_t)c * HW + iy * g.W + ix];
                }
                v[j] = val;
            }
        }
    }
    struct Out {
        float* base;
        const float* res;
        int HoWo;
        __device__ Out(const Params& p, long long n) {
            HoWo = p.g.Ho * p.g.Wo;
            const int ni = (int)n, b = ni / HoWo, pp = ni - b * HoWo;
            base = p.y + (size_t)b * p.g.Co * HoWo + pp;
            res = p.residual ? p.residual + (size_t)b * p.g.Co * HoWo + pp : nullptr;
        }
        __device__ __forceinline__ void store(const Params& p, int m, float v) {
            if (p.bias) v += p.bias[m];
            if (res) v += res[(size_t)m * HoWo];
            if (p.act_slope >= 0.0f && v < 0.0f) v *= p.act_slope;
            base[(size_t)m * HoWo] = v;
        }
        // four consecutive pixels of one image (16-byte epilogue, igemm.cuh)
        static constexpr bool kVec4 = true;
        __device__ static bool vec4_ok(const Params& p) { return ((p.g.Ho * p.g.Wo) & 3) == 0; }
        __device__ __forceinline__ void store4(const Params& p, int m, f32x4 v) {
            if (p.bias) v += p.bias[m];
            if (res) v += *reinterpret_cast<const f32x4*>(res + (size_t)m * HoWo);
            if (p.act_slope >= 0.0f) {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (v[e] < 0.0f) v[e] *= p.act_slope;
            }
            *reinterpret_cast<f32x4*>(base + (size_t)m * HoWo) = v;
        }
    };
};

// The same gather for C % 16 == 0 with buffer addressing (igemm.cuh): per thread one byte offset of the window's
// corner and a bit per tap that says whether the tap lies inside the image, both computed once per tile; per chunk
// one add and one select, and the eight channel loads differ by a scalar offset only.  Needs the tensor below
// 2 GiB and at most 32 taps (the host checks; otherwise ConvFwdLoader<true>).
struct ConvFwdBufLoader {
    using Params = ConvFwdParams;
    static const char* name() { return "ConvFwdBufLoader"; }
    static constexpr bool kHasSideOutput = false;
    const ConvGeom& g;
    buf_rsrc rs;
    unsigned pix_off;      // byte offset of x[b][0][iy0][ix0] (may lie outside the image: only used with an in-range tap)
    unsigned tap_ok;       // bit (r * kw + s): tap inside the image for this pixel
    int ck0 = 0, ctap = 0, cc0 = 0, cr = 0, cs = 0;      // chunk cursor (wave-uniform), see ConvFwdLoader::seek
    __device__ __forceinline__ void seek(int k0) {
        while (ck0 < k0) {
            ck0 += IG_BK;
            cc0 += IG_BK;
            if (cc0 >= g.C) { cc0 -= g.C; ++ctap; if (++cs == g.kw) { cs = 0; ++cr; } }
        }
    }
    __device__ ConvFwdBufLoader(const Params& p, long long n, bool n_valid) : g(p.g) {
        const int HoWo = g.Ho * g.Wo;
        const int nn = n_valid ? (int)n : 0;
        const int b = nn / HoWo, pp = nn - b * HoWo;
        const int oy = pp / g.Wo, ox = pp - oy * g.Wo;
        const int iy0 = oy * g.sh - g.ph, ix0 = ox * g.sw - g.pw;
        rs = ig_make_rsrc(p.x, (unsigned)((size_t)g.B * g.C * g.H * g.W * sizeof(float)));
        pix_off = (unsigned)(((b * g.C * g.H + iy0) * g.W + ix0) * (int)sizeof(float));
        tap_ok = 0;
        if (n_valid) {
            for (int r = 0; r < g.kh; ++r)
                for (int s = 0; s < g.kw; ++s) {
                    const int iy = iy0 + r, ix = ix0 + s;
                    if (iy >= 0 && iy < g.H && ix >= 0 && ix < g.W) tap_ok |= 1u << (r * g.kw + s);
                }
        }
    }
    __device__ __forceinline__ void load(int k0, int ksub, float (&v)[8]) {
        const int HW = g.H * g.W;
        seek(k0);
        const int tap = ctap < 32 ? ctap : 31;                               // (k0 >= K: every lane reads the sentinel)
        const bool in_k = k0 < g.kh * g.kw * g.C;
        const unsigned voff = (in_k && ((tap_ok >> tap) & 1u)) ? pix_off + (unsigned)((cr * g.W + cs) * (int)sizeof(float))
                                                                : IG_BUF_OOB;
        const int c0 = cc0 + __builtin_amdgcn_readfirstlane(ksub);          // ksub = tid >> 7 is wave-uniform
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = ig_buf_load(rs, voff, (unsigned)((c0 + 2 * j) * HW) * (unsigned)sizeof(float));
    }
    using Out = ConvFwdLoader<true>::Out;
};

// ConvFwdBufLoader with the row-wise sigmoid epilogue (ConvFwdSigParams): the 27-channel offset convolutions that no halo-tile
// kernel takes (small maps, split-K)
struct ConvFwdBufSigLoader : ConvFwdBufLoader {
    using Params = ConvFwdSigParams;
    static const char* name() { return "ConvFwdBufSigLoader"; }
    __device__ ConvFwdBufSigLoader(const Params& p, long long n, bool n_valid) : ConvFwdBufLoader(p, n, n_valid) {}
    struct Out : ConvFwdLoader<true>::Out {
        __device__ Out(const Params& p, long long n) : ConvFwdLoader<true>::Out(p, n) {}
        __device__ __forceinline__ void store(const Params& p, int m, float v) {
            if (p.bias) v += p.bias[m];
            if (m >= p.sig_from) v = conv_row_sigmoid(v);
            base[(size_t)m * HoWo] = v;
        }
        __device__ __forceinline__ void store4(const Params& p, int m, f32x4 v) {
            if (p.bias) v += p.bias[m];
            if (m >= p.sig_from) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = conv_row_sigmoid(v[e]);
            }
            *reinterpret_cast<f32x4*>(base + (size_t)m * HoWo) = v;
        }
    };
};

// ConvFwdBufLoader whose output ROWS are interleaved in quads: y[b][Cout / 4][Ho * Wo][4] -- channel m of pixel p at
// ((m >> 2) * Ho*Wo + p) * 4 + (m & 3).  The DCN column gradient (a 1x1 convolution with 9 C output rows) is written this way
// since round 6: its two consumers then read the four channels of a (pixel, tap) with ONE 16-byte load instead of four
// 4-byte loads from rows 4 Ho*Wo bytes apart, and this epilogue stores a lane's four consecutive accumulator rows as one cell
// without the LDS transpose of the pixel-major epilogue (igemm.cuh ig_epilogue_quads).  No bias, no activation, no residual.
struct ConvFwdBufQuadLoader : ConvFwdBufLoader {
    static const char* name() { return "ConvFwdBufQuadLoader"; }
    __device__ ConvFwdBufQuadLoader(const Params& p, long long n, bool n_valid) : ConvFwdBufLoader(p, n, n_valid) {}
    struct Out {
        float* base;
        size_t quad_stride;       // floats between row quads: 4 Ho*Wo
        __device__ Out(const Params& p, long long n) {
            const int HoWo = p.g.Ho * p.g.Wo;
            const int ni = (int)n, b = ni / HoWo, pp = ni - b * HoWo;
            base = p.y + (size_t)b * p.g.Co * HoWo + (size_t)pp * 4;
            quad_stride = (size_t)HoWo * 4;
        }
        __device__ __forceinline__ void store(const Params&, int m, float v) { base[(size_t)(m >> 2) * quad_stride + (m & 3)] = v; }
        static constexpr bool kVec4 = false;
        __device__ static bool vec4_ok(const Params&) { return false; }
        static constexpr bool kQuads = true;
        __device__ __forceinline__ void store_quad(const Params&, int m, f32x4 v) {
            *reinterpret_cast<f32x4*>(base + (size_t)(m >> 2) * quad_stride) = v;
        }
    };
};

// ConvFwdBufLoader for the calls that leave BatchNorm statistics (cnuda_conv2d_forward_stats): same gather, same epilogue
// stores; the parameter type carries the statistics pointer, which is what turns ig_epilogue_vec4's statistics tail on.
struct ConvFwdBufStatsLoader : ConvFwdBufLoader {
    using Params = ConvFwdStatsParams;
    static const char* name() { return "ConvFwdBufStatsLoader"; }
    __device__ ConvFwdBufStatsLoader(const Params& p, long long n, bool n_valid) : ConvFwdBufLoader(p, n, n_valid) {}
};

// ---------------------------------------------------------------------------
// A 1x1 convolution over the CHANNEL CONCATENATION of up to four tensors without the concatenation (round 6): DLA's
// Root is conv(torch.cat(children, 1)) (backends/dla.py:150-168) -- 17 copies into concatenated buffers per forward pass
// and 17 slices back out of their gradient per backward pass.  The K axis of the GEMM IS the concatenated channel axis:
// chunk k0 belongs to source s with k0[s] <= k0 < k0[s + 1] (every source has a multiple of 64 channels: a 16-deep chunk
// and a 64-column weight-gradient group never straddle two), so the loaders switch the buffer they gather from per chunk,
// the input-gradient epilogue switches the tensor it stores to per row, and nothing is copied.
// ---------------------------------------------------------------------------
constexpr int CAT_MAX = 4;
struct ConvCat {
    const float* x[CAT_MAX];
    int c[CAT_MAX];           // channels per source (multiples of 64)
    int k0[CAT_MAX + 1];      // first concatenated channel of source s; k0[n] = all channels
    int n;
};
struct ConvFwdCatParams : ConvFwdStatsParams {     // (x unused; stats nullable: the statistics tail is compiled in)
    ConvCat cat;
};
struct ConvFwdCatLoader {
    using Params = ConvFwdCatParams;
    static const char* name() { return "ConvFwdCatLoader"; }
    static constexpr bool kHasSideOutput = false;
    const Params& p;
    unsigned pix;             // byte offset of the pixel inside a channel plane
    int b;
    bool valid;
    int ck0 = 0, src = 0, cc0 = 0;      // chunk cursor (wave-uniform): source and first channel inside it
    __device__ ConvFwdCatLoader(const Params& pp, long long n, bool n_valid) : p(pp), valid(n_valid) {
        const int HW = p.g.H * p.g.W;
        const int nn = n_valid ? (int)n : 0;
        b = nn / HW;
        pix = (unsigned)(nn - b * HW) * 4u;
    }
    __device__ __forceinline__ void load(int k0, int ksub, float (&v)[8]) {
        const int HW = p.g.H * p.g.W;
        while (ck0 < k0) {
            ck0 += IG_BK;
            cc0 += IG_BK;
            if (src < p.cat.n && cc0 >= p.cat.c[src]) { cc0 = 0; ++src; }
        }
        const bool in_k = src < p.cat.n;                                     // (k0 past the last channel: the sentinel)
        const int s = in_k ? src : 0, cs = p.cat.c[s];
        const buf_rsrc rs = ig_make_rsrc(p.cat.x[s], (unsigned)((size_t)p.g.B * cs * HW * sizeof(float)));
        const unsigned voff = (valid && in_k) ? (unsigned)(b * cs * HW) * 4u + pix : IG_BUF_OOB;
        const int c0 = cc0 + __builtin_amdgcn_readfirstlane(ksub);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = ig_buf_load(rs, voff, (unsigned)((c0 + 2 * j) * HW) * (unsigned)sizeof(float));
    }
    using Out = ConvFwdLoader<true>::Out;
};

// Input gradient: gx[b][c][iy][ix] = sum_{tap,o} W[o][c][tap] * gy[b][o][(iy+ph-r)/sh][(ix+pw-s)/sw]
// (terms exist only where the divisions are exact and land inside the output).
struct ConvDgradParams {
    ConvGeom g;
    const float* gy;
    float* gx;
    int cop;   // output channels per tap on the K axis: Co rounded up to the 16-deep chunk, so that a chunk
               // never straddles two taps (padded rows carry zero weights; their loads are clamped to Co-1)
    const float* add = nullptr;   // nullable, shaped like gx (may BE gx): gx = input gradient + add (+ add2) -- the other
    const float* add2 = nullptr;  // consumers' shares of a tensor's gradient, summed in this epilogue (hip_runtime.fanout)
};
struct ConvDgradLoader {
    using Params = ConvDgradParams;
    static const char* name() { return "ConvDgradLoader"; }
    static constexpr bool kHasSideOutput = false;
    const ConvGeom& g;
    const float* gy_b;
    int iy, ix, cop;
    bool valid;
    int ck0 = 0, ctap = 0, co0 = 0, cr = 0, cs = 0;     // chunk cursor, see ConvFwdLoader::seek
    __device__ __forceinline__ void seek(int k0) {
        while (ck0 < k0) {
            ck0 += IG_BK;
            co0 += IG_BK;
            if (co0 >= cop) { co0 -= cop; ++ctap; if (++cs == g.kw) { cs = 0; ++cr; } }
        }
    }
    __device__ ConvDgradLoader(const Params& p, long long n, bool n_valid) : g(p.g), cop(p.cop), valid(n_valid) {
        const int HW = g.H * g.W;
        const int nn = n_valid ? (int)n : 0;   // N < 2^31 is checked on the host: 32-bit index math
        const int b = nn / HW, pp = nn - b * HW;
        iy = pp / g.W;
        ix = pp - iy * g.W;
        gy_b = p.gy + (size_t)b * g.Co * g.Ho * g.Wo;
    }
    __device__ __forceinline__ bool locate(int r, int s, int& off) const {
        const int ty = iy + g.ph - r, tx = ix + g.pw - s;
        if (ty < 0 || tx < 0) return false;
        if (g.sh == 1 && g.sw == 1) {          // the common case (stride > 1 mostly takes the class loader): no divisions
            if (ty >= g.Ho || tx >= g.Wo) return false;
            off = ty * g.Wo + tx;
            return true;
        }
        const int oy = ty / g.sh, ox = tx / g.sw;
        if (oy * g.sh != ty || ox * g.sw != tx || oy >= g.Ho || ox >= g.Wo) return false;
        off = oy * g.Wo + ox;
        return true;
    }
    __device__ __forceinline__ void load(int k0, int ksub, float (&v)[8]) {
        const int HoWo = g.Ho * g.Wo, K = g.kh * g.kw * cop;
        seek(k0);
        const int o0 = co0 + ksub;
        int off = 0;
        const bool ok = valid && k0 < K && locate(cr, cs, off);
        const float* ptr = gy_b + off;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int o = o0 + 2 * j < g.Co ? o0 + 2 * j : g.Co - 1;
            v[j] = ok ? ptr[(size_t)o * HoWo] : 0.0f;
        }
    }
    struct Out {
        float* base;
        const float *addb, *addc;
        int HW;
        __device__ Out(const Params& p, long long n) {
            HW = p.g.H * p.g.W;
            const int ni = (int)n, b = ni / HW, pp = ni - b * HW;
            base = p.gx + (size_t)b * p.g.C * HW + pp;
            addb = p.add ? p.add + (size_t)b * p.g.C * HW + pp : nullptr;
            addc = p.add2 ? p.add2 + (size_t)b * p.g.C * HW + pp : nullptr;
        }
        __device__ __forceinline__ void store(const Params&, int m, float v) {
            if (addb) v += addb[(size_t)m * HW];
            if (addc) v += addc[(size_t)m * HW];
            base[(size_t)m * HW] = v;
        }
        static constexpr bool kVec4 = true;
        __device__ static bool vec4_ok(const Params& p) { return ((p.g.H * p.g.W) & 3) == 0; }
        __device__ __forceinline__ void store4(const Params&, int m, f32x4 v) {
            if (addb) v += *reinterpret_cast<const f32x4*>(addb + (size_t)m * HW);
            if (addc) v += *reinterpret_cast<const f32x4*>(addc + (size_t)m * HW);
            *reinterpret_cast<f32x4*>(base + (size_t)m * HW) = v;
        }
    };
};

// The stride-1 input gradient with buffer addressing (see ConvFwdBufLoader): window corner offset + one bit per tap,
// computed once per tile; padded output-channel rows (zero weights) read channel Co-1, a scalar clamp.
struct ConvDgradBufLoader {
    using Params = ConvDgradParams;
    static const char* name() { return "ConvDgradBufLoader"; }
    static constexpr bool kHasSideOutput = false;
    const ConvGeom& g;
    buf_rsrc rs;
    unsigned pix_off, tap_ok;
    int cop;
    int ck0 = 0, ctap = 0, co0 = 0, cr = 0, cs = 0;
    __device__ __forceinline__ void seek(int k0) {
        while (ck0 < k0) {
            ck0 += IG_BK;
            co0 += IG_BK;
            if (co0 >= cop) { co0 -= cop; ++ctap; if (++cs == g.kw) { cs = 0; ++cr; } }
        }
    }
    __device__ ConvDgradBufLoader(const Params& p, long long n, bool n_valid) : g(p.g), cop(p.cop) {
        const int HW = g.H * g.W;
        const int nn = n_valid ? (int)n : 0;
        const int b = nn / HW, pp = nn - b * HW;
        const int iy = pp / g.W, ix = pp - iy * g.W;
        rs = ig_make_rsrc(p.gy, (unsigned)((size_t)g.B * g.Co * g.Ho * g.Wo * sizeof(float)));
        pix_off = (unsigned)(((b * g.Co * g.Ho + iy + g.ph) * g.Wo + ix + g.pw) * (int)sizeof(float));
        tap_ok = 0;
        if (n_valid) {
            for (int r = 0; r < g.kh; ++r)
                for (int s = 0; s < g.kw; ++s) {
                    const int ty = iy + g.ph - r, tx = ix + g.pw - s;
                    if (ty >= 0 && ty < g.Ho && tx >= 0 && tx < g.Wo) tap_ok |= 1u << (r * g.kw + s);
                }
        }
    }
    __device__ __forceinline__ void load(int k0, int ksub, float (&v)[8]) {
        const int HoWo = g.Ho * g.Wo;
        seek(k0);
        const int tap = ctap < 32 ? ctap : 31;
        const bool in_k = k0 < g.kh * g.kw * cop;
        const unsigned voff = (in_k && ((tap_ok >> tap) & 1u)) ? pix_off - (unsigned)((cr * g.Wo + cs) * (int)sizeof(float))
                                                                : IG_BUF_OOB;
        const int o0 = co0 + __builtin_amdgcn_readfirstlane(ksub);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int o = o0 + 2 * j < g.Co ? o0 + 2 * j : g.Co - 1;
            v[j] = ig_buf_load(rs, voff, (unsigned)(o * HoWo) * (unsigned)sizeof(float));
        }
    }
    using Out = ConvDgradLoader::Out;
};

// The input gradient of the concatenated 1x1 convolution (ConvCat above): one GEMM over all concatenated channels, row m
// of source s stored to that source's own gradient tensor [B][c_s][H*W] -- with the other consumers' shares of that tensor's
// gradient added per source (hip_runtime.fanout), as ConvDgradParams::add / add2 do for one tensor.
struct ConvDgradCatParams : ConvDgradParams {      // (gx, add, add2 unused)
    float* gxs[CAT_MAX];
    const float* adds[CAT_MAX];
    const float* add2s[CAT_MAX];
    int c[CAT_MAX];
    int n;
};
struct ConvDgradCatLoader : ConvDgradBufLoader {
    using Params = ConvDgradCatParams;
    static const char* name() { return "ConvDgradCatLoader"; }
    __device__ ConvDgradCatLoader(const Params& p, long long n, bool n_valid) : ConvDgradBufLoader(p, n, n_valid) {}
    struct Out {
        int b, pp, HW;
        __device__ Out(const Params& p, long long n) {
            HW = p.g.H * p.g.W;
            const int ni = (int)n;
            b = ni / HW;
            pp = ni - b * HW;
        }
        // row m of the concatenation -> its source's tensors and the element offset inside them.  The rows a wave stores with
        // one instruction lie in one 32-row tile and every source holds a multiple of 64 rows: the source is WAVE-UNIFORM,
        // found from the first lane's row with scalar compares (a per-lane choice among the parameter arrays would move them
        // to scratch memory).
        struct Dest { float* gx; const float *a1, *a2; size_t o; };
        __device__ __forceinline__ Dest locate(const Params& p, int m) const {
            const int mu = __builtin_amdgcn_readfirstlane(m);
            int s = 0, k0 = 0, cs = p.c[0];
            Dest d{p.gxs[0], p.adds[0], p.add2s[0], 0};
#define CNUDA_CAT_STEP(i)                                                                                             \
            if (s == i - 1 && i < p.n && mu >= k0 + cs) { k0 += cs; s = i; cs = p.c[i]; d.gx = p.gxs[i]; d.a1 = p.adds[i]; d.a2 = p.add2s[i]; }
            CNUDA_CAT_STEP(1) CNUDA_CAT_STEP(2) CNUDA_CAT_STEP(3)
#undef CNUDA_CAT_STEP
            static_assert(CAT_MAX == 4, "three steps");
            d.o = ((size_t)b * cs + (m - k0)) * HW + pp;
            return d;
        }
        __device__ __forceinline__ void store(const Params& p, int m, float v) {
            const Dest d = locate(p, m);
            if (d.a1) v += d.a1[d.o];
            if (d.a2) v += d.a2[d.o];
            d.gx[d.o] = v;
        }
        static constexpr bool kVec4 = true;
        __device__ static bool vec4_ok(const Params& p) { return ((p.g.H * p.g.W) & 3) == 0; }
        __device__ __forceinline__ void store4(const Params& p, int m, f32x4 v) {
            const Dest d = locate(p, m);
            if (d.a1) v += *reinterpret_cast<const f32x4*>(d.a1 + d.o);
            if (d.a2) v += *reinterpret_cast<const f32x4*>(d.a2 + d.o);
            *reinterpret_cast<f32x4*>(d.gx + d.o) = v;
        }
    };
};

// Input gradient for stride > 1, one launch per parity class (py, px) of the input pixel: inside a class
// the set of taps that can reach an output pixel is the same for every pixel ((iy + ph - r) % sh == 0), so
// the K axis only holds those taps (1 + 2 + 2 + 4 of 9 for 3x3 / stride 2 instead of 9 each -> 4x less work
// than gathering every tap and masking the misses).
struct ConvDgradClassParams {
    ConvGeom g;
    const float* gy;
    float* gx;
    int py, px, Hc, Wc, ntaps;
    int tap_r[9], tap_s[9];     // kernel coordinates of the class's taps, in packed-K order
    int tap_dy[9], tap_dx[9];   // (py + ph - r) / sh, (px + pw - s) / sw: output pixel of tap t = (qy + dy, qx + dx)
    const float *add = nullptr, *add2 = nullptr;   // as ConvDgradParams::add, add2
};
struct ConvDgradClassLoader {
    using Params = ConvDgradClassParams;
    static const char* name() { return "ConvDgradClassLoader"; }
    static constexpr bool kHasSideOutput = false;
    const Params& p;
    const float* gy_b;
    int iy, ix;
    bool valid;
    __device__ ConvDgradClassLoader(const Params& pp, long long n, bool n_valid) : p(pp), valid(n_valid) {
        const int HcWc = p.Hc * p.Wc;
        const int nn = n_valid ? (int)n : 0;   // N < 2^31 is checked on the host: 32-bit index math
        const int b = nn / HcWc, q = nn - b * HcWc;
        const int qy = q / p.Wc, qx = q - qy * p.Wc;
        iy = p.py + qy * p.g.sh;
        ix = p.px + qx * p.g.sw;
        gy_b = p.gy + (size_t)b * p.g.Co * p.g.Ho * p.g.Wo;
    }
    __device__ __forceinline__ void load(int k0, int ksub, float (&v)[8]) {
        const ConvGeom& g = p.g;
        const int HoWo = g.Ho * g.Wo, K = p.ntaps * g.Co;
        // Co % 16 == 0 is required by the host for this path: one tap per 16-deep chunk
        const int ti = k0 / g.Co, o0 = k0 - ti * g.Co + ksub;
        bool ok = valid && k0 < K;
        int off = 0;
        if (ok) {
            const int ty = iy + g.ph - p.tap_r[ti], tx = ix + g.pw - p.tap_s[ti];   // exact multiples by construction
            const int oy = ty / g.sh, ox = tx / g.sw;
            ok = ty >= 0 && tx >= 0 && oy < g.Ho && ox < g.Wo;
            off = oy * g.Wo + ox;
        }
        const float* ptr = gy_b + (size_t)o0 * HoWo + (ok ? off : 0);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = ok ? ptr[(size_t)(2 * j) * HoWo] : 0.0f;
    }
    struct Out {
        float* base;
        const float *addb, *addc;
        int HW;
        __device__ Out(const Params& p, long long n) {
            HW = p.g.H * p.g.W;
            const int HcWc = p.Hc * p.Wc;
            const int ni = (int)n, b = ni / HcWc, q = ni - b * HcWc;
            const int qy = q / p.Wc, qx = q - qy * p.Wc;
            const size_t o = (size_t)b * p.g.C * HW + (size_t)(p.py + qy * p.g.sh) * p.g.W + p.px + qx * p.g.sw;
            base = p.gx + o;
            addb = p.add ? p.add + o : nullptr;
            addc = p.add2 ? p.add2 + o : nullptr;
        }
        __device__ __forceinline__ void store(const Params&, int m, float v) {
            if (addb) v += addb[(size_t)m * HW];
            if (addc) v += addc[(size_t)m * HW];
            base[(size_t)m * HW] = v;
        }
        static constexpr bool kVec4 = false;      // a parity class's pixels are `stride` apart in memory
        __device__ static bool vec4_ok(const Params&) { return false; }
        __device__ __forceinline__ void store4(const Params&, int, f32x4) {}
    };
};

// The parity-class gather with buffer addressing: output pixel of tap t = (qy + dy_t, qx + dx_t) with
// dy_t = (py + ph - r_t) / sh exact, the same for every pixel of the class -- a per-thread base plus a scalar.
struct ConvDgradClassBufLoader {
    using Params = ConvDgradClassParams;
    static const char* name() { return "ConvDgradClassBufLoader"; }
    static constexpr bool kHasSideOutput = false;
    const Params& p;
    buf_rsrc rs;
    unsigned pix_off, tap_ok;
    int ck0 = 0, cti = 0, co0 = 0;      // chunk cursor: tap index in the class's list, first output channel
    __device__ ConvDgradClassBufLoader(const Params& pp, long long n, bool n_valid) : p(pp) {
        const ConvGeom& g = p.g;
        const int HcWc = p.Hc * p.Wc;
        const int nn = n_valid ? (int)n : 0;
        const int b = nn / HcWc, q = nn - b * HcWc;
        const int qy = q / p.Wc, qx = q - qy * p.Wc;
        rs = ig_make_rsrc(p.gy, (unsigned)((size_t)g.B * g.Co * g.Ho * g.Wo * sizeof(float)));
        pix_off = (unsigned)(((b * g.Co * g.Ho + qy) * g.Wo + qx) * (int)sizeof(float));
        tap_ok = 0;
        if (n_valid) {
            for (int t = 0; t < p.ntaps; ++t) {
                const int oy = qy + p.tap_dy[t], ox = qx + p.tap_dx[t];
                if (oy >= 0 && oy < g.Ho && ox >= 0 && ox < g.Wo) tap_ok |= 1u << t;
            }
        }
    }
    __device__ __forceinline__ void load(int k0, int ksub, float (&v)[8]) {
        const ConvGeom& g = p.g;
        const int HoWo = g.Ho * g.Wo;
        while (ck0 < k0) {                                   // Co % 16 == 0 (host): one tap per chunk
            ck0 += IG_BK;
            co0 += IG_BK;
            if (co0 >= g.Co) { co0 -= g.Co; ++cti; }
        }
        const int ti = cti < p.ntaps ? cti : 0;            // past the last tap: K padding, every lane reads the sentinel
        const int tapoff = p.tap_dy[ti] * g.Wo + p.tap_dx[ti];
        const unsigned voff = (cti < p.ntaps && ((tap_ok >> ti) & 1u)) ? pix_off + (unsigned)(tapoff * (int)sizeof(float))
                                                                        : IG_BUF_OOB;
        const int o0 = co0 + __builtin_amdgcn_readfirstlane(ksub);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = ig_buf_load(rs, voff, (unsigned)((o0 + 2 * j) * HoWo) * (unsigned)sizeof(float));
    }
    using Out = ConvDgradClassLoader::Out;
};

// All parity classes of one input gradient in ONE launch (round 6): blockIdx.y = class.  The classes share M, the pixel count
// and the row tile; their K differs (1 + 2 + 2 + 4 taps for 3x3 / stride 2), so the class with the most taps goes first.
// One launch fills the chip where four small ones each left it half empty and paid their own ramp and tail.
constexpr int MAX_CLASSES = 4;
struct ConvDgradClassSet {
    ConvDgradClassParams cls[MAX_CLASSES];
    const float* A[MAX_CLASSES];
    int Kp[MAX_CLASSES];
};
template <int BM>
__global__ __launch_bounds__(2 * IG_THREADS, (BM == 128 ? 2 : 1)) void igemm_fwd_ws_classes_kernel(
    ConvDgradClassSet s, int Mp, int M, long long N, int n_tiles, int m_tiles) {
    const int c = blockIdx.y;
    igemm_fwd_ws_body<BM, ConvDgradClassBufLoader, IG_KC>(s.cls[c], s.A[c], Mp, s.Kp[c], M, N, n_tiles, m_tiles);
}
template <int BM>
__global__ __launch_bounds__(IG_THREADS, (BM == 128 ? 3 : 1)) void igemm_fwd_classes_kernel(
    ConvDgradClassSet s, int Mp, int M, long long N, int n_tiles, int m_tiles) {
    const int c = blockIdx.y;
    igemm_fwd_body<BM, ConvDgradClassBufLoader>(s.cls[c], s.A[c], Mp, s.Kp[c], M, N, n_tiles, m_tiles);
}

// Weight gradient: gw[o][(tap,c)] = sum_{b,p} gy[b][o][p] * x[b][c][window(p, tap)]
struct ConvWParams {
    ConvGeom g;
    const float *x, *gy;
};
// MODE 2: C % 64 == 0 (the 64 columns of a workgroup share one tap: no per-element index math); 0: generic.
// (A four-group variant for C % 16 == 0 measured slower than the generic path: more registers, occupancy 3.)
template <int MODE>
struct ConvWLoader {
    using Params = ConvWParams;
    static const char* name() { return MODE == 2 ? "ConvWLoader<2>" : (MODE == 1 ? "ConvWLoader<1>" : "ConvWLoader<0>"); }
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
    __device__ ConvWLoader(const Params& pp, long long n, long long n_end) : p(pp) {
        cursor_init(n, n_end, p.g.Ho * p.g.Wo, p.g.Wo);
    }
    __device__ __forceinline__ void advance() { cursor_advance(p.g.Ho * p.g.Wo, p.g.Wo); }
    template <int NV, int STEP>
    __device__ __forceinline__ void load_g(int m0, int msub, float (&v)[NV]) {
        const ConvGeom& g = p.g;
        const int HoWo = g.Ho * g.Wo;
        const float* base = p.gy + (size_t)b_ * g.Co * HoWo + pp_;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int m = m0 + msub + STEP * i;
            v[i] = (valid_ && m < g.Co) ? base[(size_t)m * HoWo] : 0.0f;
        }
    }
    template <int NV, int STEP>
    __device__ __forceinline__ void load_b(int j0, int jsub, float (&v)[NV]) {
        const ConvGeom& g = p.g;
        const int HW = g.H * g.W, K = g.kh * g.kw * g.C;
        const int iy0 = oy_ * g.sh - g.ph, ix0 = ox_ * g.sw - g.pw;
        const float* x_b = p.x + (size_t)b_ * g.C * HW;
        if (MODE == 2) {
            // every aligned group of 64 columns has one tap (C % 64 == 0): NV/16 bounds tests, no index math
            constexpr int PER = 64 / STEP;     // values of this thread inside one 64-column group
#pragma unroll
            for (int h = 0; h < NV / PER; ++h) {
                const int jg = j0 + 64 * h;
                const int tap = jg / g.C, c0 = jg - tap * g.C + jsub;
                const int r = tap / g.kw, s = tap - r * g.kw;
                const int iy = iy0 + r, ix = ix0 + s;
                const bool ok1 = valid_ && jg < K && iy >= 0 && iy < g.H && ix >= 0 && ix < g.W;
                const float* ptr = x_b + (size_t)c0 * HW + (ok1 ? iy * g.W + ix : 0);
#pragma unroll
                for (int i = 0; i < PER; ++i) v[PER * h + i] = ok1 ? ptr[(size_t)(STEP * i) * HW] : 0.0f;
            }
            return;
        }
        int cur = -1, off = 0;
        bool ok = false;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int k = j0 + jsub + STEP * i;
            float val = 0.0f;
            if (valid_ && k < K) {
                const int tap = k / g.C, c = k - tap * g.C;
                if (tap != cur) {
                    const int r = tap / g.kw, s = tap - r * g.kw;
                    const int iy = iy0 + r, ix = ix0 + s;
                    ok = iy >= 0 && iy < g.H && ix >= 0 && ix < g.W;
                    off = iy * g.W + ix;
                    cur = tap;
                }
                if (ok) val = x_b[(size_t)c * HW + off];
            }
            v[i] = val;
        }
    }
};

// ConvWLoader<2> (C % 64 == 0) with buffer addressing: per chunk the cursor advance, one offset for the grad_y rows,
// one window-corner offset for x and a bounds test per 64-column group; every channel term is a scalar offset.
struct ConvWBufLoader {
    using Params = ConvWParams;
    static const char* name() { return "ConvWBufLoader"; }
    const Params& p;
    buf_rsrc rg, rx;
    IgPixelCursor c;
    unsigned gimg, ximg;          // byte offsets of the cursor's image in grad_y / x
    __device__ ConvWBufLoader(const Params& pp, long long n, long long n_end) : p(pp) {
        const ConvGeom& g = p.g;
        rg = ig_make_rsrc(p.gy, (unsigned)((size_t)g.B * g.Co * g.Ho * g.Wo * sizeof(float)));
        rx = ig_make_rsrc(p.x, (unsigned)((size_t)g.B * g.C * g.H * g.W * sizeof(float)));
        c.init(n, n_end, g.Ho * g.Wo, g.Wo);
        gimg = (unsigned)(c.b_ * g.Co * g.Ho * g.Wo) * 4u;
        ximg = (unsigned)(c.b_ * g.C * g.H * g.W) * 4u;
    }
    __device__ __forceinline__ void advance() {
        const ConvGeom& g = p.g;
        c.advance(g.Ho * g.Wo, g.Wo);
        if (c.crossed_) {          // (rare: a chunk that enters the next image)
            gimg += (unsigned)(c.crossed_ * g.Co * g.Ho * g.Wo) * 4u;
            ximg += (unsigned)(c.crossed_ * g.C * g.H * g.W) * 4u;
        }
    }
    template <int NV, int STEP>
    __device__ __forceinline__ void load_g(int m0, int msub, float (&v)[NV]) {
        ig_buf_rows<NV, STEP>(rg, c, gimg, p.g.Ho * p.g.Wo, m0, msub, v);
    }
    template <int NV, int STEP>
    __device__ __forceinline__ void load_b(int j0, int jsub, float (&v)[NV]) {
        const ConvGeom& g = p.g;
        const int HW = g.H * g.W, K = g.kh * g.kw * g.C;
        const int iy0 = ig_mad24(c.oy_, g.sh, -g.ph), ix0 = ig_mad24(c.ox_, g.sw, -g.pw);
        // x[b][jsub][iy0][ix0] (may lie outside the image: only used with an in-range tap)
        const unsigned corner = ximg + (unsigned)(ig_mad24(jsub, HW, ig_mad24(iy0, g.W, ix0))) * 4u;
        constexpr int PER = 64 / STEP;
#pragma unroll
        for (int h = 0; h < NV / PER; ++h) {
            const int jg = j0 + 64 * h;                       // wave-uniform: one tap per 64 columns
            const int tap = jg / g.C, c0 = jg - tap * g.C;
            const int r = tap / g.kw, s = tap - r * g.kw;
            const int iy = iy0 + r, ix = ix0 + s;
            const bool ok = c.valid_ && jg < K && (unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W;
            const unsigned voff = ok ? corner + (unsigned)((r * g.W + s) * (int)sizeof(float)) : IG_BUF_OOB;
#pragma unroll
            for (int i = 0; i < PER; ++i)
                v[PER * h + i] = ig_buf_load(rx, voff, (unsigned)((c0 + STEP * i) * HW) * (unsigned)sizeof(float));
        }
    }
};

// The same loader for channel counts that are multiples of 8 but not of 64 (DLA-34's 16 -> 32 and 32 -> 64 stride-2
// convolutions): a thread's eight columns j0 + jsub + 8 i of a 64-column group no longer share a tap, but the eight
// channels [c, c + 8) a lane phase covers never straddle one (8 | C) -- tap, window offset and bounds test per i, all
// else as above.  Replaces the pointer-arithmetic ConvWLoader<0> there (152 VGPRs, 29-58 TFLOP/s).
struct ConvWBufLoaderC8 : ConvWBufLoader {
    static const char* name() { return "ConvWBufLoaderC8"; }
    __device__ ConvWBufLoaderC8(const Params& pp, long long n, long long n_end) : ConvWBufLoader(pp, n, n_end) {}
    template <int NV, int STEP>
    __device__ __forceinline__ void load_b(int j0, int jsub, float (&v)[NV]) {
        static_assert(STEP == 8, "eight row phases: a phase's channels stay inside one tap when 8 | C");
        const ConvGeom& g = p.g;
        const int HW = g.H * g.W, K = g.kh * g.kw * g.C;
        const int iy0 = ig_mad24(c.oy_, g.sh, -g.ph), ix0 = ig_mad24(c.ox_, g.sw, -g.pw);
        const unsigned corner = ximg + (unsigned)(ig_mad24(jsub, HW, ig_mad24(iy0, g.W, ix0))) * 4u;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int jg = j0 + STEP * i;                     // wave-uniform
            const int tap = jg / g.C, c0 = jg - tap * g.C;
            const int r = tap / g.kw, s = tap - r * g.kw;
            const int iy = iy0 + r, ix = ix0 + s;
            const bool ok = c.valid_ && jg < K && (unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W;
            const unsigned voff = ok ? corner + (unsigned)((r * g.W + s) * (int)sizeof(float)) : IG_BUF_OOB;
            v[i] = ig_buf_load(rx, voff, (unsigned)(c0 * HW) * (unsigned)sizeof(float));
        }
    }
};

// The weight gradient of the concatenated 1x1 convolution: ConvWBufLoader's x operand gathered from the source that owns the
// 64-column group (sources hold multiples of 64 channels).  1x1, stride 1, no padding: the cursor's pixel is the input pixel.
struct ConvWCatParams : ConvWParams {              // (x unused)
    ConvCat cat;
};
struct ConvWCatLoader {
    using Params = ConvWCatParams;
    static const char* name() { return "ConvWCatLoader"; }
    const Params& p;
    buf_rsrc rg;
    IgPixelCursor c;
    unsigned gimg;
    __device__ ConvWCatLoader(const Params& pp, long long n, long long n_end) : p(pp) {
        const ConvGeom& g = p.g;
        rg = ig_make_rsrc(p.gy, (unsigned)((size_t)g.B * g.Co * g.Ho * g.Wo * sizeof(float)));
        c.init(n, n_end, g.Ho * g.Wo, g.Wo);
        gimg = (unsigned)(c.b_ * g.Co * g.Ho * g.Wo) * 4u;
    }
    __device__ __forceinline__ void advance() {
        const ConvGeom& g = p.g;
        c.advance(g.Ho * g.Wo, g.Wo);
        if (c.crossed_) gimg += (unsigned)(c.crossed_ * g.Co * g.Ho * g.Wo) * 4u;
    }
    template <int NV, int STEP>
    __device__ __forceinline__ void load_g(int m0, int msub, float (&v)[NV]) {
        ig_buf_rows<NV, STEP>(rg, c, gimg, p.g.Ho * p.g.Wo, m0, msub, v);
    }
    template <int NV, int STEP>
    __device__ __forceinline__ void load_b(int j0, int jsub, float (&v)[NV]) {
        const ConvGeom& g = p.g;
        const int HW = g.H * g.W;
        constexpr int PER = 64 / STEP;
#pragma unroll
        for (int h = 0; h < NV / PER; ++h) {
            const int jg = j0 + 64 * h;                       // wave-uniform: one source per 64 columns
            int s = 0;
#pragma unroll
            for (int i = 1; i < CAT_MAX; ++i) s += (i < p.cat.n && jg >= p.cat.k0[i]) ? 1 : 0;
            const int cs = p.cat.c[s], c0 = jg - p.cat.k0[s];
            const buf_rsrc rx = ig_make_rsrc(p.cat.x[s], (unsigned)((size_t)g.B * cs * HW * sizeof(float)));
            const bool ok = c.valid_ && jg < g.C;
            const unsigned voff = ok ? (unsigned)(ig_mad24(c.b_, cs, jsub) * HW + c.pp_) * 4u : IG_BUF_OOB;
#pragma unroll
            for (int i = 0; i < PER; ++i)
                v[PER * h + i] = ig_buf_load(rx, voff, (unsigned)((c0 + STEP * i) * HW) * (unsigned)sizeof(float));
        }
    }
};

int fill_geom(ConvGeom& g, int B, int C, int H, int W, int Co, int kh, int kw, int sh, int sw, int ph, int pw,
              const char* who) {
    CNUDA_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && Co > 0, "%s: empty tensor", who);
    CNUDA_REQUIRE(kh > 0 && kw > 0 && sh > 0 && sw > 0 && ph >= 0 && pw >= 0, "%s: bad kernel geometry", who);
    g = ConvGeom{B, C, H, W, Co, kh, kw, sh, sw, ph, pw, (H + 2 * ph - kh) / sh + 1, (W + 2 * pw - kw) / sw + 1};
    CNUDA_REQUIRE(H + 2 * ph >= kh && W + 2 * pw >= kw && g.Ho > 0 && g.Wo > 0, "%s: kernel larger than padded input",
                  who);
    return 0;
}

// CNUDA_BUF=0: the pointer-addressed loaders (A/B measurements; tensors of 2 GiB and more always take them)
bool buffer_addressing() {
    static const bool on = !(getenv("CNUDA_BUF") && getenv("CNUDA_BUF")[0] == '0');
    return on;
}

// ConvWBufLoader: both tensors under 2 GiB (32-bit byte offsets, the sentinel above them) and planes under 2^23
// elements (24-bit multiplies in the per-chunk address arithmetic)
bool wgrad_buffer_ok(const ConvGeom& g) {
    return buffer_addressing() && (size_t)g.B * g.C * g.H * g.W * sizeof(float) < IG_BUF_OOB &&
           (size_t)g.B * g.Co * g.Ho * g.Wo * sizeof(float) < IG_BUF_OOB && (long long)g.H * g.W < (1 << 23) &&
           (long long)g.Ho * g.Wo < (1 << 23);
}

// largest tile that still fills the chip (small feature maps: 16x16 / 32x32): the 128-row tile wants two workgroups
// per CU, the 64-row tile is still the better choice with one (round 3: the 32-row tile this used to force on the
// 512-channel level at batch 16 runs at 80 TFLOP/s; -1.5 % per inference batch)
int pick_bm(int M, long long N) {
    int bm = M > 64 ? 128 : (M > 32 ? 64 : 32);
    const long long n_tiles = (N + IG_BN - 1) / IG_BN;
    while (bm > 32 && n_tiles * ((M + bm - 1) / bm) < (bm == 128 ? 512 : 256)) bm >>= 1;
    return bm;
}

// Split-K (round 6, igemm.cuh igemm_fwd_*splitk_kernel): a forward-type GEMM whose pixel x row tiles at the NATURAL row tile
// cover less than half the chip but whose K is long keeps that tile and cuts K over grid.y -- partial slabs, a fixed-order
// reduce that runs the loader's epilogue.  pick_bm's answer to the same problem is a smaller row tile (more, emptier
// workgroups: 12-20 TFLOP/s on the ADVENT discriminator's 4 x 4 convolutions and the 512 -> 27 offset convolution).
// CNUDA_SPLITK=0 keeps pick_bm's plan (A/B measurements; tests/test_gpu_kernel_switches.py).
struct SplitK {
    int bm = 0, z = 1, split_k = 0;          // row tile, splits, K elements per split (a multiple of the chunk)
    bool on() const { return z > 1; }
};
int g_splitk_max_tiles = getenv("CNUDA_SPLITK_TILES") ? atoi(getenv("CNUDA_SPLITK_TILES")) : 128;   // cnuda_conv_set_splitk_policy (tests, measurements)
SplitK pick_splitk(int M, long long N, int Kp) {
    static const bool enabled = !(getenv("CNUDA_SPLITK") && getenv("CNUDA_SPLITK")[0] == '0');
    SplitK s;
    if (!enabled || matrix_mode() != 0) return s;
    const int bm0 = M > 64 ? 128 : (M > 32 ? 64 : 32);
    const long long tiles = ((N + IG_BN - 1) / IG_BN) * ((M + bm0 - 1) / bm0);
    const int nchunk = Kp / IG_KC;
    if (tiles >= g_splitk_max_tiles || nchunk < 32) return s;
    int z = (int)std::min<long long>(nchunk / 16, 512 / tiles);          // >= 16 chunks per split, ~two workgroups per CU
    if (z < 2) return s;
    const int per = (nchunk + z - 1) / z;
    s.bm = bm0;
    s.z = (nchunk + per - 1) / per;
    s.split_k = per * IG_KC;
    return s;
}
size_t splitk_slab_bytes(const SplitK& s, int M, long long N) {
    return s.on() ? (size_t)s.z * round_up(M, s.bm) * (size_t)((N + IG_BN - 1) / IG_BN * IG_BN) * sizeof(float) : 0;
}

// hwgrad_kernel (hwgrad.cuh) takes the weight gradient of the narrow 3x3 convolutions (the DCN offset / mask layers)
// (round 6: any row width that is a multiple of 8 -- rectangular tiles, hwgrad.cuh; the overhang of a map whose height is
// no multiple of the tile's stays under a quarter)
int hwgrad_tiles_y(const ConvGeom& g) { const int tr = HW_BN / halo_tile_width(g.W); return (g.H + tr - 1) / tr; }
bool hwgrad_ok(const ConvGeom& g) {
    if (!(matrix_mode() == 0 && g.kh == 3 && g.kw == 3 && g.sh == 1 && g.sw == 1 && g.ph == 1 && g.pw == 1 &&
          g.Co <= 32 && g.C % 16 == 0 && halo_tile_width(g.W) != 0 && wgrad_buffer_ok(g)))
        return false;
    return 4 * (hwgrad_tiles_y(g) * (HW_BN / halo_tile_width(g.W)) - g.H) <= g.H;
}

// ... and hwgrad_s2_kernel that of the narrow 3x3 / stride 2 convolutions (DLA-34's level1)
// (round 6: any output row width that is a multiple of 4 -- the last tile of a row may be ragged, its missing columns stage
// zeros -- as long as the padding stays under a quarter of the row)
bool hwgrad_s2_ok(const ConvGeom& g) {
    return matrix_mode() == 0 && g.kh == 3 && g.kw == 3 && g.sh == 2 && g.sw == 2 && g.ph == 1 && g.pw == 1 && g.Co <= 64 &&
           g.C % 16 == 0 && g.H % 2 == 0 && g.W % 2 == 0 && g.Wo % 4 == 0 &&
           4 * (round_up(g.Wo, HS_BN) - g.Wo) <= g.Wo && wgrad_buffer_ok(g);
}

struct ConvPlan {
    int T;
    bool hw;                  // weight gradient on halo tiles (hwgrad_kernel): Z = pixel-tile splits per channel group
    bool hw_s2;               // ... the stride-2 form (hwgrad_s2_kernel: 128-pixel tiles)
    int hw_tiles, hw_tiles_per_split;
    int Kf, Kpf, bmf, Mpf;   // forward:  K = T*C,  M = Co
    int Kd, Kpd, bmd, Mpd;   // dgrad:    K = T*Co, M = C
    SplitK skf, skd;         // split-K plans of the two (z == 1: none)
    int Mpw, Jp, Z, wbm, wbj; // wgrad slabs and tile shape
    long long Nf, Nd, pix_per_split;
    size_t fwd_bytes, dgrad_bytes, wgrad_bytes;
};
ConvPlan make_plan(const ConvGeom& g) {
    ConvPlan q;
    q.T = g.kh * g.kw;
    q.Nf = (long long)g.B * g.Ho * g.Wo;
    q.Nd = (long long)g.B * g.H * g.W;
    q.Kf = q.T * g.C;   q.Kpf = round_up(q.Kf, IG_KC);  q.bmf = pick_bm(g.Co, q.Nf);
    q.Kd = q.T * round_up(g.Co, IG_BK);  q.Kpd = round_up(q.Kd, IG_KC);  q.bmd = pick_bm(g.C, q.Nd);
    // (split-K only for the buffer-addressed loaders of the plain stride-1 paths; the parity-class input gradient plans per class)
    const bool sk_ok = buffer_addressing() && q.T <= 32 && g.C % IG_BK == 0 &&
                       (size_t)g.B * g.C * g.H * g.W * sizeof(float) < IG_BUF_OOB &&
                       (size_t)g.B * g.Co * g.Ho * g.Wo * sizeof(float) < IG_BUF_OOB;
    if (sk_ok) q.skf = pick_splitk(g.Co, q.Nf, q.Kpf);
    if (sk_ok && g.sh == 1 && g.sw == 1) q.skd = pick_splitk(g.C, q.Nd, q.Kpd);
    if (q.skf.on()) q.bmf = q.skf.bm;
    if (q.skd.on()) q.bmd = q.skd.bm;
    q.Mpf = round_up(g.Co, q.bmf);
    q.Mpd = round_up(g.C, q.bmd);
    q.wbj = (g.Co <= 32 || (g.C % 64 == 0 && q.Kf % 128 == 0)) ? 128 : 64;   // 64 x 128: +8-13 % where nothing is padded
    // 128 x 64 where the columns do not fill 128 (K = 9 * 64) but the output channels do: the same two accumulator
    // tiles per wave and loads per MFMA as 64 x 128 (the 64 -> 256 head convolutions at 128 x 128)
    const bool wbuf = wgrad_buffer_ok(g);     // (ConvWBufLoader only)
    q.wbm = g.Co <= 32 ? 32 : ((wbuf && q.wbj == 64 && g.C % 64 == 0 && g.Co % 128 == 0) ? 128 : 64);
    // 128 x 128 (2 x 2 accumulator tiles per wave: one fragment dword per MFMA instead of 1.5, 32 loads per 64 MFMAs
    // instead of 24 per 32 -- the 64 x 128 tile runs into the LDS: ~2300 LDS cycles per 2048-cycle chunk with three
    // workgroups per CU) where both extents allow it
    if (wbuf && q.wbm == 64 && q.wbj == 128 && g.C % 64 == 0 && g.Co % 128 == 0) q.wbm = 128;
    q.Mpw = round_up(g.Co, q.wbm);
    q.Jp = round_up(q.Kf, q.wbj);
    const long long tiles = (long long)(q.Mpw / q.wbm) * (q.Jp / q.wbj);
    const long long z = wgrad_splits(tiles, q.wbm, q.wbj, (q.Nf + WG_BP - 1) / WG_BP);
    q.pix_per_split = ((q.Nf + z - 1) / z + WG_BP - 1) / WG_BP * WG_BP;
    q.Z = (int)((q.Nf + q.pix_per_split - 1) / q.pix_per_split);
    q.hw = hwgrad_ok(g);
    q.hw_s2 = !q.hw && hwgrad_s2_ok(g);
    q.hw_tiles = q.hw_tiles_per_split = 0;
    if (q.hw || q.hw_s2) {
        // (C / 16) channel groups x Z splits of the 256- (128-) pixel tiles: two workgroups per CU, at least one tile each
        q.hw_tiles = q.hw ? g.B * hwgrad_tiles_y(g) * (g.W / halo_tile_width(g.W)) : g.B * g.Ho * (round_up(g.Wo, HS_BN) / HS_BN);
        const int groups = (g.C / 16) * (q.hw_s2 ? (g.Co + 31) / 32 : 1);
        int zz = std::max(1, 512 / groups);
        if (zz > q.hw_tiles) zz = q.hw_tiles;
        q.hw_tiles_per_split = (q.hw_tiles + zz - 1) / zz;
        q.Z = (q.hw_tiles + q.hw_tiles_per_split - 1) / q.hw_tiles_per_split;
    }
    q.fwd_bytes = carve_bytes(ig_a_bytes(q.Kpf, q.Mpf), 1) + carve_bytes(splitk_slab_bytes(q.skf, g.Co, q.Nf), 1) + 256;
    q.dgrad_bytes = carve_bytes(ig_a_bytes(q.Kpd, q.Mpd), 1) + carve_bytes(splitk_slab_bytes(q.skd, g.C, q.Nd), 1) + 256;
    if (g.sh > 1 || g.sw > 1) {
        // parity-class input gradient: the largest class (all classes have Nd / (sh sw) pixels; K at most ceil(kh / sh) *
        // ceil(kw / sw) taps) bounds the packed matrix and the split-K slabs of every class
        const long long Nc = q.Nd / (g.sh * g.sw);
        const int Kpc = round_up(ceil_div(g.kh, g.sh) * ceil_div(g.kw, g.sw) * g.Co, IG_KC);
        const SplitK sc = pick_splitk(g.C, Nc, Kpc);
        const int mp = round_up(g.C, sc.on() ? sc.bm : 32);
        // (one launch for all classes: their packed matrices side by side, each at most the largest class's, rows at most 128)
        const size_t need = (size_t)g.sh * g.sw * carve_bytes(ig_a_bytes(Kpc, std::max(round_up(g.C, 128), std::max(mp, q.Mpd))), 1) +
                            carve_bytes(splitk_slab_bytes(sc, g.C, Nc), 1) + 256;
        if (need > q.dgrad_bytes) q.dgrad_bytes = need;
    }
    // (slabs, then the bias row sums per split: [Z][Mpw] -- never less than the [Co][B] scratch of the channel-sum kernels)
    q.wgrad_bytes = carve_bytes((size_t)q.Z * q.Mpw * q.Jp, 4) +
                    carve_bytes(std::max((size_t)g.Co * g.B, (size_t)q.Z * q.Mpw), 4) + 256;
    return q;
}

// ---------------------------------------------------------------------------
// Halo-tile kernels (hconv.cuh) for 3x3 / stride 1 / padding 1: the forward and -- over grad_y, with flipped taps --
// the input gradient.
// ---------------------------------------------------------------------------
struct HconvFwd {
    using Params = ConvFwdParams;
    using Out = ConvFwdLoader<true>::Out;
    static const char* name() { return "fwd"; }
};
struct HconvFwdSig {          // (the row-wise sigmoid epilogue, see ConvFwdSigParams)
    using Params = ConvFwdSigParams;
    using Out = ConvFwdBufSigLoader::Out;
    static const char* name() { return "fwd, row sigmoid"; }
};
struct HconvDgrad {
    using Params = ConvDgradParams;
    using Out = ConvDgradLoader::Out;
    static const char* name() { return "dgrad"; }
};
// Which layers take them (CNUDA_HCONV: 0 none, 1 every eligible layer, 2 = default: the 32-row GEMMs).  Measured in
// round 4, A/B on one box, whole benched step: 32-row GEMMs only 84.0-84.3 ms, none 84.5-84.7, every eligible layer
// 85.0-85.3 -- the 64- and 128-row tiles are matrix-pipe-bound either way (PMC: 0.72-0.81 of the cycles the chip
// clocks under them, profiles/r4_pmc_conv.md) and the wave-specialised im2col kernels keep the edge there; the 27-row
// DCN offset convolutions were bound by the texture-address unit and gain 9-14 % per launch.
int g_hconv_level = getenv("CNUDA_HCONV") ? atoi(getenv("CNUDA_HCONV")) : 2;
int g_hconv_min_tiles = 128;         // cnuda_conv_set_halo_policy (tests)
int hconv_level() { return g_hconv_level; }
// kc: channels of the gathered tensor (x for the forward, grad_y for the input gradient); bm: the GEMM's row tile.
// Small problems (under 128 pixel tiles) keep the im2col kernels: nothing to gain, and their results stay bit for bit
// what the golden step fixtures were calibrated on.
// Any row width that is a multiple of 8 (round 6; rounds 4-5: 16 / 32 / 64 / 128 only): the tile is TR rows x TW columns,
// TW = the largest power of two that divides W (hconv.cuh) -- 5 column tiles on the 160- / 80- / 40-wide maps of a 640 x 640
// input.  A map whose height is no multiple of the tile's keeps the halo kernels while the overhang stays under a quarter.
bool hconv_ok(const ConvGeom& g, int kc, int bm) {
    const int lv = hconv_level();
    if (!(lv != 0 && (lv == 1 || bm == 32) && matrix_mode() == 0 && g.kh == 3 && g.kw == 3 && g.sh == 1 && g.sw == 1 &&
          g.ph == 1 && g.pw == 1 && kc % 16 == 0 && halo_tile_width(g.W) != 0 &&
          (long long)g.B * g.H * g.W >= (long long)g_hconv_min_tiles * IG_BN &&
          (size_t)g.B * kc * g.H * g.W * sizeof(float) < IG_BUF_OOB))
        return false;
    const HaloGeom hg = make_halo_geom(kc, g.H, g.W, IG_BN);
    return hg.cells <= HC_MAXCELLS * IG_THREADS && 4 * (hg.tiles_y * hg.TR - g.H) <= g.H;
}
template <int BM, int BN, class Ad>
bool hconv_launch_one(const typename Ad::Params& p, const float* src, const float* A, int Mp, int Kp, int M, long long N,
                      int m_tiles, const HaloGeom& hg, hipStream_t st) {
    const size_t lds = hconv_lds_bytes(hg, BM);
    if (!raise_dynamic_lds(reinterpret_cast<const void*>(&hconv_kernel<BM, BN, Ad>), lds)) return false;
    const int n_tiles = (int)(N / ((long long)hg.H * hg.W)) * hg.tiles_y * hg.tiles_x;
    CNUDA_LAUNCH((hconv_kernel<BM, BN, Ad>), dim3(n_tiles * m_tiles), dim3(IG_THREADS), lds, st, p, src, A, Mp, Kp, M, N,
                 n_tiles, m_tiles, hg);
    return true;
}
template <class Ad>
int launch_hconv(int bm, const typename Ad::Params& p, const float* src, int kc, const ConvGeom& g, const float* A, int Mp,
                 int Kp, int M, long long N, hipStream_t st, const char* who) {
    CNUDA_REQUIRE(N < (1ll << 31) - 256, "%s: more than 2^31 pixels per call", who);
    const int m_tiles = Mp / bm;
    // 256-pixel tiles for the narrow GEMMs when that still leaves two rounds of workgroups (and the halo fits: tile
    // columns >= 32, whole tile rows, at most HC_MAXCELLS cells per thread)
    const HaloGeom hg2 = make_halo_geom(kc, g.H, g.W, 256);
    const bool wide = bm <= 64 && hg2.TW >= 32 && g.H % hg2.TR == 0 && hg2.cells <= HC_MAXCELLS * IG_THREADS &&
                      (N / 256) * m_tiles >= 1024;
    const HaloGeom hg = wide ? hg2 : make_halo_geom(kc, g.H, g.W, 128);
    ProfScope prof(st);
    prof.name("hconv_kernel<%d, %d, %s>", bm, wide ? 256 : 128, Ad::name());
    bool ok;
    if (bm == 128) ok = hconv_launch_one<128, 128, Ad>(p, src, A, Mp, Kp, M, N, m_tiles, hg, st);
    else if (bm == 64 && wide) ok = hconv_launch_one<64, 256, Ad>(p, src, A, Mp, Kp, M, N, m_tiles, hg, st);
    else if (bm == 64) ok = hconv_launch_one<64, 128, Ad>(p, src, A, Mp, Kp, M, N, m_tiles, hg, st);
    else if (wide) ok = hconv_launch_one<32, 256, Ad>(p, src, A, Mp, Kp, M, N, m_tiles, hg, st);
    else ok = hconv_launch_one<32, 128, Ad>(p, src, A, Mp, Kp, M, N, m_tiles, hg, st);
    if (!ok) return CNUDA_ERR_INVALID_ARGUMENT;
    return check_launch(who);
}

// ---------------------------------------------------------------------------
// Input gradient of a 3x3 / stride 2 / padding 1 convolution with C = 16 input channels (DLA-34's level1, 16 -> 32 at
// 512 x 512 -> 256 x 256: backends/dla.py:233-241): 8.4 M output pixels per image batch of 32, 16 channels each, K of at
// most 4 taps x 32.  As four parity-class GEMMs on 32-row MFMA tiles half the rows are padding, a tile has 2..8 K chunks
// and its stride-2 epilogue stores 4 bytes per lane: 943 us at 20 TFLOP/s (round 5, profiles/r5_early_layers.txt) for
// 0.8 GB of compulsory traffic.  Here one thread owns one cell (qy, qx) of the grad_y grid and produces the 2 x 2 block of
// grad_x pixels it feeds into first -- rows 2 qy, 2 qy + 1, columns 2 qx, 2 qx + 1 -- for all 16 channels: every one of the
// nine taps lands in exactly one of the four pixels (row 2 qy takes kernel row 1 of grad_y row qy; row 2 qy + 1 takes
// kernel row 0 of grad_y row qy + 1 and kernel row 2 of row qy; columns alike), so the block is 9 x Co x 16 multiply-adds
// over the four grad_y cells (qy + {0, 1}, qx + {0, 1}), all coalesced along qx.  The weights of one output channel
// (9 taps x 16 channels, repacked [o][tap][c]) are wave-uniform: scalar loads, one SGPR operand per multiply-add.  The
// block's two rows are stored as 8-byte pairs.  Vector ALU at the f32 rate the MFMA has, no padding, no K chunks.
// ---------------------------------------------------------------------------
struct DgradS2Params {
    const float *gy, *wt;          // wt: [Co][9][16]
    float* gx;
    const float *add, *add2;       // nullable addends shaped like gx (may alias it: read per element before the store)
    int B, Co, H, W, Ho, Wo, QH, QW;
};
__global__ __launch_bounds__(256) void dgrad_s2_c16_pack_kernel(const float* __restrict__ w, float* __restrict__ wt, int Co) {
    const int i = blockIdx.x * 256 + threadIdx.x;        // wt[(o * 9 + tap) * 16 + c] = w[(o * 16 + c) * 9 + tap]
    if (i < Co * 144) {
        const int c = i & 15, t = (i >> 4) % 9, o = i / 144;
        wt[i] = w[(o * 16 + c) * 9 + t];
    }
}
__global__ __launch_bounds__(256) void dgrad_s2_c16_kernel(DgradS2Params p) {
    const int qx = blockIdx.x * 64 + (threadIdx.x & 63);
    const int qy = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int b = blockIdx.z;
    if (qy >= p.QH) return;                                  // (wave-uniform)
    const int HoWo = p.Ho * p.Wo, HW = p.H * p.W;
    const bool x0 = qx < p.Wo, x1 = qx + 1 < p.Wo, y0 = qy < p.Ho, y1 = qy + 1 < p.Ho;
    const float* g00 = p.gy + (size_t)b * p.Co * HoWo + (size_t)(y0 ? qy : 0) * p.Wo + (x0 ? qx : 0);
    const int d01 = x1 ? 1 : 0, d10 = y1 ? p.Wo : 0;        // clamped neighbours; their values are zeroed below
    float acc[4][16];                                        // [2 py + px][c]
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int c = 0; c < 16; ++c) acc[q][c] = 0.0f;
    for (int o = 0; o < p.Co; ++o) {
        const float* gp = g00 + (size_t)o * HoWo;
        float a = gp[0], bb = gp[d01], cc = gp[d10], dd = gp[d10 + d01];     // grad_y (qy, qx), (qy, qx+1), (qy+1, qx), (qy+1, qx+1)
        if (!(x0 && y0)) a = 0.0f;
        if (!(x1 && y0)) bb = 0.0f;
        if (!(x0 && y1)) cc = 0.0f;
        if (!(x1 && y1)) dd = 0.0f;
        const float* wo = p.wt + (size_t)o * 144;            // wave-uniform: scalar loads
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            // taps (r, s): pixel (py, px) of the block takes r = 1 | {0, 2}, s = 1 | {0, 2}
            acc[0][c] = fmaf(wo[4 * 16 + c], a, acc[0][c]);                                   // (1,1) <- (qy, qx)
            acc[1][c] = fmaf(wo[3 * 16 + c], bb, fmaf(wo[5 * 16 + c], a, acc[1][c]));         // (1,0) <- (qy, qx+1); (1,2) <- (qy, qx)
            acc[2][c] = fmaf(wo[1 * 16 + c], cc, fmaf(wo[7 * 16 + c], a, acc[2][c]));         // (0,1) <- (qy+1, qx); (2,1) <- (qy, qx)
            acc[3][c] = fmaf(wo[0 * 16 + c], dd, fmaf(wo[2 * 16 + c], cc,                     // (0,0) <- (qy+1, qx+1); (0,2) <- (qy+1, qx)
                             fmaf(wo[6 * 16 + c], bb, fmaf(wo[8 * 16 + c], a, acc[3][c]))));  // (2,0) <- (qy, qx+1); (2,2) <- (qy, qx)
        }
    }
    const int iy = 2 * qy, ix = 2 * qx;
    if (ix >= p.W) return;
    const bool two = ix + 1 < p.W, pair_ok = two && (p.W & 1) == 0;      // 8-byte stores need even row starts
#pragma unroll
    for (int py = 0; py < 2; ++py) {
        if (iy + py >= p.H) break;
        const size_t o0 = (size_t)b * 16 * HW + (size_t)(iy + py) * p.W + ix;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const size_t o = o0 + (size_t)c * HW;
            float v0 = acc[2 * py][c], v1 = acc[2 * py + 1][c];
            if (pair_ok) {
                if (p.add) { const float2 t = *reinterpret_cast<const float2*>(p.add + o); v0 += t.x; v1 += t.y; }
                if (p.add2) { const float2 t = *reinterpret_cast<const float2*>(p.add2 + o); v0 += t.x; v1 += t.y; }
                *reinterpret_cast<float2*>(p.gx + o) = make_float2(v0, v1);
            } else {
                if (p.add) { v0 += p.add[o]; if (two) v1 += p.add[o + 1]; }
                if (p.add2) { v0 += p.add2[o]; if (two) v1 += p.add2[o + 1]; }
                p.gx[o] = v0;
                if (two) p.gx[o + 1] = v1;
            }
        }
    }
}

template <class Loader> struct SplitKLoader : std::false_type {};
template <> struct SplitKLoader<ConvFwdBufLoader> : std::true_type {};
template <> struct SplitKLoader<ConvFwdBufSigLoader> : std::true_type {};
template <> struct SplitKLoader<ConvDgradBufLoader> : std::true_type {};
template <> struct SplitKLoader<ConvDgradClassBufLoader> : std::true_type {};

template <class Loader>
int launch_fwd(int bm, const typename Loader::Params& p, const float* A, int Mp, int Kp, int M, long long N,
               hipStream_t st, const char* who, const SplitK& sk = SplitK(), float* slab = nullptr) {
    CNUDA_REQUIRE(N < (1ll << 31) - IG_BN, "%s: more than 2^31 pixels per call", who);
    const int n_tiles = ceil_div(N, IG_BN), m_tiles = Mp / bm;
    const dim3 grid(n_tiles * m_tiles), block(IG_THREADS);
    ProfScope prof(st);
    if constexpr (SplitKLoader<Loader>::value) {
        if (sk.on()) {
            CNUDA_REQUIRE(slab && bm == sk.bm && matrix_mode() == 0, "%s: split-K plan without its slabs", who);
            const bool ws = wave_specialised() && bm >= 64;
            prof.name(ws ? "igemm_fwd_ws_splitk_kernel<%d, %s> x %d + reduce" : "igemm_fwd_splitk_kernel<%d, %s> x %d + reduce",
                      bm, Loader::name(), sk.z);
            const dim3 gridz(n_tiles * m_tiles, sk.z);
            if (bm == 128 && ws)
                CNUDA_LAUNCH((igemm_fwd_ws_splitk_kernel<128, Loader>), gridz, dim3(2 * IG_THREADS), 0, st, p, A, Mp, Kp, M, N, n_tiles, m_tiles, slab, sk.split_k);
            else if (bm == 64 && ws)
                CNUDA_LAUNCH((igemm_fwd_ws_splitk_kernel<64, Loader>), gridz, dim3(2 * IG_THREADS), 0, st, p, A, Mp, Kp, M, N, n_tiles, m_tiles, slab, sk.split_k);
            else if (bm == 128)
                CNUDA_LAUNCH((igemm_fwd_splitk_kernel<128, Loader>), gridz, block, 0, st, p, A, Mp, Kp, M, N, n_tiles, m_tiles, slab, sk.split_k);
            else if (bm == 64)
                CNUDA_LAUNCH((igemm_fwd_splitk_kernel<64, Loader>), gridz, block, 0, st, p, A, Mp, Kp, M, N, n_tiles, m_tiles, slab, sk.split_k);
            else
                CNUDA_LAUNCH((igemm_fwd_splitk_kernel<32, Loader>), gridz, block, 0, st, p, A, Mp, Kp, M, N, n_tiles, m_tiles, slab, sk.split_k);
            CNUDA_LAUNCH((splitk_reduce_kernel<Loader>), dim3(ceil_div(N, 256), ceil_div(M, SK_ROWS)), dim3(256), 0, st, p, slab, sk.z,
                         Mp, (long long)n_tiles * IG_BN, M, N);
            return check_launch(who);
        }
    }
    CNUDA_REQUIRE(!sk.on(), "%s: split-K plan for a loader without split-K kernels", who);
    if constexpr (std::is_same<Loader, ConvFwdBufLoader>::value || std::is_same<Loader, ConvFwdBufStatsLoader>::value ||
                  std::is_same<Loader, ConvFwdLoader<true>>::value || std::is_same<Loader, ConvFwdBufQuadLoader>::value) {
        // few chunks, several row tiles (the DCN column-gradient GEMM: a 1x1 forward with K = 64 and 9*C rows)
        static const bool shortk = !(getenv("CNUDA_SHORTK") && getenv("CNUDA_SHORTK")[0] == '0');
        if (shortk && matrix_mode() == 0 && bm == 128 && Kp <= 64 && m_tiles >= 2) {
            prof.name("igemm_fwd_shortk_kernel<128, %s, 64>", Loader::name());
            CNUDA_LAUNCH((igemm_fwd_shortk_kernel<128, Loader, 64>), dim3(n_tiles), block, 0, st, p, A, Mp, Kp, M, N,
                               n_tiles, m_tiles);
            return check_launch(who);
        }
    }
    const bool ws = matrix_mode() == 0 && wave_specialised() && bm >= 64;
    prof.name(matrix_mode() == 1 ? "igemm_fwd_kernel<%d, %s> [split bf16 x3]"
                                 : (ws ? "igemm_fwd_ws_kernel<%d, %s>" : "igemm_fwd_kernel<%d, %s>"), bm, Loader::name());
    if (matrix_mode() == 1) {
        // the caller's A buffer has ig_a_bytes() of room: split image behind the f32 matrix
        float* A3 = const_cast<float*>(A) + (size_t)Kp * Mp;
        CNUDA_LAUNCH(split_a_kernel, dim3(stream_grid((long long)(Kp / 16) * 2 * Mp, 256)), dim3(256), 0, st, A,
                           reinterpret_cast<u32x4*>(A3), Kp, Mp);
        A = A3;
        if (bm == 128)
            CNUDA_LAUNCH((igemm_fwd_kernel<128, Loader, true>), grid, block, 0, st, p, A, Mp, Kp, M, N, n_tiles, m_tiles);
        else if (bm == 64)
            CNUDA_LAUNCH((igemm_fwd_kernel<64, Loader, true>), grid, block, 0, st, p, A, Mp, Kp, M, N, n_tiles, m_tiles);
        else
            CNUDA_LAUNCH((igemm_fwd_kernel<32, Loader, true>), grid, block, 0, st, p, A, Mp, Kp, M, N, n_tiles, m_tiles);
        return check_launch(who);
    }
    // wave-specialised variant (igemm.cuh): +4-8 % on the 64- and 128-row tiles of the 128..512-channel layers,
    // neutral on the 64-channel ones, -2-4 % on the 32-row tile, which therefore keeps the 4-wave kernel
    if (wave_specialised() && bm >= 64) {
        const dim3 block2(2 * IG_THREADS);
        if (bm == 128)
            CNUDA_LAUNCH((igemm_fwd_ws_kernel<128, Loader>), grid, block2, 0, st, p, A, Mp, Kp, M, N, n_tiles, m_tiles);
        else
            CNUDA_LAUNCH((igemm_fwd_ws_kernel<64, Loader>), grid, block2, 0, st, p, A, Mp, Kp, M, N, n_tiles, m_tiles);
        return check_launch(who);
    }
    if (bm == 128)
        CNUDA_LAUNCH((igemm_fwd_kernel<128, Loader>), grid, block, 0, st, p, A, Mp, Kp, M, N, n_tiles, m_tiles);
    else if (bm == 64)
        CNUDA_LAUNCH((igemm_fwd_kernel<64, Loader>), grid, block, 0, st, p, A, Mp, Kp, M, N, n_tiles, m_tiles);
    else
        CNUDA_LAUNCH((igemm_fwd_kernel<32, Loader>), grid, block, 0, st, p, A, Mp, Kp, M, N, n_tiles, m_tiles);
    return check_launch(who);
}

// the tile variants of the buffer-addressed weight-gradient GEMM (ConvWBufLoader; ConvWCatLoader)
template <class Loader>
void launch_wgrad_buf(const ConvPlan& q, const typename Loader::Params& p, dim3 grid, float* slabs, float* bsl, hipStream_t st) {
    const dim3 blk(IG_THREADS);
    const dim3 blk2(2 * IG_THREADS);
    if (q.wbm == 128 && q.wbj == 128 && wave_specialised())
        CNUDA_LAUNCH((igemm_wgrad_ws_kernel<Loader, 128, 128>), grid, blk2, 0, st, p, slabs, q.Mpw,
                           q.Jp, q.Nf, q.pix_per_split, bsl);
    else if (q.wbm == 128 && q.wbj == 128)
        CNUDA_LAUNCH((igemm_wgrad_kernel<Loader, 128, 128>), grid, blk, 0, st, p, slabs, q.Mpw,
                           q.Jp, q.Nf, q.pix_per_split, bsl);
    else if (q.wbm == 128 && wave_specialised())
        CNUDA_LAUNCH((igemm_wgrad_ws_kernel<Loader, 128, 64>), grid, blk2, 0, st, p, slabs, q.Mpw,
                           q.Jp, q.Nf, q.pix_per_split, bsl);
    else if (q.wbm == 128)
        CNUDA_LAUNCH((igemm_wgrad_kernel<Loader, 128, 64>), grid, blk, 0, st, p, slabs, q.Mpw,
                           q.Jp, q.Nf, q.pix_per_split, bsl);
    else if (wave_specialised() && q.wbm == 64 && q.wbj == 128)
        CNUDA_LAUNCH((igemm_wgrad_ws_kernel<Loader, 64, 128>), grid, blk2, 0, st, p, slabs, q.Mpw,
                           q.Jp, q.Nf, q.pix_per_split, bsl);
    else if (wave_specialised() && q.wbm == 64)
        CNUDA_LAUNCH((igemm_wgrad_ws_kernel<Loader, 64, 64>), grid, blk2, 0, st, p, slabs, q.Mpw,
                           q.Jp, q.Nf, q.pix_per_split, bsl);
    else if (q.wbm == 64 && q.wbj == 128)
        CNUDA_LAUNCH((igemm_wgrad_kernel<Loader, 64, 128>), grid, blk, 0, st, p, slabs, q.Mpw,
                           q.Jp, q.Nf, q.pix_per_split, bsl);
    else if (q.wbm == 64)
        CNUDA_LAUNCH((igemm_wgrad_kernel<Loader, 64, 64>), grid, blk, 0, st, p, slabs, q.Mpw, q.Jp,
                           q.Nf, q.pix_per_split, bsl);
    else
        CNUDA_LAUNCH((igemm_wgrad_kernel<Loader, 32, 128>), grid, blk, 0, st, p, slabs, q.Mpw,
                           q.Jp, q.Nf, q.pix_per_split, bsl);
}

}  // namespace
}  // namespace cnuda

using namespace cnuda;


extern "C" int cnuda_conv_set_splitk_policy(int max_tiles) {
    const int prev = g_splitk_max_tiles;
    g_splitk_max_tiles = max_tiles < 0 ? 128 : max_tiles;
    return prev;
}

extern "C" int cnuda_conv_set_halo_policy(int level, int min_tiles) {
    if (level >= 0) g_hconv_level = level;
    if (min_tiles >= 1) g_hconv_min_tiles = min_tiles;
    return g_hconv_level | (g_hconv_min_tiles << 8);
}

extern "C" size_t cnuda_conv2d_workspace_bytes(int B, int C, int H, int W, int Cout, int kh, int kw, int sh, int sw,
                                               int ph, int pw) {
    ConvGeom g;
    if (fill_geom(g, B, C, H, W, Cout, kh, kw, sh, sw, ph, pw, "cnuda_conv2d_workspace_bytes")) return 0;
    const ConvPlan q = make_plan(g);
    size_t m = q.fwd_bytes;
    if (q.dgrad_bytes > m) m = q.dgrad_bytes;
    if (q.wgrad_bytes > m) m = q.wgrad_bytes;
    if (smallc_supported(C, Cout, kh, kw, sh, sw) || smallc_supported(Cout, C, kh, kw, sh, sw)) {
        const size_t sm = smallc_workspace_bytes(B, C, H, W, Cout, kh, kw, sh, ph, pw) + carve_bytes((size_t)Cout * B, 4);
        if (sm > m) m = sm;
    }
    return m;
}

extern "C" int cnuda_conv2d_forward(const float* x, const float* weight, const float* bias, float* y, int B, int C,
                                    int H, int W, int Cout, int kh, int kw, int sh, int sw, int ph, int pw, float act_slope,
                                    void* workspace, size_t workspace_bytes, cnuda_stream_t stream) {
    return cnuda_conv2d_forward_res(x, weight, bias, nullptr, y, B, C, H, W, Cout, kh, kw, sh, sw, ph, pw, act_slope,
                                    workspace, workspace_bytes, stream);
}

// Which pixel blocks a forward call of this geometry can leave BatchNorm statistics for: 0 (none: the halo-tile kernels,
// the scalar epilogues of planes / rows that are no multiple of four pixels), else the pixels per block (64 for the 64- /
// 128-row tiles, 32 for the 32-row tile) of the flattened (image, pixel) axis and *blocks_per_image = 0 -- or, for the
// LDS-tile kernels of the 3- / 16-channel layers, 64 and *blocks_per_image > 0: a block is one 64-column piece of an
// output row and never straddles images.  *rows = rows per block of the stats array.
extern "C" int cnuda_conv2d_stats_block(int B, int C, int H, int W, int Cout, int kh, int kw, int sh, int sw, int ph,
                                        int pw, int* rows, int* blocks_per_image) {
    ConvGeom g;
    if (fill_geom(g, B, C, H, W, Cout, kh, kw, sh, sw, ph, pw, "cnuda_conv2d_stats_block")) return 0;
    if (blocks_per_image) *blocks_per_image = 0;
    if (smallc_supported(C, Cout, kh, kw, sh, sw)) return smallc_stats_blocks(B, C, H, W, Cout, kh, kw, sh, ph, pw, blocks_per_image, rows);
    if (((g.Ho * g.Wo) & 3) != 0) return 0;
    const ConvPlan q = make_plan(g);
    if (q.skf.on() || hconv_ok(g, C, q.bmf)) return 0;
    // (the statistics kernels exist for the buffer-addressed loader: every DLA-34 / ResNet layer behind the stem)
    if (!(C % IG_BK == 0 && buffer_addressing() && q.T <= 32 && (size_t)B * C * H * W * sizeof(float) < IG_BUF_OOB) || matrix_mode() != 0)
        return 0;
    if (rows) *rows = q.Mpf;
    return q.bmf == 32 ? 32 : 64;
}

extern "C" int cnuda_conv2d_forward_res(const float* x, const float* weight, const float* bias, const float* residual,
                                        float* y, int B, int C, int H, int W, int Cout, int kh, int kw, int sh, int sw,
                                        int ph, int pw, float act_slope, void* workspace, size_t workspace_bytes,
                                        cnuda_stream_t stream) {
    return cnuda_conv2d_forward_stats(x, weight, bias, residual, y, nullptr, B, C, H, W, Cout, kh, kw, sh, sw, ph, pw, act_slope,
                                      workspace, workspace_bytes, stream);
}

extern "C" int cnuda_conv2d_forward_stats(const float* x, const float* weight, const float* bias, const float* residual,
                                          float* y, float* stats, int B, int C, int H, int W, int Cout, int kh, int kw,
                                          int sh, int sw, int ph, int pw, float act_slope, void* workspace,
                                          size_t workspace_bytes, cnuda_stream_t stream) {
    CNUDA_REQUIRE(x && weight && y, "cnuda_conv2d_forward: null pointer");
    ConvGeom g;
    if (int rc = fill_geom(g, B, C, H, W, Cout, kh, kw, sh, sw, ph, pw, "cnuda_conv2d_forward")) return rc;
    if (stats) {
        int rows = 0;
        CNUDA_REQUIRE(!residual && act_slope < 0.0f &&
                          cnuda_conv2d_stats_block(B, C, H, W, Cout, kh, kw, sh, sw, ph, pw, &rows, nullptr) != 0,
                      "cnuda_conv2d_forward_stats: no statistics for this call (cnuda_conv2d_stats_block says which)");
    }
    if (!residual && smallc_supported(C, Cout, kh, kw, sh, sw))
        return smallc_forward(x, weight, bias, y, B, C, H, W, Cout, kh, kw, sh, ph, pw, act_slope, 0, workspace,
                              workspace_bytes, (hipStream_t)stream, stats);
    const ConvPlan q = make_plan(g);
    CNUDA_REQUIRE(workspace && workspace_bytes >= q.fwd_bytes, "cnuda_conv2d_forward: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    Carver cv(workspace, workspace_bytes);
    ConvFwdParams p{g, x, bias, y, act_slope, residual};
    if (!q.skf.on() && hconv_ok(g, C, q.bmf)) {      // (K = 9 C is already a multiple of the chunk: the same packed size, another K order)
        const float* Ah = launch_pack(weight, reinterpret_cast<float*>(cv.take<char>(ig_a_bytes(q.Kpf, q.Mpf))),
                                      ig_a_bytes(q.Kpf, q.Mpf), Cout, C, q.T, PACK_HALO_FWD, q.Kpf, q.Mpf, 0, st);
        return launch_hconv<HconvFwd>(q.bmf, p, x, C, g, Ah, q.Mpf, q.Kpf, Cout, q.Nf, st, "cnuda_conv2d_forward");
    }
    const float* A = launch_pack(weight, reinterpret_cast<float*>(cv.take<char>(ig_a_bytes(q.Kpf, q.Mpf))),
                                 ig_a_bytes(q.Kpf, q.Mpf), Cout, C, q.T, PACK_FWD, q.Kpf, q.Mpf, 0, st);
    if (stats) {            // (cnuda_conv2d_stats_block vouched for the buffer-addressed loader)
        ConvFwdStatsParams ps;
        static_cast<ConvFwdParams&>(ps) = p;
        ps.stats = stats;
        ps.stats_mp = q.Mpf;
        return launch_fwd<ConvFwdBufStatsLoader>(q.bmf, ps, A, q.Mpf, q.Kpf, Cout, q.Nf, st, "cnuda_conv2d_forward_stats");
    }
    if (C % IG_BK == 0 && buffer_addressing() && q.T <= 32 && (size_t)B * C * H * W * sizeof(float) < IG_BUF_OOB) {
        float* slab = q.skf.on() ? reinterpret_cast<float*>(cv.take<char>(splitk_slab_bytes(q.skf, Cout, q.Nf))) : nullptr;
        return launch_fwd<ConvFwdBufLoader>(q.bmf, p, A, q.Mpf, q.Kpf, Cout, q.Nf, st, "cnuda_conv2d_forward", q.skf, slab);
    }
    CNUDA_REQUIRE(!q.skf.on(), "cnuda_conv2d_forward: split-K plan off the buffer-addressed path");
    if (C % IG_BK == 0)
        return launch_fwd<ConvFwdLoader<true>>(q.bmf, p, A, q.Mpf, q.Kpf, Cout, q.Nf, st, "cnuda_conv2d_forward");
    return launch_fwd<ConvFwdLoader<false>>(q.bmf, p, A, q.Mpf, q.Kpf, Cout, q.Nf, st, "cnuda_conv2d_forward");
}

// ---- the 1x1 convolution over a channel concatenation (ConvCat; DLA's Root) ------------------------------------------------
namespace {
int fill_cat(ConvCat& cat, const float* const* xs, const int* cs, int n, const char* who) {
    CNUDA_REQUIRE(xs && cs && n >= 2 && n <= CAT_MAX, "%s: 2 .. %d sources", who, CAT_MAX);
    int k = 0;
    for (int i = 0; i < CAT_MAX; ++i) {
        cat.x[i] = i < n ? xs[i] : nullptr;
        cat.c[i] = i < n ? cs[i] : 0;
        cat.k0[i] = k;
        if (i < n) {
            CNUDA_REQUIRE(xs[i] && cs[i] > 0 && cs[i] % 64 == 0, "%s: source %d: null or not a multiple of 64 channels", who, i);
            k += cs[i];
        }
    }
    cat.k0[CAT_MAX] = k;
    for (int i = n; i < CAT_MAX; ++i) cat.k0[i] = k;
    cat.n = n;
    return 0;
}
int cat_channels(const int* cs, int n) {
    int k = 0;
    for (int i = 0; i < n; ++i) k += cs[i];
    return k;
}
}  // namespace

extern "C" int cnuda_conv2d_cat_supported(const int* cs, int n, int B, int H, int W, int Cout) {
    if (!cs || n < 2 || n > CAT_MAX || matrix_mode() != 0) return 0;
    for (int i = 0; i < n; ++i)
        if (cs[i] <= 0 || cs[i] % 64 != 0) return 0;
    const int C = cat_channels(cs, n);
    ConvGeom g;
    if (fill_geom(g, B, C, H, W, Cout, 1, 1, 1, 1, 0, 0, "cnuda_conv2d_cat_supported")) return 0;
    int rows = 0;
    if (!cnuda_conv2d_stats_block(B, C, H, W, Cout, 1, 1, 1, 1, 0, 0, &rows, nullptr)) return 0;   // (buffer-addressed, no K split, H*W % 4 == 0)
    const ConvPlan q = make_plan(g);
    return !q.skd.on() && !q.hw && !q.hw_s2 && wgrad_buffer_ok(g) && (size_t)B * Cout * H * W * sizeof(float) < IG_BUF_OOB;
}

extern "C" int cnuda_conv2d_cat_forward(const float* const* xs, const int* cs, int n, const float* weight, const float* bias,
                                        float* y, float* stats, float act_slope, int B, int H, int W, int Cout,
                                        void* workspace, size_t workspace_bytes, cnuda_stream_t stream) {
    CNUDA_REQUIRE(weight && y, "cnuda_conv2d_cat_forward: null pointer");
    CNUDA_REQUIRE(!stats || act_slope < 0.0f, "cnuda_conv2d_cat_forward: statistics are those of the plain convolution");
    CNUDA_REQUIRE(cnuda_conv2d_cat_supported(cs, n, B, H, W, Cout), "cnuda_conv2d_cat_forward: unsupported (cnuda_conv2d_cat_supported)");
    ConvFwdCatParams p;
    if (int rc = fill_cat(p.cat, xs, cs, n, "cnuda_conv2d_cat_forward")) return rc;
    const int C = p.cat.k0[CAT_MAX];
    ConvGeom g;
    if (int rc = fill_geom(g, B, C, H, W, Cout, 1, 1, 1, 1, 0, 0, "cnuda_conv2d_cat_forward")) return rc;
    const ConvPlan q = make_plan(g);
    CNUDA_REQUIRE(workspace && workspace_bytes >= q.fwd_bytes, "cnuda_conv2d_cat_forward: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    Carver cv(workspace, workspace_bytes);
    static_cast<ConvFwdParams&>(p) = ConvFwdParams{g, nullptr, bias, y, act_slope, nullptr};
    p.stats = stats;
    p.stats_mp = q.Mpf;
    const float* A = launch_pack(weight, reinterpret_cast<float*>(cv.take<char>(ig_a_bytes(q.Kpf, q.Mpf))),
                                 ig_a_bytes(q.Kpf, q.Mpf), Cout, C, q.T, PACK_FWD, q.Kpf, q.Mpf, 0, st);
    return launch_fwd<ConvFwdCatLoader>(q.bmf, p, A, q.Mpf, q.Kpf, Cout, q.Nf, st, "cnuda_conv2d_cat_forward");
}

// grad_xs[i] = input gradient of source i (+ adds[i] + add2s[i], nullable, shaped like it; either may BE grad_xs[i]); a null
// grad_xs[i] is not allowed (a source without a gradient gets a scratch tensor from the caller).
extern "C" int cnuda_conv2d_cat_backward_data(const float* grad_y, const float* weight, float* const* grad_xs,
                                              const float* const* adds, const float* const* add2s, const int* cs, int n,
                                              int B, int H, int W, int Cout, void* workspace, size_t workspace_bytes,
                                              cnuda_stream_t stream) {
    CNUDA_REQUIRE(grad_y && weight && grad_xs, "cnuda_conv2d_cat_backward_data: null pointer");
    CNUDA_REQUIRE(cnuda_conv2d_cat_supported(cs, n, B, H, W, Cout), "cnuda_conv2d_cat_backward_data: unsupported (cnuda_conv2d_cat_supported)");
    const int C = cat_channels(cs, n);
    ConvGeom g;
    if (int rc = fill_geom(g, B, C, H, W, Cout, 1, 1, 1, 1, 0, 0, "cnuda_conv2d_cat_backward_data")) return rc;
    const ConvPlan q = make_plan(g);
    CNUDA_REQUIRE(workspace && workspace_bytes >= q.dgrad_bytes, "cnuda_conv2d_cat_backward_data: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    Carver cv(workspace, workspace_bytes);
    float* Aws = reinterpret_cast<float*>(cv.take<char>(ig_a_bytes(q.Kpd, q.Mpd)));
    const float* A = launch_pack(weight, Aws, ig_a_bytes(q.Kpd, q.Mpd), Cout, C, q.T, PACK_DGRAD, q.Kpd, q.Mpd,
                                 round_up(Cout, IG_BK), st);
    ConvDgradCatParams p;
    static_cast<ConvDgradParams&>(p) = ConvDgradParams{g, grad_y, nullptr, round_up(Cout, IG_BK), nullptr, nullptr};
    for (int i = 0; i < CAT_MAX; ++i) {
        p.gxs[i] = i < n ? grad_xs[i] : nullptr;
        p.adds[i] = (i < n && adds) ? adds[i] : nullptr;
        p.add2s[i] = (i < n && add2s) ? add2s[i] : nullptr;
        if (!p.adds[i] && p.add2s[i]) { p.adds[i] = p.add2s[i]; p.add2s[i] = nullptr; }
        p.c[i] = i < n ? cs[i] : 0;
        CNUDA_REQUIRE(i >= n || p.gxs[i], "cnuda_conv2d_cat_backward_data: source %d without a gradient tensor", i);
    }
    p.n = n;
    return launch_fwd<ConvDgradCatLoader>(q.bmd, p, A, q.Mpd, q.Kpd, C, q.Nd, st, "cnuda_conv2d_cat_backward_data");
}

extern "C" int cnuda_conv2d_cat_backward_weight(const float* const* xs, const int* cs, int n, const float* grad_y,
                                                float* grad_weight, int B, int H, int W, int Cout, void* workspace,
                                                size_t workspace_bytes, cnuda_stream_t stream) {
    CNUDA_REQUIRE(grad_y && grad_weight, "cnuda_conv2d_cat_backward_weight: null pointer");
    CNUDA_REQUIRE(cnuda_conv2d_cat_supported(cs, n, B, H, W, Cout), "cnuda_conv2d_cat_backward_weight: unsupported (cnuda_conv2d_cat_supported)");
    ConvWCatParams p;
    if (int rc = fill_cat(p.cat, xs, cs, n, "cnuda_conv2d_cat_backward_weight")) return rc;
    const int C = p.cat.k0[CAT_MAX];
    ConvGeom g;
    if (int rc = fill_geom(g, B, C, H, W, Cout, 1, 1, 1, 1, 0, 0, "cnuda_conv2d_cat_backward_weight")) return rc;
    const ConvPlan q = make_plan(g);
    CNUDA_REQUIRE(workspace && workspace_bytes >= q.wgrad_bytes, "cnuda_conv2d_cat_backward_weight: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    Carver cv(workspace, workspace_bytes);
    float* slabs = cv.take<float>((size_t)q.Z * q.Mpw * q.Jp);
    static_cast<ConvWParams&>(p) = ConvWParams{g, nullptr, grad_y};
    {
        ProfScope prof(st);
        prof.name((wave_specialised() && q.wbm >= 64) ? "igemm_wgrad_ws_kernel<ConvWCatLoader, %d, %d>" : "igemm_wgrad_kernel<ConvWCatLoader, %d, %d>",
                  q.wbm, q.wbj);
        launch_wgrad_buf<ConvWCatLoader>(q, p, dim3(q.Jp / q.wbj, q.Mpw / q.wbm, q.Z), slabs, nullptr, st);
    }
    if (int rc = check_launch("cnuda_conv2d_cat_backward_weight")) return rc;
    launch_slab_reduce(slabs, grad_weight, q.Z, q.Mpw, q.Jp, Cout, C, q.T, st, nullptr, nullptr);
    return check_launch("cnuda_conv2d_cat_backward_weight(reduce)");
}

// y = conv(x) with the output channels interleaved in quads, y[b][Cout / 4][Ho * Wo][4] (ConvFwdBufQuadLoader): the layout of
// the DCN column gradient.  cnuda_conv2d_rowquads_supported: Cout % 4 == 0 and the geometry takes the buffer-addressed
// implicit GEMM without a K split; elsewhere the caller keeps the plain layout.  Workspace: cnuda_conv2d_workspace_bytes.
extern "C" int cnuda_conv2d_rowquads_supported(int B, int C, int H, int W, int Cout, int kh, int kw, int sh, int sw, int ph,
                                               int pw) {
    ConvGeom g;
    if (fill_geom(g, B, C, H, W, Cout, kh, kw, sh, sw, ph, pw, "cnuda_conv2d_rowquads_supported")) return 0;
    if (Cout % 4 != 0 || smallc_supported(C, Cout, kh, kw, sh, sw)) return 0;
    const ConvPlan q = make_plan(g);
    return !q.skf.on() && C % IG_BK == 0 && buffer_addressing() && q.T <= 32 &&
           (size_t)B * C * H * W * sizeof(float) < IG_BUF_OOB;
}

extern "C" int cnuda_conv2d_forward_rowquads(const float* x, const float* weight, float* y, int B, int C, int H, int W,
                                             int Cout, int kh, int kw, int sh, int sw, int ph, int pw, void* workspace,
                                             size_t workspace_bytes, cnuda_stream_t stream) {
    CNUDA_REQUIRE(x && weight && y, "cnuda_conv2d_forward_rowquads: null pointer");
    CNUDA_REQUIRE(cnuda_conv2d_rowquads_supported(B, C, H, W, Cout, kh, kw, sh, sw, ph, pw),
                  "cnuda_conv2d_forward_rowquads: geometry without a quad-interleaved epilogue (cnuda_conv2d_rowquads_supported)");
    ConvGeom g;
    if (int rc = fill_geom(g, B, C, H, W, Cout, kh, kw, sh, sw, ph, pw, "cnuda_conv2d_forward_rowquads")) return rc;
    const ConvPlan q = make_plan(g);
    CNUDA_REQUIRE(workspace && workspace_bytes >= q.fwd_bytes, "cnuda_conv2d_forward_rowquads: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    Carver cv(workspace, workspace_bytes);
    ConvFwdParams p{g, x, nullptr, y, -1.0f, nullptr};
    const float* A = launch_pack(weight, reinterpret_cast<float*>(cv.take<char>(ig_a_bytes(q.Kpf, q.Mpf))),
                                 ig_a_bytes(q.Kpf, q.Mpf), Cout, C, q.T, PACK_FWD, q.Kpf, q.Mpf, 0, st);
    return launch_fwd<ConvFwdBufQuadLoader>(q.bmf, p, A, q.Mpf, q.Kpf, Cout, q.Nf, st, "cnuda_conv2d_forward_rowquads");
}

// y = conv(x) + bias with a sigmoid on the output channels >= sig_from: the offset / mask convolution of a DCN layer when the
// deformable convolution reads both out of this tensor (cnuda_dcn_v2_forward_om).  cnuda_conv2d_rowsig_supported: the
// geometries with such an epilogue compiled (C % 16 == 0, at most 32 output channels, tensors below 2 GiB, f32 matrix mode).
extern "C" int cnuda_conv2d_rowsig_supported(int B, int C, int H, int W, int Cout, int kh, int kw, int sh, int sw, int ph,
                                             int pw) {
    ConvGeom g;
    if (fill_geom(g, B, C, H, W, Cout, kh, kw, sh, sw, ph, pw, "cnuda_conv2d_rowsig_supported")) return 0;
    return (Cout <= 32 && C % IG_BK == 0 && buffer_addressing() && kh * kw <= 32 && matrix_mode() == 0 &&
            (size_t)B * C * H * W * sizeof(float) < IG_BUF_OOB && (size_t)B * Cout * g.Ho * g.Wo * sizeof(float) < IG_BUF_OOB &&
            !smallc_supported(C, Cout, kh, kw, sh, sw)) ? 1 : 0;
}
extern "C" int cnuda_conv2d_forward_rowsig(const float* x, const float* weight, const float* bias, float* y, int sig_from,
                                           int B, int C, int H, int W, int Cout, int kh, int kw, int sh, int sw, int ph, int pw,
                                           void* workspace, size_t workspace_bytes, cnuda_stream_t stream) {
    CNUDA_REQUIRE(x && weight && y && sig_from >= 0, "cnuda_conv2d_forward_rowsig: bad arguments");
    CNUDA_REQUIRE(cnuda_conv2d_rowsig_supported(B, C, H, W, Cout, kh, kw, sh, sw, ph, pw),
                  "cnuda_conv2d_forward_rowsig: geometry without a row-sigmoid epilogue (cnuda_conv2d_rowsig_supported)");
    ConvGeom g;
    if (int rc = fill_geom(g, B, C, H, W, Cout, kh, kw, sh, sw, ph, pw, "cnuda_conv2d_forward_rowsig")) return rc;
    const ConvPlan q = make_plan(g);
    CNUDA_REQUIRE(workspace && workspace_bytes >= q.fwd_bytes, "cnuda_conv2d_forward_rowsig: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    Carver cv(workspace, workspace_bytes);
    ConvFwdSigParams p;
    static_cast<ConvFwdParams&>(p) = ConvFwdParams{g, x, bias, y, -1.0f, nullptr};
    p.sig_from = sig_from;
    if (!q.skf.on() && hconv_ok(g, C, q.bmf)) {
        const float* Ah = launch_pack(weight, reinterpret_cast<float*>(cv.take<char>(ig_a_bytes(q.Kpf, q.Mpf))),
                                      ig_a_bytes(q.Kpf, q.Mpf), Cout, C, q.T, PACK_HALO_FWD, q.Kpf, q.Mpf, 0, st);
        return launch_hconv<HconvFwdSig>(q.bmf, p, x, C, g, Ah, q.Mpf, q.Kpf, Cout, q.Nf, st, "cnuda_conv2d_forward_rowsig");
    }
    const float* A = launch_pack(weight, reinterpret_cast<float*>(cv.take<char>(ig_a_bytes(q.Kpf, q.Mpf))),
                                 ig_a_bytes(q.Kpf, q.Mpf), Cout, C, q.T, PACK_FWD, q.Kpf, q.Mpf, 0, st);
    float* slab = q.skf.on() ? reinterpret_cast<float*>(cv.take<char>(splitk_slab_bytes(q.skf, Cout, q.Nf))) : nullptr;
    return launch_fwd<ConvFwdBufSigLoader>(q.bmf, p, A, q.Mpf, q.Kpf, Cout, q.Nf, st, "cnuda_conv2d_forward_rowsig", q.skf, slab);
}

extern "C" int cnuda_conv2d_backward_data(const float* grad_y, const float* weight, float* grad_x, int B, int C, int H,
                                          int W, int Cout, int kh, int kw, int sh, int sw, int ph, int pw,
                                          void* workspace, size_t workspace_bytes, cnuda_stream_t stream) {
    return cnuda_conv2d_backward_data_add(grad_y, weight, nullptr, nullptr, grad_x, B, C, H, W, Cout, kh, kw, sh, sw, ph, pw, workspace,
                                          workspace_bytes, stream);
}

// ---- apply on load (round 5): convolutions whose INPUT is another convolution's output with a train-mode BatchNorm + ReLU
// still to be applied (cnuda_bn_train_forward_stats with y == nullptr left mean / invstd).  The kernel normalises while it
// stages x -- one pass over the activation less in each direction.  Only where the staging touches every input element once:
// cnuda_conv2d_norm_input_supported says for which geometries (today: the LDS-tile kernels' 3x3 / one-row-tile instances,
// i.e. DLA-34's level0 behind the stem).  Values are bit-identical to the materialised form (same two rounded operations).
extern "C" int cnuda_conv2d_norm_input_supported(int B, int C, int H, int W, int Cout, int kh, int kw, int sh, int sw, int ph,
                                                 int pw) {
    (void)B; (void)H; (void)W;
    return ph == (kh - 1) / 2 && pw == (kw - 1) / 2 && smallc_norm_supported(C, Cout, kh, kw, sh, sw) ? 1 : 0;
}
extern "C" int cnuda_conv2d_forward_norm_input(const float* x, const float* mean, const float* invstd, const float* gamma,
                                               const float* beta, int imgs_per_group, const float* weight, const float* bias,
                                               float* y, float* stats, int B, int C, int H, int W, int Cout, int kh, int kw,
                                               int sh, int sw, int ph, int pw, float act_slope, void* workspace,
                                               size_t workspace_bytes, cnuda_stream_t stream) {
    CNUDA_REQUIRE(x && mean && invstd && gamma && beta && weight && y, "cnuda_conv2d_forward_norm_input: null pointer");
    CNUDA_REQUIRE(imgs_per_group > 0 && B % imgs_per_group == 0, "cnuda_conv2d_forward_norm_input: statistics groups");
    CNUDA_REQUIRE(cnuda_conv2d_norm_input_supported(B, C, H, W, Cout, kh, kw, sh, sw, ph, pw),
                  "cnuda_conv2d_forward_norm_input: geometry without an apply-on-load kernel");
    if (stats) {
        int rows = 0;
        CNUDA_REQUIRE(act_slope < 0.0f && cnuda_conv2d_stats_block(B, C, H, W, Cout, kh, kw, sh, sw, ph, pw, &rows, nullptr) != 0,
                      "cnuda_conv2d_forward_norm_input: no statistics for this call");
    }
    const SmallNorm nm{mean, invstd, gamma, beta, imgs_per_group};
    return smallc_forward(x, weight, bias, y, B, C, H, W, Cout, kh, kw, sh, ph, pw, act_slope, 0, workspace, workspace_bytes,
                          (hipStream_t)stream, stats, &nm);
}
extern "C" int cnuda_conv2d_backward_weight_norm_input(const float* x, const float* mean, const float* invstd,
                                                       const float* gamma, const float* beta, int imgs_per_group,
                                                       const float* grad_y, float* grad_weight, float* grad_bias, int B, int C,
                                                       int H, int W, int Cout, int kh, int kw, int sh, int sw, int ph, int pw,
                                                       void* workspace, size_t workspace_bytes, cnuda_stream_t stream) {
    CNUDA_REQUIRE(x && mean && invstd && gamma && beta && grad_y && grad_weight,
                  "cnuda_conv2d_backward_weight_norm_input: null pointer");
    CNUDA_REQUIRE(imgs_per_group > 0 && B % imgs_per_group == 0, "cnuda_conv2d_backward_weight_norm_input: statistics groups");
    CNUDA_REQUIRE(cnuda_conv2d_norm_input_supported(B, C, H, W, Cout, kh, kw, sh, sw, ph, pw),
                  "cnuda_conv2d_backward_weight_norm_input: geometry without an apply-on-load kernel");
    ConvGeom g;
    if (int rc = fill_geom(g, B, C, H, W, Cout, kh, kw, sh, sw, ph, pw, "cnuda_conv2d_backward_weight_norm_input")) return rc;
    hipStream_t st = (hipStream_t)stream;
    const SmallNorm nm{mean, invstd, gamma, beta, imgs_per_group};
    if (int rc = smallc_backward_weight(x, grad_y, grad_weight, B, C, H, W, Cout, kh, kw, sh, ph, pw, workspace, workspace_bytes,
                                        st, &nm))
        return rc;
    if (grad_bias) launch_channel_sum(grad_y, grad_bias, B, Cout, (long long)g.Ho * g.Wo, st);
    return check_launch("cnuda_conv2d_backward_weight_norm_input");
}

extern "C" int cnuda_conv2d_backward_data_add(const float* grad_y, const float* weight, const float* addend,
                                              const float* addend2, float* grad_x, int B, int C, int H, int W, int Cout,
                                              int kh, int kw, int sh, int sw, int ph, int pw, void* workspace,
                                              size_t workspace_bytes, cnuda_stream_t stream) {
    if (!addend && addend2) { addend = addend2; addend2 = nullptr; }
    CNUDA_REQUIRE(grad_y && weight && grad_x, "cnuda_conv2d_backward_data: null pointer");
    ConvGeom g;
    if (int rc = fill_geom(g, B, C, H, W, Cout, kh, kw, sh, sw, ph, pw, "cnuda_conv2d_backward_data")) return rc;
    // An addend that IS grad_x (the fan-in slots of hip_runtime.fanout accumulate in place) must be read before the
    // tile is stored: the implicit-GEMM epilogues below do that per element; the LDS-tile kernels store first and add
    // in a second pass, which would double the new gradient and lose the old content -> they take the call only when
    // no addend aliases the output.
    const bool aliased = (addend && addend == grad_x) || (addend2 && addend2 == grad_x);
    if (!aliased && sh == 1 && sw == 1 && smallc_supported(Cout, C, kh, kw, 1, 1) && kh - 1 - ph >= 0 && kw - 1 - pw >= 0) {
        // (the LDS-tile kernels of the 3- / 16-channel layers have no addend: one elementwise pass behind them)
        if (int rc = smallc_forward(grad_y, weight, nullptr, grad_x, B, Cout, g.Ho, g.Wo, C, kh, kw, 1, kh - 1 - ph,
                                    kw - 1 - pw, -1.0f, 1, workspace, workspace_bytes, (hipStream_t)stream))
            return rc;
        if (addend) if (int rc = cnuda_add(grad_x, addend, grad_x, (long long)B * C * H * W, stream)) return rc;
        return addend2 ? cnuda_add(grad_x, addend2, grad_x, (long long)B * C * H * W, stream) : 0;
    }
    const ConvPlan q = make_plan(g);
    CNUDA_REQUIRE(workspace && workspace_bytes >= q.dgrad_bytes, "cnuda_conv2d_backward_data: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    Carver cv(workspace, workspace_bytes);
    float* Aws = reinterpret_cast<float*>(cv.take<char>(ig_a_bytes(q.Kpd, q.Mpd)));
    if (C == 16 && kh == 3 && kw == 3 && sh == 2 && sw == 2 && ph == 1 && pw == 1 && Cout <= 256 && matrix_mode() == 0 &&
        (size_t)Cout * 144 * sizeof(float) <= ig_a_bytes(q.Kpd, q.Mpd)) {
        // the 16-channel full-resolution level: one thread per 2 x 2 output block (dgrad_s2_c16_kernel)
        CNUDA_LAUNCH(dgrad_s2_c16_pack_kernel, dim3((Cout * 144 + 255) / 256), dim3(256), 0, st, weight, Aws, Cout);
        DgradS2Params dp{grad_y, Aws, grad_x, addend, addend2, B, Cout, H, W, g.Ho, g.Wo, (H + 1) / 2, (W + 1) / 2};
        ProfScope prof(st);
        prof.name("dgrad_s2_c16_kernel");
        CNUDA_LAUNCH(dgrad_s2_c16_kernel, dim3((dp.QW + 63) / 64, (dp.QH + 3) / 4, B), dim3(256), 0, st, dp);
        return check_launch("cnuda_conv2d_backward_data(stride 2, 16 channels)");
    }
    const bool buf_ok = buffer_addressing() && q.T <= 32 && (size_t)B * Cout * g.Ho * g.Wo * sizeof(float) < IG_BUF_OOB;
    if ((sh > 1 || sw > 1) && H % sh == 0 && W % sw == 0 && Cout % IG_BK == 0 &&
        ceil_div(kh, sh) * ceil_div(kw, sw) <= 9) {   // taps one class can see (tap_r/tap_s hold 9)
        // K restricted to the taps a parity class can see; all classes in ONE launch (blockIdx.y = class) unless a class
        // cuts K over the grid (split-K, small maps) or the tensors need the pointer loaders
        ProfScope prof(st);   // brackets the whole class group (inner scopes find nothing armed)
        prof.name("igemm_fwd*_kernel<*, ConvDgradClassLoader> x %d parity classes", sh * sw);
        const long long Nc = (long long)B * (H / sh) * (W / sw);
        ConvDgradClassParams cps[16];
        int Kpcs[16], taps_all[16][9], ncls = 0;
        bool any_split = false;
        for (int py = 0; py < sh; ++py)
            for (int px = 0; px < sw; ++px) {
                CNUDA_REQUIRE(ncls < 16, "cnuda_conv2d_backward_data: more than 16 parity classes");
                ConvDgradClassParams& cp = cps[ncls];
                cp.g = g; cp.gy = grad_y; cp.gx = grad_x; cp.py = py; cp.px = px; cp.Hc = H / sh; cp.Wc = W / sw;
                cp.add = addend; cp.add2 = addend2;
                cp.ntaps = 0;
                for (int r = 0; r < kh; ++r)
                    for (int t = 0; t < kw; ++t)
                        // (iy + ph - r) must be a multiple of sh for every iy = py + sh*qy: decided by py alone
                        // (C++ % keeps the dividend's sign; zero is zero either way)
                        if ((py + ph - r) % sh == 0 && (px + pw - t) % sw == 0) {
                            cp.tap_r[cp.ntaps] = r; cp.tap_s[cp.ntaps] = t; taps_all[ncls][cp.ntaps] = r * kw + t;
                            cp.tap_dy[cp.ntaps] = (py + ph - r) / sh; cp.tap_dx[cp.ntaps] = (px + pw - t) / sw;
                            ++cp.ntaps;
                        }
                // ntaps == 0 (a class no tap reaches): K is all padding, the kernel writes zeros
                const int Kc = cp.ntaps * Cout;
                Kpcs[ncls] = round_up(Kc > 0 ? Kc : IG_KC, IG_KC);
                any_split = any_split || (buf_ok && pick_splitk(C, Nc, Kpcs[ncls]).on());
                ++ncls;
            }
        if (buf_ok && !any_split && ncls <= MAX_CLASSES && matrix_mode() == 0) {
            const int bm = pick_bm(C, Nc * ncls), Mp = round_up(C, bm);      // (the grid is ncls times one class's)
            // classes in order of decreasing K: the long ones start first
            int order[MAX_CLASSES];
            for (int i = 0; i < ncls; ++i) order[i] = i;
            std::stable_sort(order, order + ncls, [&](int a, int b2) { return Kpcs[a] > Kpcs[b2]; });
            ConvDgradClassSet set;
            size_t used = 0;
            for (int i = 0; i < ncls; ++i) {
                const int c = order[i];
                const size_t bytes = carve_bytes(ig_a_bytes(Kpcs[c], Mp), 1);
                CNUDA_REQUIRE(used + bytes + 256 <= workspace_bytes, "cnuda_conv2d_backward_data: workspace");
                float* dst = reinterpret_cast<float*>(reinterpret_cast<char*>(Aws) + used);
                used += bytes;
                set.cls[i] = cps[c];
                set.Kp[i] = Kpcs[c];
                set.A[i] = launch_pack_taps(weight, dst, ig_a_bytes(Kpcs[c], Mp), Cout, C, q.T, taps_all[c], cps[c].ntaps, Kpcs[c], Mp, st);
            }
            for (int i = ncls; i < MAX_CLASSES; ++i) { set.cls[i] = set.cls[0]; set.Kp[i] = set.Kp[0]; set.A[i] = set.A[0]; }
            CNUDA_REQUIRE(Nc < (1ll << 31) - IG_BN, "cnuda_conv2d_backward_data: more than 2^31 pixels per call");
            const int n_tiles = ceil_div(Nc, IG_BN), m_tiles = Mp / bm;
            const dim3 grid(n_tiles * m_tiles, ncls);
            if (wave_specialised() && bm == 128)
                CNUDA_LAUNCH((igemm_fwd_ws_classes_kernel<128>), grid, dim3(2 * IG_THREADS), 0, st, set, Mp, C, Nc, n_tiles, m_tiles);
            else if (wave_specialised() && bm == 64)
                CNUDA_LAUNCH((igemm_fwd_ws_classes_kernel<64>), grid, dim3(2 * IG_THREADS), 0, st, set, Mp, C, Nc, n_tiles, m_tiles);
            else if (bm == 128)
                CNUDA_LAUNCH((igemm_fwd_classes_kernel<128>), grid, dim3(IG_THREADS), 0, st, set, Mp, C, Nc, n_tiles, m_tiles);
            else if (bm == 64)
                CNUDA_LAUNCH((igemm_fwd_classes_kernel<64>), grid, dim3(IG_THREADS), 0, st, set, Mp, C, Nc, n_tiles, m_tiles);
            else
                CNUDA_LAUNCH((igemm_fwd_classes_kernel<32>), grid, dim3(IG_THREADS), 0, st, set, Mp, C, Nc, n_tiles, m_tiles);
            return check_launch("cnuda_conv2d_backward_data(classes)");
        }
        for (int ci = 0; ci < ncls; ++ci) {
                const ConvDgradClassParams& cp = cps[ci];
                const int Kpc = Kpcs[ci];
                const int* taps = taps_all[ci];
                const SplitK sc = buf_ok ? pick_splitk(C, Nc, Kpc) : SplitK();
                const int bm = sc.on() ? sc.bm : pick_bm(C, Nc), Mp = round_up(C, bm);
                const size_t slab_bytes = splitk_slab_bytes(sc, C, Nc);
                CNUDA_REQUIRE(carve_bytes(ig_a_bytes(Kpc, Mp), 1) + carve_bytes(slab_bytes, 1) + 256 <= workspace_bytes,
                              "cnuda_conv2d_backward_data: workspace");
                const float* A = launch_pack_taps(weight, Aws, ig_a_bytes(Kpc, Mp), Cout, C, q.T, taps, cp.ntaps, Kpc, Mp, st);
                // (the slabs of a class behind the packed matrix -- whose room is that of the plan's largest: the same carve)
                float* slab = sc.on() ? reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(Aws) + ig_a_bytes(Kpc, Mp) + 255) & ~(uintptr_t)255) : nullptr;
                if (int rc = buf_ok ? launch_fwd<ConvDgradClassBufLoader>(bm, cp, A, Mp, Kpc, C, Nc, st, "cnuda_conv2d_backward_data(class)", sc, slab)
                                    : launch_fwd<ConvDgradClassLoader>(bm, cp, A, Mp, Kpc, C, Nc, st, "cnuda_conv2d_backward_data(class)"))
                    return rc;
        }
        return 0;
    }
    if (!q.skd.on() && hconv_ok(g, Cout, q.bmd)) {   // (Co % 16 == 0: Kpd = 9 Co, no padded rows)
        const float* Ah = launch_pack(weight, Aws, ig_a_bytes(q.Kpd, q.Mpd), Cout, C, q.T, PACK_HALO_DGRAD, q.Kpd, q.Mpd, 0, st);
        ConvDgradParams ph{g, grad_y, grad_x, Cout, addend, addend2};
        return launch_hconv<HconvDgrad>(q.bmd, ph, grad_y, Cout, g, Ah, q.Mpd, q.Kpd, C, q.Nd, st, "cnuda_conv2d_backward_data");
    }
    const float* A = launch_pack(weight, Aws, ig_a_bytes(q.Kpd, q.Mpd), Cout, C, q.T, PACK_DGRAD, q.Kpd, q.Mpd,
                                 round_up(Cout, IG_BK), st);
    ConvDgradParams p{g, grad_y, grad_x, round_up(Cout, IG_BK), addend, addend2};
    if (buf_ok && sh == 1 && sw == 1) {
        float* slab = q.skd.on() ? reinterpret_cast<float*>(cv.take<char>(splitk_slab_bytes(q.skd, C, q.Nd))) : nullptr;
        return launch_fwd<ConvDgradBufLoader>(q.bmd, p, A, q.Mpd, q.Kpd, C, q.Nd, st, "cnuda_conv2d_backward_data", q.skd, slab);
    }
    CNUDA_REQUIRE(!q.skd.on(), "cnuda_conv2d_backward_data: split-K plan off the buffer-addressed path");
    return launch_fwd<ConvDgradLoader>(q.bmd, p, A, q.Mpd, q.Kpd, C, q.Nd, st, "cnuda_conv2d_backward_data");
}

extern "C" int cnuda_conv2d_backward_weight(const float* x, const float* grad_y, float* grad_weight, float* grad_bias,
                                            int B, int C, int H, int W, int Cout, int kh, int kw, int sh, int sw,
                                            int ph, int pw, void* workspace, size_t workspace_bytes,
                                            cnuda_stream_t stream) {
    CNUDA_REQUIRE(x && grad_y && grad_weight, "cnuda_conv2d_backward_weight: null pointer");
    ConvGeom g;
    if (int rc = fill_geom(g, B, C, H, W, Cout, kh, kw, sh, sw, ph, pw, "cnuda_conv2d_backward_weight")) return rc;
    if (smallc_supported(C, Cout, kh, kw, sh, sw)) {
        hipStream_t st0 = (hipStream_t)stream;
        if (int rc = smallc_backward_weight(x, grad_y, grad_weight, B, C, H, W, Cout, kh, kw, sh, ph, pw, workspace,
                                            workspace_bytes, st0))
            return rc;
        if (grad_bias) launch_channel_sum(grad_y, grad_bias, B, Cout, (long long)g.Ho * g.Wo, st0);
        return check_launch("cnuda_conv2d_backward_weight(small)");
    }
    const ConvPlan q = make_plan(g);
    CNUDA_REQUIRE(workspace && workspace_bytes >= q.wgrad_bytes, "cnuda_conv2d_backward_weight: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    Carver cv(workspace, workspace_bytes);
    float* slabs = cv.take<float>((size_t)q.Z * q.Mpw * q.Jp);
    // bias gradient: row sums of grad_y per split from the GEMM's own staging registers, summed with the slabs
    float* bsl = grad_bias ? cv.take<float>((size_t)q.Z * q.Mpw) : nullptr;
    ConvWParams p{g, x, grad_y};
    if (q.hw_s2) {
        ProfScope prof(st);
        prof.name("hwgrad_s2_kernel");
        const HwS2Params hp{x, grad_y, B, C, H, W, Cout, g.Ho, g.Wo, q.hw_tiles, q.hw_tiles_per_split};
        const size_t lds = HS_LDS_FLOATS * sizeof(float);
        CNUDA_REQUIRE(raise_dynamic_lds(reinterpret_cast<const void*>(&hwgrad_s2_kernel), lds),
                      "cnuda_conv2d_backward_weight: dynamic LDS");
        CNUDA_LAUNCH(hwgrad_s2_kernel, dim3(C / 16, q.Z, (Cout + 31) / 32), dim3(IG_THREADS), lds, st, hp, slabs, q.Mpw, q.Jp, bsl);
    } else if (q.hw) {
        ProfScope prof(st);
        const int tw = halo_tile_width(W);
        const bool side = tw < W;
        prof.name(side ? "hwgrad_kernel<%d, side>" : "hwgrad_kernel<%d>", tw);
        const HwParams hp{x, grad_y, B, C, H, W, Cout, W / tw, hwgrad_tiles_y(g), q.hw_tiles, q.hw_tiles_per_split};
        const dim3 grid(C / 16, q.Z), blk(IG_THREADS);
#define CNUDA_HWGRAD(TWV, SIDEV) do {                                                                                \
        const size_t lds = HwShape<TWV, SIDEV>::lds_floats * sizeof(float);                                           \
        CNUDA_REQUIRE(raise_dynamic_lds(reinterpret_cast<const void*>(&hwgrad_kernel<TWV, SIDEV>), lds),              \
                      "cnuda_conv2d_backward_weight: dynamic LDS");                                                   \
        CNUDA_LAUNCH((hwgrad_kernel<TWV, SIDEV>), grid, blk, lds, st, hp, slabs, q.Mpw, q.Jp, bsl);                   \
    } while (0)
        if (!side) {
            if (tw == 128) CNUDA_HWGRAD(128, false); else if (tw == 64) CNUDA_HWGRAD(64, false);
            else if (tw == 32) CNUDA_HWGRAD(32, false); else if (tw == 16) CNUDA_HWGRAD(16, false); else CNUDA_HWGRAD(8, false);
        } else {
            if (tw == 128) CNUDA_HWGRAD(128, true); else if (tw == 64) CNUDA_HWGRAD(64, true);
            else if (tw == 32) CNUDA_HWGRAD(32, true); else if (tw == 16) CNUDA_HWGRAD(16, true); else CNUDA_HWGRAD(8, true);
        }
#undef CNUDA_HWGRAD
    } else {
        ProfScope prof(st);
        const dim3 grid(q.Jp / q.wbj, q.Mpw / q.wbm, q.Z), blk(IG_THREADS);
        const bool fast = C % 64 == 0;
        const bool buf = fast && wgrad_buffer_ok(g);
        const bool buf8 = !fast && C % 8 == 0 && wgrad_buffer_ok(g) && (q.wbm == 32 || (q.wbm == 64 && q.wbj == 64));
        prof.name((wave_specialised() && fast && q.wbm >= 64) ? "igemm_wgrad_ws_kernel<%s, %d, %d>" : "igemm_wgrad_kernel<%s, %d, %d>",
                  buf ? "ConvWBufLoader" : (buf8 ? "ConvWBufLoaderC8" : (fast ? "ConvWLoader<2>" : "ConvWLoader<0>")), q.wbm, q.wbj);
        if (buf8) {
            if (q.wbm == 32)
                CNUDA_LAUNCH((igemm_wgrad_kernel<ConvWBufLoaderC8, 32, 128>), grid, blk, 0, st, p, slabs, q.Mpw, q.Jp,
                                   q.Nf, q.pix_per_split, bsl);
            else
                CNUDA_LAUNCH((igemm_wgrad_kernel<ConvWBufLoaderC8, 64, 64>), grid, blk, 0, st, p, slabs, q.Mpw, q.Jp,
                                   q.Nf, q.pix_per_split, bsl);
        } else if (buf) {
            launch_wgrad_buf<ConvWBufLoader>(q, p, grid, slabs, bsl, st);
        } else if (wave_specialised() && fast && q.wbm == 64) {
            const dim3 blk2(2 * IG_THREADS);
            if (q.wbj == 128)
                CNUDA_LAUNCH((igemm_wgrad_ws_kernel<ConvWLoader<2>, 64, 128>), grid, blk2, 0, st, p, slabs, q.Mpw,
                                   q.Jp, q.Nf, q.pix_per_split, bsl);
            else
                CNUDA_LAUNCH((igemm_wgrad_ws_kernel<ConvWLoader<2>, 64, 64>), grid, blk2, 0, st, p, slabs, q.Mpw,
                                   q.Jp, q.Nf, q.pix_per_split, bsl);
        } else if (q.wbm == 64) {
            if (fast && q.wbj == 128)
                CNUDA_LAUNCH((igemm_wgrad_kernel<ConvWLoader<2>, 64, 128>), grid, blk, 0, st, p, slabs, q.Mpw,
                                   q.Jp, q.Nf, q.pix_per_split, bsl);
            else if (fast)
                CNUDA_LAUNCH((igemm_wgrad_kernel<ConvWLoader<2>, 64, 64>), grid, blk, 0, st, p, slabs, q.Mpw, q.Jp,
                                   q.Nf, q.pix_per_split, bsl);
            else
                CNUDA_LAUNCH((igemm_wgrad_kernel<ConvWLoader<0>, 64, 64>), grid, blk, 0, st, p, slabs, q.Mpw, q.Jp,
                                   q.Nf, q.pix_per_split, bsl);
        } else {
            if (fast)
                CNUDA_LAUNCH((igemm_wgrad_kernel<ConvWLoader<2>, 32, 128>), grid, blk, 0, st, p, slabs, q.Mpw,
                                   q.Jp, q.Nf, q.pix_per_split, bsl);
            else
                CNUDA_LAUNCH((igemm_wgrad_kernel<ConvWLoader<0>, 32, 128>), grid, blk, 0, st, p, slabs, q.Mpw,
                                   q.Jp, q.Nf, q.pix_per_split, bsl);
        }
    }
    if (int rc = check_launch("cnuda_conv2d_backward_weight")) return rc;
    launch_slab_reduce(slabs, grad_weight, q.Z, q.Mpw, q.Jp, Cout, C, q.T, st, bsl, grad_bias);
    return check_launch("cnuda_conv2d_backward_weight(reduce)");
}
