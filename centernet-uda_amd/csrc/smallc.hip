// Convolutions with few channels at full image resolution and stride 1 -- the DLA-34 stem (3->16, 7x7 @512x512) and
// level0 (16->16, 3x3): backends/dla.py:233-241, 277-287.  (level1, 16->32 at stride 2, runs on igemm.cuh's forward
// kernel, dgrad_s2_c16_kernel and hwgrad_s2_kernel.)
// Their K = C*kh*kw is tiny (144-147) while the pixel count is huge, so the generic im2col-style implicit
// GEMM spends its time re-gathering the same input pixel kh*kw times and pads the 16 output channels to a
// 32-row MFMA tile.  Here the input tile is staged ONCE in LDS with its halo (zero-filled outside the
// image) and the GEMM runs on v_mfma_f32_16x16x4_f32 tiles that match 16 output channels exactly:
//
//   forward : out[o][px] = sum_k Wp[k][o] * Xh[koff(k) + pix(px)]            k = (tap, c), c fastest
//             A (weights) and the halo offset of every k live in LDS for the whole workgroup,
//             B is one ds_read_b32 per MFMA (lanes = 16 consecutive pixels).
//   weight gradient : gw[o][k] = sum_px gy[o][px] * Xh[koff(k) + pix(px)]
//             A = gy tile from LDS, B = the same halo image read with lanes = 16 consecutive k
//             (c fastest -> plane stride, odd -> conflict-free); every wave owns a quarter of the
//             tile's pixels and a private set of accumulators; partial slabs per workgroup, fixed-order
//             reduction (bit-reproducible).
// The stride-1 input gradient is the forward kernel on flipped / transposed weights.
#include "igemm.cuh"
#include "igemm_host.h"

namespace cnuda {
namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int SC_MAXK = 160;          // K = C*kh*kw padded to a multiple of 4 (147 -> 148, 144)
constexpr int SC_TW = 64;             // output tile: TH rows x 64 columns, one row per wave
constexpr int SC_TH = 4;

struct SmallGeom {
    int B, C, H, W, Co, kh, kw, s, ph, pw, Ho, Wo;
    int K, Kp;              // C*kh*kw and its multiple-of-4 padding
    int HR, HC, plane;      // halo rows / cols and the (odd) channel-plane stride of the LDS image
};

// halo image: Xh[c][hy][hx], hy in [0, HR), hx in [0, HC); input row = oy0 - ph + hy (stride 1).  The kernels are
// instantiated per (square) filter size KH, so the halo geometry is a set of compile-time constants: the staging code is
// then a straight line of buffer loads and LDS stores with immediate offsets.  (The first version derived (channel, row)
// of every halo row by a run-time division and guarded every load with a branch: ~100 scalar instructions per row, 24
// rows per wave and tile -- measured, with the MFMA loop compiled out the 16 -> 16 layer still took 509 of its 723 us.)
constexpr int SC_ROWS_PER_WAVE = 24;

// Apply on load (round 5): the input x is the OUTPUT OF A CONVOLUTION whose train-mode BatchNorm + ReLU was not applied --
// the kernels normalise while they stage: a = max(x * sc + sh, 0) with sc = invstd * gamma, sh = beta - mean * sc per
// (statistics group of the image, channel), the two operations bn_apply_kernel performs, rounded the same way (bit-identical
// to the materialised activation), and zero outside the image AFTER the transform.  mean == nullptr: x is taken as it is.
__device__ __forceinline__ float sc_bn_shift(float beta, float mean, float sc) { return __fsub_rn(beta, __fmul_rn(mean, sc)); }
__device__ __forceinline__ float sc_bn_act(float x, float sc, float sh) { return fmaxf(__fadd_rn(__fmul_rn(x, sc), sh), 0.0f); }

template <int KH> struct ScShape {
    static constexpr int HR = SC_TH - 1 + KH, HC = SC_TW - 1 + KH;
    static constexpr int NX = HC - 64;                      // halo columns beyond a wave's 64 lanes
    static constexpr int plane = (HR * HC) | 1;             // odd plane stride: lanes that differ in c hit different banks
    // channels a wave stages (wid, wid + 4, ...): at most SC_ROWS_PER_WAVE staging registers, and the extra columns of
    // all its rows in one load -- 4 channels (C <= 16) for 3x3, 1 (C <= 4) for 7x7
    static constexpr int CPW = SC_ROWS_PER_WAVE / HR < 64 / (HR * NX) ? SC_ROWS_PER_WAVE / HR : 64 / (HR * NX);
    static constexpr int ROWS = CPW * HR;
    static_assert(ROWS * NX <= 64, "the extra columns of a wave's rows fit one load");
};

// A wave stages the halo rows of channels wid + 4 * cc (cc < CPW): lane = column (one 256-byte load per row), and the
// NX columns past the 64th of all its rows in ONE more load (lane -> (row slot, column)).  Rows / columns outside the
// image read 0.0f through the buffer range check (the padding value), no compare on the data path.
// load() only ISSUES the global loads (into registers); store() writes them to LDS: the forward kernel loads the NEXT
// tile's halo before the MFMA loop of the current one.
template <int KH, bool NORM = false>
struct HaloStage {
    using S = ScShape<KH>;
    struct Regs { float v[S::ROWS]; float vx; };
    float nsc[S::CPW], nsh[S::CPW], xsc, xsh;    // NORM: scale / shift of the channels this wave stages, and of the extra lane's
    buf_rsrc rs;
    unsigned va;                 // byte offset of column ix0 + lane inside an input row, or the sentinel
    unsigned xcol;               // the extra-column lane: byte offset of ITS column, or the sentinel
    int x_c, x_hy, x_lds;        // ... its channel, halo row and LDS cell (x_lds < 0: lane unused)
    int lds0;                    // LDS cell of (channel wid, halo row 0, column lane)
    int wid;
    __device__ __forceinline__ HaloStage(const SmallGeom& g, const float* xb, int ix0, int tid) {
        const int lane = tid & 63;
        wid = __builtin_amdgcn_readfirstlane(tid >> 6);       // (uniform, and the compiler knows it: row terms stay scalar)
        rs = ig_make_rsrc(xb, (unsigned)((size_t)g.C * g.H * g.W * sizeof(float)));
        const int ixa = ix0 + lane;
        va = (ixa >= 0 && ixa < g.W) ? (unsigned)ixa * 4u : IG_BUF_OOB;
        lds0 = wid * S::plane + lane;
        const int slot = lane / S::NX, xc = 64 + lane % S::NX;
        const int cc = slot / S::HR;
        x_hy = slot - cc * S::HR;
        x_c = wid + 4 * cc;
        const int ixx = ix0 + xc;
        xcol = (ixx >= 0 && ixx < g.W) ? (unsigned)ixx * 4u : IG_BUF_OOB;
        x_lds = (slot < S::ROWS && x_c < g.C) ? x_c * S::plane + x_hy * S::HC + xc : -1;
        if (x_lds < 0) xcol = IG_BUF_OOB;
    }
    // NORM: the scale / shift of image b's statistics group
    __device__ __forceinline__ void set_norm(const SmallGeom& g, const SmallNorm& nm, int b) {
        const int grp = b / nm.imgs_per_group;
        auto coef = [&](int c, float& sc, float& sh) {
            sc = __fmul_rn(nm.invstd[grp * g.C + c], nm.gamma[c]);
            sh = sc_bn_shift(nm.beta[c], nm.mean[grp * g.C + c], sc);
        };
#pragma unroll
        for (int cc = 0; cc < S::CPW; ++cc) {
            const int c = wid + 4 * cc;
            nsc[cc] = 0.f; nsh[cc] = 0.f;
            if (c < g.C) coef(c, nsc[cc], nsh[cc]);
        }
        xsc = 0.f; xsh = 0.f;
        if (x_lds >= 0) coef(x_c, xsc, xsh);
    }
    __device__ __forceinline__ void load(const SmallGeom& g, int iy0, Regs& r) const {
#pragma unroll
        for (int cc = 0; cc < S::CPW; ++cc) {
            const int c = wid + 4 * cc;                       // wave-uniform
#pragma unroll
            for (int hy = 0; hy < S::HR; ++hy) {
                const int iy = iy0 + hy;
                const bool ok = c < g.C && iy >= 0 && iy < g.H;
                r.v[cc * S::HR + hy] = ig_buf_load(rs, ok ? va : IG_BUF_OOB, ok ? (unsigned)((c * g.H + iy) * g.W) * 4u : 0u);
            }
        }
        const int iy = iy0 + x_hy;
        const bool ok = iy >= 0 && iy < g.H;
        r.vx = ig_buf_load(rs, ok ? xcol + (unsigned)((x_c * g.H + iy) * g.W) * 4u : IG_BUF_OOB, 0u);
    }
    // iy0: the tile's first halo row in the image (NORM: which staged cells lie inside the image)
    __device__ __forceinline__ void store(const SmallGeom& g, float* __restrict__ Xh, const Regs& r, int iy0) const {
#pragma unroll
        for (int cc = 0; cc < S::CPW; ++cc) {
            if (wid + 4 * cc >= g.C) break;                   // (uniform)
#pragma unroll
            for (int hy = 0; hy < S::HR; ++hy) {
                float v = r.v[cc * S::HR + hy];
                if constexpr (NORM) {
                    const bool in = iy0 + hy >= 0 && iy0 + hy < g.H && va != IG_BUF_OOB;
                    v = in ? sc_bn_act(v, nsc[cc], nsh[cc]) : 0.0f;
                }
                Xh[lds0 + 4 * cc * S::plane + hy * S::HC] = v;
            }
        }
        if (x_lds >= 0) {
            float v = r.vx;
            if constexpr (NORM) {
                const bool in = iy0 + x_hy >= 0 && iy0 + x_hy < g.H && xcol != IG_BUF_OOB;
                v = in ? sc_bn_act(v, xsc, xsh) : 0.0f;
            }
            Xh[x_lds] = v;
        }
    }
};

// offset of GEMM row k = tap*C + c inside the halo image (relative to the pixel's top-left halo cell)
__device__ __forceinline__ int k_offset(const SmallGeom& g, int k) {
    if (k >= g.K) return 0;       // padding rows multiply zero weights
    const int tap = k / g.C, c = k - tap * g.C;
    const int r = tap / g.kw, t = tap - r * g.kw;
    return c * g.plane + r * g.HC + t;
}

// ---------------------------------------------------------------------------------------------
// forward.  grid = (tiles_x, tiles_y, B); Wp is [Kp][16*MT] (packed, zero padded), MT = Co tiles of 16.
// ---------------------------------------------------------------------------------------------
constexpr int SC_NV = 8;   // vertical tiles per workgroup (amortises the per-workgroup weight / offset fetch)
constexpr int SC_YLD = SC_TW + 4;   // row stride of the per-wave output staging tile (16-byte aligned rows)

#ifndef SC_FWD_OCC
#define SC_FWD_OCC 2
#endif
template <int MT, int KH, bool NORM = false>
__global__ __launch_bounds__(IG_THREADS, SC_FWD_OCC) void smallc_fwd_kernel(SmallGeom g, const float* __restrict__ x,
                                                               const float* __restrict__ Wp,
                                                               const int* __restrict__ koff_tab,
                                                               const float* __restrict__ bias, float* __restrict__ y,
                                                               float act_slope, float* __restrict__ stats, SmallNorm nm) {
    extern __shared__ __align__(16) float smem[];
    float* Xh = smem;
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);   // (uniform, and the compiler knows it)
    const int kq = lane >> 4, il = lane & 15;
    const int ox0 = blockIdx.x * SC_TW, b = blockIdx.z;
    const float* xb = x + (size_t)b * g.C * g.H * g.W;
    const int ksteps = g.Kp / 4;

    // A operand (weights) and the halo offset of every GEMM row k, constant for the whole kernel, live in LDS: in
    // registers they took 80-120 VGPRs, which the halo prefetch below needs (one extra conflict-free LDS read per
    // four MFMAs instead)
    float* Ws = smem + g.C * g.plane + 16 + 4 * (16 * MT * SC_YLD);         // [Kp][16 * MT]
    int* Ks = reinterpret_cast<int*>(Ws + g.Kp * 16 * MT);                  // [Kp]
    for (int i = tid; i < g.Kp * 16 * MT; i += IG_THREADS) Ws[i] = Wp[i];
    for (int i = tid; i < g.Kp; i += IG_THREADS) Ks[i] = koff_tab[i];
    const int HoWo = g.Ho * g.Wo;
    // per-wave output staging tile [16 * MT][SC_YLD]: the accumulators have 16 consecutive pixels across lanes (64-byte
    // runs per store); staged through LDS every lane writes four consecutive pixels and a wave instruction covers
    // whole 256-byte rows
    float* Ys = smem + g.C * g.plane + 16 + wid * (16 * MT * SC_YLD);
    const bool vec = (g.Wo & 3) == 0;
    using S = ScShape<KH>;
    HaloStage<KH, NORM> hs(g, xb, ox0 - g.pw, tid);
    if constexpr (NORM) hs.set_norm(g, nm, b);
    typename HaloStage<KH, NORM>::Regs halo;
    hs.load(g, (blockIdx.y * SC_NV) * SC_TH - g.ph, halo);
#pragma unroll 1
    for (int vt = 0; vt < SC_NV; ++vt) {
    const int oy0 = (blockIdx.y * SC_NV + vt) * SC_TH;
    if (oy0 >= g.Ho) break;
    __syncthreads();                                // the previous tile's fragment reads are done
    hs.store(g, Xh, halo, oy0 - g.ph);
    __syncthreads();
    if (vt + 1 < SC_NV && oy0 + SC_TH < g.Ho)       // next tile's halo: in flight under this tile's MFMAs
        hs.load(g, (oy0 + SC_TH) - g.ph, halo);

    const int oy = oy0 + wid;                       // one output row per wave
    float* yb = y + (size_t)b * g.Co * HoWo + (size_t)oy * g.Wo;
    if (oy < g.Ho) {
        constexpr int NP = SC_TW / 16;                  // four 16-pixel tiles along the row, processed together
        const int pbase0 = wid * S::HC + il;         // (stride 1 only: smallc_supported)
        f32x4 acc[MT][NP];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int pt = 0; pt < NP; ++pt) acc[m][pt] = f32x4{0.f, 0.f, 0.f, 0.f};
        // explicit two-deep software pipeline (the compiler emits read -> wait -> MFMA per step otherwise, and each
        // step then pays two dependent LDS round trips: row offset, then operand): the operands of step ks + 1 and
        // the row offset of step ks + 2 are read before the MFMAs of step ks issue
        auto offs = [&](int ks) { return Ks[(ks < ksteps ? ks : ksteps - 1) * 4 + kq] + pbase0; };
        auto frag = [&](int ks, int ko, float (&fa)[MT], float (&fb)[NP]) {
            const int k = (ks < ksteps ? ks : ksteps - 1) * 4 + kq;
#pragma unroll
            for (int m = 0; m < MT; ++m) fa[m] = Ws[k * (16 * MT) + m * 16 + il];
#pragma unroll
            for (int pt = 0; pt < NP; ++pt) fb[pt] = Xh[ko + pt * 16];
        };
        auto mma = [&](const float (&fa)[MT], const float (&fb)[NP]) {
#pragma unroll
            for (int pt = 0; pt < NP; ++pt)
#pragma unroll
                for (int m = 0; m < MT; ++m)
                    acc[m][pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[m], fb[pt], acc[m][pt], 0, 0, 0);
        };
        float a0[MT], b0[NP], a1[MT], b1[NP];
        int ko0 = offs(0), ko1 = offs(1);
        frag(0, ko0, a0, b0);
        for (int ks = 0; ks < ksteps; ks += 2) {
            ko0 = offs(ks + 2);
            frag(ks + 1, ko1, a1, b1);
            __builtin_amdgcn_sched_barrier(0);
            mma(a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            ko1 = offs(ks + 3);
            frag(ks + 2, ko0, a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            if (ks + 1 < ksteps) mma(a1, b1);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (vec) {
#pragma unroll
            for (int pt = 0; pt < NP; ++pt)
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) Ys[(m * 16 + kq * 4 + r) * SC_YLD + pt * 16 + il] = acc[m][pt][r];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // same wave: LDS operations complete in order
            const int c4 = lane & 15, rsub = lane >> 4;
            const int ox = ox0 + 4 * c4;
#pragma unroll
            for (int it = 0; it < 4 * MT; ++it) {
                const int o = it * 4 + rsub;
                f32x4 v = *reinterpret_cast<const f32x4*>(Ys + o * SC_YLD + 4 * c4);
                if (o < g.Co && ox < g.Wo) {
                    if (bias) v += bias[o];
                    if (act_slope >= 0.0f) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) if (v[e] < 0.0f) v[e] *= act_slope;
                    }
                    *reinterpret_cast<f32x4*>(yb + (size_t)o * HoWo + ox) = v;
                }
                if (stats) {
                    // BatchNorm statistics of what was stored (igemm.cuh, "BatchNorm statistics"): the wave's row segment
                    // (<= 64 pixels of output row oy) is one block, numbered (image, row, column tile); the 16 lanes of an
                    // output channel meet through two quad permutes, a half-row and a row mirror
                    if (!(o < g.Co && ox < g.Wo)) v = f32x4{0.f, 0.f, 0.f, 0.f};
                    float s1 = (v[0] + v[1]) + (v[2] + v[3]);
                    float s2 = (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
                    auto row16 = [](float t) {
                        t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0xB1, 0xf, 0xf, false));
                        t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0x4E, 0xf, 0xf, false));
                        t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0x141, 0xf, 0xf, false));
                        t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0x140, 0xf, 0xf, false));
                        return t;
                    };
                    s1 = row16(s1);
                    s2 = row16(s2);
                    if (c4 == 0)
                        reinterpret_cast<float2*>(stats)[(((size_t)b * g.Ho + oy) * gridDim.x + blockIdx.x) * (16 * MT) + o] =
                            make_float2(s1, s2);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // reads done before the next tile's staging
        } else {
#pragma unroll
        for (int pt = 0; pt < NP; ++pt) {
            const int ox = ox0 + pt * 16 + il;
            if (ox < g.Wo) {
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int o = m * 16 + kq * 4 + r;
                        if (o < g.Co) {
                            float v = acc[m][pt][r];
                            if (bias) v += bias[o];
                            if (act_slope >= 0.0f && v < 0.0f) v *= act_slope;
                            yb[(size_t)o * HoWo + ox] = v;
                        }
                    }
            }
        }
        }
    }
    }
}

// ---------------------------------------------------------------------------------------------
// weight gradient.  grid = (tiles_x, tiles_y, B); slab per workgroup: [16*MT][Kp16] (Kp16 = K padded to 16)
// ---------------------------------------------------------------------------------------------
// NT: column tiles (of 16 k) the instance computes -- >= Kp16 / 16; tiles past the real ones read halo cell 0 and are
// never written out.  A compile-time count keeps the MFMA loop free of branches: the first version tested `nt < ntiles`
// per MFMA, which serialised every MFMA behind its own LDS read (read, wait, multiply; nine times per k-step).
template <int MT, int KH, int NT, bool NORM = false>
__global__ __launch_bounds__(IG_THREADS, (MT == 1 ? 3 : 2)) void smallc_wgrad_kernel(SmallGeom g, const float* __restrict__ x,
                                                                 const float* __restrict__ gy,
                                                                 const int* __restrict__ koff_tab,
                                                                 float* __restrict__ slabs, int Kp16, SmallNorm nm) {
    extern __shared__ __align__(16) float smem[];
    float* Xh = smem;                                   // [C][plane]
    float* Gs = Xh + g.C * g.plane + 16;                // [16*MT][SC_TH*SC_TW + 1]  (o, pixel of the tile)
    constexpr int GLD = SC_TH * SC_TW + 1;
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);   // (uniform, and the compiler knows it)
    const int kq = lane >> 4, il = lane & 15;
    const int ox0 = blockIdx.x * SC_TW, b = blockIdx.z;
    const int HoWo = g.Ho * g.Wo;
    const float* xb = x + (size_t)b * g.C * g.H * g.W;
    const float* gb = gy + (size_t)b * g.Co * HoWo;
    const int ntiles = Kp16 / 16;
    HaloStage<KH, NORM> hs(g, xb, ox0 - g.pw, tid);
    if constexpr (NORM) hs.set_norm(g, nm, b);

    // this lane's column k = nt*16 + il of every column tile -> halo cell of its pixel 0 (B operand, lanes = columns):
    // row `wid` of the tile, pixel kq of a k-step
    int kaddr[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) kaddr[nt] = (nt < ntiles ? koff_tab[nt * 16 + il] : 0) + wid * ScShape<KH>::HC + kq;

    f32x4 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[m][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    // staging registers of one tile: the halo rows and the grad_y rows (o, ty) = wid, wid + 4, ... of this wave.  Both are
    // buffer loads issued together (one exposed round trip per tile at most); with one row tile of output channels
    // (MT == 1) the NEXT tile's loads go out before this tile's MFMA loop and the round trip hides behind it.
#ifndef SC_WG_PREFETCH
#define SC_WG_PREFETCH 1
#endif
    constexpr bool PF = MT == 1 && SC_WG_PREFETCH;
    typename HaloStage<KH, NORM>::Regs halo;
    float gv[4 * MT * SC_TH];
    const buf_rsrc grs = ig_make_rsrc(gb, (unsigned)((size_t)g.Co * HoWo * sizeof(float)));
    const unsigned gva = ox0 + lane < g.Wo ? (unsigned)(ox0 + lane) * 4u : IG_BUF_OOB;
    auto tile_load = [&](int oy0) {
        hs.load(g, oy0 - g.ph, halo);
#pragma unroll
        for (int i = 0; i < 4 * MT * SC_TH; ++i) {
            const int row = wid + 4 * i;
            const int o = row / SC_TH, ty = row - o * SC_TH;     // wave-uniform
            const int oy = oy0 + ty;
            const bool ok = o < g.Co && oy < g.Ho;
            gv[i] = ig_buf_load(grs, ok ? gva : IG_BUF_OOB, ok ? (unsigned)(o * HoWo + oy * g.Wo) * 4u : 0u);
        }
    };
    if (PF) tile_load(blockIdx.y * SC_NV * SC_TH);
#pragma unroll 1
    for (int vt = 0; vt < SC_NV; ++vt) {
    const int oy0 = (blockIdx.y * SC_NV + vt) * SC_TH;
    if (oy0 >= g.Ho) break;
    const bool more = vt + 1 < SC_NV && oy0 + SC_TH < g.Ho;
    __syncthreads();                                    // the previous tile's fragment reads are done
    if (!PF) tile_load(oy0);
    hs.store(g, Xh, halo, oy0 - g.ph);
    // gy tile: Gs[o][row*64 + col], zero outside the image / beyond Co
#pragma unroll
    for (int i = 0; i < 4 * MT * SC_TH; ++i) {
        const int row = wid + 4 * i;
        const int o = row / SC_TH, ty = row - o * SC_TH;
        Gs[o * GLD + ty * SC_TW + lane] = gv[i];
    }
    __syncthreads();
    if (PF && more) tile_load(oy0 + SC_TH);

    // wave `wid` reduces over row `wid` of the tile: 64 pixels = 16 MFMA k-steps of 4 pixels (lane's pixel 4 ps + kq), in an
    // explicit two-deep pipeline: the fragments of step ps + 1 are read while the MFMAs of step ps run
    {
        const float* ga = Gs + il * GLD + wid * SC_TW + kq;
        float a0[MT], b0[NT], a1[MT], b1[NT];
        auto frag = [&](int ps, float (&fa)[MT], float (&fb)[NT]) {
#pragma unroll
            for (int m = 0; m < MT; ++m) fa[m] = ga[m * 16 * GLD + ps * 4];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) fb[nt] = Xh[kaddr[nt] + ps * 4];
        };
        auto mma = [&](const float (&fa)[MT], const float (&fb)[NT]) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int m = 0; m < MT; ++m)
                    acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[m], fb[nt], acc[m][nt], 0, 0, 0);
        };
        frag(0, a0, b0);
#pragma unroll
        for (int ps = 0; ps < SC_TW / 4; ps += 2) {
            frag(ps + 1, a1, b1);
            __builtin_amdgcn_sched_barrier(0);
            mma(a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            if (ps + 2 < SC_TW / 4) frag(ps + 2, a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            mma(a1, b1);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    }
    // sum the four waves' partials through LDS in a fixed order, then write the workgroup's slab
    __syncthreads();
    float* red = smem;                                   // reuse: [4 waves][16*MT][Kp16]
    const int slab_elems = 16 * MT * Kp16;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
            if (nt < ntiles)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    red[wid * slab_elems + (m * 16 + kq * 4 + r) * Kp16 + nt * 16 + il] = acc[m][nt][r];
    __syncthreads();
    const size_t wg = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    float* slab = slabs + wg * slab_elems;
    for (int e = tid; e < slab_elems; e += IG_THREADS)
        slab[e] = ((red[e] + red[slab_elems + e]) + red[2 * slab_elems + e]) + red[3 * slab_elems + e];
}

// gw[o][c][tap] = sum_z slabs[z][o][tap*C + c] in two fixed-order stages (bit-reproducible): thousands of workgroup
// slabs (one per tile column of 8 tiles) against a few thousand outputs -- one workgroup per 64 outputs walking all of
// them took 0.35 ms; stage 1 spreads the slabs over SC_RED_GROUPS workgroups per 64 outputs, stage 2 adds the groups.
constexpr int SC_RED_GROUPS = 64;
__global__ __launch_bounds__(256) void smallc_slab_reduce1_kernel(const float* __restrict__ slabs,
                                                                  float* __restrict__ partial, int Z, int slab_elems) {
    __shared__ float part[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + lane, grp = blockIdx.y;
    const int per = (Z + SC_RED_GROUPS - 1) / SC_RED_GROUPS;
    const int z0 = grp * per, z1 = min(Z, z0 + per);
    float s = 0.0f;
    if (e < slab_elems)
        for (int z = z0 + w; z < z1; z += 4) s += slabs[(size_t)z * slab_elems + e];
    part[w][lane] = s;
    __syncthreads();
    if (w == 0 && e < slab_elems)
        partial[(size_t)grp * slab_elems + e] = ((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane];
}
__global__ void smallc_slab_reduce2_kernel(const float* __restrict__ partial, float* __restrict__ gw, int slab_elems,
                                           int Kp16, int Co, int C, int T) {
    const int K = T * C;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Co * K) return;
    const int o = i / K, k = i - o * K;
    float s = 0.0f;
    for (int g = 0; g < SC_RED_GROUPS; ++g) s += partial[(size_t)g * slab_elems + (size_t)o * Kp16 + k];
    const int tap = k / C, c = k - tap * C;
    gw[((size_t)o * C + c) * T + tap] = s;
}

// Wp[k = tap*C + c][m] = W[m][c][tap]  (forward)  or, for the stride-1 input gradient computed as a
// forward conv of grad_y:  Wp[k = tap*Co + o][m = c] = W[o][c][T-1-tap]   (flipped taps, swapped channels)
__global__ void smallc_pack_kernel(const float* __restrict__ W, float* __restrict__ Wp, int Co, int C, int T, int Kp,
                                   int Mp, int transposed) {
    const int total = Kp * Mp;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int k = i / Mp, m = i - k * Mp;
        float v = 0.0f;
        if (!transposed) {
            if (k < T * C && m < Co) { const int tap = k / C, c = k - tap * C; v = W[((size_t)m * C + c) * T + tap]; }
        } else {
            if (k < T * Co && m < C) { const int tap = k / Co, o = k - tap * Co; v = W[((size_t)o * C + m) * T + (T - 1 - tap)]; }
        }
        Wp[i] = v;
    }
}

__global__ void smallc_koff_kernel(SmallGeom g, int* __restrict__ tab, int n) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) tab[k] = k_offset(g, k);
}

bool fill_small(SmallGeom& g, int B, int C, int H, int W, int Co, int kh, int kw, int s, int ph, int pw) {
    g.B = B; g.C = C; g.H = H; g.W = W; g.Co = Co; g.kh = kh; g.kw = kw; g.s = s; g.ph = ph; g.pw = pw;
    g.Ho = (H + 2 * ph - kh) / s + 1;
    g.Wo = (W + 2 * pw - kw) / s + 1;
    g.K = C * kh * kw;
    g.Kp = (g.K + 3) / 4 * 4;
    g.HR = (SC_TH - 1) * s + kh;
    g.HC = (SC_TW - 1) * s + kw;
    g.plane = g.HR * g.HC;
    if ((g.plane & 1) == 0) g.plane += 1;    // odd plane stride: lanes that differ in c hit different banks
    return true;
}

}  // namespace

// ---- entry points used by conv.hip -----------------------------------------------------------------
bool smallc_supported(int C, int Co, int kh, int kw, int sh, int sw) {
    // stride 2 needs a halo tile of ~75 KB (one workgroup per CU): measured slower than the generic kernels
    // (square 3x3 / 7x7 filters: the halo geometry is compiled in, ScShape)
    const int cpw = (C + 3) / 4;
    return sh == 1 && sw == 1 && kh == kw && (kh == 3 || kh == 7) && C <= 16 && Co <= 32 && C * kh * kw <= SC_MAXK - 12 &&
           cpw <= (kh == 3 ? ScShape<3>::CPW : ScShape<7>::CPW);
}
bool smallc_norm_supported(int C, int Co, int kh, int kw, int sh, int sw) {
    // (one row tile of output channels, 3x3: DLA-34's level0 behind the stem's BatchNorm; K = 9 C must fill more than four
    // column tiles -- the weight-gradient instance compiled with the transform is the nine-tile one)
    return smallc_supported(C, Co, kh, kw, sh, sw) && kh == 3 && Co <= 16 && (C * 9 + 15) / 16 > 4;
}
size_t smallc_workspace_bytes(int B, int C, int H, int W, int Co, int kh, int kw, int s, int ph, int pw) {
    SmallGeom g;
    fill_small(g, B, C, H, W, Co, kh, kw, s, ph, pw);
    const int mt = (Co + 15) / 16, Kp16 = (g.K + 15) / 16 * 16;
    const size_t tiles = (size_t)ceil_div(g.Wo, SC_TW) * ceil_div(g.Ho, SC_TH * SC_NV) * B;
    return carve_bytes((size_t)g.Kp * 16 * mt, 4) + carve_bytes(SC_MAXK, 4) + carve_bytes(tiles * 16 * mt * Kp16, 4) +
           carve_bytes((size_t)SC_RED_GROUPS * 16 * mt * Kp16, 4) + 512;
}

// BatchNorm statistics of a forward call (stats argument of smallc_forward): one block per (image, output row, 64-column
// tile) -- *blocks_per_image of them, *rows channels each; 0 when the scalar epilogue would run (Wo % 4 != 0)
int smallc_stats_blocks(int B, int C, int H, int W, int Co, int kh, int kw, int s, int ph, int pw, int* blocks_per_image,
                        int* rows) {
    SmallGeom g;
    fill_small(g, B, C, H, W, Co, kh, kw, s, ph, pw);
    if ((g.Wo & 3) != 0) return 0;
    if (blocks_per_image) *blocks_per_image = g.Ho * ceil_div(g.Wo, SC_TW);
    if (rows) *rows = 16 * ((Co + 15) / 16);
    return SC_TW;
}

int smallc_forward(const float* x, const float* w, const float* bias, float* y, int B, int C, int H, int W, int Co,
                   int kh, int kw, int s, int ph, int pw, float act_slope, int transposed, void* ws, size_t ws_bytes,
                   hipStream_t st, float* stats, const SmallNorm* norm) {
    // transposed: computes the stride-1 input gradient: x := grad_y [B, Co_orig, H, W], w is the ORIGINAL
    // weight [Co_orig = C here][C_orig = Co here][kh][kw]; the caller passes C/Co already swapped.
    SmallGeom g;
    fill_small(g, B, C, H, W, Co, kh, kw, s, ph, pw);
    const int mt = (Co + 15) / 16;
    Carver cv(ws, ws_bytes);
    float* Wp = cv.take<float>((size_t)g.Kp * 16 * mt);
    int* koff = cv.take<int>(SC_MAXK);
    CNUDA_REQUIRE(cv.ok(), "smallc_forward: workspace too small");
    CNUDA_LAUNCH(smallc_koff_kernel, dim3(1), dim3(SC_MAXK), 0, st, g, koff, SC_MAXK);
    // pack: forward W[Co][C][T]; transposed: original W[C_here... ] see kernel comment
    if (!transposed)
        CNUDA_LAUNCH(smallc_pack_kernel, dim3(32), dim3(256), 0, st, w, Wp, Co, C, kh * kw, g.Kp, 16 * mt, 0);
    else
        CNUDA_LAUNCH(smallc_pack_kernel, dim3(32), dim3(256), 0, st, w, Wp, /*Co_orig=*/C, /*C_orig=*/Co,
                           kh * kw, g.Kp, 16 * mt, 1);
    // halo image + 4 staging tiles + weights [Kp][16 mt] + row offsets [Kp]
    const size_t lds = (size_t)(g.C * g.plane + 16 + 4 * 16 * mt * SC_YLD + g.Kp * 16 * mt + g.Kp) * sizeof(float);
    const dim3 grid(ceil_div(g.Wo, SC_TW), ceil_div(g.Ho, SC_TH * SC_NV), B);
    ProfScope prof(st);
    prof.name("smallc_fwd_kernel<%d>", mt);
    CNUDA_REQUIRE(kh == kw && (kh == 3 || kh == 7) && g.plane == (kh == 3 ? ScShape<3>::plane : ScShape<7>::plane),
                  "smallc_forward: filter size without a compiled halo geometry");
    const SmallNorm none{nullptr, nullptr, nullptr, nullptr, 1};
#define CNUDA_SC_FWD(MTV, KHV) \
    CNUDA_LAUNCH((smallc_fwd_kernel<MTV, KHV>), grid, dim3(IG_THREADS), lds, st, g, x, Wp, koff, bias, y, act_slope, stats, none)
    if (norm) {
        CNUDA_REQUIRE(!transposed && smallc_norm_supported(C, Co, kh, kw, s, s) && norm->mean && norm->invstd && norm->gamma &&
                      norm->beta && norm->imgs_per_group > 0, "smallc_forward: apply-on-load is not compiled for this geometry");
        CNUDA_LAUNCH((smallc_fwd_kernel<1, 3, true>), grid, dim3(IG_THREADS), lds, st, g, x, Wp, koff, bias, y, act_slope, stats,
                     *norm);
    } else if (mt == 1) { if (kh == 3) CNUDA_SC_FWD(1, 3); else CNUDA_SC_FWD(1, 7); }
    else         { if (kh == 3) CNUDA_SC_FWD(2, 3); else CNUDA_SC_FWD(2, 7); }
#undef CNUDA_SC_FWD
    return check_launch("smallc_forward");
}

int smallc_backward_weight(const float* x, const float* gy, float* gw, int B, int C, int H, int W, int Co, int kh,
                           int kw, int s, int ph, int pw, void* ws, size_t ws_bytes, hipStream_t st,
                           const SmallNorm* norm) {
    SmallGeom g;
    fill_small(g, B, C, H, W, Co, kh, kw, s, ph, pw);
    const int mt = (Co + 15) / 16, Kp16 = (g.K + 15) / 16 * 16;
    const dim3 grid(ceil_div(g.Wo, SC_TW), ceil_div(g.Ho, SC_TH * SC_NV), B);
    const size_t tiles = (size_t)grid.x * grid.y * grid.z;
    Carver cv(ws, ws_bytes);
    (void)cv.take<float>((size_t)g.Kp * 16 * mt);
    int* koff = cv.take<int>(SC_MAXK);
    float* slabs = cv.take<float>(tiles * 16 * mt * Kp16);
    float* partial = cv.take<float>((size_t)SC_RED_GROUPS * 16 * mt * Kp16);
    CNUDA_REQUIRE(cv.ok(), "smallc_backward_weight: workspace too small");
    CNUDA_LAUNCH(smallc_koff_kernel, dim3(1), dim3(SC_MAXK), 0, st, g, koff, SC_MAXK);
    const size_t stage = (size_t)(g.C * g.plane + 16) + (size_t)16 * mt * (SC_TH * SC_TW + 1);
    const size_t red = (size_t)4 * 16 * mt * Kp16;
    const size_t lds = (stage > red ? stage : red) * sizeof(float);
    CNUDA_REQUIRE(kh == kw && (kh == 3 || kh == 7) && g.plane == (kh == 3 ? ScShape<3>::plane : ScShape<7>::plane),
                  "smallc_backward_weight: filter size without a compiled halo geometry");
    // column tiles the instance computes: all of a 16-channel 3x3 (9) / a 3-channel 7x7 (10) filter, or 4 for the small
    // shapes (K <= 64)
    const SmallNorm none{nullptr, nullptr, nullptr, nullptr, 1};
    const int nt_real = Kp16 / 16, nt_full = kh == 3 ? 9 : 10;
    CNUDA_REQUIRE(nt_real <= nt_full, "smallc_backward_weight: K exceeds the compiled column tiles");
    {
        ProfScope prof(st);
        prof.name("smallc_wgrad_kernel<%d>", mt);
#define CNUDA_SC_WG(MTV, KHV, NTV) do {                                                                              \
        CNUDA_REQUIRE(raise_dynamic_lds(reinterpret_cast<const void*>(smallc_wgrad_kernel<MTV, KHV, NTV>), lds),       \
                      "smallc_backward_weight: dynamic LDS");                                                         \
        CNUDA_LAUNCH((smallc_wgrad_kernel<MTV, KHV, NTV>), grid, dim3(IG_THREADS), lds, st, g, x, gy, koff, slabs, Kp16, none); \
    } while (0)
#define CNUDA_SC_WG_KH(MTV) do {                                                                                     \
        if (kh == 3) { if (nt_real <= 4) CNUDA_SC_WG(MTV, 3, 4); else CNUDA_SC_WG(MTV, 3, 9); }                       \
        else         { if (nt_real <= 4) CNUDA_SC_WG(MTV, 7, 4); else CNUDA_SC_WG(MTV, 7, 10); }                      \
    } while (0)
        if (norm) {
            CNUDA_REQUIRE(smallc_norm_supported(C, Co, kh, kw, s, s) && nt_real > 4 && norm->mean && norm->invstd &&
                          norm->gamma && norm->beta && norm->imgs_per_group > 0,
                          "smallc_backward_weight: apply-on-load is not compiled for this geometry");
            CNUDA_REQUIRE(raise_dynamic_lds(reinterpret_cast<const void*>(smallc_wgrad_kernel<1, 3, 9, true>), lds),
                          "smallc_backward_weight: dynamic LDS");
            CNUDA_LAUNCH((smallc_wgrad_kernel<1, 3, 9, true>), grid, dim3(IG_THREADS), lds, st, g, x, gy, koff, slabs, Kp16, *norm);
        } else if (mt == 1) CNUDA_SC_WG_KH(1); else CNUDA_SC_WG_KH(2);
#undef CNUDA_SC_WG_KH
#undef CNUDA_SC_WG
    }
    if (int rc = check_launch("smallc_backward_weight")) return rc;
    const int slab_elems = 16 * mt * Kp16;
    CNUDA_LAUNCH(smallc_slab_reduce1_kernel, dim3((slab_elems + 63) / 64, SC_RED_GROUPS), dim3(256), 0, st, slabs,
                       partial, (int)tiles, slab_elems);
    CNUDA_LAUNCH(smallc_slab_reduce2_kernel, dim3((Co * g.K + 255) / 256), dim3(256), 0, st, partial, gw, slab_elems,
                       Kp16, Co, C, kh * kw);
    return check_launch("smallc_backward_weight(reduce)");
}

}  // namespace cnuda
