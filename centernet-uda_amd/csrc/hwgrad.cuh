// Halo-tile weight gradient for 3x3 / stride 1 / padding 1 convolutions with few output channels (Co <= 32: the 27-channel
// offset / mask convolutions of the 16 DCN layers, libs/DCNv2/dcn_v2.py:104-110) on the fp32 MFMA.
//
//   gw[o][c][r][s] = sum over (b, y, x) of  gy[b][o][y][x] * x[b][c][y + r - 1][x + s - 1]
//
// The im2col-style kernel (igemm_wgrad_kernel<., 32, 128>) gathers every tap's column tile from global memory: nine
// 4-byte loads of (nearly) the same input element, each feeding ONE 32-row MFMA tile -- 20 loads and 20 LDS stores per
// thread for 16 MFMAs per wave, 59-66 TFLOP/s.  Here a workgroup owns a group of 16 input channels and a range of
// 256-pixel tiles (rectangles of one image, full rows where W <= 128 is a power of two): per tile it stages the group's INPUT rows once with their one-pixel halo
// (16-byte loads) and the grad_y tile [32][256] once, and every wave reduces a quarter of the tile's pixels for all nine
// taps on v_mfma_f32_16x16x4_f32 (rows = output channels, two tiles of 16; columns = the 16 channels; k = 4 consecutive
// pixels): the tap only shifts the LDS address of the B fragment.  One A read serves nine taps, one B read two row tiles:
// 11 LDS reads per 18 MFMAs, no global gather at all.  The four waves' partial sums meet through LDS in a fixed order
// (wave 0 + 1 + 2 + 3) and the workgroup writes its 32 x 144 block of the split's slab; slab_reduce_* adds the splits
// (bitwise reproducible, no float atomics) -- the slab layout, the bias row sums and the reduction launch are those of
// igemm_wgrad_kernel.
//
// Host-side conditions (conv.hip hwgrad_ok): 3x3, stride 1, padding 1, Co <= 32, C % 16 == 0, W % 8 == 0 (round 6: the
// tile is a rectangle of TR rows x TW columns, TW the largest power of two dividing W; rounds 4-5 took W in {16, 32, 64,
// 128} with tiles of full rows), both tensors below 2 GiB, f32 matrix mode.
#pragma once
#include "igemm.cuh"

namespace cnuda {

constexpr int HW_BN = 256;                       // pixels per tile
// Tile = TR rows x TW columns of one image, TW = the largest power of two that divides the row width (hconv.cuh
// halo_tile_width).  SIDE: the tile does not span the row (TW < W) -- its left / right neighbour columns are staged as one
// more 16-byte cell on either side (zero at the image edge) instead of the two constant zero columns.  A tile may hang
// over the last image row: the rows below the image stage zeros (input and grad_y alike: nothing is added).
template <int TW, bool SIDE> struct HwShape {
    static constexpr int TR = HW_BN / TW;        // image rows per tile
    // LDS row: tile column j at j + 4 (16-byte aligned interior), the halo columns at 3 and TW + 4
    static constexpr int RS = TW + 8;
    // channel-plane stride == 2 (mod 32): a B fragment read has lanes (pixel kq, channel il) -> cell il * PL + kq + const,
    // and the 32 lanes of a half-wave (kq in {0, 1}) then fall on 32 different banks.  Same for the grad_y rows.
    static constexpr int PL = ((TR + 2) * RS + 29) / 32 * 32 + 2;
    static constexpr int GLD = HW_BN + 2;
    static constexpr int CPR = TW / 4 + (SIDE ? 2 : 0), CPP = (TR + 2) * CPR, XCELLS = 16 * CPP;   // 16-byte cells per row / plane / group
    static constexpr int XPER = (XCELLS + IG_THREADS - 1) / IG_THREADS;
    static constexpr int GPER = 32 * (HW_BN / 4) / IG_THREADS;                       // 8
    static constexpr size_t lds_floats = (size_t)16 * PL + (size_t)32 * GLD;
    static_assert(PL % 32 == 2 && PL >= (TR + 2) * RS && GLD % 32 == 2, "bank layout");
    static_assert(32 * 146 <= (int)lds_floats, "the cross-wave reduction reuses the staging area");
};

struct HwParams {
    const float* x;         // [B][C][H][W]
    const float* gy;        // [B][Co][H][W]
    int B, C, H, W, Co;
    int tiles_x, tiles_y;   // W / TW, ceil(H / TR)
    int n_tiles, tiles_per_split;
};

template <int TW, bool SIDE>
__global__ __launch_bounds__(IG_THREADS, 2) void hwgrad_kernel(HwParams p, float* __restrict__ slabs, int Mp, int Jp,
                                                               float* __restrict__ bslab) {
    using S = HwShape<TW, SIDE>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const Hs = smem;                          // [16][PL]
    float* const Gs = smem + 16 * S::PL;             // [32][GLD]
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kq = lane >> 4, il = lane & 15;
    const int grp = blockIdx.x, z = blockIdx.y;
    const int W = p.W, HWp = p.H * W, tiles_per_image = p.tiles_x * p.tiles_y;
    const int t0 = z * p.tiles_per_split;
    const int t1 = t0 + p.tiles_per_split < p.n_tiles ? t0 + p.tiles_per_split : p.n_tiles;
    const buf_rsrc rx = ig_make_rsrc(p.x, (unsigned)((size_t)p.B * p.C * HWp * sizeof(float)));
    const buf_rsrc rg = ig_make_rsrc(p.gy, (unsigned)((size_t)p.B * p.Co * HWp * sizeof(float)));

    // input cells of this thread: byte offset relative to (image, channel group, tile origin) -- negative for the row above
    // and the cell left of the tile, so the tile's origin is added per lane and only the non-negative (image, group) term is
    // the scalar offset -- LDS cell, the cell's window row (0 = the row above the tile; rows beyond the image are decided
    // per tile) and whether it is the cell left / right of the tile (decided per tile by two scalar flags)
    unsigned xv[S::XPER];
    int xl[S::XPER], xrow[S::XPER], xside[S::XPER];
#pragma unroll
    for (int i = 0; i < S::XPER; ++i) {
        const int e = tid + i * IG_THREADS;
        xv[i] = IG_BUF_OOB; xl[i] = -1; xrow[i] = 0; xside[i] = 0;
        if (e < S::XCELLS) {
            const int c = e / S::CPP, rem = e - c * S::CPP, row = rem / S::CPR, q = rem - row * S::CPR;
            const int qq = SIDE ? q - 1 : q;                                                    // (-1: left of the tile, TW / 4: right)
            xv[i] = (unsigned)((c * HWp + (row - 1) * W + 4 * qq) * (int)sizeof(float));        // (row 0: one row above the tile)
            xl[i] = c * S::PL + row * S::RS + 4 + 4 * qq;
            xrow[i] = row;
            xside[i] = qq < 0 ? 1 : (qq == TW / 4 ? 2 : 0);
        }
    }
    // grad_y cells: rows wid, wid + 4, ... (wave-uniform), lane = 16-byte cell of the tile's 256 pixels: tile row 4 lane / TW
    const int grow = (4 * lane) / TW;
    const unsigned gvoff = (unsigned)((grow * W + (4 * lane) % TW) * (int)sizeof(float));
    // a tile that spans the row: the halo columns left and right of the image are zero for every tile, written once
    if (!SIDE)
        for (int e = tid; e < 16 * (S::TR + 2) * 2; e += IG_THREADS) {
            const int side = e & 1, cr = e >> 1, c = cr / (S::TR + 2), row = cr - c * (S::TR + 2);
            Hs[c * S::PL + row * S::RS + (side ? TW + 4 : 3)] = 0.0f;
        }

    f32x4 acc[2][9];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    // bias gradient (bslab != nullptr): row sums of grad_y from the staging registers, by the workgroups of channel group 0
    const bool do_bias = bslab != nullptr && grp == 0;
    float bs[S::GPER];
#pragma unroll
    for (int i = 0; i < S::GPER; ++i) bs[i] = 0.0f;

    f32x4 xr[S::XPER], gr[S::GPER];
    auto tile_load = [&](int t) {
        const int b = t / tiles_per_image, tt = t - b * tiles_per_image, ty = tt / p.tiles_x, tx = tt - ty * p.tiles_x;
        const int y0 = ty * S::TR, x0 = tx * TW;
        const bool top_ok = y0 > 0, left_ok = x0 > 0, right_ok = x0 + TW < W;
        const int rows_in = p.H - y0;                 // window rows 1 .. rows_in lie inside the image (row r = image row y0 + r - 1)
        const unsigned xs = (unsigned)(((b * p.C + grp * 16) * HWp) * (int)sizeof(float));
        const unsigned ys = (unsigned)((y0 * W + x0) * (int)sizeof(float));
#pragma unroll
        for (int i = 0; i < S::XPER; ++i) {
            const bool out = xl[i] < 0 || (xrow[i] == 0 && !top_ok) || xrow[i] > rows_in ||
                             (SIDE && ((xside[i] == 1 && !left_ok) || (xside[i] == 2 && !right_ok)));
            xr[i] = ig_buf_load4(rx, out ? IG_BUF_OOB : xv[i] + ys, xs);
        }
        const bool g_in = grow < rows_in;
#pragma unroll
        for (int i = 0; i < S::GPER; ++i) {
            const int m = wid + 4 * i;
            const bool ok = m < p.Co;
            gr[i] = ig_buf_load4(rg, ok && g_in ? gvoff : IG_BUF_OOB,
                                 ok ? (unsigned)(((b * p.Co + m) * HWp + y0 * W + x0) * (int)sizeof(float)) : 0u);
        }
    };
    auto tile_store = [&]() {
#pragma unroll
        for (int i = 0; i < S::XPER; ++i)
            if (xl[i] >= 0) {        // (plane stride == 2 mod 4: 8-byte aligned cells)
                *reinterpret_cast<float2*>(Hs + xl[i]) = make_float2(xr[i][0], xr[i][1]);
                *reinterpret_cast<float2*>(Hs + xl[i] + 2) = make_float2(xr[i][2], xr[i][3]);
            }
#pragma unroll
        for (int i = 0; i < S::GPER; ++i) {
            float* d = Gs + (wid + 4 * i) * S::GLD + 4 * lane;
            *reinterpret_cast<float2*>(d) = make_float2(gr[i][0], gr[i][1]);
            *reinterpret_cast<float2*>(d + 2) = make_float2(gr[i][2], gr[i][3]);
            if (do_bias) bs[i] += (gr[i][0] + gr[i][1]) + (gr[i][2] + gr[i][3]);
        }
    };

    // wave `wid` reduces pixels [64 wid, 64 wid + 64) of the tile: 16 k-steps of 4 pixels.  Pixel 64 wid sits at
    // (row, column) = (64 wid / TW, 64 wid % TW); step ks moves on by 4 ks pixels -- a compile-time (row, column) delta.
    const int prow = (64 * wid) / TW, pcol = (64 * wid) % TW;
    const float* const bbase = Hs + il * S::PL + prow * S::RS + pcol + 3 + kq;      // tap (0, 0) of pixel kq of step 0
    const float* const abase = Gs + il * S::GLD + 64 * wid + kq;

    if (t0 < t1) tile_load(t0);
    for (int t = t0; t < t1; ++t) {
        __syncthreads();                 // the previous tile's fragment reads are done
        tile_store();
        __syncthreads();
        if (t + 1 < t1) tile_load(t + 1);          // in flight under this tile's MFMAs
        float a0[2], b0[9], a1[2], b1[9];
        auto frag = [&](int ks, float (&fa)[2], float (&fb)[9]) {
            const int d = ((4 * ks) / TW) * S::RS + (4 * ks) % TW;      // (compile-time: ks is unrolled)
            fa[0] = abase[4 * ks];
            fa[1] = abase[16 * S::GLD + 4 * ks];
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int s = 0; s < 3; ++s) fb[r * 3 + s] = bbase[d + r * S::RS + s];
        };
        auto mma = [&](const float (&fa)[2], const float (&fb)[9]) {
#pragma unroll
            for (int tp = 0; tp < 9; ++tp)
#pragma unroll
                for (int m = 0; m < 2; ++m)
                    acc[m][tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[m], fb[tp], acc[m][tp], 0, 0, 0);
        };
        frag(0, a0, b0);
#pragma unroll
        for (int ks = 0; ks < 16; ks += 2) {
            frag(ks + 1, a1, b1);
            __builtin_amdgcn_sched_barrier(0);
            mma(a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            if (ks + 2 < 16) frag(ks + 2, a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            mma(a1, b1);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // the four waves' partial sums, added in the fixed order ((w0 + w1) + w2) + w3 through LDS
    constexpr int RLD = 146;
    float* const red = smem;             // [32][RLD]: row = output channel, column = tap * 16 + channel of the group
    for (int w = 0; w < 4; ++w) {
        __syncthreads();
        if (wid == w) {
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int tp = 0; tp < 9; ++tp)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float* d = red + (m * 16 + kq * 4 + r) * RLD + tp * 16 + il;
                        *d = w == 0 ? acc[m][tp][r] : *d + acc[m][tp][r];
                    }
        }
    }
    __syncthreads();
    float* const slab = slabs + (size_t)z * Mp * Jp;
    for (int e = tid; e < 32 * 144; e += IG_THREADS) {
        const int m = e / 144, jj = e - m * 144, tp = jj >> 4, c = jj & 15;
        if (m < Mp) slab[(size_t)m * Jp + tp * p.C + grp * 16 + c] = red[m * RLD + jj];
    }
    if (do_bias) {
#pragma unroll
        for (int i = 0; i < S::GPER; ++i) {
            float v = bs[i];
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
            const int m = wid + 4 * i;
            if (lane == 0 && m < Mp) bslab[(size_t)z * Mp + m] = v;
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// The same reduction for a 3x3 / STRIDE 2 / padding 1 convolution with <= 32 output channels (DLA-34's level1, 16 -> 32
// at 512 x 512 -> 256 x 256, backends/dla.py:233-241): the im2col-style kernel pads its K = 144 columns to 256 and gathers
// 4 bytes at a time (35 TFLOP/s).  Tile = 128 consecutive output pixels of one output row; its three input rows
// (2 oy - 1 .. 2 oy + 1) are staged DE-INTERLEAVED by column parity -- a 16-byte cell of four input columns goes as two
// 8-byte stores into the even and the odd plane -- so that the four consecutive output pixels of an MFMA k-step read
// consecutive LDS cells for every tap: tap column s = 1 is the even plane at ox, s = 2 the odd plane at ox, s = 0 the odd
// plane at ox - 1 (the column left of the tile: a real column for the tile's right-hand neighbours, zero at the image
// edge).  Everything else -- rows = output channels in two tiles of 16, columns = 16 input channels, k = pixels, the waves'
// fixed-order sum, slabs, bias row sums -- as hwgrad_kernel.
// Host-side conditions (conv.hip hwgrad_s2_ok): 3x3, stride 2, padding 1, even H and W, Co <= 64 (blockIdx.z: 32-row block), C % 16 == 0,
// Wo % 4 == 0 (round 6: the last tile of an output row may be ragged -- its missing columns stage zeros; round 5: Wo % 128
// == 0), both tensors below 2 GiB, f32 matrix mode.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int HS_BN = 128;                                   // output pixels per tile
constexpr int HS_PW = HS_BN + 8;                             // a parity plane of one row: output column j at j + 4, column -1 at 3
constexpr int HS_ROW = 2 * HS_PW;                            // even plane | odd plane
constexpr int HS_PL = (3 * HS_ROW + 29) / 32 * 32 + 2;       // channel stride == 2 (mod 32), see HwShape
constexpr int HS_GLD = HS_BN + 2;
constexpr size_t HS_LDS_FLOATS = (size_t)16 * HS_PL + (size_t)32 * HS_GLD;
static_assert(32 * 146 <= (int)HS_LDS_FLOATS, "the cross-wave reduction reuses the staging area");

struct HwS2Params {
    const float* x;         // [B][C][H][W], H = 2 Ho, W = 2 Wo
    const float* gy;        // [B][Co][Ho][Wo]
    int B, C, H, W, Co, Ho, Wo;
    int n_tiles, tiles_per_split;
};

__global__ __launch_bounds__(IG_THREADS, 2) void hwgrad_s2_kernel(HwS2Params p, float* __restrict__ slabs, int Mp, int Jp,
                                                                  float* __restrict__ bslab) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const Hs = smem;                          // [16][3 rows][even | odd][HS_PW]
    float* const Gs = smem + 16 * HS_PL;             // [32][HS_GLD]
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kq = lane >> 4, il = lane & 15;
    const int grp = blockIdx.x, z = blockIdx.y;
    const int m_off = blockIdx.z * 32;               // output channels [m_off, m_off + 32) (two row blocks for 33..64 channels)
    const int HWp = p.H * p.W, HoWo = p.Ho * p.Wo, halves = (p.Wo + HS_BN - 1) / HS_BN;   // (the last tile of a row may be ragged)
    const int t0 = z * p.tiles_per_split;
    const int t1 = t0 + p.tiles_per_split < p.n_tiles ? t0 + p.tiles_per_split : p.n_tiles;
    const buf_rsrc rx = ig_make_rsrc(p.x, (unsigned)((size_t)p.B * p.C * HWp * sizeof(float)));
    const buf_rsrc rg = ig_make_rsrc(p.gy, (unsigned)((size_t)p.B * p.Co * HoWo * sizeof(float)));

    // input cells of this thread: 16 channels x 3 rows x 64 cells of four input columns -> twelve per thread; byte offset
    // relative to (image, group, row 2 oy, column 256 h) -- the row above is at -W, added per lane -- and the LDS cell
    constexpr int XPER = 16 * 3 * 64 / IG_THREADS;
    unsigned xv[XPER];
    int xl[XPER], xcol[XPER];
    bool xtop[XPER];
#pragma unroll
    for (int i = 0; i < XPER; ++i) {
        const int e = tid + i * IG_THREADS;
        const int c = e / 192, rem = e - c * 192, row = rem >> 6, q = rem & 63;
        xv[i] = (unsigned)((c * HWp + (row - 1) * p.W + 4 * q) * (int)sizeof(float));
        xl[i] = c * HS_PL + row * HS_ROW + 4 + 2 * q;
        xtop[i] = row == 0;
        xcol[i] = 4 * q;                                         // first of the cell's four input columns, relative to the tile
    }
    // the column left of the tile (odd plane, index -1): one element per (channel, row), threads 0..47
    const int hc = tid / 3, hr = tid - hc * 3;
    const bool hon = tid < 48;
    const int hl = hc * HS_PL + hr * HS_ROW + HS_PW + 3;
    // grad_y cells: 32 rows x 32 cells -> four per thread (row tid / 32 + 8 i, cell tid % 32)
    constexpr int GPER = 32 * (HS_BN / 4) / IG_THREADS;
    const int gm0 = tid >> 5, gq = tid & 31;

    f32x4 acc[2][9];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool do_bias = bslab != nullptr && grp == 0;
    float bs[GPER];
#pragma unroll
    for (int i = 0; i < GPER; ++i) bs[i] = 0.0f;

    f32x4 xr[XPER], gr[GPER];
    float hv = 0.0f;
    auto tile_load = [&](int t) {
        const int b = t / (p.Ho * halves), rem = t - b * (p.Ho * halves), oy = rem / halves, h = rem - oy * halves;
        const bool top_ok = oy > 0;                              // (input row 2 oy - 1; the row below, 2 oy + 1, always exists)
        const unsigned xs = (unsigned)(((b * p.C + grp * 16) * HWp) * (int)sizeof(float));
        const unsigned ys = (unsigned)((2 * oy * p.W + 2 * HS_BN * h) * (int)sizeof(float));
        const int cols_in = p.W - 2 * HS_BN * h, ocols_in = p.Wo - HS_BN * h;      // columns of the tile inside the image
#pragma unroll
        for (int i = 0; i < XPER; ++i)
            xr[i] = ig_buf_load4(rx, ((xtop[i] && !top_ok) || xcol[i] >= cols_in) ? IG_BUF_OOB : xv[i] + ys, xs);
        {
            const bool ok = hon && h > 0 && (hr > 0 || top_ok);
            hv = ig_buf_load(rx, ok ? (unsigned)((hc * HWp + (hr - 1) * p.W - 1) * (int)sizeof(float)) + ys : IG_BUF_OOB, xs);
        }
        const unsigned gs = (unsigned)(((b * p.Co) * HoWo + oy * p.Wo + HS_BN * h) * (int)sizeof(float));
#pragma unroll
        for (int i = 0; i < GPER; ++i) {
            const int m = m_off + gm0 + 8 * i;
            gr[i] = ig_buf_load4(rg, (m < p.Co && 4 * gq < ocols_in) ? (unsigned)((m * HoWo + 4 * gq) * (int)sizeof(float)) : IG_BUF_OOB, gs);
        }
    };
    auto tile_store = [&]() {
#pragma unroll
        for (int i = 0; i < XPER; ++i) {                         // columns 4q, 4q + 2 -> even plane; 4q + 1, 4q + 3 -> odd plane
            *reinterpret_cast<float2*>(Hs + xl[i]) = make_float2(xr[i][0], xr[i][2]);
            *reinterpret_cast<float2*>(Hs + xl[i] + HS_PW) = make_float2(xr[i][1], xr[i][3]);
        }
        if (hon) Hs[hl] = hv;
#pragma unroll
        for (int i = 0; i < GPER; ++i) {
            float* d = Gs + (gm0 + 8 * i) * HS_GLD + 4 * gq;
            *reinterpret_cast<float2*>(d) = make_float2(gr[i][0], gr[i][1]);
            *reinterpret_cast<float2*>(d + 2) = make_float2(gr[i][2], gr[i][3]);
            if (do_bias) bs[i] += (gr[i][0] + gr[i][1]) + (gr[i][2] + gr[i][3]);
        }
    };

    // wave `wid` reduces output pixels [32 wid, 32 wid + 32) of the tile: 8 k-steps of 4 pixels
    const float* const bbase = Hs + il * HS_PL + 4 + 32 * wid + kq;
    const float* const abase = Gs + il * HS_GLD + 32 * wid + kq;

    if (t0 < t1) tile_load(t0);
    for (int t = t0; t < t1; ++t) {
        __syncthreads();
        tile_store();
        __syncthreads();
        if (t + 1 < t1) tile_load(t + 1);
        float a0[2], b0[9], a1[2], b1[9];
        auto frag = [&](int ks, float (&fa)[2], float (&fb)[9]) {
            fa[0] = abase[4 * ks];
            fa[1] = abase[16 * HS_GLD + 4 * ks];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const float* row = bbase + r * HS_ROW + 4 * ks;
                fb[r * 3 + 0] = row[HS_PW - 1];                  // input column 2 ox - 1: odd plane at ox - 1
                fb[r * 3 + 1] = row[0];                          //              2 ox    : even plane at ox
                fb[r * 3 + 2] = row[HS_PW];                      //              2 ox + 1: odd plane at ox
            }
        };
        auto mma = [&](const float (&fa)[2], const float (&fb)[9]) {
#pragma unroll
            for (int tp = 0; tp < 9; ++tp)
#pragma unroll
                for (int m = 0; m < 2; ++m)
                    acc[m][tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[m], fb[tp], acc[m][tp], 0, 0, 0);
        };
        frag(0, a0, b0);
#pragma unroll
        for (int ks = 0; ks < 8; ks += 2) {
            frag(ks + 1, a1, b1);
            __builtin_amdgcn_sched_barrier(0);
            mma(a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            if (ks + 2 < 8) frag(ks + 2, a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            mma(a1, b1);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    constexpr int RLD = 146;
    float* const red = smem;
    for (int w = 0; w < 4; ++w) {
        __syncthreads();
        if (wid == w) {
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int tp = 0; tp < 9; ++tp)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float* d = red + (m * 16 + kq * 4 + r) * RLD + tp * 16 + il;
                        *d = w == 0 ? acc[m][tp][r] : *d + acc[m][tp][r];
                    }
        }
    }
    __syncthreads();
    float* const slab = slabs + (size_t)z * Mp * Jp;
    for (int e = tid; e < 32 * 144; e += IG_THREADS) {
        const int m = e / 144, jj = e - m * 144, tp = jj >> 4, c = jj & 15;
        if (m_off + m < Mp) slab[(size_t)(m_off + m) * Jp + tp * p.C + grp * 16 + c] = red[m * RLD + jj];
    }
    if (do_bias) {
#pragma unroll
        for (int i = 0; i < GPER; ++i) {
            float v = bs[i];
#pragma unroll
            for (int d = 16; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);      // the 32 lanes of a row: one half of the wave
            const int m = m_off + gm0 + 8 * i;
            if ((tid & 31) == 0 && m < Mp) bslab[(size_t)z * Mp + m] = v;
        }
    }
}

}  // namespace cnuda
