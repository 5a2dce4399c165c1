// Halo-tile implicit GEMM for 3x3 / stride 1 / padding 1 convolutions (forward, and the input gradient, which is the
// same convolution over grad_y with flipped taps) on the fp32 MFMA.
//
// The im2col-style kernels of igemm.cuh gather every tap's B tile from global memory: nine loads of (nearly) the same
// input element per channel, nine LDS stores, and -- for the narrow GEMMs (27 output rows: the DCN offset
// convolutions) -- a texture-address unit that is as busy as the matrix pipe.  Here the K axis is ordered
// (16-channel group, tap, channel in group): a workgroup stages the INPUT tile of a channel group once -- its 128
// consecutive pixels (TR full rows of an image row width W <= 128) plus a one-pixel halo, 16-byte loads and LDS
// stores -- and the nine chunks of that group read their B fragments straight from that halo tile at (row + r,
// column + s): lane-consecutive ds_read_b32 like the im2col tile's, one wave-uniform address term per chunk.
// Per channel group a thread issues 2..6 16-byte loads instead of 72 4-byte ones.  The A (weight) tile is staged per
// chunk exactly as in igemm_fwd_kernel; the packed weights use the same K order (PACK_HALO_FWD / PACK_HALO_DGRAD).
//
// Pixel tile = BN (128 or 256) pixels of ONE image as a rectangle of TR rows x TW columns, TW = the largest power of two
// that divides the row width W (at most 128): a 128- / 64- / 32- / 16-wide map is tiled by full rows as in round 4, a
// 160- / 80- / 40-wide one (640 x 640 inputs) by 5 column tiles of 32 / 16 / 8 (round 6).  A tile that does not span the
// row stages four more columns on either side (the neighbouring tile's, or zero at the image edge) instead of the two
// constant zero columns; a tile that hangs over the last image row (H % TR != 0) stages zeros there and does not store.
// The four pixels of a lane's 16-byte epilogue store lie in one tile row (4 | TW).  Host-side conditions (conv.hip
// hconv_ok): 3x3, stride 1, padding 1, gathered channels % 16 == 0, W % 8 == 0, tensor below 2 GiB, f32 matrix mode.
#pragma once
#include "igemm.cuh"

namespace cnuda {

// Block tiles BM x BN (four waves; each wave TM x TN accumulator tiles of 32 x 32).  With the B staging nearly free the
// pixel tile can be 256 wide for the narrow GEMMs: twice the MFMAs per barrier and per A-tile load, half the tiles
// (prologue and store tail per tile) -- a 64 x 256 tile runs the 64 -> 64 layers like the 128 x 128 tile runs the wide ones.
template <int BM, int BN> struct HcTile;
template <> struct HcTile<128, 128> { static constexpr int WM = 2, WN = 2, TM = 2, TN = 2; };
template <> struct HcTile<64, 128>  { static constexpr int WM = 2, WN = 2, TM = 1, TN = 2; };
template <> struct HcTile<32, 128>  { static constexpr int WM = 1, WN = 4, TM = 1, TN = 1; };
template <> struct HcTile<64, 256>  { static constexpr int WM = 2, WN = 2, TM = 1, TN = 4; };
template <> struct HcTile<32, 256>  { static constexpr int WM = 1, WN = 4, TM = 1, TN = 2; };

struct HaloGeom {
    int Kc, H, W;              // gathered channels, plane size
    int TW, tw_shift;          // tile columns (a power of two that divides W), log2
    int TR;                    // tile rows (BN / TW)
    int tiles_x, tiles_y;      // W / TW, ceil(H / TR)
    int side;                  // 1: TW < W -- the tile's left / right neighbour columns are staged (one 16-byte cell each side)
    int RS, PL;                // LDS row stride and plane size of the halo tile, floats
    int cpr, cpp, cells;       // 16-byte cells per row (TW / 4 + 2 side), per plane ((TR + 2) * cpr), per channel group (16 * cpp)
};
constexpr int HC_MAXCELLS = 8;      // per thread: W = 128, 256-pixel tile -> 16 * 4 * 32 / 256

// tile width of a map W columns wide: the largest power of two that divides W, at most 128 (0: none of at least 8)
inline int halo_tile_width(int W) {
    int tw = 128;
    while (tw >= 8 && W % tw != 0) tw >>= 1;
    return tw >= 8 ? tw : 0;
}
inline HaloGeom make_halo_geom(int Kc, int H, int W, int bn) {
    HaloGeom h;
    h.Kc = Kc; h.H = H; h.W = W;
    h.TW = halo_tile_width(W);
    h.tw_shift = 0;
    while ((1 << h.tw_shift) < h.TW) ++h.tw_shift;
    h.TR = bn / h.TW;
    h.tiles_x = W / h.TW;
    h.tiles_y = (H + h.TR - 1) / h.TR;
    h.side = h.TW < W ? 1 : 0;
    // column index of tile column j is j + 4 (16-byte aligned interior), the halo columns are 3 and TW + 4.  A 32-pixel
    // MFMA column block spans 32 / TW tile rows: TW = 16 -> a stride of 48 puts the second row 16 banks away from the
    // first, TW = 8 -> a stride of 40 the four rows 8 banks apart
    h.RS = h.TW == 16 ? 48 : (h.TW == 8 ? 40 : h.TW + 8);
    h.PL = (h.TR + 2) * h.RS;
    h.cpr = h.TW / 4 + 2 * h.side;
    h.cpp = (h.TR + 2) * h.cpr;
    h.cells = 16 * h.cpp;
    return h;
}
// ONE halo buffer (occupancy: the MFMA loop wants three to four waves per SIMD; a second buffer for W = 128 costs two
// of four resident workgroups) + two A stages; the epilogue's staging tiles must fit as well
inline size_t hconv_lds_bytes(const HaloGeom& h, int bm) {
    size_t fl = (size_t)16 * h.PL + 2 * IG_KC * bm;
    if (fl < (size_t)4 * IG_EPI_WAVE) fl = 4 * IG_EPI_WAVE;
    return fl * sizeof(float);
}

// One 16-deep chunk = one tap (r, s) of one channel group: A fragments from the staged weight tile, B fragments from
// the halo tile.  Same two-deep register pipeline as ig_mma_chunk.
template <int BM, int BN>
__device__ __forceinline__ void hc_mma_chunk(const float* __restrict__ As, const float* __restrict__ Hb,
                                             f32x16 (&acc)[HcTile<BM, BN>::TM][HcTile<BM, BN>::TN], int wm_off,
                                             const int (&boff)[HcTile<BM, BN>::TN], int PL, int lane) {
    using T = HcTile<BM, BN>;
    const int kl = lane >> 5, il = lane & 31;
    const float* ap = As + kl * BM + wm_off + il;
    const float* bp[T::TN];
#pragma unroll
    for (int j = 0; j < T::TN; ++j) bp[j] = Hb + kl * PL + boff[j];
    float a[2][T::TM], b[2][T::TN];
    auto frag = [&](int kk, float (&fa)[T::TM], float (&fb)[T::TN]) {
#pragma unroll
        for (int i = 0; i < T::TM; ++i) fa[i] = ap[kk * BM + i * 32];
#pragma unroll
        for (int j = 0; j < T::TN; ++j) fb[j] = bp[j][kk * PL];
    };
    auto mma = [&](const float (&fa)[T::TM], const float (&fb)[T::TN]) {
#pragma unroll
        for (int i = 0; i < T::TM; ++i)
#pragma unroll
            for (int j = 0; j < T::TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
    };
    frag(0, a[0], b[0]);
#pragma unroll
    for (int kk = 0; kk < IG_KC; kk += 4) {
        frag(kk + 2, a[1], b[1]);
        __builtin_amdgcn_sched_barrier(0);
        mma(a[0], b[0]);
        __builtin_amdgcn_sched_barrier(0);
        if (kk + 4 < IG_KC) frag(kk + 4, a[0], b[0]);
        __builtin_amdgcn_sched_barrier(0);
        mma(a[1], b[1]);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// ig_epilogue_vec4 for these tiles: every wave stages its 32 x 32 tiles through LDS and stores 16 bytes per lane
// (tile pixel px -> flat pixel index: row px / TW, column px % TW of the tile at (image b, row y0, column x0))
template <int BM, int BN, class Ad>
__device__ __forceinline__ void hc_epilogue_vec4(const typename Ad::Params& p, float* __restrict__ stage,
                                                 const f32x16 (&acc)[HcTile<BM, BN>::TM][HcTile<BM, BN>::TN], int m0,
                                                 int b, int y0, int x0, const HaloGeom& hg, int wm_off, int wn_off, int lane,
                                                 int M) {
    using T = HcTile<BM, BN>;
    const int col = lane & 31, cg = lane & 7, rsub = lane >> 3;
#pragma unroll
    for (int j = 0; j < T::TN; ++j) {
        const int px = wn_off + j * 32 + 4 * cg, y = y0 + (px >> hg.tw_shift);
        const bool n_ok = y < hg.H;
        const long long n = n_ok ? (long long)(b * hg.H + y) * hg.W + x0 + (px & (hg.TW - 1)) : 0;
        typename Ad::Out out(p, n);
#pragma unroll
        for (int i = 0; i < T::TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) stage[mfma_row(r, lane) * IG_EPI_LD + col] = acc[i][j][r];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int row = it * 8 + rsub;
                const f32x4 v = *reinterpret_cast<const f32x4*>(stage + row * IG_EPI_LD + 4 * cg);
                const int m = m0 + wm_off + i * 32 + row;
                if (m < M && n_ok) out.store4(p, m, v);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
}

// Ad: { using Params; using Out; static const char* name(); }  -- the epilogue contract of igemm.cuh's loaders.
template <int BM, int BN, class Ad>
__global__ __launch_bounds__(IG_THREADS, 2) void hconv_kernel(
    typename Ad::Params p, const float* __restrict__ src, const float* __restrict__ A, int Mp, int Kp, int M, long long N,
    int n_tiles, int m_tiles, HaloGeom hg) {
    using T = HcTile<BM, BN>;
    extern __shared__ __attribute__((aligned(16))) float smem[];      // Hs[16 * PL] | As[2][16 * BM]; reused by the epilogue
    const int PL = hg.PL, RS = hg.RS;
    float* const Hs = smem;
    float* const Asb = smem + 16 * PL;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wg = xcd_remap(blockIdx.x, n_tiles * m_tiles);
    const int m0 = (wg % m_tiles) * BM;
    const int wm_off = (wid / T::WN) * (T::TM * 32), wn_off = (wid % T::WN) * (T::TN * 32);
    const int HW = hg.H * hg.W;
    // pixel tile -> (image, tile row, tile column), image-major
    int nt = wg / m_tiles;
    const int tx = nt % hg.tiles_x; nt /= hg.tiles_x;
    const int ty = nt % hg.tiles_y, b = nt / hg.tiles_y;
    const int y0 = ty * hg.TR, x0 = tx * hg.TW;

    // halo cells of this thread: global byte offset (channel 0 of the group; sentinel for rows outside the image) and LDS slot
    const buf_rsrc rs = ig_make_rsrc(src, (unsigned)((size_t)N * hg.Kc * sizeof(float)));
    unsigned voff[HC_MAXCELLS];
    int loff[HC_MAXCELLS];
    const int ncell = (hg.cells + IG_THREADS - 1) / IG_THREADS;           // uniform: 3..8
#pragma unroll
    for (int i = 0; i < HC_MAXCELLS; ++i) {
        const int e = tid + i * IG_THREADS;
        voff[i] = IG_BUF_OOB;
        loff[i] = -1;
        if (i < ncell && e < hg.cells) {
            const int c = e / hg.cpp, rem = e - c * hg.cpp;
            const int row = rem / hg.cpr, q = rem - row * hg.cpr;
            const int iy = y0 - 1 + row, ix = x0 + 4 * (q - hg.side);     // (side: cell 0 holds the four columns left of the tile)
            loff[i] = c * PL + row * RS + 4 + 4 * (q - hg.side);
            if (iy >= 0 && iy < hg.H && ix >= 0 && ix < hg.W)
                voff[i] = (unsigned)(((b * hg.Kc + c) * HW + iy * hg.W + ix) * (int)sizeof(float));
        }
    }
    // a tile that spans the row: the halo columns left and right of the image are zero for every group -- written once,
    // never overwritten (with side cells the staging itself writes them: the neighbour's columns or the range check's zeros)
    if (!hg.side)
        for (int e = tid; e < 16 * (hg.TR + 2) * 2; e += IG_THREADS) {
            const int side = e & 1, cr = e >> 1;                             // cr over (channel, row)
            const int c = cr / (hg.TR + 2), row = cr - c * (hg.TR + 2);
            Hs[c * PL + row * RS + (side ? hg.TW + 4 : 3)] = 0.0f;
        }
    // B fragment offsets of this lane inside a plane: pixel -> (row, column + 3); the tap adds r * RS + s
    int boff[T::TN];
#pragma unroll
    for (int j = 0; j < T::TN; ++j) {
        const int px = wn_off + j * 32 + (lane & 31);
        boff[j] = (px >> hg.tw_shift) * RS + (px & (hg.TW - 1)) + 3;
    }

    f32x16 acc[T::TM][T::TN];
#pragma unroll
    for (int i = 0; i < T::TM; ++i)
#pragma unroll
        for (int j = 0; j < T::TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    f32x4 hreg[HC_MAXCELLS];
    auto halo_load = [&](int g) {
        const unsigned soff = (unsigned)(g * 16 * HW) * (unsigned)sizeof(float);
#pragma unroll
        for (int i = 0; i < HC_MAXCELLS; ++i)
            if (i < ncell) hreg[i] = ig_buf_load4(rs, voff[i], soff);
    };
    auto halo_store = [&]() {
#pragma unroll
        for (int i = 0; i < HC_MAXCELLS; ++i)
            if (i < ncell && loff[i] >= 0) *reinterpret_cast<f32x4*>(Hs + loff[i]) = hreg[i];
    };
    f32x4 ra[ig_a_per<BM>()];
    const IgABuf<BM> abuf(A, Mp, Kp, m0, tid);
    const int nchunk = Kp / IG_KC, G = hg.Kc >> 4;

    halo_load(0);
    abuf.load(0, ra);
    halo_store();
    ig_store_a<BM>(Asb, tid, ra);
    if (1 < nchunk) abuf.load(IG_KC, ra);
    __syncthreads();
    int g = 0, tap = 0, tr = 0, ts = 0;                                  // chunk c = 9 g + tap, tap = 3 tr + ts
    for (int c = 0; c < nchunk; ++c) {
        const bool more = g + 1 < G;
        if (tap == 0 && more) halo_load(g + 1);
        hc_mma_chunk<BM, BN>(Asb + (c & 1) * IG_KC * BM, Hs + tr * RS + ts, acc, wm_off, boff, PL, lane);
        if (c + 1 < nchunk) {
            ig_store_a<BM>(Asb + ((c + 1) & 1) * IG_KC * BM, tid, ra);
            if (c + 2 < nchunk) abuf.load((c + 2) * IG_KC, ra);
        }
        __syncthreads();
        if (tap == 8 && more) {          // every wave has read the group's last fragments: the next group's tile moves in
            halo_store();                // (its loads went out nine chunks ago; the other resident workgroups fill the gap)
            __syncthreads();
        }
        if (++ts == 3) { ts = 0; if (++tr == 3) { tr = 0; } }
        if (++tap == 9) { tap = 0; ++g; }
    }
    if (Ad::Out::vec4_ok(p)) {
        hc_epilogue_vec4<BM, BN, Ad>(p, smem + wid * IG_EPI_WAVE, acc, m0, b, y0, x0, hg, wm_off, wn_off, lane, M);
        return;
    }
#pragma unroll
    for (int j = 0; j < T::TN; ++j) {
        const int px = wn_off + j * 32 + (lane & 31), y = y0 + (px >> hg.tw_shift);
        if (y >= hg.H) continue;
        typename Ad::Out out(p, (long long)(b * hg.H + y) * hg.W + x0 + (px & (hg.TW - 1)));
#pragma unroll
        for (int i = 0; i < T::TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm_off + i * 32 + mfma_row(r, lane);
                if (m < M) out.store(p, m, acc[i][j][r]);
            }
    }
}

}  // namespace cnuda
