// Implicit-GEMM building blocks on the fp32 MFMA (v_mfma_f32_32x32x2_f32) for
// gfx950.  fp32 in / fp32 accumulate is exact f32 (a k-ordered fma chain), so
// the 1e-4 parity budget of the hot path is spent on summation order only.
//
// GEMM view shared by every user of this header:
//     D[m][n] = sum_k A[k][m] * B[k][n]
//   A: a small "packed" matrix in global memory, [Kp][Mp] row-major (m
//      contiguous), zero padded to multiples of (BK, BM) by a pack kernel.
//   B: produced on the fly by a Loader (im2col gather, transposed-conv gather,
//      bilinear deformable sampling, or a plain row read), n = flattened
//      (batch, y, x) "pixel" so that consecutive lanes touch consecutive
//      addresses of an NCHW plane.
// Workgroup = 256 threads = 4 waves; block tile BM x 128 x 16; every wave owns
// TM x TN tiles of 32x32.  LDS holds As[BK][BM] and Bs[BK][BN]: both operand
// reads are lane-consecutive ds_read_b32 (conflict-free, MI355X_MICROARCH LDS
// table), fragment maps per cdna_hip_programming.md section 3:
//   A operand: lane l holds A[k = l>>5][i = l&31]; B: B[k = l>>5][j = l&31]
//   D: col j = l&31, row i = (reg&3) + 8*(reg>>2) + 4*(l>>5).
#pragma once
#include <type_traits>
#include <utility>
#include "common.h"

namespace cnuda {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int IG_BN = 128;
constexpr int IG_BK = 16;   // K granularity of the loaders / packing
constexpr int IG_KC = 16;   // K depth staged in LDS per barrier pair (two loader calls)
constexpr int IG_THREADS = 256;

template <int BM> struct IgTile;
template <> struct IgTile<128> { static constexpr int WM = 2, WN = 2, TM = 2, TN = 2; };
template <> struct IgTile<64>  { static constexpr int WM = 2, WN = 2, TM = 1, TN = 2; };
template <> struct IgTile<32>  { static constexpr int WM = 1, WN = 4, TM = 1, TN = 1; };

// XCD-aware, bijective remap of a linear workgroup id: workgroups are dealt
// round-robin over the 8 XCDs (b and b+8 share an L2), so give every XCD a
// contiguous run of tiles -- neighbouring pixel tiles share halo rows and the
// M tiles of one pixel tile share the whole gathered B panel.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    constexpr int NX = 8;
    const int per = nwg / NX, rem = nwg % NX;
    const int x = bid % NX, q = bid / NX;
    // XCD x owns `per` tiles (+1 for the first `rem` XCDs)
    const int start = x * per + (x < rem ? x : rem);
    return start + q;
}

// row m of D tile element `reg` for this lane
__device__ __forceinline__ int mfma_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

// One K chunk of MFMAs.  The operand fragments of k-step s+1 are read from LDS BEFORE the MFMAs of step s are
// issued (explicit two-deep register pipeline): left to itself the compiler emits read -> s_waitcnt lgkmcnt(0) ->
// MFMA per step, and the ~100-cycle LDS round trip then idles the matrix pipe between steps (measured with
// SQ_VALU_MFMA_BUSY_CYCLES: 46 / 56 / 67 % busy for the 32 / 64 / 128-row tiles = 1 / 2 / 4 MFMAs per wait).
// (LDA: row stride of the A image in LDS -- BM, unless a narrower tile is computed from a wider staged image: the short-K
// kernel's last row tile)
template <int BM, int KC = IG_KC, int LDA = BM>
__device__ __forceinline__ void ig_mma_chunk(const float* __restrict__ As, const float* __restrict__ Bs,
                                             f32x16 (&acc)[IgTile<BM>::TM][IgTile<BM>::TN],
                                             int wm_off, int wn_off, int lane) {
    using T = IgTile<BM>;
    const int kl = lane >> 5, il = lane & 31;
    const float* ap = As + kl * LDA + wm_off + il;
    const float* bp = Bs + kl * IG_BN + wn_off + il;
    float a[2][T::TM], b[2][T::TN];
    auto frag = [&](int kk, float (&fa)[T::TM], float (&fb)[T::TN]) {
#pragma unroll
        for (int i = 0; i < T::TM; ++i) fa[i] = ap[kk * LDA + i * 32];
#pragma unroll
        for (int j = 0; j < T::TN; ++j) fb[j] = bp[kk * IG_BN + j * 32];
    };
    auto mma = [&](const float (&fa)[T::TM], const float (&fb)[T::TN]) {
#pragma unroll
        for (int i = 0; i < T::TM; ++i)
#pragma unroll
            for (int j = 0; j < T::TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
    };
    // sched_barrier(0): nothing is rescheduled across it, so the reads stay ahead of the MFMAs they overlap
    frag(0, a[0], b[0]);
#pragma unroll
    for (int kk = 0; kk < KC; kk += 4) {
        frag(kk + 2, a[1], b[1]);
        __builtin_amdgcn_sched_barrier(0);
        mma(a[0], b[0]);
        __builtin_amdgcn_sched_barrier(0);
        if (kk + 4 < KC) frag(kk + 4, a[0], b[0]);
        __builtin_amdgcn_sched_barrier(0);
        mma(a[1], b[1]);
        __builtin_amdgcn_sched_barrier(0);
    }
}


// ---------------------------------------------------------------------------
// Split-operand mode (X3).  Every f32 operand x is cut into three bf16 pieces x = x1 + x2 + x3 by TRUNCATION
// (8 + 8 + 8 mantissa bits: the cut is exact, no bit of x is dropped) and the product a*b is evaluated on the
// bf16 MFMA (v_mfma_f32_32x32x16_bf16, f32 accumulate, 16x the f32 MFMA rate) as the six leading partial
// products a1b1 + a1b2 + a2b1 + a1b3 + a2b2 + a3b1; the three dropped ones are below 2^-23 |a*b|, the size of
// one f32 rounding.  LDS images are [piece][k half][row] of 16-byte cells (8 bf16 = the k positions one lane
// feeds to the MFMA), so the cell writes and the fragment reads are both lane-consecutive ds_*_b128.
// K order inside a 16-deep chunk is permuted (position 8h + j holds k = h + 2j, the loaders' own order); A and
// B use the same permutation, so the sum is unchanged.  A non-finite operand turns into NaN (inf - inf in the
// cut) where the f32 MFMA would give inf or NaN: finite inputs only, like everything else on this path.
// ---------------------------------------------------------------------------
using bf16x8 = __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;   // one 16-byte cell (a native vector: stays in registers)

// NV floats -> three bf16 pieces each, packed pairwise (NV/2 dwords per piece)
template <int NV>
__device__ __forceinline__ void x3_split(const float (&v)[NV], unsigned (&o)[3][NV / 2]) {
    unsigned pc[3][NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const unsigned u1 = __float_as_uint(v[j]) & 0xFFFF0000u;
        const float r1 = v[j] - __uint_as_float(u1);             // exact
        const unsigned u2 = __float_as_uint(r1) & 0xFFFF0000u;
        const float r2 = r1 - __uint_as_float(u2);               // exact, at most 8 significant bits
        pc[0][j] = u1; pc[1][j] = u2; pc[2][j] = __float_as_uint(r2);
    }
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int j = 0; j < NV / 2; ++j)     // low half = element 2j, high half = element 2j+1 (their top 16 bits)
            o[q][j] = __builtin_amdgcn_perm(pc[q][2 * j + 1], pc[q][2 * j], 0x07060302u);
}

// operand fragments of one chunk are u32x4 [tile][piece]
template <int BM>
__device__ __forceinline__ void ig_read_frag_x3(const u32x4* __restrict__ As, const u32x4* __restrict__ Bs,
                                                u32x4 (&a)[IgTile<BM>::TM][3], u32x4 (&b)[IgTile<BM>::TN][3],
                                                int wm_off, int wn_off, int lane) {
    using T = IgTile<BM>;
    const int kl = lane >> 5, il = lane & 31;
    // As: [3][2][BM], Bs: [3][2][IG_BN]; order = the order the MFMAs below first need them in
#pragma unroll
    for (int q = 2; q >= 0; --q) {
#pragma unroll
        for (int i = 0; i < T::TM; ++i) a[i][q] = As[(q * 2 + kl) * BM + wm_off + i * 32 + il];
#pragma unroll
        for (int j = 0; j < T::TN; ++j) b[j][2 - q] = Bs[((2 - q) * 2 + kl) * IG_BN + wn_off + j * 32 + il];
    }
}
template <int BM>
__device__ __forceinline__ void ig_mma_frag_x3(const u32x4 (&a)[IgTile<BM>::TM][3], const u32x4 (&b)[IgTile<BM>::TN][3],
                                               f32x16 (&acc)[IgTile<BM>::TM][IgTile<BM>::TN]) {
    using T = IgTile<BM>;
    auto mm = [&](const u32x4& x, const u32x4& y, f32x16& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, x), __builtin_bit_cast(bf16x8, y), c, 0, 0, 0);
    };
    // smallest partial products first; the tiles rotate so that consecutive MFMAs hit different accumulators
#pragma unroll
    for (int t = 0; t < 6; ++t) {
        constexpr int qa[6] = {2, 1, 0, 1, 0, 0}, qb[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
        for (int i = 0; i < T::TM; ++i)
#pragma unroll
            for (int j = 0; j < T::TN; ++j) mm(a[i][qa[t]], b[j][qb[t]], acc[i][j]);
    }
}

// A operand of the X3 kernels: split once per launch by split_a_kernel into 16-byte cells
//   A3[chunk c][piece q][k half h][m]  = bf16 piece q of A[16c + h + 2j][m], j = 0..7
// so that a workgroup's A tile of one chunk is 6*BM consecutive-in-m cells copied verbatim to LDS.
template <int BM> constexpr int ig_a3_per() { return (6 * BM + IG_THREADS - 1) / IG_THREADS; }   // 3 / 2 / 1 cells per thread
template <int BM>
__device__ __forceinline__ void ig_load_a_x3(const u32x4* __restrict__ A3, int Mp, int k0, int m0, int tid,
                                             u32x4 (&r)[ig_a3_per<BM>()]) {
    const u32x4* base = A3 + (size_t)(k0 >> 4) * 6 * Mp + m0;
#pragma unroll
    for (int i = 0; i < ig_a3_per<BM>(); ++i) {
        const int e = tid + i * IG_THREADS;
        // (cells past 6*BM: an address inside the tile, the store below skips them.  Until round 6 this was `tid`, which for
        // the 32-row tile -- 192 cells, 256 threads -- is itself past the tile: threads 192..255 read up to 1 KB past the
        // chunk, on the last chunk past the image; pack.hip kPackSlack tells how that surfaced)
        const int ec = (6 * BM) % IG_THREADS == 0 ? e : (e < 6 * BM ? e : e % (6 * BM));
        r[i] = base[(size_t)(ec / BM) * Mp + ec % BM];
    }
}
template <int BM>
__device__ __forceinline__ void ig_store_a_x3(u32x4* __restrict__ As, int tid, const u32x4 (&r)[ig_a3_per<BM>()]) {
#pragma unroll
    for (int i = 0; i < ig_a3_per<BM>(); ++i) {
        const int e = tid + i * IG_THREADS;
        if ((6 * BM) % IG_THREADS == 0 || e < 6 * BM) As[e] = r[i];
    }
}
static __global__ void split_a_kernel(const float* __restrict__ A, u32x4* __restrict__ A3, int Kp, int Mp) {
    const long long total = (long long)(Kp / 16) * 2 * Mp;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int m = (int)(i % Mp), h = (int)(i / Mp) & 1, c = (int)(i / (2 * Mp));
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = A[(size_t)(16 * c + h + 2 * j) * Mp + m];
        unsigned o[3][4];
        x3_split<8>(v, o);
#pragma unroll
        for (int q = 0; q < 3; ++q) A3[((size_t)(c * 3 + q) * 2 + h) * Mp + m] = u32x4{o[q][0], o[q][1], o[q][2], o[q][3]};
    }
}
// bytes of a packed A operand buffer: the f32 [Kp][Mp] matrix followed by room for its split image
inline size_t ig_a_bytes(size_t Kp, size_t Mp) { return Kp * Mp * 10; }

// Stage the A tile rows [k0, k0+BK) x cols [m0, m0+BM) of the packed matrix: 16-byte loads and LDS stores
// (four consecutive m per thread; Mp and m0 are multiples of 32, the buffers 256-byte aligned).
using f32x4 = __attribute__((ext_vector_type(4))) float;
template <int BM, int KC = IG_KC> constexpr int ig_a_per() { return (BM * KC / 4 + IG_THREADS - 1) / IG_THREADS; }   // 2 / 1 / 1 (half the threads)
template <int BM, int KC = IG_KC>
__device__ __forceinline__ void ig_load_a(const float* __restrict__ A, int Mp, int k0, int m0, int tid,
                                          f32x4 (&r)[ig_a_per<BM, KC>()]) {
    constexpr int CELLS = BM * KC / 4;
#pragma unroll
    for (int i = 0; i < ig_a_per<BM, KC>(); ++i) {
        const int e = tid + i * IG_THREADS;
        const int ec = CELLS % IG_THREADS == 0 ? e : (e < CELLS ? e : e - CELLS);      // idle threads re-read a valid cell
        const int kk = ec / (BM / 4), m = (ec % (BM / 4)) * 4;
        r[i] = *reinterpret_cast<const f32x4*>(A + (size_t)(k0 + kk) * Mp + m0 + m);
    }
}
template <int BM, int KC = IG_KC>
__device__ __forceinline__ void ig_store_a(float* __restrict__ As, int tid, const f32x4 (&r)[ig_a_per<BM, KC>()]) {
    constexpr int CELLS = BM * KC / 4;
#pragma unroll
    for (int i = 0; i < ig_a_per<BM, KC>(); ++i) {
        const int e = tid + i * IG_THREADS;
        if (CELLS % IG_THREADS == 0 || e < CELLS) reinterpret_cast<f32x4*>(As)[e] = r[i];
    }
}

// ---------------------------------------------------------------------------
// Buffer addressing.  The f32 MFMA runs on the same lanes as the vector ALU: a VALU instruction issued by ANY wave of
// a SIMD takes its cycles from the matrix pipe (measured, DESIGN.md section 10: kernel time = MFMA time + the
// producers' time), so the staging waves must compute addresses on the SCALAR unit.  buffer_load adds a per-lane
// 32-bit byte offset (VGPR, computed once per tile), a wave-uniform byte offset (SGPR: channel / chunk terms, scalar
// adds) and range-checks the per-lane part against num_records: a lane whose offset is the sentinel reads 0.0f,
// which is the padding value, without a compare or a select.
// ---------------------------------------------------------------------------
using buf_rsrc = __amdgpu_buffer_rsrc_t;
constexpr unsigned IG_BUF_OOB = 0x80000000u;                 // per-lane offset past any tensor this path accepts (< 2 GiB)
__device__ __forceinline__ buf_rsrc ig_make_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float ig_buf_load(buf_rsrc r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ f32x4 ig_buf_load4(buf_rsrc r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}
// A tile through a buffer: the per-thread cell offsets are fixed for the whole launch, (k0, m0) is a scalar offset
template <int BM, int KC = IG_KC>
struct IgABuf {
    buf_rsrc rs;
    unsigned voff[ig_a_per<BM, KC>()];
    int Mp;
    __device__ __forceinline__ IgABuf(const float* A, int Mp_, int Kp, int m0, int tid) : Mp(Mp_) {
        rs = ig_make_rsrc(A + m0, (unsigned)(((size_t)Kp * Mp_ - m0) * sizeof(float)));
        constexpr int CELLS = BM * KC / 4;
#pragma unroll
        for (int i = 0; i < ig_a_per<BM, KC>(); ++i) {
            const int e = tid + i * IG_THREADS;
            const int ec = CELLS % IG_THREADS == 0 ? e : (e < CELLS ? e : e - CELLS);
            voff[i] = (unsigned)(((ec / (BM / 4)) * Mp_ + (ec % (BM / 4)) * 4) * (int)sizeof(float));
        }
    }
    __device__ __forceinline__ void load(int k0, f32x4 (&r)[ig_a_per<BM, KC>()]) const {
        const unsigned soff = (unsigned)(k0 * Mp) * (unsigned)sizeof(float);
#pragma unroll
        for (int i = 0; i < ig_a_per<BM, KC>(); ++i) r[i] = ig_buf_load4(rs, voff[i], soff);
    }
};


// ---------------------------------------------------------------------------
// Epilogue with 16-byte stores.  An MFMA accumulator tile has the pixel axis across LANES (lane & 31 = column), so
// the straightforward epilogue stores one dword per lane, 16 instructions per 32x32 tile, each covering two 128-byte
// row segments.  Here every wave stages its tile through LDS (the operand buffers are free after the K loop) and
// reads it back with four consecutive pixels per lane: a quarter of the store instructions, 16 bytes each
// (the store tail is issue-bound, not bandwidth-bound: cdna_hip_programming.md T21).  Out::store4(p, m, float4)
// receives four consecutive pixels n..n+3 of one image row-major plane (the host guarantees that the pixel count per
// image is a multiple of four, else kVec4 epilogues are not used).
// ---------------------------------------------------------------------------
constexpr int IG_EPI_LD = 36;                                  // floats per staged row (32 + pad, 16-byte aligned)
constexpr int IG_EPI_WAVE = 32 * IG_EPI_LD;                    // floats per wave
// BatchNorm statistics from the epilogue (Params with a `stats` member: the forward loaders).  The layer that follows a
// convolution in DLA-34 is a train-mode BatchNorm whose first pass re-reads the whole output for sum(x) / sum(x^2) per
// channel (bn_reduce_kernel<0>: 1.2 ms of a benched step).  Here the tile is in registers anyway -- after the LDS
// transpose a lane holds four consecutive pixels of one channel row and eight lanes share the row -- so each wave
// leaves, per channel row of its tile, the two sums over ITS pixels (TN x 32 = IG_STAT_PX<BM> of them) at
//   stats[(pixel block)][row m][2],   pixel block = first pixel / IG_STAT_PX<BM>, rows padded to Mp,
// one 8-byte store per row from the lane with cg == 0 (consecutive rows: 64 contiguous bytes per instruction).  The
// sums are taken over the values that are STORED (bias included); lanes past N contribute nothing.  Fixed order (a lane's
// four pixels, its TN tiles in sequence, then the row's eight lanes by quad permutes and a half-row mirror):
// bit-reproducible; bn_fold_stats_kernel adds the blocks of a statistics group in double precision, in order.
template <class P, class = void> struct IgHasStats : std::false_type {};
template <class P> struct IgHasStats<P, std::void_t<decltype(std::declval<P>().stats)>> : std::true_type {};
template <int BM> constexpr int ig_stat_px() { return IgTile<BM>::TN * 32; }
template <int BM, class Loader>
__device__ __forceinline__ void ig_epilogue_vec4(const typename Loader::Params& p, float* __restrict__ stage,
                                                 const f32x16 (&acc)[IgTile<BM>::TM][IgTile<BM>::TN], int m0, long long n0,
                                                 int wm_off, int wn_off, int lane, int M, long long N) {
    using T = IgTile<BM>;
    const int col = lane & 31, cg = lane & 7, rsub = lane >> 3;
    constexpr bool kStats = IgHasStats<typename Loader::Params>::value;
    bool stats_on = false;
    if constexpr (kStats) stats_on = p.stats != nullptr;
    float ssum[T::TM][4], ssq[T::TM][4];
#pragma unroll
    for (int i = 0; i < T::TM; ++i)
#pragma unroll
        for (int it = 0; it < 4; ++it) { ssum[i][it] = 0.0f; ssq[i][it] = 0.0f; }
#pragma unroll
    for (int j = 0; j < T::TN; ++j) {
        const long long n = n0 + wn_off + j * 32 + 4 * cg;
        typename Loader::Out out(p, n < N ? n : 0);
#pragma unroll
        for (int i = 0; i < T::TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) stage[mfma_row(r, lane) * IG_EPI_LD + col] = acc[i][j][r];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // same wave: LDS ops complete in order
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int row = it * 8 + rsub;
                const f32x4 v = *reinterpret_cast<const f32x4*>(stage + row * IG_EPI_LD + 4 * cg);
                const int m = m0 + wm_off + i * 32 + row;
                if (m < M && n < N) out.store4(p, m, v);
                if constexpr (kStats) {
                    if (stats_on) {
                        f32x4 w = v;
                        if (p.bias && m < M) w += p.bias[m];
                        if (!(n < N)) w = f32x4{0.f, 0.f, 0.f, 0.f};
                        // (per lane: its four pixels of this tile; the eight lanes of a row meet once, after the last tile)
                        ssum[i][it] += (w[0] + w[1]) + (w[2] + w[3]);
                        ssq[i][it] += (w[0] * w[0] + w[1] * w[1]) + (w[2] * w[2] + w[3] * w[3]);
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // reads done before the next tile overwrites
        }
    }
    if constexpr (kStats) {
        if (stats_on) {
            // the eight lanes (cg = 0..7) of a row: two quad permutes and a half-row mirror on the vector ALU (an LDS-crossbar
            // shuffle per step measured +4 % on the 128-row tile: three dependent ~60-cycle round trips per row and tile)
            auto row8 = [](float v) {
                v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));   // quad_perm [1,0,3,2]
                v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false));   // quad_perm [2,3,0,1]
                v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, false));  // row_half_mirror
                return v;
            };
            const long long blk = (n0 + wn_off) / ig_stat_px<BM>();
            float2* dst = reinterpret_cast<float2*>(p.stats) + blk * p.stats_mp + m0 + wm_off;
#pragma unroll
            for (int i = 0; i < T::TM; ++i)
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const float a = row8(ssum[i][it]), b = row8(ssq[i][it]);
                    if (cg == 0) dst[i * 32 + it * 8 + rsub] = make_float2(a, b);
                }
        }
    }
}

// Epilogue for outputs whose ROWS are interleaved in quads (Out::kQuads: y[b][M / 4][pixel][4], the DCN column gradient's
// layout since round 6).  A lane of a 32x32 MFMA tile holds, per register quad, four CONSECUTIVE rows of one pixel column
// (mfma_row): exactly one 16-byte cell of that layout -- no LDS transpose, four stores per tile and lane, 512 contiguous
// bytes per half wave.  M % 4 == 0 (the host's condition for the layout): a quad is inside the matrix or outside it.
template <class O, class = void> struct IgHasQuads : std::false_type {};
template <class O> struct IgHasQuads<O, std::void_t<decltype(O::kQuads)>> : std::true_type {};
template <int BM, class Loader>
__device__ __forceinline__ void ig_epilogue_quads(const typename Loader::Params& p,
                                                  const f32x16 (&acc)[IgTile<BM>::TM][IgTile<BM>::TN], int m0, long long n0,
                                                  int wm_off, int wn_off, int lane, int M, long long N) {
    using T = IgTile<BM>;
#pragma unroll
    for (int j = 0; j < T::TN; ++j) {
        const long long n = n0 + wn_off + j * 32 + (lane & 31);
        if (n >= N) continue;
        typename Loader::Out out(p, n);
#pragma unroll
        for (int i = 0; i < T::TM; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int m = m0 + wm_off + i * 32 + 8 * q + 4 * (lane >> 5);
                if (m < M) out.store_quad(p, m, f32x4{acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]});
            }
    }
}

// raw-load storage of two-phase loaders (Loader::kHasSideOutput): Loader::Raw, else an empty placeholder
template <class Loader, bool TWO_PHASE> struct IgRaw { struct type {}; };
template <class Loader> struct IgRaw<Loader, true> { using type = typename Loader::Raw; };

// Generic forward-type kernel.  Loader contract:
//   Loader(const Params&, long long n, bool n_valid)   per-thread pixel setup
//   void load(int k0, int ksub, float (&v)[8])         v[j] = B[k0 + ksub + 2j][n]
//   or, when Loader::kHasSideOutput: load_raw(k0, ksub, Raw&) issues the loads of a chunk and finish(Raw&, v) turns
//   them into values one chunk later, when the kernel stores that chunk to LDS
// Epilogue contract:
//   void store(const Params&, int m, long long n, float value)
// Split-K epilogue (round 6): the accumulator tile as it stands -- no bias, no activation -- into split z's slab
// [Mp][Np] (Np = n_tiles * 128: every tile whole, no bounds tests); splitk_reduce_kernel adds the slabs in order and runs
// the loader's own epilogue (Out::store) on the sums.
template <int BM>
__device__ __forceinline__ void ig_store_slab(float* __restrict__ slab, const f32x16 (&acc)[IgTile<BM>::TM][IgTile<BM>::TN],
                                              int Mp, long long Np, int m0, long long n0, int wm_off, int wn_off, int lane) {
    using T = IgTile<BM>;
    float* const d = slab + (size_t)blockIdx.y * Mp * Np;
#pragma unroll
    for (int j = 0; j < T::TN; ++j) {
        const long long n = n0 + wn_off + j * 32 + (lane & 31);
#pragma unroll
        for (int i = 0; i < T::TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) d[(size_t)(m0 + wm_off + i * 32 + mfma_row(r, lane)) * Np + n] = acc[i][j][r];
    }
}

template <int BM, class Loader, bool X3 = false, bool SPLITK = false>
__device__ __forceinline__ void igemm_fwd_body(
    const typename Loader::Params& p, const float* __restrict__ A, int Mp, int Kp, int M, long long N,
    int n_tiles, int m_tiles, float* __restrict__ slab = nullptr, int split_k = 0) {
    using T = IgTile<BM>;
    static_assert(!(X3 && SPLITK), "split-K runs on the f32 pipe");
    // two LDS stages: chunk k+1 is written while chunk k's fragments are still being read, one barrier per chunk
    // (X3: three bf16 pieces per operand, 1.5x the bytes)
    constexpr int LDS_A = X3 ? 3 * 2 * BM * 4 : IG_KC * BM, LDS_B = X3 ? 3 * 2 * IG_BN * 4 : IG_KC * IG_BN;   // floats
    constexpr int LDS_AB = 2 * LDS_A + 2 * LDS_B;
    static_assert(LDS_AB >= 4 * IG_EPI_WAVE, "the operand buffers hold the epilogue staging tiles");
    __shared__ __attribute__((aligned(16))) float smem[LDS_AB];    // operand stages; reused by the vec4 epilogue
    float (*As)[LDS_A] = reinterpret_cast<float (*)[LDS_A]>(smem);
    float (*Bs)[LDS_B] = reinterpret_cast<float (*)[LDS_B]>(smem + 2 * LDS_A);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wg = xcd_remap(blockIdx.x, n_tiles * m_tiles);
    const int m0 = (wg % m_tiles) * BM;
    const long long n0 = (long long)(wg / m_tiles) * IG_BN;
    const int wm_off = (wid / T::WN) * (T::TM * 32), wn_off = (wid % T::WN) * (T::TN * 32);

    const int nl = tid & (IG_BN - 1), ksub = tid >> 7;  // pixel within tile, k parity
    Loader ld(p, n0 + nl, n0 + nl < N);
    if constexpr (Loader::kHasSideOutput) { if (m0 != 0) ld.disable_col(); }

    f32x16 acc[T::TM][T::TN];
#pragma unroll
    for (int i = 0; i < T::TM; ++i)
#pragma unroll
        for (int j = 0; j < T::TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // Kp is a multiple of IG_KC (the pack kernels zero-pad); the loaders return 0 past the real K
    constexpr int NH = IG_KC / IG_BK;               // loader calls per chunk
    f32x4 ra[ig_a_per<BM>()];
    float rb[NH][8];
    u32x4 ra3[ig_a3_per<BM>()];                     // X3: the pre-split A cells of a chunk
    typename IgRaw<Loader, Loader::kHasSideOutput>::type raw[NH];   // two-phase loaders keep raw loads here
    const IgABuf<BM> abuf(A, Mp, Kp, m0, tid);      // (unused by the X3 variant, whose A cells are pre-split)
    auto stage_store = [&](int buf) {
        if constexpr (X3) ig_store_a_x3<BM>(reinterpret_cast<u32x4*>(As[buf]), tid, ra3);
        else ig_store_a<BM>(As[buf], tid, ra);
        if constexpr (Loader::kHasSideOutput) {
#pragma unroll
            for (int h = 0; h < NH; ++h) ld.finish(raw[h], rb[h]);
        }
        if constexpr (X3) {
            static_assert(NH == 1, "X3 stages one 16-deep MFMA step per chunk");
            unsigned o[3][4];
            x3_split<8>(rb[0], o);
#pragma unroll
            for (int pc = 0; pc < 3; ++pc)
                reinterpret_cast<u32x4*>(Bs[buf])[(pc * 2 + ksub) * IG_BN + nl] = u32x4{o[pc][0], o[pc][1], o[pc][2], o[pc][3]};
        } else {
#pragma unroll
            for (int h = 0; h < NH; ++h)
#pragma unroll
                for (int j = 0; j < 8; ++j) Bs[buf][(h * IG_BK + ksub + 2 * j) * IG_BN + nl] = rb[h][j];
        }
    };
    auto stage_load = [&](int k0) {
        if constexpr (X3) ig_load_a_x3<BM>(reinterpret_cast<const u32x4*>(A), Mp, k0, m0, tid, ra3);
        else abuf.load(k0, ra);
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            if constexpr (Loader::kHasSideOutput) ld.load_raw(k0 + h * IG_BK, ksub, raw[h]);
            else ld.load(k0 + h * IG_BK, ksub, rb[h]);
        }
    };
    if constexpr (X3) {
        // Fully pipelined: per iteration k a wave (1) issues the LDS reads of chunk k+1's fragments into the spare
        // register set, (2) runs the MFMAs of chunk k from the set read one iteration ago, (3) splits and stores
        // chunk k+2 (global loads issued one iteration ago) into the LDS stage chunk k came from (free: its
        // fragments have been in registers since the last barrier), (4) issues the global loads of chunk k+3; one
        // barrier.  The body is straight-line (loads past the last chunk are clamped to it and
        // what they stage is never consumed), so the scheduler may thread (3) through the MFMA stream: an MFMA
        // holds the SIMD's vector issue for 8 of its 32 cycles, the other 24 take ~6 VALU instructions.
        const int nchunk = Kp / IG_KC, klast = Kp - IG_KC;
        auto kclamp = [&](int c) { const int k = c * IG_KC; return k < klast ? k : klast; };
        auto frag_read = [&](int st, u32x4 (&fa)[T::TM][3], u32x4 (&fb)[T::TN][3]) {
            ig_read_frag_x3<BM>(reinterpret_cast<const u32x4*>(As[st]), reinterpret_cast<const u32x4*>(Bs[st]), fa, fb,
                                wm_off, wn_off, lane);
        };
        u32x4 fa0[T::TM][3], fb0[T::TN][3], fa1[T::TM][3], fb1[T::TN][3];
        stage_load(0);
        stage_store(0);
        stage_load(kclamp(1));
        __syncthreads();
        frag_read(0, fa0, fb0);
        stage_store(1);
        stage_load(kclamp(2));
        __syncthreads();
        auto step = [&](int c, u32x4 (&ca)[T::TM][3], u32x4 (&cb)[T::TN][3], u32x4 (&na)[T::TM][3], u32x4 (&nb)[T::TN][3]) {
            frag_read((c + 1) & 1, na, nb);
            ig_mma_frag_x3<BM>(ca, cb, acc);
            stage_store(c & 1);
#pragma unroll
            for (int i = 0; i < 6 * T::TM * T::TN; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                      // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, BM == 128 ? 3 : (BM == 64 ? 5 : 6), 0);   // VALU beside it
            }
            stage_load(kclamp(c + 3));
            __syncthreads();
        };
        // The bf16 MFMA's accumulate does not round to nearest: every step loses a little toward zero (measured
        // -3e-6 of the result after 2304 coherent K terms, linear in K; the f32 MFMA shows none).  So the MFMAs
        // accumulate into a partial sum that is folded into the result with rounded f32 adds every X3_FOLD chunks:
        // the bias then scales with the partial sum's size and stays below one ulp of the result.
        constexpr int X3_FOLD = 8;
        f32x16 total[T::TM][T::TN];
#pragma unroll
        for (int i = 0; i < T::TM; ++i)
#pragma unroll
            for (int j = 0; j < T::TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) total[i][j][r] = 0.0f;
        auto fold = [&]() {
#pragma unroll
            for (int i = 0; i < T::TM; ++i)
#pragma unroll
                for (int j = 0; j < T::TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) { total[i][j][r] += acc[i][j][r]; acc[i][j][r] = 0.0f; }
        };
        int c = 0;
        for (; c + 1 < nchunk; c += 2) {
            step(c, fa0, fb0, fa1, fb1);
            step(c + 1, fa1, fb1, fa0, fb0);
            if (((c + 2) & (X3_FOLD - 1)) == 0) fold();
        }
        if (c < nchunk) ig_mma_frag_x3<BM>(fa0, fb0, acc);
#pragma unroll
        for (int i = 0; i < T::TM; ++i)
#pragma unroll
            for (int j = 0; j < T::TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] += total[i][j][r];
    } else {
    // (SPLITK: blockIdx.y takes the K range [kb, ke), split_k elements each)
    const int kb = SPLITK ? (int)blockIdx.y * split_k : 0;
    const int ke = SPLITK ? (kb + split_k < Kp ? kb + split_k : Kp) : Kp;
    stage_load(kb);
    stage_store(0);
    if (kb + IG_KC < ke) stage_load(kb + IG_KC);
    __syncthreads();
    int cur = 0;
    for (int k0 = kb; k0 < ke; k0 += IG_KC) {
        ig_mma_chunk<BM>(As[cur], Bs[cur], acc, wm_off, wn_off, lane);
        if (k0 + IG_KC < ke) {
            stage_store(cur ^ 1);                      // chunk k+1 (its global loads were issued one chunk ago)
            if (k0 + 2 * IG_KC < ke) stage_load(k0 + 2 * IG_KC);
        }
        __syncthreads();
        cur ^= 1;
    }
    }
    if constexpr (SPLITK) {
        ig_store_slab<BM>(slab, acc, Mp, (long long)n_tiles * IG_BN, m0, n0, wm_off, wn_off, lane);
        return;
    }
    if constexpr (IgHasQuads<typename Loader::Out>::value) {
        ig_epilogue_quads<BM, Loader>(p, acc, m0, n0, wm_off, wn_off, lane, M, N);
        return;
    }
    if constexpr (Loader::Out::kVec4) {
        if (Loader::Out::vec4_ok(p)) {      // (uniform; the K loop ended with a barrier: the operand buffers are free)
            ig_epilogue_vec4<BM, Loader>(p, smem + wid * IG_EPI_WAVE, acc, m0, n0, wm_off, wn_off, lane, M, N);
            return;
        }
    }
    // epilogue: lane owns pixel column (lane&31) of each tile
#pragma unroll
    for (int j = 0; j < T::TN; ++j) {
        const long long n = n0 + wn_off + j * 32 + (lane & 31);
        if (n >= N) continue;
        typename Loader::Out out(p, n);
#pragma unroll
        for (int i = 0; i < T::TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm_off + i * 32 + mfma_row(r, lane);
                if (m < M) out.store(p, m, acc[i][j][r]);
            }
    }
}

template <int BM, class Loader, bool X3 = false>
__global__ __launch_bounds__(IG_THREADS, (BM == 128 ? (X3 ? 2 : 3) : 1)) void igemm_fwd_kernel(
    typename Loader::Params p, const float* __restrict__ A, int Mp, int Kp, int M, long long N,
    int n_tiles, int m_tiles) {
    igemm_fwd_body<BM, Loader, X3, false>(p, A, Mp, Kp, M, N, n_tiles, m_tiles);
}
// Split-K form (grid.y = splits): problems whose pixel x row tiles do not fill the chip but whose K is long -- the 4 x 4
// convolutions of the ADVENT discriminator on 20 x 20 ... 5 x 5 maps (uda/adversarial_entropy_minimization.py:51-68), the
// 512 -> 27 offset convolution of the 16 x 16 level -- keep their natural row tile and cut K instead of shrinking the tile
// until the grid is large enough (a 32-row tile runs at 12-20 TFLOP/s there).
template <int BM, class Loader>
__global__ __launch_bounds__(IG_THREADS, (BM == 128 ? 3 : 1)) void igemm_fwd_splitk_kernel(
    typename Loader::Params p, const float* __restrict__ A, int Mp, int Kp, int M, long long N,
    int n_tiles, int m_tiles, float* __restrict__ slab, int split_k) {
    igemm_fwd_body<BM, Loader, false, true>(p, A, Mp, Kp, M, N, n_tiles, m_tiles, slab, split_k);
}
// y[m][n] = epilogue(sum over z of slab[z][m][n]): the splits in order (bit-reproducible), then the loader's own
// Out::store (bias, residual, activation / addends / strided scatter).  Thread = pixel n, blockIdx.y = 8 rows.
constexpr int SK_ROWS = 8;
template <class Loader>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(typename Loader::Params p, const float* __restrict__ slab, int Z,
                                                            int Mp, long long Np, int M, long long N) {
    const long long n = (long long)blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    typename Loader::Out out(p, n);
    const int m_end = ((int)blockIdx.y + 1) * SK_ROWS < M ? ((int)blockIdx.y + 1) * SK_ROWS : M;
    for (int m = blockIdx.y * SK_ROWS; m < m_end; ++m) {
        float v = 0.0f;
        for (int z = 0; z < Z; ++z) v += slab[((size_t)z * Mp + m) * Np + n];
        out.store(p, m, v);
    }
}

// ---------------------------------------------------------------------------
// Wave-specialised variant of the f32 forward-type kernel: 8 waves per workgroup.  Waves 4-7 (threads 256..511) are
// PRODUCERS -- they run the Loader, stage the next chunk's A and B tiles into LDS and issue the loads after it;
// waves 0-3 are CONSUMERS -- fragment reads and MFMAs only, each owning the same TM x TN tiles as in
// igemm_fwd_kernel.  Every SIMD then hosts vector/memory-only waves next to matrix-only waves, which the hardware
// runs concurrently, instead of waves that alternate between the two kinds of work.  Same LDS image, same two
// stages, one barrier per chunk (all 8 waves); the two roles never hold registers at the same time, so the
// kernel needs fewer registers than the 4-wave one (78 vs 113 for the 128-row tile).
// ---------------------------------------------------------------------------
template <int BM, class Loader, int KC, bool SPLITK = false>
__device__ __forceinline__ void igemm_fwd_ws_body(
    const typename Loader::Params& p, const float* __restrict__ A, int Mp, int Kp, int M, long long N,
    int n_tiles, int m_tiles, float* __restrict__ slab = nullptr, int split_k = 0) {
    using T = IgTile<BM>;
    // (SPLITK: blockIdx.y takes the K range [kb, ke), split_k elements each)
    const int kb = SPLITK ? (int)blockIdx.y * split_k : 0;
    const int ke = SPLITK ? (kb + split_k < Kp ? kb + split_k : Kp) : Kp;
    constexpr int LDS_AB = 2 * KC * BM + 2 * KC * IG_BN;
    static_assert(LDS_AB >= 4 * IG_EPI_WAVE, "the operand buffers hold the epilogue staging tiles");
    __shared__ __attribute__((aligned(16))) float smem[LDS_AB];    // operand stages; reused by the vec4 epilogue
    float (*As)[KC * BM] = reinterpret_cast<float (*)[KC * BM]>(smem);
    float (*Bs)[KC * IG_BN] = reinterpret_cast<float (*)[KC * IG_BN]>(smem + 2 * KC * BM);
    const bool producer = threadIdx.x >= IG_THREADS;            // wave-uniform
    const int tid = threadIdx.x & (IG_THREADS - 1), lane = tid & 63, wid = tid >> 6;
    const int wg = xcd_remap(blockIdx.x, n_tiles * m_tiles);
    const int m0 = (wg % m_tiles) * BM;
    const long long n0 = (long long)(wg / m_tiles) * IG_BN;
    if (producer) {
        const int nl = tid & (IG_BN - 1), ksub = tid >> 7;
        Loader ld(p, n0 + nl, n0 + nl < N);
        if constexpr (Loader::kHasSideOutput) { if (m0 != 0) ld.disable_col(); }
        // ONE register stage: the loads of chunk k + 2 are issued right after chunk k + 1 went to LDS.  (A second
        // stage -- loads issued two chunks ahead -- measured 0.4 ms per step slower once buffer addressing made the
        // loads cheap: it costs the third workgroup per CU, 86 instead of <= 80 registers.  DESIGN.md section 10.)
        static_assert(!Loader::kHasSideOutput, "two-phase loaders keep per-chunk state (the current tap's weights)");
        constexpr int NH = KC / IG_BK;                 // loader calls per chunk
        struct Regs {
            f32x4 ra[ig_a_per<BM, KC>()];
            float rb[NH][8];
            typename IgRaw<Loader, Loader::kHasSideOutput>::type raw[NH];
        };
        Regs r0;
        const IgABuf<BM, KC> abuf(A, Mp, Kp, m0, tid);
        auto stage_store = [&](int buf, Regs& r) {
            ig_store_a<BM, KC>(As[buf], tid, r.ra);
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                if constexpr (Loader::kHasSideOutput) ld.finish(r.raw[h], r.rb[h]);
#pragma unroll
                for (int j = 0; j < 8; ++j) Bs[buf][(h * IG_BK + ksub + 2 * j) * IG_BN + nl] = r.rb[h][j];
            }
        };
        auto stage_load = [&](int k0, Regs& r) {
            abuf.load(k0, r.ra);
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                if constexpr (Loader::kHasSideOutput) ld.load_raw(k0 + h * IG_BK, ksub, r.raw[h]);
                else ld.load(k0 + h * IG_BK, ksub, r.rb[h]);
            }
        };
        stage_load(kb, r0);
        stage_store(0, r0);
        if (kb + KC < ke) stage_load(kb + KC, r0);
        __syncthreads();
        int c1 = 0;
        for (int k0 = kb; k0 < ke; k0 += KC) {
            if (k0 + KC < ke) {
                stage_store(c1 ^ 1, r0);
                if (k0 + 2 * KC < ke) stage_load(k0 + 2 * KC, r0);
            }
            __syncthreads();
            c1 ^= 1;
        }
        return;
    }
    const int wm_off = (wid / T::WN) * (T::TM * 32), wn_off = (wid % T::WN) * (T::TN * 32);
    f32x16 acc[T::TM][T::TN];
#pragma unroll
    for (int i = 0; i < T::TM; ++i)
#pragma unroll
        for (int j = 0; j < T::TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    __syncthreads();
    int cur = 0;
#ifdef IG_FOLD
    // Measurement build only (-DIG_FOLD=n, profiles/experiments/r6_blocked_accumulation.md): the MFMA's k-ordered chain is
    // cut every n chunks -- the running tile is added into a second accumulator set with rounded adds and cleared -- which
    // makes the sum blocked like the CPU reference's (oneDNN), at the price of TM x TN x 16 more registers per consumer wave.
    f32x16 tot[T::TM][T::TN];
#pragma unroll
    for (int i = 0; i < T::TM; ++i)
#pragma unroll
        for (int j = 0; j < T::TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) tot[i][j][r] = 0.0f;
    int since = 0;
#endif
    for (int k0 = kb; k0 < ke; k0 += KC) {
        ig_mma_chunk<BM, KC>(As[cur], Bs[cur], acc, wm_off, wn_off, lane);
#ifdef IG_FOLD
        if (++since == IG_FOLD) {
            since = 0;
#pragma unroll
            for (int i = 0; i < T::TM; ++i)
#pragma unroll
                for (int j = 0; j < T::TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) { tot[i][j][r] += acc[i][j][r]; acc[i][j][r] = 0.0f; }
        }
#endif
        __syncthreads();
        cur ^= 1;
    }
#ifdef IG_FOLD
#pragma unroll
    for (int i = 0; i < T::TM; ++i)
#pragma unroll
        for (int j = 0; j < T::TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] += tot[i][j][r];
#endif
    if constexpr (SPLITK) {
        ig_store_slab<BM>(slab, acc, Mp, (long long)n_tiles * IG_BN, m0, n0, wm_off, wn_off, lane);
        return;
    }
    if constexpr (IgHasQuads<typename Loader::Out>::value) {
        ig_epilogue_quads<BM, Loader>(p, acc, m0, n0, wm_off, wn_off, lane, M, N);
        return;
    }
    if constexpr (Loader::Out::kVec4) {
        if (Loader::Out::vec4_ok(p)) {      // (the producers wrote their last stage before the last barrier)
            ig_epilogue_vec4<BM, Loader>(p, smem + wid * IG_EPI_WAVE, acc, m0, n0, wm_off, wn_off, lane, M, N);
            return;
        }
    }
#pragma unroll
    for (int j = 0; j < T::TN; ++j) {
        const long long n = n0 + wn_off + j * 32 + (lane & 31);
        if (n >= N) continue;
        typename Loader::Out out(p, n);
#pragma unroll
        for (int i = 0; i < T::TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm_off + i * 32 + mfma_row(r, lane);
                if (m < M) out.store(p, m, acc[i][j][r]);
            }
    }
}

template <int BM, class Loader, int KC = IG_KC>
__global__ __launch_bounds__(2 * IG_THREADS, (BM == 128 ? 2 : 1)) void igemm_fwd_ws_kernel(
    typename Loader::Params p, const float* __restrict__ A, int Mp, int Kp, int M, long long N,
    int n_tiles, int m_tiles) {
    igemm_fwd_ws_body<BM, Loader, KC>(p, A, Mp, Kp, M, N, n_tiles, m_tiles);
}
template <int BM, class Loader, int KC = IG_KC>
__global__ __launch_bounds__(2 * IG_THREADS, (BM == 128 ? 2 : 1)) void igemm_fwd_ws_splitk_kernel(
    typename Loader::Params p, const float* __restrict__ A, int Mp, int Kp, int M, long long N,
    int n_tiles, int m_tiles, float* __restrict__ slab, int split_k) {
    igemm_fwd_ws_body<BM, Loader, KC, true>(p, A, Mp, Kp, M, N, n_tiles, m_tiles, slab, split_k);
}

// ---------------------------------------------------------------------------
// Short-K variant (Kp <= KP = 64: the DCN column-gradient GEMM, a 1x1 convolution with K = 64 and 9*C = 576 output
// rows).  With four chunks per tile the pipelined kernels spend as long in prologue and epilogue as in the K loop
// (76-88 TFLOP/s).  Here a workgroup owns one pixel tile for ALL row tiles: the gathered B tile (KP x 128) is staged
// once and stays in LDS, every row tile copies its whole A tile (KP x BM) next to it, runs its KP/2 k-steps without a
// barrier in between and stores; two workgroups per CU (64 KB each) cover each other's copy and store phases.
// ---------------------------------------------------------------------------
template <int BM, class Loader, int KP>
__global__ __launch_bounds__(IG_THREADS, 2) void igemm_fwd_shortk_kernel(
    typename Loader::Params p, const float* __restrict__ A, int Mp, int Kp, int M, long long N, int n_tiles, int m_tiles) {
    using T = IgTile<BM>;
    static_assert(KP % IG_KC == 0 && KP * BM >= 4 * IG_EPI_WAVE, "the A buffer holds the epilogue staging tiles");
    __shared__ __attribute__((aligned(16))) float As[KP * BM];      // reused by the vec4 epilogue
    __shared__ __attribute__((aligned(16))) float Bs[KP * IG_BN];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const long long n0 = (long long)xcd_remap(blockIdx.x, n_tiles) * IG_BN;
    const int wm_off = (wid / T::WN) * (T::TM * 32), wn_off = (wid % T::WN) * (T::TN * 32);
    {   // B tile: every chunk gathered once
        const int nl = tid & (IG_BN - 1), ksub = tid >> 7;
        Loader ld(p, n0 + nl, n0 + nl < N);
        float rb[KP / IG_BK][8];
#pragma unroll
        for (int c = 0; c < KP / IG_BK; ++c) ld.load(c * IG_BK, ksub, rb[c]);     // (rows past Kp: the loader returns 0)
#pragma unroll
        for (int c = 0; c < KP / IG_BK; ++c)
#pragma unroll
            for (int j = 0; j < 8; ++j) Bs[(c * IG_BK + ksub + 2 * j) * IG_BN + nl] = rb[c][j];
    }
    constexpr int CELLS = KP * BM / 4, PER = CELLS / IG_THREADS;
    static_assert(CELLS % IG_THREADS == 0, "whole 16-byte cells per thread");
    const buf_rsrc ra = ig_make_rsrc(A, (unsigned)((size_t)Kp * Mp * sizeof(float)));
    f32x4 cell[PER];
    auto load_a = [&](int mt) {
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int e = tid + i * IG_THREADS, kk = e / (BM / 4), m = (e % (BM / 4)) * 4;
            // rows past Kp (Kp < KP): the per-lane offset leaves the buffer, the load returns 0
            cell[i] = ig_buf_load4(ra, kk < Kp ? (unsigned)((kk * Mp + m) * (int)sizeof(float)) : IG_BUF_OOB,
                                   (unsigned)(mt * BM) * (unsigned)sizeof(float));
        }
    };
    load_a(0);
    for (int mt = 0; mt < m_tiles; ++mt) {
        const int m0 = mt * BM;
        __syncthreads();                 // the previous row tile's epilogue is done with As (and Bs is complete)
#pragma unroll
        for (int i = 0; i < PER; ++i) reinterpret_cast<f32x4*>(As)[tid + i * IG_THREADS] = cell[i];
        __syncthreads();
        // the next row tile's A loads go out BEFORE this tile's stores: vmcnt counts in issue order, so a load issued
        // after the epilogue would wait for every one of its stores to drain
        if (mt + 1 < m_tiles) load_a(mt + 1);
        if constexpr (BM == 128) {
            // The LAST row tile when at most 64 of its rows exist (M = 9 * 64 = 576 = 4.5 tiles for the 64-channel DCN layers):
            // computed as a 64 x 128 tile over the four waves -- half the MFMAs of a 128-row tile whose upper half would be
            // all padding (round 6: a tenth of this kernel's matrix work)
            if (M - m0 <= 64) {
                using T2 = IgTile<64>;
                const int wm2 = (wid / T2::WN) * (T2::TM * 32), wn2 = (wid % T2::WN) * (T2::TN * 32);
                f32x16 acc2[T2::TM][T2::TN];
#pragma unroll
                for (int i = 0; i < T2::TM; ++i)
#pragma unroll
                    for (int j = 0; j < T2::TN; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc2[i][j][r] = 0.0f;
#pragma unroll
                for (int c = 0; c < KP / IG_KC; ++c)
                    ig_mma_chunk<64, IG_KC, BM>(As + c * IG_KC * BM, Bs + c * IG_KC * IG_BN, acc2, wm2, wn2, lane);
                if constexpr (IgHasQuads<typename Loader::Out>::value) {
                    ig_epilogue_quads<64, Loader>(p, acc2, m0, n0, wm2, wn2, lane, M, N);     // (no staging: As is not touched)
                } else {
                __syncthreads();
                if (Loader::Out::vec4_ok(p)) {
                    ig_epilogue_vec4<64, Loader>(p, As + wid * IG_EPI_WAVE, acc2, m0, n0, wm2, wn2, lane, M, N);
                } else {
#pragma unroll
                    for (int j = 0; j < T2::TN; ++j) {
                        const long long n = n0 + wn2 + j * 32 + (lane & 31);
                        if (n >= N) continue;
                        typename Loader::Out out(p, n);
#pragma unroll
                        for (int i = 0; i < T2::TM; ++i)
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                const int m = m0 + wm2 + i * 32 + mfma_row(r, lane);
                                if (m < M) out.store(p, m, acc2[i][j][r]);
                            }
                    }
                }
                }
                continue;
            }
        }
        f32x16 acc[T::TM][T::TN];
#pragma unroll
        for (int i = 0; i < T::TM; ++i)
#pragma unroll
            for (int j = 0; j < T::TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
#pragma unroll
        for (int c = 0; c < KP / IG_KC; ++c)
            ig_mma_chunk<BM>(As + c * IG_KC * BM, Bs + c * IG_KC * IG_BN, acc, wm_off, wn_off, lane);
        if constexpr (IgHasQuads<typename Loader::Out>::value) {
            ig_epilogue_quads<BM, Loader>(p, acc, m0, n0, wm_off, wn_off, lane, M, N);      // (no staging: As is not touched)
        } else {
        __syncthreads();                 // every wave has read its last fragments: As becomes the staging area
        if (Loader::Out::vec4_ok(p)) {
            ig_epilogue_vec4<BM, Loader>(p, As + wid * IG_EPI_WAVE, acc, m0, n0, wm_off, wn_off, lane, M, N);
        } else {
#pragma unroll
            for (int j = 0; j < T::TN; ++j) {
                const long long n = n0 + wn_off + j * 32 + (lane & 31);
                if (n >= N) continue;
                typename Loader::Out out(p, n);
#pragma unroll
                for (int i = 0; i < T::TM; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = m0 + wm_off + i * 32 + mfma_row(r, lane);
                        if (m < M) out.store(p, m, acc[i][j][r]);
                    }
            }
        }
        }
    }
}

// ---------------------------------------------------------------------------
// Weight-gradient-type kernel:  D[m][j] = sum_n  G[m][n] * B[j][n]
// (m = output channel, j = column of the packed K axis, n = pixel).  The pixel
// range is split over blockIdx.z; every workgroup writes one fp32 partial slab
// [Mp x Jp] and a second kernel sums the slabs in a fixed order (bitwise
// reproducible, no float atomics).
// Tile 64(m) x 64(j), 64 pixels per chunk; LDS images are [pixel][m|j] with an
// odd row stride so that the pixel-major stores are conflict-free.
// Loader contract (WLoader): per-thread cursor over the pixel axis
//   WLoader(const Params&, long long n, long long n_end)   cursor at pixel n (one-time integer divisions)
//   void advance()                                          move the cursor WG_BP pixels forward (no division)
//   template<int NV, int STEP> void load_b(int j0, int jsub, float (&v)[NV])   v[i] = B[j0 + jsub + STEP*i][cursor]
//   template<int NV, int STEP> void load_g(int m0, int msub, float (&v)[NV])   v[i] = G[m0 + msub + STEP*i][cursor]
// ---------------------------------------------------------------------------
constexpr int WG_BM = 64, WG_BJ = 64, WG_BP = 32;   // pixels per chunk (2 LDS stages of 32 instead of 1 of 64)

// Pixel cursor of the buffer-addressed weight-gradient loaders: (image, pixel inside the image, output row, column)
// of the thread's pixel in the current 32-pixel chunk, advanced without divisions or 64-bit arithmetic (these run on
// the staging waves between MFMAs: every vector instruction here is a matrix-pipe cycle lost).
struct IgPixelCursor {
    int left_;                 // pixels from the thread's pixel to the end of the split (> 0: the pixel is valid)
    int b_, pp_, oy_, ox_;
    int crossed_;              // images entered by the last advance() (the loaders move their image bases by it)
    bool valid_;
    __device__ __forceinline__ void init(long long n, long long n_end, int HoWo, int Wo) {
        const long long d = n_end - n;
        left_ = d > 0x7fffffffll ? 0x7fffffff : (d < 0 ? 0 : (int)d);
        valid_ = left_ > 0;
        const long long nn = valid_ ? n : 0;
        b_ = (int)(nn / HoWo);
        pp_ = (int)(nn - (long long)b_ * HoWo);
        oy_ = pp_ / Wo;
        ox_ = pp_ - oy_ * Wo;
        crossed_ = 0;
    }
    __device__ __forceinline__ void advance(int HoWo, int Wo) {
        left_ = left_ > WG_BP ? left_ - WG_BP : 0;
        valid_ = left_ > 0;
        pp_ += WG_BP;
        ox_ += WG_BP;
        crossed_ = 0;
        while (ox_ >= Wo) { ox_ -= Wo; ++oy_; }
        while (pp_ >= HoWo) { pp_ -= HoWo; ++b_; ++crossed_; oy_ = pp_ / Wo; ox_ = pp_ - oy_ * Wo; }
    }
};
// 24-bit multiply-add (full rate; v_mul_lo_u32 is quarter rate): a, b below 2^24 -- channel / row counts and plane
// sizes of tensors under 2 GiB
__device__ __forceinline__ int ig_mad24(int a, int b, int c) { return __mul24(a, b) + c; }
// NV rows (row0 + sub + STEP * i) of a [B][R][HoWo] tensor at the cursor's pixel.  `image_base` is the byte offset
// of the cursor's image (kept by the loader: += R * HoWo * 4 per image entered).  The per-lane offset carries the
// image, the thread's row phase and the pixel, the row stride is a scalar offset.  The range check covers the whole
// address (measured on gfx950: per-lane + scalar offset against num_records), so rows past R read the next image or,
// on the last image, 0.0f -- they only feed slab rows / columns that slab_reduce_kernel never reads; a thread whose
// pixel lies past the split's end reads the sentinel, i.e. 0.0f.
template <int NV, int STEP>
__device__ __forceinline__ void ig_buf_rows(buf_rsrc rs, const IgPixelCursor& c, unsigned image_base, int HoWo, int row0,
                                            int sub, float (&v)[NV]) {
    const unsigned voff = c.valid_ ? image_base + (unsigned)ig_mad24(sub, HoWo, c.pp_) * 4u : IG_BUF_OOB;
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = ig_buf_load(rs, voff, (unsigned)((row0 + STEP * i) * HoWo) * (unsigned)sizeof(float));
}

// bias row sums of a weight-gradient workgroup: thread (pl, sub) holds the sums of rows sub + STEP * i over its pixel
// lane's chunks; the 32 pixel lanes of a row group are one half of a wave -> xor butterfly 16..1, lane pl == 0 stores
template <int NG, int STEP>
__device__ __forceinline__ void ig_wgrad_store_bias(float (&bs)[NG], float* __restrict__ dst, int pl, int sub) {
    static_assert(WG_BP == 32, "a row group's pixel lanes are one half of a wave");
#pragma unroll
    for (int i = 0; i < NG; ++i) {
        float v = bs[i];
#pragma unroll
        for (int d = 16; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
        if (pl == 0) dst[sub + STEP * i] = v;
    }
}

// BM x BJ = 64 x 128 (waves 2 x 2, two accumulator tiles each: when the column count is a multiple of 128),
// 64 x 64 (waves 2 x 2) or 32 x 128 (waves 1 x 4, for layers with <= 32 output channels:
// the 16-channel stem / level-0 convs and the 27-channel DCN offset convs would waste 2-4x on a 64-row tile)
template <class WLoader, int BM, int BJ>
__global__ __launch_bounds__(IG_THREADS) void igemm_wgrad_kernel(
    typename WLoader::Params p, float* __restrict__ slabs, int Mp, int Jp, long long N, long long pix_per_split,
    float* __restrict__ bslab) {
    constexpr int GLD = BM + 1, BLD = BJ + 1;          // odd row strides: conflict-free pixel-major stores
    constexpr int STEP = IG_THREADS / WG_BP;            // rows (channels / columns) covered per pass
    // 32x32 accumulator tiles per wave: TM x TJ (2 x 2 for the 128 x 128 tile: 4 fragment dwords per 4 MFMAs; else one
    // row of tiles side by side along j)
    constexpr int TM = (BM == 128 && BJ == 128) ? 2 : 1, TJ = (BM / 32) * (BJ / 32) / 4 / TM;
    constexpr int WJ = BJ / 32 / TJ, NG = BM / STEP, NB = BJ / STEP;
    static_assert((BM / 32 / TM) * WJ == 4, "four waves tile the block");
    // two LDS stages, one barrier per pixel chunk (32 pixels): chunk k+1 is stored while chunk k is consumed
    __shared__ float Gs[2][WG_BP * GLD];
    __shared__ float Bs[2][WG_BP * BLD];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int j0 = blockIdx.x * BJ, m0 = blockIdx.y * BM;
    const long long n_begin = (long long)blockIdx.z * pix_per_split;
    long long n_end = n_begin + pix_per_split;
    if (n_end > N) n_end = N;
    const int pl = tid % WG_BP, sub = tid / WG_BP;  // pixel within chunk, row phase (0..STEP-1)
    const int wm_off = (wid / WJ) * 32 * TM, wj_off = (wid % WJ) * 32 * TJ;
    WLoader ld(p, n_begin + pl, n_end);
    f32x16 acc[TM][TJ];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int t = 0; t < TJ; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][t][r] = 0.0f;
    float rg[NG], rb[NB];
    // bias gradient (bslab != nullptr): the row sums of G over this split's pixels, taken by the workgroups of the first
    // column tile from the values they stage anyway -- the G tile passes through these registers once per chunk, so the
    // separate pass over grad_y (channel_sum_*: 38 launch pairs and 0.7 ms per benched step) is not needed.  Fixed order:
    // a thread's chunks in sequence, then a butterfly over its 32 pixel lanes; the splits are summed by slab_reduce_*.
    const bool do_bias = bslab != nullptr && blockIdx.x == 0;
    float bs[NG];
#pragma unroll
    for (int i = 0; i < NG; ++i) bs[i] = 0.0f;
    auto stage_store = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NG; ++i) Gs[buf][pl * GLD + sub + STEP * i] = rg[i];
#pragma unroll
        for (int i = 0; i < NB; ++i) Bs[buf][pl * BLD + sub + STEP * i] = rb[i];
        if (do_bias) {
#pragma unroll
            for (int i = 0; i < NG; ++i) bs[i] += rg[i];
        }
    };
    auto stage_load = [&]() {
        ld.template load_g<NG, STEP>(m0, sub, rg);
        ld.template load_b<NB, STEP>(j0, sub, rb);
    };
    stage_load();
    stage_store(0);
    if (n_begin + WG_BP < n_end) { ld.advance(); stage_load(); }
    __syncthreads();
    int cur = 0;
    const int kl = lane >> 5, il = lane & 31;
    for (long long nb = n_begin; nb < n_end; nb += WG_BP) {
        {   // fragments of the next two k-steps are in flight while this pair's MFMAs run (see ig_mma_chunk)
            const float* gp = Gs[cur] + kl * GLD + wm_off + il;
            const float* bp = Bs[cur] + kl * BLD + wj_off + il;
            float a[2][2][TM], b[2][2][TJ];
            auto frag = [&](int kk, float (&fa)[2][TM], float (&fb)[2][TJ]) {
#pragma unroll
                for (int i = 0; i < TM; ++i) { fa[0][i] = gp[kk * GLD + i * 32]; fa[1][i] = gp[(kk + 2) * GLD + i * 32]; }
#pragma unroll
                for (int t = 0; t < TJ; ++t) { fb[0][t] = bp[kk * BLD + t * 32]; fb[1][t] = bp[(kk + 2) * BLD + t * 32]; }
            };
            auto mma = [&](const float (&fa)[2][TM], const float (&fb)[2][TJ]) {
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int t = 0; t < TJ; ++t)
                            acc[i][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[h][i], fb[h][t], acc[i][t], 0, 0, 0);
            };
            frag(0, a[0], b[0]);
#pragma unroll
            for (int kk = 0; kk < WG_BP; kk += 8) {
                frag(kk + 4, a[1], b[1]);
                __builtin_amdgcn_sched_barrier(0);
                mma(a[0], b[0]);
                __builtin_amdgcn_sched_barrier(0);
                if (kk + 8 < WG_BP) frag(kk + 8, a[0], b[0]);
                __builtin_amdgcn_sched_barrier(0);
                mma(a[1], b[1]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (nb + WG_BP < n_end) {
            stage_store(cur ^ 1);
            if (nb + 2 * WG_BP < n_end) { ld.advance(); stage_load(); }
        }
        __syncthreads();
        cur ^= 1;
    }
    float* slab = slabs + (size_t)blockIdx.z * Mp * Jp;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int t = 0; t < TJ; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm_off + i * 32 + mfma_row(r, lane);
                const int j = j0 + wj_off + t * 32 + (lane & 31);
                slab[(size_t)m * Jp + j] = acc[i][t][r];
            }
    if (do_bias) ig_wgrad_store_bias<NG, STEP>(bs, bslab + (size_t)blockIdx.z * Mp + m0, pl, sub);
}

// Wave-specialised weight-gradient kernel (see igemm_fwd_ws_kernel): threads 256..511 run the two loaders and the
// LDS stores, threads 0..255 only read fragments and issue MFMAs.  Same tiles, LDS image and slab output as
// igemm_wgrad_kernel.
template <class WLoader, int BM, int BJ>
__global__ __launch_bounds__(2 * IG_THREADS) void igemm_wgrad_ws_kernel(
    typename WLoader::Params p, float* __restrict__ slabs, int Mp, int Jp, long long N, long long pix_per_split,
    float* __restrict__ bslab) {
    constexpr int GLD = BM + 1, BLD = BJ + 1;
    constexpr int STEP = IG_THREADS / WG_BP;
    constexpr int TM = (BM == 128 && BJ == 128) ? 2 : 1, TJ = (BM / 32) * (BJ / 32) / 4 / TM;     // see igemm_wgrad_kernel
    constexpr int WJ = BJ / 32 / TJ, NG = BM / STEP, NB = BJ / STEP;
    static_assert((BM / 32 / TM) * WJ == 4, "four waves tile the block");
    __shared__ float Gs[2][WG_BP * GLD];
    __shared__ float Bs[2][WG_BP * BLD];
    const bool producer = threadIdx.x >= IG_THREADS;            // wave-uniform
    const int tid = threadIdx.x & (IG_THREADS - 1), lane = tid & 63, wid = tid >> 6;
    const int j0 = blockIdx.x * BJ, m0 = blockIdx.y * BM;
    const long long n_begin = (long long)blockIdx.z * pix_per_split;
    long long n_end = n_begin + pix_per_split;
    if (n_end > N) n_end = N;
    if (producer) {
        const int pl = tid % WG_BP, sub = tid / WG_BP;
        WLoader ld(p, n_begin + pl, n_end);
        struct Regs { float rg[NG], rb[NB]; };
        const bool do_bias = bslab != nullptr && blockIdx.x == 0;       // (see igemm_wgrad_kernel: bias row sums)
        float bs[NG];
#pragma unroll
        for (int i = 0; i < NG; ++i) bs[i] = 0.0f;
        auto stage_store = [&](int buf, const Regs& r) {
#pragma unroll
            for (int i = 0; i < NG; ++i) Gs[buf][pl * GLD + sub + STEP * i] = r.rg[i];
#pragma unroll
            for (int i = 0; i < NB; ++i) Bs[buf][pl * BLD + sub + STEP * i] = r.rb[i];
            if (do_bias) {
#pragma unroll
                for (int i = 0; i < NG; ++i) bs[i] += r.rg[i];
            }
        };
        auto stage_load = [&](Regs& r) {
            ld.template load_g<NG, STEP>(m0, sub, r.rg);
            ld.template load_b<NB, STEP>(j0, sub, r.rb);
        };
        const int nchunk = (int)((n_end - n_begin + WG_BP - 1) / WG_BP);
        Regs r0;    // one register stage (a second one measured +0.7 ms per step: DESIGN.md section 10)
        stage_load(r0);
        stage_store(0, r0);
        if (1 < nchunk) { ld.advance(); stage_load(r0); }
        __syncthreads();
        int cur = 0;
        for (int c = 0; c < nchunk; ++c) {
            if (c + 1 < nchunk) {
                stage_store(cur ^ 1, r0);
                if (c + 2 < nchunk) { ld.advance(); stage_load(r0); }
            }
            __syncthreads();
            cur ^= 1;
        }
        if (do_bias) ig_wgrad_store_bias<NG, STEP>(bs, bslab + (size_t)blockIdx.z * Mp + m0, pl, sub);
        return;
    }
    const int wm_off = (wid / WJ) * 32 * TM, wj_off = (wid % WJ) * 32 * TJ;
    f32x16 acc[TM][TJ];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int t = 0; t < TJ; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][t][r] = 0.0f;
    __syncthreads();
    int cur = 0;
    const int kl = lane >> 5, il = lane & 31;
    for (long long nb = n_begin; nb < n_end; nb += WG_BP) {
        const float* gp = Gs[cur] + kl * GLD + wm_off + il;
        const float* bp = Bs[cur] + kl * BLD + wj_off + il;
        float a[2][2][TM], b[2][2][TJ];
        auto frag = [&](int kk, float (&fa)[2][TM], float (&fb)[2][TJ]) {
#pragma unroll
            for (int i = 0; i < TM; ++i) { fa[0][i] = gp[kk * GLD + i * 32]; fa[1][i] = gp[(kk + 2) * GLD + i * 32]; }
#pragma unroll
            for (int t = 0; t < TJ; ++t) { fb[0][t] = bp[kk * BLD + t * 32]; fb[1][t] = bp[(kk + 2) * BLD + t * 32]; }
        };
        auto mma = [&](const float (&fa)[2][TM], const float (&fb)[2][TJ]) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int t = 0; t < TJ; ++t)
                        acc[i][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[h][i], fb[h][t], acc[i][t], 0, 0, 0);
        };
        frag(0, a[0], b[0]);
#pragma unroll
        for (int kk = 0; kk < WG_BP; kk += 8) {
            frag(kk + 4, a[1], b[1]);
            __builtin_amdgcn_sched_barrier(0);
            mma(a[0], b[0]);
            __builtin_amdgcn_sched_barrier(0);
            if (kk + 8 < WG_BP) frag(kk + 8, a[0], b[0]);
            __builtin_amdgcn_sched_barrier(0);
            mma(a[1], b[1]);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
        cur ^= 1;
    }
    float* slab = slabs + (size_t)blockIdx.z * Mp * Jp;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int t = 0; t < TJ; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm_off + i * 32 + mfma_row(r, lane);
                const int j = j0 + wj_off + t * 32 + (lane & 31);
                slab[(size_t)m * Jp + j] = acc[i][t][r];
            }
}

}  // namespace cnuda
