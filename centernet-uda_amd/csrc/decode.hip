// Detection decode for gfx950: 3x3 (or kxk) max-pool NMS + two-stage top-K +
// box assembly.  Replaces backends/decode.py:6-76 of the reference.
//
// Stage 1 (one 1,024-thread workgroup per (b, c) plane): the plane is read once from HBM into LDS, the NMS runs from
// LDS into registers, and the K best of the strictly positive survivors (after the NMS 8/9 of a map is exactly 0) are
// found by a 4,096-bin histogram of their leading score bits, a suffix scan for the bin of the K-th largest and a
// rank sort of the few keys at or above it.  Flat or pathological maps (plateaus at the threshold, fewer than K
// positive scores) take an exact 8-bit MSB radix select over the NMS'd plane in LDS instead.  Stage 2 (one workgroup
// per image) selects the K best of the C*K candidates the same way and assembles the boxes.
//
// Order key: high 32 bits = order-preserving image of the fp32 score, low 32
// bits = ~index, so "larger key" == "higher score, or equal score and lower
// index" -- a strict total order (the reference's torch.topk leaves ties
// unspecified; see include/centernet_uda_hip.h).
#include "common.h"

namespace cnuda {
namespace {

constexpr int kMaxK = 1024;
constexpr int kSelSlack = 64;

__device__ __forceinline__ uint32_t float_order_bits(float v) {
    v += 0.0f;  // -0.0 -> +0.0 so that both zeros tie like they do for torch.topk
    uint32_t u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float order_bits_float(uint32_t u) {
    u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
    return __uint_as_float(u);
}
__device__ __forceinline__ uint64_t make_key(float score, uint32_t index) {
    return ((uint64_t)float_order_bits(score) << 32) | (uint64_t)(0xffffffffu - index);
}

struct SelectScratch {
    int hist[256];
    uint64_t prefix;     // decided high bits of the K-th key
    int remaining;       // how many keys are still to be taken from the current bucket
    int done;
    int out_count;
    uint64_t sel[kMaxK + kSelSlack];   // selected keys (+ rank_sort_desc's zero padding)
};

// Sorts n <= NT unique keys s.sel[0..n) descending, in place, by counting.  The compare matrix is tiled: a thread
// keeps kSortKeys keys in registers and streams every P-th key of the array past them (P, a power of two <= 64, threads
// per block of kSortKeys keys: adjacent lanes of one wave; eight independent LDS reads per batch, the lanes of one part
// read one address, neighbouring parts neighbouring banks); the P partial counts are added with shuffles and a key's
// rank is its final position.  n * n / NT compares and n * n / (kSortKeys * NT) LDS reads per thread and, in place, two
// barriers -- a bitonic network over pow2(n) keys takes log^2 stages with a workgroup barrier each (36 for 256 keys).
// With `out` the keys of rank < K go straight to out[rank] (global memory), without any barrier.  The keys must be
// followed by zeros up to the end of the last batch: rank_sort_prepare, any time after n is known and before the
// barrier that publishes the keys.
constexpr int kSortKeys = 4, kSortBatch = 8;
template <int NT>
__device__ __forceinline__ int rank_sort_shift(int n) {             // log2(threads per block of kSortKeys keys)
    const int blocks = (max(n, 1) + kSortKeys - 1) / kSortKeys;
    int sh = 0;
    while (sh < 6 && (blocks << (sh + 1)) <= NT) ++sh;
    return sh;
}
template <int NT>
__device__ __forceinline__ void rank_sort_prepare(SelectScratch& s, int n) {
    static_assert(kMaxK + kSortBatch * (kSortKeys * NT / kMaxK) <= kMaxK + kSelSlack && kSortBatch * 64 <= kMaxK,
                  "the zero padding fits s.sel");
    const int P = 1 << rank_sort_shift<NT>(n);
    for (int i = n + threadIdx.x; i < n + kSortBatch * P; i += NT) s.sel[i] = 0;
}
template <int NT>
__device__ __forceinline__ void rank_sort_desc(SelectScratch& s, int n, uint64_t* __restrict__ out = nullptr, int K = 0) {
    const int tid = threadIdx.x;
    const int sh = rank_sort_shift<NT>(n), P = 1 << sh;
    const int q0 = (tid >> sh) * kSortKeys, part = tid & (P - 1);
    uint64_t key[kSortKeys];
    int rank[kSortKeys];
#pragma unroll
    for (int r = 0; r < kSortKeys; ++r) { key[r] = s.sel[min(q0 + r, kMaxK - 1)]; rank[r] = 0; }
    for (int j = part; j < n; j += kSortBatch * P) {                 // (threads without keys run along: shuffles below)
        uint64_t x[kSortBatch];
#pragma unroll
        for (int u = 0; u < kSortBatch; ++u) x[u] = s.sel[j + u * P];
#pragma unroll
        for (int u = 0; u < kSortBatch; ++u)
#pragma unroll
            for (int r = 0; r < kSortKeys; ++r) rank[r] += x[u] > key[r] ? 1 : 0;
    }
    for (int o = 1; o < P; o <<= 1)
#pragma unroll
        for (int r = 0; r < kSortKeys; ++r) rank[r] += __shfl_xor(rank[r], o, 64);
    if (out) {
        if (part == 0) {
#pragma unroll
            for (int r = 0; r < kSortKeys; ++r)
                if (q0 + r < n && rank[r] < K) out[rank[r]] = key[r];
        }
        return;
    }
    __syncthreads();
    if (part == 0) {
#pragma unroll
        for (int r = 0; r < kSortKeys; ++r)
            if (q0 + r < n) s.sel[rank[r]] = key[r];
    }
    __syncthreads();
}

// Block-wide exact top-K of n unique 64-bit keys, result sorted descending in
// s.sel[0..K).  key_at(i) must be cheap and deterministic (called once per pass).
template <int NT, typename KeyAt>
__device__ void block_topk(KeyAt key_at, int n, int K, int KP, SelectScratch& s) {
    const int tid = threadIdx.x;
    if (tid == 0) { s.prefix = 0; s.remaining = K; s.done = 0; s.out_count = 0; }
    // threshold search: after the loop every key >= s.prefix (compared on the
    // decided bits) belongs to the top K.
    int shift = 56;
    uint64_t decided_mask = 0;
    for (int pass = 0; pass < 8; ++pass, shift -= 8) {
        if (tid < 256) s.hist[tid] = 0;
        __syncthreads();
        if (s.done) break;
        const uint64_t prefix = s.prefix;
        for (int i = tid; i < n; i += NT) {
            const uint64_t k = key_at(i);
            if ((k & decided_mask) == prefix) atomicAdd(&s.hist[(int)((k >> shift) & 0xff)], 1);
        }
        __syncthreads();
        if (tid < 64) {
            // suffix counts over 256 bins, 4 bins per lane (lane 63 owns the top bins)
            const int base = (63 - tid) * 4;  // lane 0 -> bins 252..255
            int c3 = s.hist[base + 3], c2 = s.hist[base + 2], c1 = s.hist[base + 1], c0 = s.hist[base];
            int local = c0 + c1 + c2 + c3;
            int incl = local;  // inclusive scan over lanes 0..tid (higher bins first)
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                int t = __shfl_up(incl, o, 64);
                if (tid >= o) incl += t;
            }
            const int above = incl - local;  // keys in strictly higher bins than this lane's four
            const int rem = s.remaining;
            // walk this lane's bins from high to low
            int acc = above;
            int cs[4] = {c3, c2, c1, c0};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int cnt = cs[j];
                if (acc < rem && rem <= acc + cnt) {
                    const int digit = base + 3 - j;
                    s.prefix = prefix | ((uint64_t)digit << shift);
                    s.remaining = rem - acc;
                    if (cnt == rem - acc) s.done = 1;  // whole bucket is selected
                }
                acc += cnt;
            }
        }
        decided_mask |= (uint64_t)0xff << shift;
        __syncthreads();
    }
    __syncthreads();
    // collect: keys whose decided bits are >= the threshold prefix
    {
        // decided_mask covers exactly the digits fixed when the loop ended
        const uint64_t prefix = s.prefix;
        const uint64_t m = decided_mask;
        for (int i = tid; i < KP; i += NT) s.sel[i] = 0;
        __syncthreads();
        for (int i = tid; i < n; i += NT) {
            const uint64_t k = key_at(i);
            if ((k & m) >= prefix) {
                const int pos = atomicAdd(&s.out_count, 1);
                if (pos < KP) s.sel[pos] = k;
            }
        }
        __syncthreads();
    }
    // the selected keys (exactly K of them: keys are unique) sit unordered in s.sel[0..K), zeros behind them
    static_assert(kMaxK <= kSortKeys * NT, "rank_sort_desc holds kSortKeys keys per thread");
    const int cnt = min(s.out_count, KP);
    rank_sort_prepare<NT>(s, cnt);
    __syncthreads();
    rank_sort_desc<NT>(s, cnt);
}

__device__ __forceinline__ float nms_value(const float* __restrict__ plane, int H, int W, int y, int x, int pad) {
    const float v = plane[y * W + x];
    float m = v;
    // -inf padding == clamped coordinates for a max: the clamped neighbour is an element of the window anyway.
    // Unconditional loads (no branch per neighbour): the (2*pad+1)^2 loads of a pixel are issued together.
    for (int dy = -pad; dy <= pad; ++dy) {
        const int yy = min(max(y + dy, 0), H - 1);
        for (int dx = -pad; dx <= pad; ++dx) {
            const int xx = min(max(x + dx, 0), W - 1);
            m = fmaxf(m, plane[yy * W + xx]);
        }
    }
    // keep = 1 - ceil(hmax - heat)  (decode.py:12), NOT (hmax == heat): identical
    // for scores in [0,1], reproduced literally for anything else (Q9)
    const float keep = 1.0f - ceilf(m - v);
    return v * keep;
}

// 3x3 NMS of four horizontally consecutive pixels (x0 % 4 == 0, W % 4 == 0): three rows of six values are
// loaded once (a dwordx4 and two edge scalars per row, clamped like nms_value), the vertical maxima are shared
// by the four horizontal windows: 4.5 loads and 5 max per pixel instead of 9 and 8.
__device__ __forceinline__ void nms_quad(const float* __restrict__ plane, int H, int W, int y, int x0, float (&out)[4]) {
    float col[6];      // vertical max of columns x0-1 .. x0+4
    float mid[4];
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy) {
        const int yy = min(max(y + dy, 0), H - 1);
        const float* row = plane + (size_t)yy * W;
        const float4 c = *reinterpret_cast<const float4*>(row + x0);
        const float l = row[max(x0 - 1, 0)], r = row[min(x0 + 4, W - 1)];
        const float v[6] = {l, c.x, c.y, c.z, c.w, r};
#pragma unroll
        for (int j = 0; j < 6; ++j) col[j] = dy == -1 ? v[j] : fmaxf(col[j], v[j]);
        if (dy == 0) { mid[0] = c.x; mid[1] = c.y; mid[2] = c.z; mid[3] = c.w; }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float m = fmaxf(fmaxf(col[j], col[j + 1]), col[j + 2]);
        out[j] = mid[j] * (1.0f - ceilf(m - mid[j]));          // decode.py:12, see nms_value
    }
}

// The same from a plane in LDS, addressed by the quad's pixel index i (x0 = i % W): the rows above and below are i -/+ W
// unless the quad sits in the first / last row, the edge columns i - 1 and i + 4 unless it starts / ends its row.
__device__ __forceinline__ void nms_quad_lds(const float* __restrict__ p, int i, int x0, int W, int HW, float (&out)[4]) {
    const int rows[3] = {i >= W ? i - W : i, i, i < HW - W ? i + W : i};
    const int lo = x0 == 0 ? 0 : -1, hi = x0 + 4 == W ? 3 : 4;
    float col[6];
    float mid[4];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const float4 c = *reinterpret_cast<const float4*>(p + rows[d]);
        const float v[6] = {p[rows[d] + lo], c.x, c.y, c.z, c.w, p[rows[d] + hi]};
#pragma unroll
        for (int j = 0; j < 6; ++j) col[j] = d == 0 ? v[j] : fmaxf(col[j], v[j]);
        if (d == 1) { mid[0] = c.x; mid[1] = c.y; mid[2] = c.z; mid[3] = c.w; }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float m = fmaxf(fmaxf(col[j], col[j + 1]), col[j + 2]);
        out[j] = mid[j] * (1.0f - ceilf(m - mid[j]));          // decode.py:12, see nms_value
    }
}

__global__ void nms_kernel(const float* __restrict__ heat, float* __restrict__ out,
                           long long planes, int H, int W, int pad) {
    const long long total = planes * H * W;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(i % W);
        const int y = (int)((i / W) % H);
        const long long p = i / ((long long)W * H);
        out[i] = nms_value(heat + p * H * W, H, W, y, x, pad);
    }
}

// Exact top-K of a plane whose NMS'd scores already sit in LDS as order-preserving 32-bit images (`bits[i]`, pixel
// i): an 8-bit MSB radix search for the K-th largest value (per-wave histograms: no cross-wave atomics; zeros --
// 8/9 of an NMS'd noise map -- are counted by ballot instead of hammering one bin), then one ordered pass: every
// value above the threshold is taken, and of the values EQUAL to it the ones with the lowest pixel indices (the
// order the 64-bit keys define).  Nothing is re-read from global memory and the NMS is not recomputed: plateaus
// (a trained model's background is clamped to exactly 1e-4, so whole regions survive the NMS) and maps with fewer
// than K positive scores cost the same as any other map.  Result: s.sel[0..K) sorted descending (rank sort).
template <int NT>
__device__ void lds_plane_topk(const uint32_t* __restrict__ bits, int n, int K, int KP, SelectScratch& s,
                               int* __restrict__ whist /* [NT/64][256] */, int* __restrict__ wcnt /* [2][NT/64] */) {
    constexpr int NW = NT / 64;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const uint32_t zero_bits = 0x80000000u;            // float_order_bits(+0.0f)
    __shared__ uint32_t thr_prefix;
    __shared__ int thr_remaining, thr_done;
    if (tid == 0) { thr_prefix = 0; thr_remaining = K; thr_done = 0; }
    uint32_t decided = 0;
    int shift = 24;
    for (int pass = 0; pass < 4; ++pass, shift -= 8) {
        for (int i = tid; i < NW * 256; i += NT) whist[i] = 0;
        __syncthreads();
        if (thr_done) break;
        const uint32_t prefix = thr_prefix;
        int zeros = 0;
        for (int i = tid; i < n + (NT - 1); i += NT) {       // (every lane runs every iteration: ballots)
            const bool in = i < n;
            const uint32_t v = in ? bits[i] : 0u;
            const bool live = in && (v & decided) == prefix;
            const bool z = live && v == zero_bits;
            zeros += __popcll(__ballot(z));
            // one round of leader matching first: the lanes that share the first live lane's digit (a plateau: a whole
            // wave of equal background scores) are counted with one ballot instead of 64 serialised LDS atomics
            bool todo = live && !z;
            const int digit = (int)((v >> shift) & 0xff);
            const unsigned long long act = __ballot(todo);
            if (act) {
                const int leader = __ffsll((long long)act) - 1;
                const int ld = __shfl(digit, leader, 64);
                const unsigned long long same = __ballot(todo && digit == ld);
                if (lane == leader) atomicAdd(&whist[wid * 256 + ld], __popcll(same));
                todo = todo && digit != ld;
            }
            if (todo) atomicAdd(&whist[wid * 256 + digit], 1);
        }
        if (lane == 0 && zeros) atomicAdd(&whist[wid * 256 + (int)((zero_bits >> shift) & 0xff)], zeros);
        __syncthreads();
        for (int b = tid; b < 256; b += NT) {                 // fold the wave histograms into s.hist
            int t = 0;
#pragma unroll
            for (int w = 0; w < NW; ++w) t += whist[w * 256 + b];
            s.hist[b] = t;
        }
        __syncthreads();
        if (tid < 64) {      // suffix counts over 256 bins, 4 bins per lane (as block_topk)
            const int base = (63 - tid) * 4;
            const int c3 = s.hist[base + 3], c2 = s.hist[base + 2], c1 = s.hist[base + 1], c0 = s.hist[base];
            const int local = c0 + c1 + c2 + c3;
            int incl = local;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int t = __shfl_up(incl, o, 64);
                if (tid >= o) incl += t;
            }
            const int rem = thr_remaining;
            int acc = incl - local;
            const int cs[4] = {c3, c2, c1, c0};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int cnt = cs[j];
                if (acc < rem && rem <= acc + cnt) {
                    thr_prefix = prefix | ((uint32_t)(base + 3 - j) << shift);
                    thr_remaining = rem - acc;
                    if (cnt == rem - acc) thr_done = 1;       // the whole bucket is selected
                }
                acc += cnt;
            }
        }
        decided |= 0xffu << shift;
        __syncthreads();
    }
    __syncthreads();
    // values whose decided bits exceed the threshold's are in; of those that equal it, `need` are -- all of them when
    // the search stopped early (done), else the lowest pixel indices first
    const uint32_t T = thr_prefix, m = decided;
    const int need = thr_remaining;
    const int per_wave = (n + NW - 1) / NW;                  // wave w owns pixels [w*per_wave, (w+1)*per_wave)
    const int lo = wid * per_wave, hi = min(n, lo + per_wave);
    int gt = 0, eq = 0;
    for (int i = lo + lane; i < hi + 63; i += 64) {
        const uint32_t v = i < hi ? bits[i] & m : 0u;
        gt += __popcll(__ballot(i < hi && v > T));
        eq += __popcll(__ballot(i < hi && v == T));
    }
    if (lane == 0) { wcnt[wid] = gt; wcnt[NW + wid] = eq; }
    for (int i = tid; i < KP; i += NT) s.sel[i] = 0;
    __syncthreads();
    int gt_base = 0, eq_base = 0, gt_total = 0, eq_total = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        gt_base += w < wid ? wcnt[w] : 0;
        eq_base += w < wid ? wcnt[NW + w] : 0;
        gt_total += wcnt[w];
        eq_total += wcnt[NW + w];
    }
    const int cnt = min(gt_total + min(eq_total, need), KP);
    rank_sort_prepare<NT>(s, cnt);
    for (int i = lo + lane; i < hi + 63; i += 64) {
        const bool in = i < hi;
        const uint32_t full = in ? bits[i] : 0u;
        const uint32_t v = full & m;
        const bool g = in && v > T, e = in && v == T;
        const unsigned long long mg = __ballot(g), me = __ballot(e);
        const unsigned long long below = (1ull << lane) - 1ull;
        if (g) s.sel[gt_base + __popcll(mg & below)] = ((uint64_t)full << 32) | (uint64_t)(0xffffffffu - (uint32_t)i);
        const int er = eq_base + __popcll(me & below);
        if (e && er < need) s.sel[gt_total + er] = ((uint64_t)full << 32) | (uint64_t)(0xffffffffu - (uint32_t)i);
        gt_base += __popcll(mg);
        eq_base += __popcll(me);
    }
    __syncthreads();
    rank_sort_desc<NT>(s, cnt);
}

// ---- threshold by histogram -----------------------------------------------------------------------------------------
// Both stages pick the K largest of n keys the same way: a 4,096-bin histogram of 12 leading score bits in LDS (one
// pass, the atomics spread over the bins), a suffix scan for the bin that holds the K-th largest key, and a rank sort
// of the keys at or above that bin -- a couple of hundred at most on any real map.  When more than kMaxK keys share
// that bin (plateaus) or fewer than K keys were counted, the exact radix selects above take over.
constexpr int kBins = 4096;          // stage 1: 12 bits below the sign
constexpr int kBins2 = 8192;         // stage 2: 13 bits from the sign down (its LDS has the room)
// scans hist[0..NT * BPT) from the top bin down, BPT bins per thread.  Returns through LDS (uniform values): the bin of
// the K-th largest counted key, the number of keys at or above it (INT_MAX when fewer than K keys were counted).
template <int NT, int BPT>
__device__ __forceinline__ void hist_threshold(const int* __restrict__ hist, int K, int* __restrict__ wave_tot,
                                               int& bin, int& nsel) {
    static_assert(BPT % 4 == 0, "16-byte reads");
    __shared__ int thr_bin, thr_sel;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (tid == 0) { thr_bin = 0; thr_sel = 0x7fffffff; }
    // thread t owns bins BPT * (NT - 1 - t) .. + BPT - 1: the inclusive scan over t counts the keys from the top bin down
    const int bin_base = (NT - 1 - tid) * BPT;
    int c[BPT];
#pragma unroll
    for (int g = 0; g < BPT / 4; ++g) {
        const int4 t = reinterpret_cast<const int4*>(hist + bin_base)[g];
        c[4 * g] = t.x; c[4 * g + 1] = t.y; c[4 * g + 2] = t.z; c[4 * g + 3] = t.w;
    }
    int local = 0;
#pragma unroll
    for (int j = 0; j < BPT; ++j) local += c[j];
    // inclusive scan over the wave in the data-parallel-primitive network: shifts inside the rows of 16, then the last
    // lane of row 0 / 2 into row 1 / 3 and lane 31 into rows 2 and 3
    int incl = local;
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xf, 0xf, false);      // row_shr:1
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xf, 0xf, false);      // row_shr:2
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xf, 0xf, false);      // row_shr:4
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xf, 0xf, false);      // row_shr:8
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x142, 0xa, 0xf, false);      // row_bcast:15 -> rows 1, 3
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x143, 0xc, 0xf, false);      // row_bcast:31 -> rows 2, 3
    if (lane == 63) wave_tot[wid] = incl;
    __syncthreads();
    int acc = incl - local;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) acc += w < wid ? wave_tot[w] : 0;
#pragma unroll
    for (int j = BPT - 1; j >= 0; --j) {
        if (acc < K && K <= acc + c[j]) { thr_bin = bin_base + j; thr_sel = acc + c[j]; }
        acc += c[j];
    }
    __syncthreads();
    bin = thr_bin;
    nsel = thr_sel;
}

// Stage 2's selection: the K largest of the n unique keys key_at(0..n), sorted, by threshold + rank sort.  Up to three
// histogram levels of 13 key bits each, from the sign bit down: a level whose K-th-key bin leaves more than kRankMax
// keys selected is refined inside that bin by the next 13 bits.  The keys are a concatenation of SORTED lists (stage
// 1's output), so equal digits come in runs: a run adds its length to its bin with two atomics -- minus its first index
// at the head, plus its last index + 1 at the tail -- instead of one contended atomic per key.  On success s.sel[0..K)
// holds the K largest keys in descending order; false (nothing written) when fewer than K keys exist or more than
// kMaxK keys still tie after 39 bits -- block_topk handles those.
constexpr int kRankMax = 512;
template <int NT, typename KeyAt>
__device__ __forceinline__ bool merge_select(KeyAt key_at, int n, int K, int* __restrict__ hist,
                                             int* __restrict__ wave_tot, SelectScratch& s) {
    const int tid = threadIdx.x, lane = tid & 63;
    uint64_t mask = 0, prefix = 0;                    // leading bits decided so far, and their value at the threshold
    int need = K, taken = 0, nsel = 0, shift = 51;
    constexpr int kDigit = kBins2 - 1, kBPT = kBins2 / NT;
    bool found = false;
    if (tid == 0) s.out_count = 0;
    for (int level = 0; level < 3 && !found; ++level, shift -= 13) {
#pragma unroll
        for (int g = 0; g < kBPT / 4; ++g) reinterpret_cast<int4*>(hist)[g * NT + tid] = make_int4(0, 0, 0, 0);
        __syncthreads();
        for (int i = tid; i < n; i += NT) {
            const uint64_t k = key_at(i);
            if ((k & mask) != prefix) continue;
            const int d = (int)(k >> shift) & kDigit;
            const uint64_t kp = i > 0 ? key_at(i - 1) : 0, kn = i + 1 < n ? key_at(i + 1) : 0;
            const bool head = !(i > 0 && (kp & mask) == prefix && ((int)(kp >> shift) & kDigit) == d);
            const bool tail = !(i + 1 < n && (kn & mask) == prefix && ((int)(kn >> shift) & kDigit) == d);
            if (head || tail) atomicAdd(&hist[d], (tail ? i + 1 : 0) - (head ? i : 0));
        }
        __syncthreads();
        int tb, cnt;
        hist_threshold<NT, kBPT>(hist, need, wave_tot, tb, cnt);
        if (cnt == 0x7fffffff) return false;
        const int at_tb = hist[tb];
        mask |= (uint64_t)kDigit << shift;
        prefix |= (uint64_t)tb << shift;
        nsel = taken + cnt;
        found = nsel <= (level == 2 ? kMaxK : kRankMax);
        taken += cnt - at_tb;                         // the keys above the bin are in for good
        need -= cnt - at_tb;
        __syncthreads();                              // every thread has read hist[tb] before the next level clears it
    }
    if (!found) return false;
    rank_sort_prepare<NT>(s, nsel);
    // selected: the decided bits are at or above the threshold's; a wave takes its slots in s.sel with one atomic
    int mine = 0;
    for (int i0 = 0; i0 < n; i0 += NT) {
        const int i = i0 + tid;
        mine += __popcll(__ballot(i < n && (key_at(min(i, n - 1)) & mask) >= prefix));
    }
    int at = 0;
    if (lane == 0 && mine) at = atomicAdd(&s.out_count, mine);
    at = __builtin_amdgcn_readfirstlane(at);
    for (int i0 = 0; i0 < n; i0 += NT) {
        const int i = i0 + tid;
        const uint64_t k = key_at(min(i, n - 1));
        const bool in = i < n && (k & mask) >= prefix;
        const unsigned long long m = __ballot(in);
        if (in) s.sel[at + __popcll(m & ((1ull << lane) - 1ull))] = k;
        at += __popcll(m);
    }
    __syncthreads();
    rank_sort_desc<NT>(s, nsel);
    return true;
}

// Stage 2's first choice.  The C lists arrive sorted, K entries each, so a bound on the K-th largest key is cheap: the
// ceil(K / C) best entries of every list are at least K keys, and the K-th largest of THEM (a rank sort of <= K + C keys)
// is a lower bound T0 of the K-th largest overall.  One binary search per list counts its keys >= T0, a scan gives each
// list its place in s.sel, the lists' heads are copied and the <= kMaxK survivors rank-sorted.  No pass over all C*K
// candidates (the histogram path makes two to four): identical classes leave about 2 K keys, one dominant class
// K + 2 C.  false (nothing usable in s.sel) when C or the survivors do not fit -- merge_select takes over.
template <int NT, typename KeyAt>
__device__ __forceinline__ bool seed_select(KeyAt key_at, int C, int K, int* __restrict__ cnt /* [NT] */,
                                            int* __restrict__ off /* [NT] */, int* __restrict__ wave_tot,
                                            SelectScratch& s) {
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int m = (K + C - 1) / C, ns = m * C;
    if (C > NT || ns > kMaxK) return false;
    rank_sort_prepare<NT>(s, ns);
    for (int i = tid; i < ns; i += NT) {
        const int c = i / m;
        s.sel[i] = key_at(c * K + (i - c * m));
    }
    __syncthreads();
    rank_sort_desc<NT>(s, ns);
    const uint64_t t0 = s.sel[K - 1];
    __syncthreads();                                  // everyone holds T0: s.sel is free again
    int mine = 0;
    if (tid < C) {
        int lo = 0, hi = K;                           // first position of list `tid` whose key is below T0
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (key_at(tid * K + mid) >= t0) lo = mid + 1; else hi = mid;
        }
        mine = lo;
    }
    // exclusive scan of the counts over the lists (wave scan in the DPP network + the wave totals)
    int incl = mine;
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xf, 0xf, false);      // row_shr:1
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xf, 0xf, false);      // row_shr:2
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xf, 0xf, false);      // row_shr:4
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xf, 0xf, false);      // row_shr:8
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x142, 0xa, 0xf, false);      // row_bcast:15 -> rows 1, 3
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x143, 0xc, 0xf, false);      // row_bcast:31 -> rows 2, 3
    if (lane == 63) wave_tot[wid] = incl;
    __syncthreads();
    int before = incl - mine, nsel = 0;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) {
        before += w < wid ? wave_tot[w] : 0;
        nsel += wave_tot[w];
    }
    if (nsel > kMaxK) return false;
    cnt[tid] = mine;
    off[tid] = before;
    rank_sort_prepare<NT>(s, nsel);
    __syncthreads();
    for (int c = wid; c < C; c += NT / 64) {          // a wave per list: its head goes to its place
        const int n = cnt[c], at = off[c];
        for (int j = lane; j < n; j += 64) s.sel[at + j] = key_at(c * K + j);
    }
    __syncthreads();
    rank_sort_desc<NT>(s, nsel);
    return true;
}

// Stage 1: one 1,024-thread workgroup per (b, c) plane.
//   A  the plane is copied from HBM into LDS once (coalesced 16-byte loads);
//   B  every thread evaluates the NMS of its pixels from LDS into registers (up to 32: planes up to 181 x 181 fit the
//      128 KB of dynamic LDS); the strictly positive scores are the keys (after the NMS 8/9 of a map is exactly zero;
//      first histogram level: score bits 30..19, exponent + 4 mantissa bits, 16 bins per octave);
//   C  hist_threshold;  D  the keys at or above the threshold bin are compacted (wave ballots, one LDS atomic per wave)
//      and rank-sorted straight into the candidate list.
// Maps with fewer than K positive scores or a plateau at the threshold (a trained model's background is clamped to
// exactly 1e-4: whole regions survive the NMS) write the NMS'd score bits over the raw plane and run lds_plane_topk;
// planes that do not fit the LDS run block_topk with the NMS recomputed from the L1/L2-resident plane.
// Round 6 -- SUB-PLANES: with few planes (the reference's 6 classes x 16 images = 96 workgroups on 256 CUs) a plane is cut
// into `nsub` bands of `sub_rows` rows, one workgroup each: the band's rows plus the row above and below it (the NMS
// window's reach: pad == 1) go to LDS as a small plane of its own, the halo rows' pixels are not candidates (score 0 on
// the fast path, the lowest possible order image on the general one), and every band leaves ITS K best with their pixel
// indices in the whole plane.  The union holds the plane's K best; stage 2 sees nsub lists per class, in band order --
// position (class, band, rank) still orders ties by class, then pixel index, as before.
constexpr int kPlaneThreads = 1024;
constexpr int kPlanePer = 32;        // pixels per thread held in registers
__global__ __launch_bounds__(kPlaneThreads) void plane_topk_kernel(const float* __restrict__ heat,
                                                                   uint64_t* __restrict__ cand, int Hfull, int W, int K,
                                                                   int KP, int pad, int lds_plane, int nsub, int sub_rows) {
    __shared__ SelectScratch s;
    __shared__ int hist[kBins];                       // later: the per-wave histograms of lds_plane_topk
    __shared__ int wave_tot[2 * (kPlaneThreads / 64)];
    extern __shared__ float plane_lds[];              // [HW] the raw plane, then (general path) its NMS'd score bits
    const int tid = threadIdx.x, lane = tid & 63;
    // band `sub` of plane `pl`: rows [r0, r1) + the halo row above (has_top) / below (has_bot); H, HW: the band's own
    // small plane, halo rows included; goff: pixel index in the whole plane = index in the band's plane + goff
    const int pl = nsub > 1 ? (int)blockIdx.x / nsub : (int)blockIdx.x, sub = nsub > 1 ? (int)blockIdx.x - pl * nsub : 0;
    const int r0 = sub * sub_rows, r1 = min(Hfull, r0 + sub_rows);
    const int has_top = (nsub > 1 && r0 > 0) ? 1 : 0, has_bot = (nsub > 1 && r1 < Hfull) ? 1 : 0;
    const int H = nsub > 1 ? (r1 - r0) + has_top + has_bot : Hfull;
    const int HW = H * W;
    const int goff = (r0 - has_top) * W;
    const int real_lo = has_top * W, real_hi = HW - has_bot * W;       // the band's own pixels: [real_lo, real_hi)
    const float* plane = heat + (size_t)pl * Hfull * W + goff;
    uint64_t* dst = cand + (size_t)blockIdx.x * K;
    if (!lds_plane) {
        block_topk<kPlaneThreads>([&](int i) { return make_key(nms_value(plane, H, W, i / W, i % W, pad), (uint32_t)i); },
                                  HW, K, KP, s);
        for (int i = tid; i < K; i += kPlaneThreads) dst[i] = s.sel[i];
        return;
    }
    // A
    if ((HW & 3) == 0) {
        for (int i = tid * 4; i < HW; i += kPlaneThreads * 4)
            *reinterpret_cast<float4*>(plane_lds + i) = *reinterpret_cast<const float4*>(plane + i);
    } else {
        for (int i = tid; i < HW; i += kPlaneThreads) plane_lds[i] = plane[i];
    }
    reinterpret_cast<int4*>(hist)[tid] = make_int4(0, 0, 0, 0);
    if (tid == 0) s.out_count = 0;
    __syncthreads();
    // B (a strictly positive float's own bits order like the float: bits 30..19 are the bin)
    const bool quads = pad == 1 && (W & 3) == 0;        // the reference's default 3x3 window on 4-aligned rows
    float v[kPlanePer];
    // pixel of v[u]: quads -- thread owns quads of consecutive pixels, v[4*q + j] <-> (q * NT + tid) * 4 + j
    auto pixel = [&](int u) { return quads ? ((u >> 2) * kPlaneThreads + tid) * 4 + (u & 3) : u * kPlaneThreads + tid; };
    const int rounds = (HW + 4 * kPlaneThreads - 1) / (4 * kPlaneThreads);   // groups of 4 registers in use (uniform)
    int x0 = (tid * 4) % W;                           // column of the thread's quad; the next one is 4 * NT pixels on
    const int x_step = (4 * kPlaneThreads) % W;
#pragma unroll
    for (int q = 0; q < kPlanePer / 4; ++q) {
        float o[4] = {0.f, 0.f, 0.f, 0.f};
        if (q < rounds) {
            if (quads) {
                const int i = (q * kPlaneThreads + tid) * 4;
                if (i < HW) nms_quad_lds(plane_lds, i, x0, W, HW, o);
                x0 += x_step;
                x0 -= x0 >= W ? W : 0;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int i = (4 * q + j) * kPlaneThreads + tid;
                    const int y = i / W;
                    if (i < HW) o[j] = nms_value(plane_lds, H, W, y, i - y * W, pad);
                }
            }
            if (nsub > 1) {            // halo rows are another band's pixels (a quad never straddles rows: W % 4 == 0)
                const int i0 = quads ? (q * kPlaneThreads + tid) * 4 : 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int i = quads ? i0 + j : (4 * q + j) * kPlaneThreads + tid;
                    if (i < real_lo || i >= real_hi) o[j] = 0.0f;
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (o[j] > 0.0f) atomicAdd(&hist[__float_as_uint(o[j]) >> 19], 1);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) v[4 * q + j] = o[j];
    }
    __syncthreads();
    // C
    int tb, nsel;
    hist_threshold<kPlaneThreads, kBins / kPlaneThreads>(hist, K, wave_tot, tb, nsel);
    if (nsel <= kMaxK && tb > 0) {
        // D: "bin >= tb" is one float compare against the bin's lower edge (bin 0 -- subnormal scores -- is left to the
        // general path); a wave takes its slots in s.sel with one atomic: the rank sort does not care about the order
        const float edge = __uint_as_float((uint32_t)tb << 19);
        rank_sort_prepare<kPlaneThreads>(s, nsel);
        int mine = 0;
#pragma unroll
        for (int q = 0; q < kPlanePer / 4; ++q) {
            if (q < rounds) {
#pragma unroll
                for (int j = 0; j < 4; ++j) mine += __popcll(__ballot(v[4 * q + j] >= edge));
            }
        }
        int at = 0;
        if (lane == 0 && mine) at = atomicAdd(&s.out_count, mine);
        at = __builtin_amdgcn_readfirstlane(at);
        const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
        for (int q = 0; q < kPlanePer / 4; ++q) {
            if (q < rounds) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float x = v[4 * q + j];
                    const unsigned long long m = __ballot(x >= edge);
                    if (x >= edge) s.sel[at + __popcll(m & below)] = make_key(x, (uint32_t)(pixel(4 * q + j) + goff));
                    at += __popcll(m);
                }
            }
        }
        __syncthreads();
        rank_sort_desc<kPlaneThreads>(s, nsel, dst, K);
        return;
    }
    // general case, LDS-resident (see lds_plane_topk): every thread is past its reads of the raw plane (barriers of C)
#pragma unroll
    for (int u = 0; u < kPlanePer; ++u) {
        const int pix = pixel(u);
        // (a halo pixel: order image 0, below every score's -- never selected while the band has K pixels of its own)
        if (pix < HW) reinterpret_cast<uint32_t*>(plane_lds)[pix] = (pix >= real_lo && pix < real_hi) ? float_order_bits(v[u]) : 0u;
    }
    __syncthreads();
    static_assert(sizeof(hist) >= (kPlaneThreads / 64) * 256 * sizeof(int), "the per-wave histograms fit");
    lds_plane_topk<kPlaneThreads>(reinterpret_cast<const uint32_t*>(plane_lds), HW, K, KP, s, hist, wave_tot);
    // (low word = ~index: index + goff <-> low word - goff; the order of the keys is unchanged)
    for (int i = tid; i < K; i += kPlaneThreads) dst[i] = s.sel[i] - (uint64_t)(uint32_t)goff;
}

// Stage 2: per image, top-K over C*K candidates + box assembly.  kThreads: 1,024; the 256- / 512-thread instances exist for the
// measurement of round 6 (cnuda_decode_set_stage2_threads; 900 candidates, B = 16: 8.98 / 7.01 / 7.42 us with 256 / 512 /
// 1,024 threads -- the kernel waits for its chain of dependent phases, not for its barriers: profiles/r6_decode_bands_ab.txt).
template <int kThreads>
__global__ __launch_bounds__(kThreads) void merge_decode_kernel(
    const uint64_t* __restrict__ cand, const float* __restrict__ wh, const float* __restrict__ reg,
    float* __restrict__ dets, int64_t* __restrict__ inds,
    int C /* candidate lists per image: classes x bands */, int H, int W, int K, int KP, int wh_ch, int rotated, int lds_cand,
    int nsub) {
    __shared__ SelectScratch s;
    extern __shared__ uint64_t cand_lds[];            // [C*K] this image's candidates (0 bytes: too many, read in place)
    const int b = blockIdx.x;
    const int HW = H * W;
    const int n = C * K;
    const uint64_t* cb = cand + (size_t)b * n;
    if (lds_cand) {
        // one pipelined pass over the candidates (16-byte loads, four in flight per thread): the selection below reads
        // every key three to five times and the box assembly once more, each a dependent trip to L2 otherwise
        const int n2 = n >> 1;                        // cb is 16-byte aligned (256-byte workspace base, n * 8 per image
        const bool aligned = (((size_t)b * n) & 1) == 0;   // ... when b * n is even)
        if (aligned) {
            const uint4* src = reinterpret_cast<const uint4*>(cb);
            uint4* dst4 = reinterpret_cast<uint4*>(cand_lds);
            for (int i = threadIdx.x; i < n2; i += 4 * kThreads) {
                uint4 t[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) t[u] = src[min(i + u * kThreads, n2 - 1)];
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (i + u * kThreads < n2) dst4[i + u * kThreads] = t[u];
            }
            if ((n & 1) && threadIdx.x == 0) cand_lds[n - 1] = cb[n - 1];
        } else {
            for (int i = threadIdx.x; i < n; i += kThreads) cand_lds[i] = cb[i];
        }
        __syncthreads();
    }
    // (the body twice, so that each copy knows its address space: LDS reads or global loads, not flat ones)
    auto body = [&](const uint64_t* __restrict__ cb) {
    // second-stage key: same score bits, position c*K+rank as the index
    auto key2 = [&](int i) { return (cb[i] & 0xffffffff00000000ull) | (uint64_t)(0xffffffffu - (uint32_t)i); };
    // the same selection as stage 1, from the sign bit down (candidates of a map with fewer than K positive scores are
    // zero or, outside the reference's use, negative)
    {
        __shared__ int hist[kBins2];                  // (seed_select: two arrays of one entry per list)
        __shared__ int wave_tot[kThreads / 64];
        static_assert(kBins2 >= 2 * kThreads, "the lists' counts and places fit the histogram");
        // few candidates (the reference's 6 classes: 900): one histogram pass is cheaper than two sorts (6.8 vs 8.4 us);
        // many (80 classes: 12,000): the seed bound spares the passes over all of them (17.5 -> 13.6 us)
        const bool seeded = n > 4096 && seed_select<kThreads>(key2, C, K, hist, hist + kThreads, wave_tot, s);
        if (!seeded && !merge_select<kThreads>(key2, n, K, hist, wave_tot, s)) block_topk<kThreads>(key2, n, K, KP, s);
    }
    const int ncol = rotated ? 7 : 6;
    for (int k = threadIdx.x; k < K; k += kThreads) {
        const uint64_t key = s.sel[k];
        const uint32_t pos = 0xffffffffu - (uint32_t)(key & 0xffffffffu);
        const int cls = (int)(pos / (uint32_t)(K * nsub));
        const float score = order_bits_float((uint32_t)(key >> 32));
        const uint32_t idx = 0xffffffffu - (uint32_t)(cb[pos] & 0xffffffffu);
        float xs = (float)(int)(idx % (uint32_t)W);
        float ys = (float)(int)(idx / (uint32_t)W);
        if (reg) {
            xs += reg[((size_t)b * 2 + 0) * HW + idx];
            ys += reg[((size_t)b * 2 + 1) * HW + idx];
        } else {
            xs += 0.5f;
            ys += 0.5f;
        }
        const float w = wh[((size_t)b * wh_ch + 0) * HW + idx];
        const float h = wh[((size_t)b * wh_ch + 1) * HW + idx];
        float* d = dets + ((size_t)b * K + k) * ncol;
        if (!rotated) {
            d[0] = xs - w / 2.0f;
            d[1] = ys - h / 2.0f;
            d[2] = xs + w / 2.0f;
            d[3] = ys + h / 2.0f;
            d[4] = score;
            d[5] = (float)cls;
        } else {
            const float a = wh[((size_t)b * wh_ch + 2) * HW + idx];
            float sg = 1.0f / (1.0f + expf(-a));
            sg = fminf(fmaxf(sg, 1e-4f), 1.0f - 1e-4f);
            d[0] = xs;
            d[1] = ys;
            d[2] = w;
            d[3] = h;
            d[4] = sg * 360.0f - 180.0f;
            d[5] = score;
            d[6] = (float)cls;
        }
        if (inds) inds[(size_t)b * K + k] = (int64_t)idx;
    }
    };
    if (lds_cand) body(cand_lds);
    else body(cb);
}

constexpr int kMaxSub = 4;
// bands per plane (plane_topk_kernel): while the planes alone leave CUs idle -- twice the workgroups per step, as long as
// every band keeps >= 16 rows, >= 2048 pixels and >= 4 K pixels of its own (the general path never runs out of real
// pixels) and the grid stays within two workgroups per CU.  3x3 window and 4-aligned rows only (the halo is one row).
// Measured (round 6, B = 16, C = 6, 128 x 128, rocprofv3, profiles/r6_decode_bands_ab.txt): stage 1 takes 12.1 / 11.0 / 15.1 us
// with 1 / 2 / 4 bands -- the kernel is a chain of latency-bound phases (barriers, dependent LDS round trips), NOT bound by
// its pixels: half the pixels save 9 %, and four bands put two 1,024-thread workgroups on a CU, each 1.4x slower.  Default 2.
constexpr int kDefaultSub = 2;
int g_decode_stage2_threads = 0;     // CNUDA_DECODE_STAGE2_THREADS (measurements): 0 = by the candidate count
int g_decode_max_sub = kDefaultSub;  // cnuda_decode_set_max_bands (tests, measurements)
int pick_bands(int planes, int H, int W, int K, int pad, bool lds_plane) {
    int s = 1;
    if (!lds_plane || pad != 1 || (W & 3)) return 1;
    while (s * 2 <= g_decode_max_sub && planes * s * 2 <= 512 && H % (s * 2) == 0 && H / (s * 2) >= 16 &&
           (long long)(H / (s * 2)) * W >= 2048 && (long long)(H / (s * 2)) * W >= 4ll * K)
        s *= 2;
    return s;
}

int next_pow2(int v) {
    int p = 2;
    while (p < v) p <<= 1;
    return p;
}

}  // namespace
}  // namespace cnuda

using namespace cnuda;

extern "C" int cnuda_decode_set_stage2_threads(int threads) {
    const int prev = g_decode_stage2_threads;
    g_decode_stage2_threads = (threads == 256 || threads == 512 || threads == 1024) ? threads : 0;
    return prev;
}
extern "C" int cnuda_decode_set_max_bands(int max_bands) {
    const int prev = g_decode_max_sub;
    g_decode_max_sub = max_bands < 1 ? kDefaultSub : (max_bands > kMaxSub ? kMaxSub : max_bands);
    return prev;
}

extern "C" size_t cnuda_decode_workspace_bytes(int B, int C, int H, int W, int K) {
    (void)H; (void)W;
    return (size_t)B * C * K * kMaxSub * sizeof(uint64_t) + 256;     // stage-1 candidates (up to kMaxSub bands per plane)
}

extern "C" int cnuda_nms(const float* heat, float* out, int B, int C, int H, int W, int nms_size,
                         cnuda_stream_t stream) {
    CNUDA_REQUIRE(heat && out, "cnuda_nms: null pointer");
    CNUDA_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, "cnuda_nms: empty tensor");
    CNUDA_REQUIRE(nms_size >= 1 && (nms_size & 1), "cnuda_nms: nms_size must be odd, got %d", nms_size);
    const long long total = (long long)B * C * H * W;
    CNUDA_LAUNCH(nms_kernel, dim3(stream_grid(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       heat, out, (long long)B * C, H, W, (nms_size - 1) / 2);
    return check_launch("cnuda_nms");
}

extern "C" int cnuda_decode_detection(const float* heat, const float* wh, const float* reg,
                                      float* dets, int64_t* inds,
                                      int B, int C, int H, int W, int K, int wh_ch, int rotated, int nms_size,
                                      void* workspace, size_t workspace_bytes, cnuda_stream_t stream) {
    CNUDA_REQUIRE(heat && wh && dets, "cnuda_decode_detection: null pointer");
    CNUDA_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, "cnuda_decode_detection: empty tensor");
    CNUDA_REQUIRE(nms_size >= 1 && (nms_size & 1), "cnuda_decode_detection: nms_size must be odd, got %d", nms_size);
    // torch.topk raises when k exceeds the row length (decode.py:18)
    CNUDA_REQUIRE(K >= 1 && (long long)K <= (long long)H * W, "selected index k out of range (K=%d, H*W=%d)", K, H * W);
    CNUDA_REQUIRE(K <= kMaxK, "cnuda_decode_detection: K=%d exceeds the supported maximum %d", K, kMaxK);
    CNUDA_REQUIRE(wh_ch >= (rotated ? 3 : 2), "cnuda_decode_detection: wh has %d channels", wh_ch);
    CNUDA_REQUIRE(workspace && workspace_bytes >= cnuda_decode_workspace_bytes(B, C, H, W, K),
                  "cnuda_decode_detection: workspace too small");
    const int KP = next_pow2(K);
    const int pad = (nms_size - 1) / 2;
    uintptr_t base = ((uintptr_t)workspace + 255) & ~(uintptr_t)255;
    uint64_t* cand = reinterpret_cast<uint64_t*>(base);
    hipStream_t st = (hipStream_t)stream;
    // dynamic LDS: the plane's score bits (for the LDS-resident general path) when they fit beside the static
    // arrays (about 27 KB) in the CU's 160 KB: planes up to 180 x 180
    const size_t bits_bytes = (size_t)H * W * sizeof(uint32_t);
    const int lds_plane = bits_bytes <= 128 * 1024 ? 1 : 0;
    CNUDA_REQUIRE(raise_dynamic_lds(reinterpret_cast<const void*>(plane_topk_kernel), 128 * 1024, 128 * 1024) &&
                      raise_dynamic_lds(reinterpret_cast<const void*>(merge_decode_kernel<1024>), 112 * 1024, 112 * 1024) &&
                      raise_dynamic_lds(reinterpret_cast<const void*>(merge_decode_kernel<512>), 112 * 1024, 112 * 1024) &&
                      raise_dynamic_lds(reinterpret_cast<const void*>(merge_decode_kernel<256>), 112 * 1024, 112 * 1024),
                  "cnuda_decode_detection: dynamic LDS");
    const int nsub = pick_bands(B * C, H, W, K, pad, lds_plane != 0), sub_rows = H / nsub;
    const size_t band_bytes = nsub > 1 ? (size_t)(sub_rows + 2) * W * sizeof(uint32_t) : bits_bytes;
    CNUDA_LAUNCH(plane_topk_kernel, dim3(B * C * nsub), dim3(kPlaneThreads), lds_plane ? band_bytes : 0, st, heat, cand,
                       H, W, K, KP, pad, lds_plane, nsub, sub_rows);
    int rc = check_launch("cnuda_decode_detection(stage 1)");
    if (rc) return rc;
    // stage 2 keeps an image's C*K candidates in LDS when they fit beside its 42 KB of static arrays
    const size_t cand_bytes = (size_t)C * nsub * K * sizeof(uint64_t);
    const int lds_cand = cand_bytes <= 112 * 1024 ? 1 : 0;
    const int small2 = g_decode_stage2_threads ? g_decode_stage2_threads : 1024;      // (256 / 512: measured no faster, see above)
    if (small2 == 256)
        CNUDA_LAUNCH(merge_decode_kernel<256>, dim3(B), dim3(256), lds_cand ? cand_bytes : 0, st,
                           cand, wh, reg, dets, inds, C * nsub, H, W, K, KP, wh_ch, rotated ? 1 : 0, lds_cand, nsub);
    else if (small2 == 512)
        CNUDA_LAUNCH(merge_decode_kernel<512>, dim3(B), dim3(512), lds_cand ? cand_bytes : 0, st,
                           cand, wh, reg, dets, inds, C * nsub, H, W, K, KP, wh_ch, rotated ? 1 : 0, lds_cand, nsub);
    else
        CNUDA_LAUNCH(merge_decode_kernel<1024>, dim3(B), dim3(1024), lds_cand ? cand_bytes : 0, st,
                           cand, wh, reg, dets, inds, C * nsub, H, W, K, KP, wh_ch, rotated ? 1 : 0, lds_cand, nsub);
    return check_launch("cnuda_decode_detection(stage 2)");
}
