// Detection decode for gfx950: 3x3 (or kxk) max-pool NMS + two-stage top-K +
// box assembly.  Replaces backends/decode.py:6-76 of the reference.
//
// Stage 1 (one 512-thread workgroup per (b, c) plane): the plane is read once from HBM (neighbours come from
// L1).  After the NMS almost every score is exactly 0, so the strictly positive ones are first compacted into
// LDS (wave ballots, one LDS atomic per wave and 64 pixels); when at least K and at most 2048 survive -- every
// real heat map -- the K best are simply the head of an in-LDS bitonic sort of those few keys.  Otherwise
// (flat or pathological maps) an 8-bit MSB radix select over all pixels' 64-bit order keys runs instead (LDS
// histograms + one-wave suffix scan; its LDS atomics serialise on the all-zero bucket, which is why it is not
// the common path).  Stage 2 (one workgroup per image) runs the radix select over the C*K candidates and
// assembles the boxes.
//
// Order key: high 32 bits = order-preserving image of the fp32 score, low 32
// bits = ~index, so "larger key" == "higher score, or equal score and lower
// index" -- a strict total order (the reference's torch.topk leaves ties
// unspecified; see include/centernet_uda_hip.h).
#include "common.h"

namespace cnuda {
namespace {

constexpr int kThreads = 1024;
constexpr int kMaxK = 1024;

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
    uint64_t sel[kMaxK];
};

// Sorts n <= NT unique keys sel[0..n) (LDS) descending, in place, by counting: key q is served by P = NT / pow2(n)
// adjacent lanes (at most a wave) which compare it with every P-th key -- broadcast LDS reads, no barrier inside the
// loop -- and add their counts through shuffles; a key's rank is its final position.  n / P iterations and two
// barriers, where a bitonic network over pow2(n) keys takes log^2 stages with a workgroup barrier each (36 for 256).
// With `out` the keys of rank < K go straight to out[rank] (global memory) instead.
template <int NT>
__device__ __forceinline__ void rank_sort_desc(uint64_t* __restrict__ sel, int n, uint64_t* __restrict__ out = nullptr,
                                               int K = 0) {
    const int tid = threadIdx.x;
    int sh = 0;                                           // log2(P)
    while (sh < 6 && ((n << (sh + 1)) <= NT)) ++sh;
    const int P = 1 << sh;
    const int q = tid >> sh, part = tid & (P - 1);
    const bool have = q < n;
    const uint64_t key = have ? sel[q] : 0;
    int rank = 0;
    for (int j = part; j < n; j += P) rank += sel[j] > key ? 1 : 0;
    for (int o = 1; o < P; o <<= 1) rank += __shfl_xor(rank, o, 64);
    if (out) {
        if (have && part == 0 && rank < K) out[rank] = key;
        return;
    }
    __syncthreads();
    if (have && part == 0) sel[rank] = key;
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
    static_assert(kMaxK <= NT, "one key per thread at least");
    rank_sort_desc<NT>(s.sel, min(s.out_count, KP));
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
// than K or more than kPool survivors cost the same as any other map.  Result: s.sel[0..K) sorted descending.
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
    rank_sort_desc<NT>(s.sel, min(gt_total + min(eq_total, need), KP));
}

// Bitonic sort (descending) of a 2048-key LDS pool by 1024 threads, two keys per thread in registers (positions 2t,
// 2t+1): stride 1 is a compare inside the thread, strides 2..64 exchange with lane t ^ (stride/2) by shuffle, only
// strides >= 128 cross waves and go through LDS -- 10 of the 66 stages need workgroup barriers.  Ends with the
// sorted keys in `pool` (barrier included).
constexpr int kPool = 2048;          // positive-score candidates the fast path sorts
__device__ __forceinline__ void pool_sort_desc(uint64_t* __restrict__ pool, int tid) {
    uint64_t k0 = pool[2 * tid], k1 = pool[2 * tid + 1];
    const int p0 = 2 * tid;
    for (int size = 2; size <= kPool; size <<= 1) {
        const bool desc = (p0 & size) == 0;            // same for both keys of the thread (size >= 2)
        for (int stride = size >> 1; stride >= 128; stride >>= 1) {
            __syncthreads();                           // previous readers of the pool are done
            pool[p0] = k0; pool[p0 + 1] = k1;
            __syncthreads();
            const uint64_t o0 = pool[p0 ^ stride], o1 = pool[(p0 + 1) ^ stride];
            const bool lower = (p0 & stride) == 0;
            const bool take_max = lower == desc;
            k0 = take_max ? (k0 > o0 ? k0 : o0) : (k0 < o0 ? k0 : o0);
            k1 = take_max ? (k1 > o1 ? k1 : o1) : (k1 < o1 ? k1 : o1);
        }
        for (int stride = size >> 1 < 64 ? size >> 1 : 64; stride >= 2; stride >>= 1) {
            const uint64_t o0 = __shfl_xor(k0, stride >> 1, 64), o1 = __shfl_xor(k1, stride >> 1, 64);
            const bool lower = (p0 & stride) == 0;
            const bool take_max = lower == desc;
            k0 = take_max ? (k0 > o0 ? k0 : o0) : (k0 < o0 ? k0 : o0);
            k1 = take_max ? (k1 > o1 ? k1 : o1) : (k1 < o1 ? k1 : o1);
        }
        {   // stride 1: the thread's own pair
            const uint64_t hi = k0 > k1 ? k0 : k1, lo = k0 > k1 ? k1 : k0;
            k0 = desc ? hi : lo;
            k1 = desc ? lo : hi;
        }
    }
    __syncthreads();
    pool[p0] = k0; pool[p0 + 1] = k1;
    __syncthreads();
}

// Stage 1.
constexpr int kPlaneThreads = 1024;
static_assert(kPool == 2 * kPlaneThreads, "two keys per thread");
__global__ __launch_bounds__(kPlaneThreads) void plane_topk_kernel(const float* __restrict__ heat,
                                                                   uint64_t* __restrict__ cand, int H, int W, int K,
                                                                   int KP, int pad, int lds_plane) {
    __shared__ SelectScratch s;
    __shared__ uint64_t pool[kPool];
    __shared__ int count;
    __shared__ int wave_count[kPlaneThreads / 64];
    extern __shared__ uint32_t plane_bits[];          // [HW] order bits of the NMS'd scores (0 bytes: plane too large)
    const int tid = threadIdx.x, lane = tid & 63;
    const int HW = H * W;
    const float* plane = heat + (size_t)blockIdx.x * HW;
    uint64_t* dst = cand + (size_t)blockIdx.x * K;
    if (tid == 0) count = 0;
    __syncthreads();
    // compaction of the strictly positive NMS'd scores without LDS atomics (16 waves hitting one LDS counter
    // serialise: 21 of 28 us in the first version): a super-chunk of 16 pixels per thread is evaluated into
    // registers, every wave counts its survivors with ballots, the 16 wave totals are scanned through LDS and
    // every wave then writes its keys at its own offsets.
    constexpr int kPer = 16, kWaves = kPlaneThreads / 64;
    const bool quads = pad == 1 && (W & 3) == 0;        // the reference's default 3x3 window on 4-aligned rows
    const bool lds_bits = lds_plane != 0;              // host: the plane's score bits fit the dynamic LDS
    __shared__ int wave_count2[2 * (kPlaneThreads / 64)];
    const int wid = tid >> 6;
    int filled = 0;                                   // survivors of the previous super-chunks (uniform)
    for (int s0 = 0; s0 < HW; s0 += kPer * kPlaneThreads) {
        float v[kPer];
        int mine = 0;
        if (quads) {
            // thread owns 4 quads of consecutive pixels: v[4*q + j] <-> pixel s0 + (q * NT + tid) * 4 + j
#pragma unroll
            for (int q = 0; q < kPer / 4; ++q) {
                const int i = s0 + (q * kPlaneThreads + tid) * 4;
                float o[4] = {0.f, 0.f, 0.f, 0.f};
                if (i < HW) { const int y = i / W; nms_quad(plane, H, W, y, i - y * W, o); }
#pragma unroll
                for (int j = 0; j < 4; ++j) v[4 * q + j] = o[j];
            }
        } else {
#pragma unroll
            for (int u = 0; u < kPer; ++u) {
                const int i = s0 + u * kPlaneThreads + tid;
                const int y = i / W;
                v[u] = i < HW ? nms_value(plane, H, W, y, i - y * W, pad) : 0.0f;
            }
        }
#pragma unroll
        for (int u = 0; u < kPer; ++u) mine += __popcll(__ballot(v[u] > 0.0f));
        if (lane == 0) wave_count[wid] = mine;
        __syncthreads();
        int base = filled, total = 0;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) {
            const int c = wave_count[w];
            base += w < wid ? c : 0;
            total += c;
        }
#pragma unroll
        for (int u = 0; u < kPer; ++u) {
            const bool pos = v[u] > 0.0f;
            const unsigned long long m = __ballot(pos);
            const int slot = base + __popcll(m & ((1ull << lane) - 1ull));
            const int pix = quads ? s0 + ((u >> 2) * kPlaneThreads + tid) * 4 + (u & 3) : s0 + u * kPlaneThreads + tid;
            if (lds_bits && pix < HW) plane_bits[pix] = float_order_bits(v[u]);
            if (pos && slot < kPool) pool[slot] = make_key(v[u], (uint32_t)pix);
            base += __popcll(m);
        }
        filled += total;
        __syncthreads();                              // wave_count is reused by the next super-chunk
    }
    if (tid == 0) count = filled;
    __syncthreads();
    const int M = count;
    if (M >= K && M <= kPool) {
        // every key outside the pool has score <= 0 < the pool's: the top K are the K largest pool keys.  They are not
        // found by sorting the pool (66 bitonic stages over 2,048 keys: 31.5 of this kernel's 56 thousand cycles) but
        // by a threshold: a 4,096-bin histogram of the 12 score bits below the sign (exponent + 4 mantissa bits: 16
        // bins per octave) laid over the pool's own 16 KB once the keys are in registers, a suffix scan for the bin
        // holding the K-th key, and a rank sort of the few keys at or above that bin (a noise map: ~190 of 1,800).
        const int i0 = 2 * tid, i1 = i0 + 1;
        const uint64_t k0 = i0 < M ? pool[i0] : 0, k1 = i1 < M ? pool[i1] : 0;
        __syncthreads();
        int* hist = reinterpret_cast<int*>(pool);
        static_assert(sizeof(pool) == 4 * kPlaneThreads * sizeof(int), "four bins per thread");
        reinterpret_cast<int4*>(hist)[tid] = make_int4(0, 0, 0, 0);
        __syncthreads();
        const int b0 = (int)(k0 >> 51) & 0xfff, b1 = (int)(k1 >> 51) & 0xfff;
        if (i0 < M) atomicAdd(&hist[b0], 1);
        if (i1 < M) atomicAdd(&hist[b1], 1);
        __syncthreads();
        // thread t owns bins 4 * (1023 - t) .. + 3: the inclusive scan over t counts the keys from the top bin down
        const int bin_base = (kPlaneThreads - 1 - tid) * 4;
        const int4 c = reinterpret_cast<const int4*>(hist)[kPlaneThreads - 1 - tid];
        const int local = c.x + c.y + c.z + c.w;
        int incl = local;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(incl, o, 64);
            if (lane >= o) incl += t;
        }
        if (lane == 63) wave_count[wid] = incl;
        __syncthreads();
        int acc = incl - local;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) acc += w < wid ? wave_count[w] : 0;
        __shared__ int thr_bin, thr_sel;
        {
            const int cs[4] = {c.w, c.z, c.y, c.x};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (acc < K && K <= acc + cs[j]) { thr_bin = bin_base + 3 - j; thr_sel = acc + cs[j]; }
                acc += cs[j];
            }
        }
        __syncthreads();
        const int tb = thr_bin, nsel = thr_sel;
        if (nsel <= kMaxK) {
            const bool s0 = i0 < M && b0 >= tb, s1 = i1 < M && b1 >= tb;
            const unsigned long long m0 = __ballot(s0), m1 = __ballot(s1);
            const unsigned long long below = (1ull << lane) - 1ull;
            if (lane == 0) wave_count2[wid] = __popcll(m0) + __popcll(m1);
            __syncthreads();
            int at = 0;
#pragma unroll
            for (int w = 0; w < kWaves; ++w) at += w < wid ? wave_count2[w] : 0;
            if (s0) s.sel[at + __popcll(m0 & below)] = k0;
            if (s1) s.sel[at + __popcll(m0) + __popcll(m1 & below)] = k1;
            __syncthreads();
            rank_sort_desc<kPlaneThreads>(s.sel, nsel, dst, K);
            return;
        }
        // a plateau at the threshold (more than kMaxK keys share its 12 bits): sort the whole pool instead
        pool[i0] = k0;
        pool[i1] = k1;
        __syncthreads();
        pool_sort_desc(pool, tid);
        for (int i = tid; i < K; i += kPlaneThreads) dst[i] = pool[i];
        return;
    }
    if (lds_bits) {
        // general case, LDS-resident (see lds_plane_topk): the whole plane's NMS'd scores are already in LDS; the
        // sorting pool's memory is free again and serves as the per-wave histograms
        static_assert(sizeof(pool) >= (kPlaneThreads / 64) * 256 * sizeof(int), "histograms fit the pool");
        lds_plane_topk<kPlaneThreads>(plane_bits, HW, K, KP, s, reinterpret_cast<int*>(pool), wave_count2);
        for (int i = tid; i < K; i += kPlaneThreads) dst[i] = s.sel[i];
        return;
    }
    // planes too large for the LDS: exact select with the NMS recomputed per pass from the L1/L2-resident plane
    block_topk<kPlaneThreads>([&](int i) { return make_key(nms_value(plane, H, W, i / W, i % W, pad), (uint32_t)i); },
                              HW, K, KP, s);
    for (int i = tid; i < K; i += kPlaneThreads) dst[i] = s.sel[i];
}

// Stage 2: per image, top-K over C*K candidates + box assembly.
__global__ __launch_bounds__(kThreads) void merge_decode_kernel(
    const uint64_t* __restrict__ cand, const float* __restrict__ wh, const float* __restrict__ reg,
    float* __restrict__ dets, int64_t* __restrict__ inds,
    int C, int H, int W, int K, int KP, int wh_ch, int rotated) {
    __shared__ SelectScratch s;
    const int b = blockIdx.x;
    const int HW = H * W;
    const int n = C * K;
    const uint64_t* cb = cand + (size_t)b * n;
    // second-stage key: same score bits, position c*K+rank as the index
    auto key2 = [&](int i) { return (cb[i] & 0xffffffff00000000ull) | (uint64_t)(0xffffffffu - (uint32_t)i); };
    if (n <= kPool) {
        // few classes (the reference's default 6 x 150 = 900 candidates).  Every class list arrives SORTED (stage 1), so
        // no sort is needed: a candidate's final rank is its rank in its own list plus, for every other list, the number
        // of entries above it -- one binary search per list in LDS (keys are unique: ranks are too).  Candidates whose
        // running rank reaches K drop out at once.  (The 66-stage bitonic sort this replaces: 16 -> ~6 us per call.)
        __shared__ uint64_t pool2[kPool];
        for (int i = threadIdx.x; i < n; i += kThreads) pool2[i] = key2(i);
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += kThreads) {
            const uint64_t key = pool2[i];
            const int c = i / K;
            int rank = i - c * K;
            for (int cc = 0; cc < C && rank < K; ++cc) {
                if (cc == c) continue;
                const uint64_t* list = pool2 + cc * K;
                int lo = 0, hi = K;                       // first position whose key is below `key`
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (list[mid] > key) lo = mid + 1; else hi = mid;
                }
                rank += lo;
            }
            if (rank < K) s.sel[rank] = key;
        }
        __syncthreads();
    } else {
        block_topk<kThreads>(key2, n, K, KP, s);
    }
    const int ncol = rotated ? 7 : 6;
    for (int k = threadIdx.x; k < K; k += kThreads) {
        const uint64_t key = s.sel[k];
        const uint32_t pos = 0xffffffffu - (uint32_t)(key & 0xffffffffu);
        const int cls = (int)(pos / (uint32_t)K);
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
}

int next_pow2(int v) {
    int p = 2;
    while (p < v) p <<= 1;
    return p;
}

}  // namespace
}  // namespace cnuda

using namespace cnuda;

extern "C" size_t cnuda_decode_workspace_bytes(int B, int C, int H, int W, int K) {
    (void)H; (void)W;
    return (size_t)B * C * K * sizeof(uint64_t) + 256;     // stage-1 candidates
}

extern "C" int cnuda_nms(const float* heat, float* out, int B, int C, int H, int W, int nms_size,
                         cnuda_stream_t stream) {
    CNUDA_REQUIRE(heat && out, "cnuda_nms: null pointer");
    CNUDA_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, "cnuda_nms: empty tensor");
    CNUDA_REQUIRE(nms_size >= 1 && (nms_size & 1), "cnuda_nms: nms_size must be odd, got %d", nms_size);
    const long long total = (long long)B * C * H * W;
    hipLaunchKernelGGL(nms_kernel, dim3(stream_grid(total, 256)), dim3(256), 0, (hipStream_t)stream,
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
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(plane_topk_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL(plane_topk_kernel, dim3(B * C), dim3(kPlaneThreads), lds_plane ? bits_bytes : 0, st, heat, cand,
                       H, W, K, KP, pad, lds_plane);
    int rc = check_launch("cnuda_decode_detection(stage 1)");
    if (rc) return rc;
    hipLaunchKernelGGL(merge_decode_kernel, dim3(B), dim3(kThreads), 0, st,
                       cand, wh, reg, dets, inds, C, H, W, K, KP, wh_ch, rotated ? 1 : 0);
    return check_launch("cnuda_decode_detection(stage 2)");
}
