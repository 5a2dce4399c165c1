// Host-side helpers shared by conv.hip and dcn.hip (packing of the small weight
// operand, split-K slab reduction, workspace carving).
#pragma once
#include <stdlib.h>

#include "common.h"

namespace cnuda {

enum PackMode {
    PACK_FWD = 0,    // dst[k = tap*C + c][m = o]        (forward / wgrad column order)
    PACK_DGRAD = 1,  // dst[k = tap*Cpad + o][m = c]     (transposed conv; Cpad = Co rounded up to 16)
    // (2: tap subsets, launch_pack_taps)
    // halo-tile kernels (hconv.cuh; 3x3, stride 1, padding 1): K ordered (16-channel group, tap, channel in group), so that
    // the nine taps of a channel group -- which all read the same LDS halo tile -- are consecutive chunks
    PACK_HALO_FWD = 3,    // dst[k = (g*9 + tap)*16 + ci][m = o] = W[o][16g + ci][tap]                     (C % 16 == 0)
    PACK_HALO_DGRAD = 4,  // dst[k = (g*9 + tap)*16 + oi][m = c] = W[16g + oi][c][8 - tap]: the input gradient of a
                          // 3x3 / stride 1 / padding 1 convolution is that convolution over grad_y with flipped taps (Co % 16 == 0)
};

inline int round_up(int v, int q) { return (v + q - 1) / q * q; }

// Pixel splits of a weight-gradient launch: the workgroup count (tiles x splits) should fill the chip's resident
// slots a whole number of times -- 256 CUs x the workgroups of that tile shape one CU holds (set by the tile's LDS
// image: 64x64 33 KB -> 4, 64x128 / 128x64 / 32x128 50 / 50 / 41 KB -> 3, 128x128 66 KB -> 2).  The flat target of
// 1024 workgroups left the three-per-CU shapes with 1.33 rounds: a second round on a third of the chip.
inline long long wgrad_splits(long long tiles, int bm, int bj, long long max_z) {
    const int per_cu = (bm == 128 && bj == 128) ? 2 : ((bm == 64 && bj == 64) ? 4 : 3);
    long long z = (256ll * per_cu) / tiles;      // exactly one round (two rounds measured the same)
    if (z > max_z) z = max_z;
    return z < 1 ? 1 : z;
}

// W is the reference layout [Co][C][T] (T = kh*kw).  The packed image is [Kp][Mp], zero padded.  Returns the
// buffer the GEMM must read: `dst` (the caller's workspace, freshly packed) or -- when the caller announced a
// weight identity for this call (cnuda_pack_stamp) and a pack cache is attached (cnuda_pack_cache_attach) -- a slot
// of the cache, packed only when the weights have changed since it was filled.  `room` = bytes available at dst
// (the split-operand mode keeps a second image behind the f32 matrix: ig_a_bytes()).
const float* launch_pack(const float* W, float* dst, size_t room, int Co, int C, int T, PackMode mode,
                         int Kp, int Mp, int Cpad, hipStream_t st);

// dst[k = ti*Co + o][m = c] = W[o][c][taps[ti]], zero padded to [Kp][Mp]; same return convention
const float* launch_pack_taps(const float* W, float* dst, size_t room, int Co, int C, int T, const int* taps, int ntaps,
                              int Kp, int Mp, hipStream_t st);

// gw[o][c][tap] = sum_z slabs[z][o][tap*C + c]; with bslab ([Z][Mp]: the GEMM's per-split row sums of grad_y, written by
// igemm_wgrad_*_kernel when it is given the pointer) also gb[o] = sum_z bslab[z][o] -- the bias gradient, same launch
void launch_slab_reduce(const float* slabs, float* gw, int Z, int Mp, int Jp,
                        int Co, int C, int T, hipStream_t st, const float* bslab = nullptr, float* gb = nullptr);

// out[c] = sum over (b, hw) of x[b][c][hw]   (bias gradients of the paths without a weight-gradient GEMM)
void launch_channel_sum(const float* x, float* out, int B, int C, long long HW, hipStream_t st);

// small-channel, full-resolution convolutions with LDS halo tiles (smallc.hip)
// Apply on load: the input is a convolution output whose train-mode BatchNorm + ReLU is applied while the kernel stages it
// (smallc.hip); mean / invstd [groups][C] as cnuda_bn_train_forward* saved them, gamma / beta [C].
struct SmallNorm {
    const float *mean, *invstd, *gamma, *beta;
    int imgs_per_group;
};
bool smallc_norm_supported(int C, int Co, int kh, int kw, int sh, int sw);     // the instances compiled with the transform
bool smallc_supported(int C, int Co, int kh, int kw, int sh, int sw);
size_t smallc_workspace_bytes(int B, int C, int H, int W, int Co, int kh, int kw, int s, int ph, int pw);
int smallc_forward(const float* x, const float* w, const float* bias, float* y, int B, int C, int H, int W, int Co,
                   int kh, int kw, int s, int ph, int pw, float act_slope, int transposed, void* ws, size_t ws_bytes,
                   hipStream_t st, float* stats = nullptr, const SmallNorm* norm = nullptr);
int smallc_stats_blocks(int B, int C, int H, int W, int Co, int kh, int kw, int s, int ph, int pw, int* blocks_per_image,
                        int* rows);
int smallc_backward_weight(const float* x, const float* gy, float* gw, int B, int C, int H, int W, int Co, int kh,
                           int kw, int s, int ph, int pw, void* ws, size_t ws_bytes, hipStream_t st,
                           const SmallNorm* norm = nullptr);

struct Carver {
    uintptr_t cur, end;
    Carver(void* ws, size_t bytes) : cur(((uintptr_t)ws + 255) & ~(uintptr_t)255), end((uintptr_t)ws + bytes) {}
    template <typename T> T* take(size_t count) {
        T* p = reinterpret_cast<T*>(cur);
        cur = (cur + count * sizeof(T) + 255) & ~(uintptr_t)255;
        return p;
    }
    bool ok() const { return cur <= end; }
};
inline size_t carve_bytes(size_t count, size_t elem) { return (count * elem + 255) / 256 * 256; }

}  // namespace cnuda
