// Small helper kernels around the implicit-GEMM cores: weight packing, split-K
// slab reduction, per-channel sums.
#include "igemm_host.h"

namespace cnuda {
namespace {

__global__ void pack_kernel(const float* __restrict__ W, float* __restrict__ dst, int Co, int C, int T,
                            int mode, int Kp, int Mp, int Cpad) {
    const long long total = (long long)Kp * Mp;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int k = (int)(i / Mp), m = (int)(i % Mp);
        int o = -1, c = -1, tap = -1;
        if (mode == PACK_FWD) {
            if (k < T * C && m < Co) { tap = k / C; c = k % C; o = m; }
        } else {   // PACK_DGRAD: Cpad = Co rounded up to the K chunk, rows o >= Co stay zero
            if (k < T * Cpad && m < C) { tap = k / Cpad; o = k % Cpad; c = m; if (o >= Co) o = -1; }
        }
        dst[i] = (o >= 0) ? W[((size_t)o * C + c) * T + tap] : 0.0f;
    }
}

struct TapList { int n; int tap[9]; };
// dst[k = ti*Co + o][m = c] = W[o][c][taps.tap[ti]]   (transposed-conv operand restricted to a tap subset)
__global__ void pack_taps_kernel(const float* __restrict__ W, float* __restrict__ dst, int Co, int C, int T,
                                 TapList taps, int Kp, int Mp) {
    const long long total = (long long)Kp * Mp;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int k = (int)(i / Mp), m = (int)(i % Mp);
        float v = 0.0f;
        if (k < taps.n * Co && m < C) {
            const int ti = k / Co, o = k - ti * Co;
            v = W[((size_t)o * C + m) * T + taps.tap[ti]];
        }
        dst[i] = v;
    }
}

// gw[o][c][tap] = sum_z slabs[z][o][tap*C + c].  Workgroup = 64 consecutive outputs x 4 waves; wave w sums
// slabs w, w+4, w+8, ... (coalesced 256-B rows), the four partial sums are added in wave order through LDS:
// a fixed summation order (bit-reproducible) with 4x the parallelism of one thread per output.
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ gw,
                                                          int Z, int Mp, int Jp, int Co, int C, int T) {
    __shared__ float part[4][64];
    const long long total = (long long)Co * C * T;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long long i = (long long)blockIdx.x * 64 + lane;
    float s = 0.0f;
    if (i < total) {
        // enumerate in slab order (j = tap*C + c contiguous): coalesced reads, strided (small) writes
        const int K = T * C;
        const int j = (int)(i % K), o = (int)(i / K);
        const size_t off = (size_t)o * Jp + j;
        for (int z = w; z < Z; z += 4) s += slabs[(size_t)z * Mp * Jp + off];
    }
    part[w][lane] = s;
    __syncthreads();
    if (w == 0 && i < total) {
        const int K = T * C;
        const int j = (int)(i % K), o = (int)(i / K);
        const int tap = j / C, c = j - tap * C;
        gw[((size_t)o * C + c) * T + tap] = ((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane];
    }
}

// grid (channel, batch image): fixed-order block tree per (c, b) plane, then a second tiny kernel sums
// the B partials in order -> reproducible, and B times more workgroups than one per channel
__global__ __launch_bounds__(256) void channel_sum_partial_kernel(const float* __restrict__ x,
                                                                  float* __restrict__ partial, int C, long long HW) {
    __shared__ float red[16];
    const int c = blockIdx.x, b = blockIdx.y;
    const float* p = x + ((size_t)b * C + c) * HW;
    float s = 0.0f;
    if ((HW & 3) == 0) {
        for (long long i = threadIdx.x * 4; i < HW; i += 1024) {
            const float4 v = *reinterpret_cast<const float4*>(p + i);
            s += (v.x + v.y) + (v.z + v.w);
        }
    } else {
        for (long long i = threadIdx.x; i < HW; i += 256) s += p[i];
    }
    s = block_sum(s, red);
    if (threadIdx.x == 0) partial[(size_t)c * gridDim.y + b] = s;
}
__global__ void channel_sum_final_kernel(const float* __restrict__ partial, float* __restrict__ out, int C, int B) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float s = 0.0f;
    for (int b = 0; b < B; ++b) s += partial[(size_t)c * B + b];
    out[c] = s;
}
__global__ __launch_bounds__(256) void channel_sum_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                          int B, int C, long long HW) {
    __shared__ float red[16];
    const int c = blockIdx.x;
    float s = 0.0f;
    for (int b = 0; b < B; ++b) {
        const float* p = x + ((size_t)b * C + c) * HW;
        for (long long i = threadIdx.x; i < HW; i += 256) s += p[i];
    }
    s = block_sum(s, red);
    if (threadIdx.x == 0) out[c] = s;
}

}  // namespace

void launch_pack(const float* W, float* dst, int Co, int C, int T, PackMode mode, int Kp, int Mp, int Cpad,
                 hipStream_t st) {
    const long long total = (long long)Kp * Mp;
    hipLaunchKernelGGL(pack_kernel, dim3(stream_grid(total, 256)), dim3(256), 0, st, W, dst, Co, C, T, (int)mode, Kp,
                       Mp, Cpad);
}

void launch_pack_taps(const float* W, float* dst, int Co, int C, int T, const int* taps, int ntaps, int Kp, int Mp,
                      hipStream_t st) {
    TapList tl;
    tl.n = ntaps;
    for (int i = 0; i < 9; ++i) tl.tap[i] = i < ntaps ? taps[i] : 0;
    const long long total = (long long)Kp * Mp;
    hipLaunchKernelGGL(pack_taps_kernel, dim3(stream_grid(total, 256)), dim3(256), 0, st, W, dst, Co, C, T, tl, Kp, Mp);
}

void launch_slab_reduce(const float* slabs, float* gw, int Z, int Mp, int Jp, int Co, int C, int T, hipStream_t st) {
    const long long total = (long long)Co * C * T;
    hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)((total + 63) / 64)), dim3(256), 0, st, slabs, gw, Z, Mp, Jp,
                       Co, C, T);
}

void launch_channel_sum(const float* x, float* out, int B, int C, long long HW, hipStream_t st, float* scratch) {
    if (scratch && B > 1) {
        hipLaunchKernelGGL(channel_sum_partial_kernel, dim3(C, B), dim3(256), 0, st, x, scratch, C, HW);
        hipLaunchKernelGGL(channel_sum_final_kernel, dim3((C + 63) / 64), dim3(64), 0, st, scratch, out, C, B);
    } else {
        hipLaunchKernelGGL(channel_sum_kernel, dim3(C), dim3(256), 0, st, x, out, B, C, HW);
    }
}

}  // namespace cnuda
