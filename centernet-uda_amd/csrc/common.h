// Shared device/host helpers for the gfx950 (CDNA4) kernels of this library.
// Wavefront = 64 lanes everywhere; no other target is supported.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/centernet_uda_hip.h"

namespace cnuda {

constexpr int kWave = 64;

void set_error(const char* fmt, ...);

// Every entry point funnels its launch through this check: a failed launch is
// an error code for the caller, never a printf (the reference only printf'd,
// libs/DCNv2/src/cuda/dcn_v2_im2col_cuda.cu:346-350).
inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}

// Optional in-library kernel timer (bench.py's roofline leg): when armed through
// cnuda_prof_arm(tag), the next ProfScope brackets exactly one main-kernel launch
// with hipEvents on the launch stream.  An entry point made of several kernels opens a
// ProfGroup first: every ProfScope inside it is recorded under the armed tag, told apart
// by `sub` (bits 24.. of the collected tag).  Inert (two int compares) otherwise.
// The launcher NAMES what it launched (`name()`: the kernel template instance it selected, printf-style), so the
// report never depends on a second copy of the selection rules outside the library.
class ProfScope {
public:
    explicit ProfScope(hipStream_t st, int sub = 0);
    ~ProfScope();
    bool active() const { return rec_ >= 0; }
    void name(const char* fmt, ...);
private:
    hipStream_t st_;
    int rec_;
};
constexpr int kProfNameLen = 96;
bool prof_active();      // the kernel timer is enabled (cnuda_prof_enable(n > 0))
class ProfGroup {
public:
    ProfGroup();
    ~ProfGroup();
};

// Launch log (tests/test_zz_kernel_coverage.py): while enabled through cnuda_launch_log_enable(1), every kernel
// launch of the library records its host function pointer; cnuda_launch_log_collect resolves the distinct pointers to
// the kernels' own symbol names (hipKernelNameRefByPtr, demangled) -- the names a rocprofv3 kernel trace shows.  One
// predictable branch per launch otherwise.
extern bool g_launch_log_on;
void launch_log(const void* host_function);
#define CNUDA_LAUNCH(kernel, grid, block, lds, st, ...)                                               \
    do {                                                                                              \
        if (::cnuda::g_launch_log_on) ::cnuda::launch_log(reinterpret_cast<const void*>(&kernel));    \
        hipLaunchKernelGGL(kernel, grid, block, lds, st, __VA_ARGS__);                                \
    } while (0)

#define CNUDA_REQUIRE(cond, ...)                 \
    do {                                         \
        if (!(cond)) {                           \
            ::cnuda::set_error(__VA_ARGS__);     \
            return CNUDA_ERR_INVALID_ARGUMENT;   \
        }                                        \
    } while (0)

// Dynamic LDS beyond 64 KiB is opt-in per kernel AND per device (hipFuncAttributeMaxDynamicSharedMemorySize): raised
// once per (kernel, current device), the return code checked.  `lds` <= 64 KiB: nothing to do.  false = the runtime
// refused (the error text is set).
bool raise_dynamic_lds(const void* host_function, size_t lds, size_t limit = 160 * 1024);

// Matrix-pipe mode of the implicit GEMMs: 0 = f32 MFMA (v_mfma_f32_32x32x2_f32), 1 = split operands on the bf16
// MFMA (igemm.cuh, "X3").  Process-wide; set through cnuda_set_matrix_mode() or CNUDA_MATRIX_MODE at load time.
int matrix_mode();
// Wave-specialised (producer / consumer, 8-wave) variants of the f32 implicit-GEMM kernels for the 64- and 128-row
// tiles: on unless CNUDA_WS=0 (read once).  +3-8 % per launch where it applies, ~1.2 % of the benched step.
bool wave_specialised();

inline int ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }

// grid cap for grid-stride memory-bound kernels: 256 CUs x 8 resident blocks
constexpr int kMaxStreamBlocks = 2048;
inline int stream_grid(long long work_items, int block) {
    long long g = (work_items + block - 1) / block;
    if (g < 1) g = 1;
    return (int)(g > kMaxStreamBlocks ? kMaxStreamBlocks : g);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Block-wide sum for blockDim.x a multiple of 64 and <= 1024; result valid in
// every thread.  `red` must hold >= 16 elements of T.
template <typename T>
__device__ __forceinline__ T block_sum(T v, T* red) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) red[wid] = v;
    __syncthreads();
    T r = (lane < nw) ? red[lane] : T(0);
    r = wave_sum(r);
    return r;
}

}  // namespace cnuda
