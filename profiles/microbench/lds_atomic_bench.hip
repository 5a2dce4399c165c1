#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
// mode 0: ds_add_f32 (no return); 1: ds_read_b32 + add + ds_write_b32; 2: ds_read_b128+4 add+ds_write_b128
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, int stride) {
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    float v = 1.0f + lane;
    int a = (wid * 2048 + lane * stride) & 8191;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            float* p = lds + ((a + j * 64 * (MODE == 2 ? 4 : 1)) & 8191 & ~(MODE == 2 ? 3 : 0));
            if (MODE == 0) __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else if (MODE == 1) { float c = *(volatile float*)p; *(volatile float*)p = c + v; }
            else { f4 c = *(f4*)p; asm volatile("" : "+v"(c)); c += v; *(f4*)p = c; asm volatile("" ::: "memory"); }
        }
    }
    __syncthreads();
    out[blockIdx.x * 256 + threadIdx.x] = lds[threadIdx.x];
}
template <int MODE> void run(const char* name, int stride) {
    float* out; hipMalloc(&out, 1024 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 1000;
    hipLaunchKernelGGL(k<MODE>, dim3(512), dim3(256), 32768, 0, out, 10, stride);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(512), dim3(256), 32768, 0, out, iters, stride);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // 512 WGs over 256 CUs: 2 WGs per CU -> per CU: 2*4 waves * iters*16 instrs
    double instr_per_cu = 2.0 * 4 * iters * 16;
    printf("%-28s stride %2d: %.3f ms  -> %.1f cycles per wave-instruction per CU (at 2.4 GHz)\n", name, stride, ms, ms * 1e-3 * 2.4e9 / instr_per_cu);
}
int main() {
    for (int stride : {1, 2, 32, 0}) {
        run<0>("ds_add_f32", stride);
        run<1>("ds_read+add+ds_write b32", stride);
        run<2>("ds_read+4add+ds_write b128", stride ? stride * 4 : 0);
    }
    return 0;
}
