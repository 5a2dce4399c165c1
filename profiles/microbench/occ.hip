#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(256, 2) void k(float* o) { extern __shared__ float s[]; s[threadIdx.x] = 1; __syncthreads(); o[threadIdx.x] = s[255 - threadIdx.x]; }
int main() {
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int bytes : {65536, 78000, 79872, 80592, 81920, 82000}) {
        int n = -1; hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k, 256, bytes);
        printf("dyn LDS %d B: %d blocks per CU (%s)\n", bytes, n, hipGetErrorString(e));
    }
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("sharedMemPerMultiprocessor %zu maxSharedMemoryPerBlock %zu regs/CU %d\n", p.maxSharedMemoryPerMultiProcessor, p.sharedMemPerBlock, p.regsPerMultiprocessor);
    return 0;
}
