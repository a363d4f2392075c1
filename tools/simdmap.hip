// Which SIMD does wave w of a 512-thread workgroup run on?  (HW_REG_HW_ID: wave slot [3:0], SIMD [5:4], CU [11:8], SE ..)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(512, 1) k(unsigned* out) {
    extern __shared__ unsigned char smem[];
    unsigned id = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = id;
    if (threadIdx.x == 100000) smem[0] = 1;
}
int main() {
    unsigned* d;
    hipMalloc(&d, 64 * 8 * 4);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    hipLaunchKernelGGL(k, dim3(64), dim3(512), 150 * 1024, 0, d);
    unsigned h[64 * 8];
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    for (int b = 0; b < 6; ++b) {
        printf("block %d:", b);
        for (int w = 0; w < 8; ++w) printf("  w%d simd %u slot %u cu %u", w, (h[b * 8 + w] >> 4) & 3, h[b * 8 + w] & 15, (h[b * 8 + w] >> 8) & 15);
        printf("\n");
    }
    return 0;
}
