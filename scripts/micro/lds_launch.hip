// microbenchmark: cost of launching N workgroups that each request L bytes of dynamic LDS (empty kernel)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(256) empty_k(int* out, int flag) {
    extern __shared__ unsigned char smem[];
    if (flag) { smem[threadIdx.x] = 1; out[blockIdx.x] = smem[0]; }
}
__global__ void __launch_bounds__(512) empty_k512(int* out, int flag) {
    extern __shared__ unsigned char smem[];
    if (flag) { smem[threadIdx.x] = 1; out[blockIdx.x] = smem[0]; }
}
int main() {
    int* d; hipMalloc(&d, 1 << 20);
    hipFuncSetAttribute((const void*)empty_k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)empty_k512, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grids[] = {256, 512, 1024, 4096};
    const int ldss[] = {0, 8 << 10, 16 << 10, 32 << 10, 64 << 10, 80 << 10, 128 << 10, 160 << 10};
    for (int g : grids) for (int l : ldss) {
        for (int i = 0; i < 5; ++i) empty_k<<<g, 256, l>>>(d, 0);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int i = 0; i < 50; ++i) empty_k<<<g, 256, l>>>(d, 0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("grid %5d block 256 lds %6d : %8.2f us per launch\n", g, l, ms * 1000 / 50);
    }
    for (int l : ldss) {
        for (int i = 0; i < 5; ++i) empty_k512<<<512, 512, l>>>(d, 0);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int i = 0; i < 50; ++i) empty_k512<<<512, 512, l>>>(d, 0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("grid   512 block 512 lds %6d : %8.2f us per launch\n", l, ms * 1000 / 50);
    }
    return 0;
}
