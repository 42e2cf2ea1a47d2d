// Issue interval of the 16-bit MFMA shapes on this box: register-only loops, 4 independent accumulators per wave, 4 waves per SIMD;
// reports cycles per instruction and SIMD at the clock the loop ran at (s_memtime counts at 100 MHz: the wall clock is used instead,
// with the 2.4 GHz peak clock and the measured TF side by side).
// hipcc --offload-arch=gfx950 -O3 mfma_rates.hip -o /tmp/mfma_rates && /tmp/mfma_rates
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ void __launch_bounds__(256) k(float* out, int iters) {
    f32x16 A[4]; f32x4 a[4];
    for (int j = 0; j < 4; ++j) { for (int i = 0; i < 16; ++i) A[j][i] = 0.f; for (int i = 0; i < 4; ++i) a[j][i] = 0.f; }
    bf16x8 ab, bb; s16x4 as, bs;
    for (int i = 0; i < 8; ++i) { ab[i] = (__bf16)(threadIdx.x * 1e-3f); bb[i] = (__bf16)1.0f; }
    for (int i = 0; i < 4; ++i) { as[i] = (short)threadIdx.x; bs[i] = 0x3f80; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if constexpr (KIND == 0) A[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, A[j], 0, 0, 0);
                else if constexpr (KIND == 1) a[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, a[j], 0, 0, 0);
                else if constexpr (KIND == 2) A[j] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(as, bs, A[j], 0, 0, 0);
                else a[j] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(as, bs, a[j], 0, 0, 0);
            }
    }
    float s = 0;
    for (int j = 0; j < 4; ++j) { for (int i = 0; i < 16; ++i) s += A[j][i]; for (int i = 0; i < 4; ++i) s += a[j][i]; }
    if (s == 12345.f) out[0] = s;
}
template <int KIND>
void run(const char* name, double flops_per) {
    float* d; hipMalloc(&d, 4);
    const int iters = 2000, grid = 256 * 4;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<KIND><<<grid, 256>>>(d, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<KIND><<<grid, 256>>>(d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double per_simd = (double)4 * iters * 64;            // instructions per SIMD (4 waves)
    printf("%-28s %.3f ms  %7.1f TF   %.1f ns per instruction and SIMD = %.1f cycles at 2.4 GHz\n", name, ms,
           (double)grid * 4 * iters * 64 * flops_per / (ms * 1e-3) / 1e12, ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4);
    hipFree(d);
}
int main() {
    run<0>("v_mfma_f32_32x32x16_bf16", 2.0 * 32 * 32 * 16);
    run<1>("v_mfma_f32_16x16x32_bf16", 2.0 * 16 * 16 * 32);
    run<2>("v_mfma_f32_32x32x8_bf16_1k", 2.0 * 32 * 32 * 8);
    run<3>("v_mfma_f32_16x16x16_bf16_1k", 2.0 * 16 * 16 * 16);
    return 0;
}
