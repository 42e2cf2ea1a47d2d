// Achievable MFMA issue rate on this box: register-only loops, NACC independent accumulators per wave,
// WAVES waves per SIMD.  hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NACC, int KIND>
__global__ void __launch_bounds__(256) k(float* out, int iters) {
    f32x16 acc[NACC];
    for (int j = 0; j < NACC; ++j) for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.0f;
    bf16x8 ab, bb;
    for (int i = 0; i < 8; ++i) { ab[i] = (__bf16)a; bb[i] = (__bf16)b; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int j = 0; j < NACC; ++j) {
                if constexpr (KIND == 0) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
                else acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc[j], 0, 0, 0);
            }
    }
    float s = 0;
    for (int j = 0; j < NACC; ++j) for (int i = 0; i < 16; ++i) s += acc[j][i];
    if (s == 12345.f) out[0] = s;
}
template <int NACC, int KIND>
void run(int wgs_per_cu, const char* name) {
    float* d; hipMalloc(&d, 4);
    const int iters = 2000, grid = 256 * wgs_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC, KIND><<<grid, 256>>>(d, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NACC, KIND><<<grid, 256>>>(d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops_per = KIND == 0 ? 2.0 * 32 * 32 * 2 : 2.0 * 32 * 32 * 16;
    const double tf = (double)grid * 4 * iters * 16 * NACC * flops_per / (ms * 1e-3) / 1e12;
    printf("%s nacc=%d waves/simd=%d : %.3f ms  %.1f TF\n", name, NACC, wgs_per_cu, ms, tf);
    hipFree(d);
}
int main() {
    run<1, 0>(1, "f32 32x32x2 "); run<4, 0>(1, "f32 32x32x2 "); run<1, 0>(2, "f32 32x32x2 "); run<4, 0>(2, "f32 32x32x2 "); run<4, 0>(4, "f32 32x32x2 ");
    run<1, 1>(1, "bf16 32x32x16"); run<4, 1>(1, "bf16 32x32x16"); run<1, 1>(2, "bf16 32x32x16"); run<4, 1>(2, "bf16 32x32x16"); run<4, 1>(4, "bf16 32x32x16");
    return 0;
}
