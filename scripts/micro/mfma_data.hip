// Does the sustained MFMA rate depend on the operand DATA?  Register-only loops (no memory in the timed loop), 4 waves per
// SIMD, 4 independent accumulators, operands either constants or random values held in registers.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int KIND>
__global__ void __launch_bounds__(256) k(const float* src, float* out, int iters, int random) {
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
    float a[16], b[16];
    bf16x8 ab[4], bb[4];
    for (int i = 0; i < 16; ++i) {
        a[i] = random ? src[(threadIdx.x * 16 + i) & 4095] : 0.5f;
        b[i] = random ? src[(threadIdx.x * 16 + i + 2048) & 4095] : 1.0f;
    }
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 8; ++e) { ab[i][e] = (__bf16)a[(4 * i + e) & 15]; bb[i][e] = (__bf16)b[(4 * i + e + 3) & 15]; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if constexpr (KIND == 0) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[(u + j) & 15], acc[j], 0, 0, 0);
                else acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab[(u + j) & 3], bb[u & 3], acc[j], 0, 0, 0);
            }
    }
    float s = 0;
    for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) s += acc[j][i];
    if (s == 12345.f) out[0] = s;
}
template <int KIND>
void run(const float* d, float* o, int random, const char* name) {
    const int iters = 3000, grid = 256 * 4;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<KIND><<<grid, 256>>>(d, o, 10, random); hipDeviceSynchronize();
    hipEventRecord(e0); k<KIND><<<grid, 256>>>(d, o, iters, random); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double fl = KIND == 0 ? 4096.0 : 32768.0;
    printf("%s %s operands: %.3f ms  %.1f TF\n", name, random ? "random  " : "constant", ms, (double)grid * 4 * iters * 64 * fl / (ms * 1e-3) / 1e12);
}
int main() {
    float* h = (float*)malloc(4096 * 4); for (int i = 0; i < 4096; ++i) h[i] = (float)rand() / RAND_MAX - 0.5f;
    float *d, *o; hipMalloc(&d, 4096 * 4); hipMalloc(&o, 4); hipMemcpy(d, h, 4096 * 4, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
        run<0>(d, o, 0, "f32  32x32x2 "); run<0>(d, o, 1, "f32  32x32x2 ");
        run<1>(d, o, 0, "bf16 32x32x16"); run<1>(d, o, 1, "bf16 32x32x16");
    }
    return 0;
}
