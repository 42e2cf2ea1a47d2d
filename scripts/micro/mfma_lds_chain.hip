// One wave per workgroup, W workgroups per CU: per "tile" 16 ds_read_b128 (both operands of 32 fp32 MFMAs) from LDS holding
// random data (16-byte pieces XOR-swizzled with the row: conflict-free), wait, 32 DEPENDENT v_mfma_f32_32x32x2_f32.  How much of the fp32 MFMA peak does that pattern reach?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int CHAINS, int PRIO>
__global__ void __launch_bounds__(64) k(const float* src, float* out, int tiles) {
    // PRIO 1: static asymmetric priority by hardware wave slot (breaks the lockstep of identical waves sharing a SIMD)
    // PRIO 2: raise priority for the MFMA phase only
    if (PRIO == 1) { if (__builtin_amdgcn_s_getreg(6148) & 1) __builtin_amdgcn_s_setprio(3); }
    __shared__ __attribute__((aligned(16))) float lds[4096];     // 16 KiB: two operand tiles of 32 x 64 fp32
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = src[i];
    __syncthreads();
    const int lane = threadIdx.x, r = lane & 31, hh = lane >> 5;
    f32x16 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c) for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
    for (int t = 0; t < tiles; ++t) {
        float a[32], b[32];
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const float4 x = *reinterpret_cast<const float4*>(&lds[r * 64 + ((((4 * hh + g + t) & 15) ^ (r & 15)) * 4)]);
            const float4 y = *reinterpret_cast<const float4*>(&lds[2048 + r * 64 + ((((4 * hh + g + t) & 15) ^ (r & 15)) * 4)]);
            a[4 * g] = x.x; a[4 * g + 1] = x.y; a[4 * g + 2] = x.z; a[4 * g + 3] = x.w;
            b[4 * g] = y.x; b[4 * g + 1] = y.y; b[4 * g + 2] = y.z; b[4 * g + 3] = y.w;
        }
        if (PRIO == 2) __builtin_amdgcn_s_setprio(3);
#pragma unroll
        for (int i = 0; i < 32; ++i) acc[i % CHAINS] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[i], acc[i % CHAINS], 0, 0, 0);
        if (PRIO == 2) __builtin_amdgcn_s_setprio(0);
    }
    float s = 0;
    for (int c = 0; c < CHAINS; ++c) for (int i = 0; i < 16; ++i) s += acc[c][i];
    if (s == 12345.f) out[0] = s;
}
template <int CHAINS, int PRIO>
void run(int wpc, const float* d, float* o) {
    const int tiles = 400, grid = 256 * wpc;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<CHAINS, PRIO><<<grid, 64>>>(d, o, 10); hipDeviceSynchronize();
    hipEventRecord(e0); k<CHAINS, PRIO><<<grid, 64>>>(d, o, tiles); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("chains %d prio-mode %d, %2d waves/CU: %.3f ms  %.1f TF\n", CHAINS, PRIO, wpc, ms, (double)grid * tiles * 32 * 4096.0 / (ms * 1e-3) / 1e12);
}
int main() {
    float* h = (float*)malloc(4096 * 4); for (int i = 0; i < 4096; ++i) h[i] = (float)rand() / RAND_MAX - 0.5f;
    float *d, *o; hipMalloc(&d, 4096 * 4); hipMalloc(&o, 4); hipMemcpy(d, h, 4096 * 4, hipMemcpyHostToDevice);
    for (int wpc : {4, 8, 16}) { run<1, 0>(wpc, d, o); run<1, 1>(wpc, d, o); run<1, 2>(wpc, d, o); }
    return 0;
}
