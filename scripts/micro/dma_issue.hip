// microbenchmark: what one global_load_lds_dwordx4 (1 KiB per wave instruction, L2-resident source) costs the ISSUING wave, as a
// function of how many waves of the workgroup issue at the same time and how many instructions each issues back to back; and
// whether MFMA work of OTHER waves on the same SIMD proceeds meanwhile.
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/dma_issue.hip -o /tmp/dma_issue && /tmp/dma_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// issuers = number of waves (0..15 of 16) that issue `per` DMA instructions; the other waves run `mfmas` MFMAs (0 = idle).
// out[wave] = cycles the wave spent in its section (s_memtime), averaged over `iters` barrier-separated repetitions.
__global__ void __launch_bounds__(1024) k(const unsigned char* src, unsigned long long* out, int issuers, int per, int mfmas, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t base = (uint32_t)(uintptr_t)smem;
    const unsigned char* p = src + (size_t)blockIdx.x * 65536 + wave * 4096;
    f32x16 acc = {0};
    bf16x8 a = {1, 1, 1, 1, 1, 1, 1, 1}, b = a;
    unsigned long long tot = 0;
    for (int it = 0; it < iters; ++it) {
        __builtin_amdgcn_s_barrier();
        const unsigned long long t0 = __builtin_readcyclecounter();
        if (wave < issuers) {
            for (int i = 0; i < per; ++i) {
                const uint32_t dst = __builtin_amdgcn_readfirstlane(base + wave * 4096 + (i & 3) * 1024);
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"((uint32_t)lane * 16u + (uint32_t)(i & 3) * 1024u), "s"(p), "s"(dst) : "memory");
            }
        } else {
            for (int i = 0; i < mfmas; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        }
        const unsigned long long t1 = __builtin_readcyclecounter();
        tot += t1 - t0;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (lane == 0) out[blockIdx.x * 16 + wave] = tot / iters;
    if (acc[0] == 12345.f) out[0] = 1;
}

int main() {
    unsigned char* src; unsigned long long* out;
    hipMalloc(&src, 256 * 65536 + 4096); hipMemset(src, 1, 256 * 65536 + 4096);
    hipMalloc(&out, 256 * 16 * 8);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    unsigned long long h[256 * 16];
    const int cfg[][3] = {{1, 1, 0}, {1, 4, 0}, {4, 1, 0}, {4, 4, 0}, {16, 1, 0}, {16, 2, 0}, {16, 5, 0}, {8, 2, 0}, {8, 5, 0}, {8, 5, 16}, {8, 2, 16}, {0, 0, 16}, {4, 5, 16}};
    for (auto& c : cfg) {
        for (int grid : {1, 256}) {
            k<<<grid, 1024, 65536>>>(src, out, c[0], c[1], c[2], 200);
            hipDeviceSynchronize();
            hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
            double iss = 0, oth = 0; int ni = 0, no = 0;
            for (int g = 0; g < grid; ++g) for (int w = 0; w < 16; ++w) { if (w < c[0]) { iss += h[g * 16 + w]; ++ni; } else { oth += h[g * 16 + w]; ++no; } }
            printf("grid %3d: %2d waves x %d DMA, others %2d MFMAs: issuing waves %7.0f cycles (%.0f per instruction), other waves %7.0f cycles\n", grid, c[0], c[1], c[2],
                   ni ? iss / ni : 0.0, (ni && c[1]) ? iss / ni / c[1] : 0.0, no ? oth / no : 0.0);
        }
    }
    return 0;
}
