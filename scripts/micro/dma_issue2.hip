// microbenchmark 2: what makes one LDS-DMA instruction cost the issuing wave ~120 cycles?  Variants of the issue sequence:
//   V0  save m0 / set m0 / s_nop / global_load_lds_dwordx4 / restore m0     (glds16_saddr of the library, round 2)
//   V1  set m0 / s_nop / DMA                                                 (m0 left clobbered)
//   V2  set m0 once per group of 4, the four DMAs use the instruction offset (0, 1024, 2048, 3072): no M0 write in between
//   V3  V2 with the group's M0 save / restore (what a library routine could ship)
//   V4  8 MFMAs issued first, then V2's DMAs (does the wave's own matrix work hide the issue?)
//   V5  V2's DMAs first, then 8 MFMAs
//   V6  8 MFMAs only
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/dma_issue2.hip -o /tmp/dma_issue2 && /tmp/dma_issue2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int V>
__global__ void __launch_bounds__(1024) k(const unsigned char* src, unsigned long long* out, int issuers, int groups, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t base = (uint32_t)(uintptr_t)smem;
    const unsigned char* p = src + (size_t)blockIdx.x * 65536 + wave * 4096;
    f32x16 acc0 = {0}, acc1 = {0}, acc2 = {0}, acc3 = {0};
    bf16x8 a = {1, 1, 1, 1, 1, 1, 1, 1}, b = a;
    const uint32_t voff = (uint32_t)lane * 16u;
    const uint32_t dst = __builtin_amdgcn_readfirstlane(base + wave * 4096);
    unsigned long long tot = 0;
    for (int it = 0; it < iters; ++it) {
        __builtin_amdgcn_s_barrier();
        const unsigned long long t0 = __builtin_readcyclecounter();
        if (wave < issuers) {
            if constexpr (V == 4) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc2, 0, 0, 0); acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc3, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc2, 0, 0, 0); acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc3, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (V != 6) {
                for (int g = 0; g < groups; ++g) {
                    unsigned keep;
                    if constexpr (V == 0) {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                                         : "=&s"(keep) : "v"(voff + i * 1024u), "s"(p), "s"(dst + i * 1024u) : "memory");
                    } else if constexpr (V == 1) {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff + i * 1024u), "s"(p), "s"(dst + i * 1024u) : "memory");
                    } else if constexpr (V == 3) {
                        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\t"
                                     "global_load_lds_dwordx4 %1, %2 offset:1024\n\tglobal_load_lds_dwordx4 %1, %2 offset:2048\n\t"
                                     "global_load_lds_dwordx4 %1, %2 offset:3072\n\ts_mov_b32 m0, %0"
                                     : "=&s"(keep) : "v"(voff), "s"(p), "s"(dst) : "memory");
                    } else {
                        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1\n\t"
                                     "global_load_lds_dwordx4 %0, %1 offset:1024\n\tglobal_load_lds_dwordx4 %0, %1 offset:2048\n\t"
                                     "global_load_lds_dwordx4 %0, %1 offset:3072"
                                     : : "v"(voff), "s"(p), "s"(dst) : "memory");
                    }
                }
            }
            if constexpr (V == 5 || V == 6) {
                __builtin_amdgcn_sched_barrier(0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc2, 0, 0, 0); acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc3, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc2, 0, 0, 0); acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc3, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        const unsigned long long t1 = __builtin_readcyclecounter();
        tot += t1 - t0;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (lane == 0) out[blockIdx.x * 16 + wave] = tot / iters;
    if (acc0[0] + acc1[0] + acc2[0] + acc3[0] == 12345.f) out[0] = 1;
}

template <int V>
void run(const unsigned char* src, unsigned long long* out, const char* name) {
    hipFuncSetAttribute((const void*)k<V>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    unsigned long long h[256 * 16];
    for (int issuers : {1, 4, 8, 16})
        for (int groups : {1, 2}) {
            k<V><<<256, 1024, 65536>>>(src, out, issuers, groups, 200);
            hipDeviceSynchronize();
            hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
            double iss = 0; int ni = 0;
            for (int g = 0; g < 256; ++g) for (int w = 0; w < issuers; ++w) { iss += h[g * 16 + w]; ++ni; }
            printf("%-34s %2d waves x %d DMA: %7.0f cycles per wave (%.0f per DMA instruction)\n", name, issuers, 4 * groups, iss / ni, iss / ni / (4 * groups));
        }
}

int main() {
    unsigned char* src; unsigned long long* out;
    hipMalloc(&src, 256 * 65536 + 8192); hipMemset(src, 1, 256 * 65536 + 8192);
    hipMalloc(&out, 256 * 16 * 8);
    run<0>(src, out, "V0 save/set/nop/DMA/restore");
    run<1>(src, out, "V1 set/nop/DMA");
    run<2>(src, out, "V2 set once, 4 x DMA offset:");
    run<3>(src, out, "V3 V2 + save/restore");
    run<6>(src, out, "V6 8 MFMA only");
    run<4>(src, out, "V4 8 MFMA then V2");
    run<5>(src, out, "V5 V2 then 8 MFMA");
    return 0;
}
