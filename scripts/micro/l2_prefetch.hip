// Can scalar loads (a path into the L2 that does not go through the L1 vector cache and its ~64 outstanding requests) warm the L2 for the
// LDS-DMA stream of the same workgroup?  Every workgroup streams its private 512 KiB region over and over (256 workgroups: 128 MiB in
// total -- beyond the L2s, inside the Infinity Cache): NDMA waves fetch it by LDS-DMA, 2 KiB per wave and iteration; the other waves are
// idle (mode 0) or touch, AHEAD iterations ahead, one dword of every 128-byte line the DMA waves will fetch (mode 1).
// hipcc --offload-arch=gfx950 -O3 l2_prefetch.hip -o /tmp/l2_prefetch && /tmp/l2_prefetch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ __forceinline__ void glds16(const void* g, uint32_t lds) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(g), "s"(lds) : "memory", "m0");
}

template <int NDMA, int MODE, int AHEAD>
__global__ void __launch_bounds__(1024) k(const unsigned char* base, int iters, float* out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr size_t REGION = 512 << 10;
    constexpr int STEP = NDMA * 2048;                    // bytes per iteration
    constexpr int NSTEP = REGION / STEP;
    const unsigned char* reg = base + (size_t)blockIdx.x * REGION;
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
    unsigned acc = 0;
    if (wave < NDMA) {
        for (int it = 0; it < iters; ++it) {
            const unsigned char* src = reg + (size_t)(it % NSTEP) * STEP + wave * 2048;
            const uint32_t dst = __builtin_amdgcn_readfirstlane(lds0 + (it & 3) * 32768 + wave * 2048);
            glds16(src + lane * 16, dst);
            glds16(src + 1024 + lane * 16, dst + 1024);
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (MODE == 1) {
        // the 16 - NDMA prefetch waves share the lines of an iteration: STEP / 128 lines, one s_load_dword each
        constexpr int NPF = 16 - NDMA, LINES = STEP / 128;
        const int me = wave - NDMA;
        for (int it = 0; it < iters; ++it) {
            const unsigned char* src = reg + (size_t)((it + AHEAD) % NSTEP) * STEP;
            for (int l = me; l < LINES; l += NPF) {
                unsigned v;
                asm volatile("s_load_dword %0, %1, %2" : "=s"(v) : "s"(src), "s"(l * 128) : "memory");
                if ((l & 31) == 31) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
                asm volatile("" :: "s"(v));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    if (iters < 0) out[0] = (float)acc + smem[lane];
}

template <int NDMA, int MODE, int AHEAD>
void run(const unsigned char* d, float* o, const char* name) {
    const int grid = 256, iters = 6000;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k<NDMA, MODE, AHEAD>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NDMA, MODE, AHEAD><<<grid, 1024, 131072>>>(d, 600, o);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NDMA, MODE, AHEAD><<<grid, 1024, 131072>>>(d, iters, o);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)grid * iters * NDMA * 2048;
    printf("%-44s %.3f ms  %.2f TB/s  = %.1f B/clk/CU at 2.1 GHz\n", name, ms, bytes / (ms * 1e-3) / 1e12, bytes / 256 / (ms * 1e-3 * 2.1e9));
}

int main() {
    unsigned char* d; float* o;
    const size_t total = (size_t)256 * (512 << 10) + (1 << 20);
    hipMalloc(&d, total); hipMemset(d, 1, total); hipMalloc(&o, 4);
    run<12, 0, 0>(d, o, "12 DMA waves, 4 idle");
    run<12, 1, 4>(d, o, "12 DMA waves, 4 touch 4 iterations ahead");
    run<12, 1, 8>(d, o, "12 DMA waves, 4 touch 8 iterations ahead");
    run<12, 1, 16>(d, o, "12 DMA waves, 4 touch 16 iterations ahead");
    run<8, 0, 0>(d, o, "8 DMA waves, 8 idle");
    run<8, 1, 8>(d, o, "8 DMA waves, 8 touch 8 iterations ahead");
    return 0;
}
