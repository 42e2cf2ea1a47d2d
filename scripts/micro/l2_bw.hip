// L2 -> CU delivery rate on this box: every workgroup streams a small (L2-resident) region over and over, either with
// LDS-DMA (global_load_lds_dwordx4) or with ordinary global_load_dwordx4 into registers.
// hipcc --offload-arch=gfx950 -O3 l2_bw.hip -o /tmp/l2_bw && /tmp/l2_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void glds16(const void* g, uint32_t lds) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(g), "s"(lds) : "memory", "m0");
}

// region_bytes per workgroup-set: WGs with the same (blockIdx.x / share) read the same region
template <int MODE>
__global__ void __launch_bounds__(512) k(const unsigned char* base, size_t region_bytes, int share, int iters, float* out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned char* reg = base + (size_t)(blockIdx.x / share) * region_bytes;
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
    const int nchunk = (int)(region_bytes / 16384);          // 16 KiB per workgroup step
    u32x4 acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        const unsigned char* src = reg + (size_t)((it + blockIdx.x) % nchunk) * 16384 + wave * 2048;
        if (MODE == 0) {
            glds16(src + lane * 16, __builtin_amdgcn_readfirstlane(lds0 + (it & 3) * 16384 + wave * 2048));
            glds16(src + 1024 + lane * 16, __builtin_amdgcn_readfirstlane(lds0 + (it & 3) * 16384 + wave * 2048 + 1024));
            if ((it & 3) == 3) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        } else {
            const u32x4 a = *reinterpret_cast<const u32x4*>(src + lane * 16);
            const u32x4 b = *reinterpret_cast<const u32x4*>(src + 1024 + lane * 16);
            acc ^= a; acc ^= b;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc.x == 0x12345678u) out[0] = 1.f;
}
template <int MODE>
void run(size_t region, int share, int wg_per_cu, const char* name) {
    const int grid = 256 * wg_per_cu, iters = 4000;
    unsigned char* d; float* o;
    const size_t total = region * ((grid + share - 1) / share);
    hipMalloc(&d, total); hipMemset(d, 1, total); hipMalloc(&o, 4);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<grid, 512, 65536>>>(d, region, share, 100, o);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<grid, 512, 65536>>>(d, region, share, iters, o);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-10s region %6zu KiB shared by %3d WGs, %d WG/CU: %.3f ms  %.2f TB/s\n", name, region / 1024, share, wg_per_cu, ms,
           (double)grid * iters * 16384 / (ms * 1e-3) / 1e12);
    hipFree(d); hipFree(o);
}
int main() {
    for (int wpc = 1; wpc <= 2; ++wpc) {
        run<0>(256 << 10, 1, wpc, "lds-dma");      // private 256 KiB per WG: total 64-128 MB -> MALL / HBM
        run<0>(256 << 10, 32, wpc, "lds-dma");     // 32 WGs share 256 KiB: total 2-4 MB -> L2
        run<0>(2 << 20, 256, wpc, "lds-dma");      // everyone shares 2 MiB
        run<1>(256 << 10, 1, wpc, "vgpr-load");
        run<1>(256 << 10, 32, wpc, "vgpr-load");
        run<1>(2 << 20, 256, wpc, "vgpr-load");
    }
    return 0;
}
