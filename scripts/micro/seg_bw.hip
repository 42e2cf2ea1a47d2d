// What does the length of the contiguous piece per matrix row cost on the L2 -> LDS path?  (axis-0 weight-gradient slabs:
// rows = features, pitch = N * 2 B; a chunk of 32 / 64 / 128 minibatch columns is 64 / 128 / 256 B per row.)
// Every workgroup (16 waves) streams a [ROWS][pitch] slab along the row: per step every wave issues 2 LDS-DMA instructions
// (1 KiB each = 1024 / SEG rows x SEG bytes), the workgroup 32 KiB; ring of 4 steps in LDS, counted vmcnt.
//   MODE 0: step s fetches bytes [s*SEG, (s+1)*SEG) of each of the workgroup's 32768 / SEG rows
//   MODE 1 (SEG = 64 only): the two instructions of a wave fetch the two 64 B halves of the SAME 128 B lines (8 rows each,
//           16 rows... per wave), i.e. a step covers 128 B of half as many rows -- same bytes, the halves requested back to back
// 8 row sets of a 4096-row matrix, 4 workgroups per XCD walk the same set (as the windows of one block row do).
// hipcc --offload-arch=gfx950 -O3 seg_bw.hip -o /tmp/seg_bw && /tmp/seg_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ __forceinline__ void glds16(const void* g, uint32_t lds) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(g), "s"(lds) : "memory", "m0");
}

template <int SEG, int MODE>
__global__ void __launch_bounds__(1024) k(const unsigned char* base, int rows_total, size_t pitch, int steps, float* out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
    constexpr int LPR = SEG / 16;            // lanes per row
    constexpr int RPI = 64 / LPR;            // rows per instruction
    const int set_rows = 32768 / SEG * (MODE == 1 ? 1 : 1);
    const int nsets = rows_total / (MODE == 1 ? 16 * 2 * 8 : set_rows);
    const int set = (blockIdx.x >> 3) % nsets;
    size_t off0, off1;
    if (MODE == 0) {
        const int r0 = set * set_rows + (2 * wave) * RPI + lane / LPR, r1 = r0 + RPI;
        off0 = (size_t)r0 * pitch + (lane % LPR) * 16;
        off1 = (size_t)r1 * pitch + (lane % LPR) * 16;
    } else {                                  // 16 rows per wave, both instructions on the same rows: lanes 4 per row, halves 0 / 1
        const int r = set * 256 + wave * 16 + lane / 4;
        off0 = (size_t)r * pitch + (lane % 4) * 16;
        off1 = off0 + 64;
    }
    const int step_bytes = MODE == 1 ? 128 : SEG;
    const int wrap = (int)(pitch / step_bytes);
    for (int s = 0; s < steps; ++s) {
        const size_t col = (size_t)((s + (blockIdx.x & 7) * 16) % wrap) * step_bytes;   // XCDs a little apart, WGs of an XCD in step
        const uint32_t dst = __builtin_amdgcn_readfirstlane(lds0 + (s & 3) * 32768 + wave * 2048);
        glds16(base + off0 + col, dst);
        glds16(base + off1 + col, dst + 1024);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (steps < 0) out[0] = smem[lane];
}

template <int SEG, int MODE>
void run(const unsigned char* d, int rows, size_t pitch, float* o, const char* name) {
    const int grid = 256, steps = 4000;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k<SEG, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<SEG, MODE><<<grid, 1024, 131072>>>(d, rows, pitch, 200, o);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<SEG, MODE><<<grid, 1024, 131072>>>(d, rows, pitch, steps, o);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)grid * steps * 32768;
    printf("%-28s %.3f ms  %.2f TB/s  = %.1f B/clk/CU at 2.1 GHz\n", name, ms, bytes / (ms * 1e-3) / 1e12, bytes / 256 / (ms * 1e-3 * 2.1e9));
}

int main() {
    const int rows = 4096; const size_t pitch = 16384;
    unsigned char* d; float* o;
    hipMalloc(&d, (size_t)rows * pitch + 65536); hipMemset(d, 1, (size_t)rows * pitch + 65536); hipMalloc(&o, 4);
    run<1024, 0>(d, rows, pitch, o, "1024 B per row");
    run<512, 0>(d, rows, pitch, o, "512 B per row");
    run<256, 0>(d, rows, pitch, o, "256 B per row");
    run<128, 0>(d, rows, pitch, o, "128 B per row");
    run<64, 0>(d, rows, pitch, o, "64 B per row");
    run<64, 1>(d, rows, pitch, o, "64 B x 2 halves back to back");
    run<32, 0>(d, rows, pitch, o, "32 B per row");
    return 0;
}
