// does raw_buffer_store_b128 / load_b128 (sc1) through a make_buffer_rsrc descriptor move 16 bytes per lane at voffset?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int i4 __attribute__((ext_vector_type(4)));
template <int FLAGS>
__global__ void k(int* buf, int* out) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(buf, 0, 0x7ffffff0, FLAGS);
    const int lane = threadIdx.x;
    for (int q = 0; q < 4; ++q) {
        i4 v = {1000 * q + lane * 4 + 0, 1000 * q + lane * 4 + 1, 1000 * q + lane * 4 + 2, 1000 * q + lane * 4 + 3};
        __builtin_amdgcn_raw_buffer_store_b128(v, r, (unsigned)((q * 64 + lane) * 16), 0, 16);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int q = 0; q < 4; ++q) {
        i4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (unsigned)((q * 64 + lane) * 16), 0, 16);
        out[(q * 64 + lane) * 4 + 0] = v.x; out[(q * 64 + lane) * 4 + 1] = v.y; out[(q * 64 + lane) * 4 + 2] = v.z; out[(q * 64 + lane) * 4 + 3] = v.w;
    }
}
template <int FLAGS> void run(const char* name) {
    int *buf, *out; hipMalloc(&buf, 1 << 16); hipMalloc(&out, 1 << 16); hipMemset(buf, 0xff, 1 << 16);
    k<FLAGS><<<1, 64>>>(buf, out); hipDeviceSynchronize();
    static int h[1024], g[1024]; hipMemcpy(h, out, 4096, hipMemcpyDeviceToHost); hipMemcpy(g, buf, 4096, hipMemcpyDeviceToHost);
    int bad_l = 0, bad_m = 0;
    for (int q = 0; q < 4; ++q) for (int l = 0; l < 64; ++l) for (int e = 0; e < 4; ++e) {
        const int want = 1000 * q + l * 4 + e;
        bad_l += h[(q * 64 + l) * 4 + e] != want; bad_m += g[(q * 64 + l) * 4 + e] != want;
    }
    printf("%s: loaded-back mismatches %d, memory-image mismatches %d  (mem[0..7] = %d %d %d %d %d %d %d %d)\n", name, bad_l, bad_m, g[0], g[1], g[2], g[3], g[4], g[5], g[6], g[7]);
}
int main() { run<0x00020000>("flags 0x00020000"); run<0x00027000>("flags 0x00027000"); return 0; }
