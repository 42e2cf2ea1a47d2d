// probe: semantics of ds_read_b64_tr_b16 on gfx950.  LDS holds u16 value = its own element index; every lane
// issues the read at byte address base[lane]; we dump what each lane receives.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void probe(const int* addr, unsigned short* out) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int a = addr[threadIdx.x];   // BYTE address
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)((__attribute__((address_space(3))) char*)lds + a));
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (unsigned short)v[j];
}
int main() {
    int h_addr[64]; unsigned short h_out[256];
    int *d_addr; unsigned short* d_out;
    hipMalloc(&d_addr, sizeof(h_addr)); hipMalloc(&d_out, sizeof(h_out));
    // experiment A: every lane points at lane*8 bytes (natural packing: 16 lanes x 8 B = one 4x16 b16 matrix per group)
    for (int l = 0; l < 64; ++l) h_addr[l] = l * 8;
    hipMemcpy(d_addr, h_addr, sizeof(h_addr), hipMemcpyHostToDevice);
    probe<<<1, 64>>>(d_addr, d_out); hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
    printf("A: addr = lane*8\n");
    for (int l = 0; l < 64; ++l) printf("lane %2d: %5d %5d %5d %5d\n", l, h_out[4*l], h_out[4*l+1], h_out[4*l+2], h_out[4*l+3]);
    // experiment B: rows of 128 B (64 elements): lane t in group g16: row = 4*g16 + t/4, col0 = 4*(t%4)
    for (int l = 0; l < 64; ++l) { int g = l >> 4, t = l & 15; h_addr[l] = ((4 * g + t / 4) * 64 + 4 * (t % 4)) * 2; }
    hipMemcpy(d_addr, h_addr, sizeof(h_addr), hipMemcpyHostToDevice);
    probe<<<1, 64>>>(d_addr, d_out); hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
    printf("B: row stride 64 elements\n");
    for (int l = 0; l < 64; ++l) printf("lane %2d: %5d %5d %5d %5d\n", l, h_out[4*l], h_out[4*l+1], h_out[4*l+2], h_out[4*l+3]);
    return 0;
}
