// bsmm_updat_tr.h -- weight-gradient kernel for feature_axis = 1 (activations (N, C): the contraction index n is the
// STRIDED one for both operands), bsize 32, 16-bit storage types.
//
//   DW[w][ci][ko] = alpha * sum_p sum_n X_p[n][c*32+ci] * DY_p[n][k*32+ko] + beta * DW[w][ci][ko],  (c,k) = updat_lut[w]
//
// The first version gathered 16 two-byte elements per fragment at an 8 KiB stride (2.2 ms on 4096^2 / 20% / N=8192).
// Here each wave streams 32-row slabs of its two operands into a private LDS ring with LDS-DMA (coalesced 64-byte row
// pieces, 4 instructions per slab pair, nothing in VGPRs) and builds the MFMA fragments with the gfx950 transposing read
// ds_read_b64_tr_b16: within a 16-lane group, lane t receives column t of the 4 x 16 (b16) matrix the group's lanes
// point at (probed on hardware: profiles/r01_tr_probe.log) -- i.e. four consecutive n for one feature, exactly the
// K-major fragment the matrix core wants.  A 64-byte row stride is bank-conflict free for this access, so the slab image
// is the plain DMA image.
//   MFMA roles: A[ci][n] (X slab), B[n][ko] (DY slab)  ->  D[ci][ko]; K = n, 16 per instruction.
//   256 threads = 4 waves per weight block; wave v takes 32-row chunks v, v+4, ...; partial tiles are summed through LDS.
#pragma once
#include "bsmm_common.h"
#include "bsmm_updat.h"

namespace bsmm {

typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr int UT_D = 4;              // ring depth (chunks)
constexpr int UT_SLOT = 2 * 2048;    // bytes per ring slot: X slab (32 rows x 64 B) + DY slab
constexpr int UT_LDS = 4 * UT_D * UT_SLOT;

__device__ __forceinline__ uint2 ds_tr16(const unsigned char* p) {
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
    return __builtin_bit_cast(uint2, v);
}

template <class DT>
__global__ void __launch_bounds__(256, 2)
updat32_a1_tr_kernel(PtrList8 Xs, PtrList8 Es, typename DT::T* __restrict__ DW, const int32_t* __restrict__ lut,
                     int blocks, int N, int Cf, int Kf, int pcount, float alpha, float beta) {
    typedef typename DT::T T;
    static_assert(DT::is16, "transposing-read kernel: 16-bit storage types");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // [4 waves][UT_D][X slab | DY slab]; reused for the reduction
    const int w = updat_block(blockIdx.x, blocks);
    if (w < 0) return;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lut[2 * w], k = lut[2 * w + 1];

    unsigned char* ring = smem + wave * (UT_D * UT_SLOT);
    const uint32_t ring_addr = lds_addr_of(ring);
    // DMA: instruction i (0,1) of a slab covers rows 16i .. 16i+15: lane -> (row 16i + (lane >> 2), 16-byte piece lane & 3)
    const int drow = lane >> 2, dpiece = lane & 3;
    // fragment reads: 16-lane group g16 = lane >> 4 -> features 16*(g16&1) .. +15, K half h = g16 >> 1;
    // lane t of the group points at row (t >> 2) of the 4-row band, 8 bytes at feature 16*(g16&1) + 4*(t & 3)
    const int g16 = lane >> 4, t16 = lane & 15;
    const int h = g16 >> 1;
    const int rd_base = (t16 >> 2) * 64 + (16 * (g16 & 1) + 4 * (t16 & 3)) * 2;

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;

    const int nchunks = (N + 31) >> 5;
    for (int p = 0; p < pcount; ++p) {
        const T* X = static_cast<const T*>(Xs.p[p]) + c * 32 + dpiece * 8;
        const T* E = static_cast<const T*>(Es.p[p]) + k * 32 + dpiece * 8;
        // chunk index q (0,1,2,...) of THIS wave covers rows (wave + 4q)*32 ..; rows past N are clamped (and masked below)
        auto issue = [&](int q, int pos) {
            const int n0 = (wave + 4 * q) * 32;
            const uint32_t slot = __builtin_amdgcn_readfirstlane(ring_addr + pos * UT_SLOT);
            const int r0 = min(n0 + drow, N - 1), r1 = min(n0 + 16 + drow, N - 1);
            glds16_asm(X + (size_t)r0 * Cf, slot);
            glds16_asm(X + (size_t)r1 * Cf, slot + 1024);
            glds16_asm(E + (size_t)r0 * Kf, slot + 2048);
            glds16_asm(E + (size_t)r1 * Kf, slot + 3072);
        };
        const int myq = (nchunks > wave) ? (nchunks - wave + 3) / 4 : 0;   // chunks this wave owns
#pragma unroll
        for (int d = 0; d < UT_D - 1; ++d) issue(d, d);   // (chunks past the end are clamped re-reads: harmless)
        int rd_pos = 0, wr_pos = UT_D - 1;
        for (int q = 0; q < myq; ++q) {
            issue(q + UT_D - 1, wr_pos);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (UT_D - 1)) : "memory");   // chunk q has landed
            const unsigned char* slot = ring + rd_pos * UT_SLOT;
            rd_pos = (rd_pos + 1 == UT_D) ? 0 : rd_pos + 1;
            wr_pos = (wr_pos + 1 == UT_D) ? 0 : wr_pos + 1;
            const int n0 = (wave + 4 * q) * 32;
            uint4 a[2], b[2];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {     // K sub-block kk: rows 16kk + 8h + {0..3 | 4..7}
                const unsigned char* sp = slot + (16 * kk + 8 * h) * 64 + rd_base;
                const uint2 a0 = ds_tr16(sp), a1 = ds_tr16(sp + 4 * 64);
                const uint2 b0 = ds_tr16(sp + 2048), b1 = ds_tr16(sp + 2048 + 4 * 64);
                a[kk] = make_uint4(a0.x, a0.y, a1.x, a1.y);
                b[kk] = make_uint4(b0.x, b0.y, b1.x, b1.y);
            }
            if (n0 + 32 > N) {   // ragged tail: rows >= N were clamped re-reads -> zero their contribution (A side suffices)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const int nb = n0 + 16 * kk + 8 * h;   // K index j of this lane's fragment is row nb + j
                    uint32_t* u = reinterpret_cast<uint32_t*>(&a[kk]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const uint32_t lo = (nb + 2 * j < N) ? 0xffffu : 0u, hi = (nb + 2 * j + 1 < N) ? 0xffff0000u : 0u;
                        u[j] &= (lo | hi);
                    }
                }
            }
            acc = DT::mfma32(a[0], b[0], acc);
            acc = DT::mfma32(a[1], b[1], acc);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the clamped tail prefetches before the ring is reused
    }

    __syncthreads();   // all waves done with their rings: reuse the memory for the cross-wave reduction
    float* red = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) red[wave * 1024 + reg * 64 + lane] = acc[reg];
    __syncthreads();
    for (int slot = threadIdx.x; slot < 1024; slot += 256) {
        const float sum = red[slot] + red[1024 + slot] + red[2048 + slot] + red[3072 + slot];
        const int reg = slot >> 6, ln = slot & 63;
        const int ci = (reg & 3) + 8 * (reg >> 2) + 4 * (ln >> 5), ko = ln & 31;
        const size_t idx = (size_t)w * 1024 + ci * 32 + ko;
        float out = alpha * sum;
        if (beta != 0.f) out += beta * DT::to_f32(DW[idx]);
        DW[idx] = DT::from_f32(out);
    }
}

// Small minibatches (round 4): ONE WAVE per weight block.  With a few dozen to a few hundred rows the kernel above spends its time on
// everything but the products: four waves cut two chunks between them, meet in LDS behind two barriers, and at 64 KiB of LDS per block
// only 512 blocks are resident at once (6.4 rounds of ~2 us at the bench layout: 16 us at N = 64).  Here a workgroup is four independent
// waves = four consecutive blocks of the z-ordered lookup table (neighbours share X rows / DY rows in the L2), each streaming its
// operands through a private ring of UTS_D slots (32 rows x (64 B + 64 B) per slot: 8 KiB per wave, five workgroups per CU, every block
// of the bench layout resident in one round), counted `vmcnt`, no barrier, no reduction; the wave writes its block itself.
// (Tried: one-wave workgroups with three ring slots, two chunks in flight per wave -- not faster, 6.1 / 11.0 / 18.5 against 5.4 / 9.4 / 18.2 us at
// N = 64 / 256 / 512: from ~400 rows on the kernel moves its 128 B per (block, row) at the ~12 TB/s the L2 -> LDS path delivers chip-wide.)
constexpr int UTS_D = 2;
constexpr int UTS_LDS = 4 * UTS_D * UT_SLOT;      // 32 KiB

template <class DT>
__global__ void __launch_bounds__(256, 4)
updat32_a1_small_kernel(PtrList8 Xs, PtrList8 Es, typename DT::T* __restrict__ DW, const int32_t* __restrict__ lut,
                        int blocks, int N, int Cf, int Kf, int pcount, float alpha, float beta, const float* __restrict__ gate, int q64 = 0) {
    // (q64, round 6: the blocks are the quadrants of 64 x 64 blocks, 4 w64 + 2 (row half) + (column half), bsmm_api.hip::updat64 -- every element goes
    //  to its place in the 64 x 64 block and the gate is the 64-block's, as in updat2_reduce_kernel: short minibatches at bsize 64 ran the streaming
    //  kernel before, 35 us at N = 64 where this one takes 6)
    typedef typename DT::T T;
    static_assert(DT::is16, "transposing-read kernel: 16-bit storage types");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int groups = (blocks + 3) >> 2;                                  // workgroups with work; XCD x walks a contiguous range of them
    const int g = updat_block(blockIdx.x, groups);
    if (g < 0) return;
    const int w = 4 * g + wave;
    if (w >= blocks) return;                                               // (no barrier below: a wave may leave alone)
    const int c = lut[2 * w], k = lut[2 * w + 1];
    unsigned char* ring = smem + wave * (UTS_D * UT_SLOT);
    const uint32_t ring_addr = lds_addr_of(ring);
    const int drow = lane >> 2, dpiece = lane & 3;
    const int g16 = lane >> 4, t16 = lane & 15;
    const int h = g16 >> 1;
    const int rd_base = (t16 >> 2) * 64 + (16 * (g16 & 1) + 4 * (t16 & 3)) * 2;

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const int nchunks = (N + 31) >> 5;
    for (int p = 0; p < pcount; ++p) {
        const T* X = static_cast<const T*>(Xs.p[p]) + c * 32 + dpiece * 8;
        const T* E = static_cast<const T*>(Es.p[p]) + k * 32 + dpiece * 8;
        auto issue = [&](int q, int pos) {                                 // chunk q = rows 32 q ..; rows past N are clamped re-reads (masked below)
            const int n0 = q * 32;
            const uint32_t slot = __builtin_amdgcn_readfirstlane(ring_addr + pos * UT_SLOT);
            const int r0 = min(n0 + drow, N - 1), r1 = min(n0 + 16 + drow, N - 1);
            glds16_asm(X + (size_t)r0 * Cf, slot);
            glds16_asm(X + (size_t)r1 * Cf, slot + 1024);
            glds16_asm(E + (size_t)r0 * Kf, slot + 2048);
            glds16_asm(E + (size_t)r1 * Kf, slot + 3072);
        };
#pragma unroll
        for (int d = 0; d < UTS_D - 1; ++d) issue(d, d);
        int rd_pos = 0, wr_pos = UTS_D - 1;
        for (int q = 0; q < nchunks; ++q) {
            issue(q + UTS_D - 1, wr_pos);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (UTS_D - 1)) : "memory");   // chunk q has landed
            const unsigned char* slot = ring + rd_pos * UT_SLOT;
            rd_pos = (rd_pos + 1 == UTS_D) ? 0 : rd_pos + 1;
            wr_pos = (wr_pos + 1 == UTS_D) ? 0 : wr_pos + 1;
            const int n0 = q * 32;
            uint4 a[2], b[2];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const unsigned char* sp = slot + (16 * kk + 8 * h) * 64 + rd_base;
                const uint2 a0 = ds_tr16(sp), a1 = ds_tr16(sp + 4 * 64);
                const uint2 b0 = ds_tr16(sp + 2048), b1 = ds_tr16(sp + 2048 + 4 * 64);
                a[kk] = make_uint4(a0.x, a0.y, a1.x, a1.y);
                b[kk] = make_uint4(b0.x, b0.y, b1.x, b1.y);
            }
            if (n0 + 32 > N) {   // ragged tail: rows >= N were clamped re-reads -> zero their contribution (A side suffices)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const int nb = n0 + 16 * kk + 8 * h;
                    uint32_t* u = reinterpret_cast<uint32_t*>(&a[kk]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const uint32_t lo = (nb + 2 * j < N) ? 0xffffu : 0u, hi = (nb + 2 * j + 1 < N) ? 0xffff0000u : 0u;
                        u[j] &= (lo | hi);
                    }
                }
            }
            acc = DT::mfma32(a[0], b[0], acc);
            acc = DT::mfma32(a[1], b[1], acc);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the clamped tail prefetch, before the next pair re-primes the ring
    }
    // D[ci][ko]: col = ko = lane & 31, row ci = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    const float a_eff = gate ? alpha * gate[q64 ? w >> 2 : w] : alpha;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int ci = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
        const size_t idx = q64 ? (size_t)(w >> 2) * 4096 + (size_t)(32 * ((w >> 1) & 1) + ci) * 64 + 32 * (w & 1) + (lane & 31)
                               : (size_t)w * 1024 + ci * 32 + (lane & 31);
        float out = a_eff * acc[reg];
        if (beta != 0.f) out += beta * DT::to_f32(DW[idx]);
        DW[idx] = DT::from_f32(out);
    }
}

// bsize 16 version: 16x16 blocks, v_mfma_f32_16x16x32 (K = 32 minibatch rows per instruction).  Slabs are 32 rows x
// 32 B (one 1 KiB DMA instruction each); lane (ci = lane & 15, q = lane >> 4) needs rows 8q .. 8q+7 of feature ci: two
// transposing reads (rows 8q+{0..3}, 8q+{4..7}); 16-lane group q covers exactly the 16 features.
constexpr int UT16_SLOT = 2 * 1024;
constexpr int UT16_D = 8;
constexpr int UT16_LDS = 4 * UT16_D * UT16_SLOT;   // 64 KiB (also holds the 4 KiB reduction buffer)

template <class DT>
__global__ void __launch_bounds__(256, 2)
updat16_a1_tr_kernel(PtrList8 Xs, PtrList8 Es, typename DT::T* __restrict__ DW, const int32_t* __restrict__ lut,
                     int blocks, int N, int Cf, int Kf, int pcount, float alpha, float beta) {
    typedef typename DT::T T;
    static_assert(DT::is16, "transposing-read kernel: 16-bit storage types");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int w = updat_block(blockIdx.x, blocks);
    if (w < 0) return;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lut[2 * w], k = lut[2 * w + 1];
    unsigned char* ring = smem + wave * (UT16_D * UT16_SLOT);
    const uint32_t ring_addr = lds_addr_of(ring);
    const int drow = lane >> 1, dpiece = lane & 1;                 // DMA: lane -> (row, 16-byte piece)
    const int q = lane >> 4, t16 = lane & 15;
    const int rd_base = (8 * q + (t16 >> 2)) * 32 + (4 * (t16 & 3)) * 2;

    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nchunks = (N + 31) >> 5;
    for (int p = 0; p < pcount; ++p) {
        const T* X = static_cast<const T*>(Xs.p[p]) + c * 16 + dpiece * 8;
        const T* E = static_cast<const T*>(Es.p[p]) + k * 16 + dpiece * 8;
        auto issue = [&](int qq, int pos) {
            const int row = min((wave + 4 * qq) * 32 + drow, N - 1);     // rows past N are clamped (masked below)
            const uint32_t slot = __builtin_amdgcn_readfirstlane(ring_addr + pos * UT16_SLOT);
            glds16_asm(X + (size_t)row * Cf, slot);
            glds16_asm(E + (size_t)row * Kf, slot + 1024);
        };
        const int myq = (nchunks > wave) ? (nchunks - wave + 3) / 4 : 0;
#pragma unroll
        for (int d = 0; d < UT16_D - 1; ++d) issue(d, d);
        int rd_pos = 0, wr_pos = UT16_D - 1;
        for (int qq = 0; qq < myq; ++qq) {
            issue(qq + UT16_D - 1, wr_pos);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (UT16_D - 1)) : "memory");
            const unsigned char* slot = ring + rd_pos * UT16_SLOT;
            rd_pos = (rd_pos + 1) & (UT16_D - 1);
            wr_pos = (wr_pos + 1) & (UT16_D - 1);
            const int n0 = (wave + 4 * qq) * 32;
            const uint2 a0 = ds_tr16(slot + rd_base), a1 = ds_tr16(slot + rd_base + 4 * 32);
            const uint2 b0 = ds_tr16(slot + 1024 + rd_base), b1 = ds_tr16(slot + 1024 + rd_base + 4 * 32);
            uint4 a = make_uint4(a0.x, a0.y, a1.x, a1.y);
            const uint4 b = make_uint4(b0.x, b0.y, b1.x, b1.y);
            if (n0 + 32 > N) {
                const int nb = n0 + 8 * q;
                uint32_t* u = reinterpret_cast<uint32_t*>(&a);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const uint32_t lo = (nb + 2 * e < N) ? 0xffffu : 0u, hi = (nb + 2 * e + 1 < N) ? 0xffff0000u : 0u;
                    u[e] &= (lo | hi);
                }
            }
            acc = DT::mfma16(a, b, acc);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) red[wave * 256 + reg * 64 + lane] = acc[reg];
    __syncthreads();
    {
        const int slot = threadIdx.x;
        const float sum = red[slot] + red[256 + slot] + red[512 + slot] + red[768 + slot];
        const int reg = slot >> 6, ln = slot & 63;
        const int ci = 4 * (ln >> 4) + reg, ko = ln & 15;
        const size_t idx = (size_t)w * 256 + ci * 16 + ko;
        float out = alpha * sum;
        if (beta != 0.f) out += beta * DT::to_f32(DW[idx]);
        DW[idx] = DT::from_f32(out);
    }
}

}  // namespace bsmm
