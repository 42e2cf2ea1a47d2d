// bsmm_updat_win.h -- windowed weight-gradient kernel.  Round 4: only the bsize-16 kernel (updat16_win_kernel, BASELINE configs[2]) is
// left; the bsize-32 kernels this file was written for in round 1 (updat32_a1_win_kernel: 8x8 / 16x16-block windows, 8 or 16 waves;
// updat32_a0_win_kernel) were superseded on both feature axes by the streaming kernel (bsmm_updat_v2.h, rounds 2 - 3: 122 -> 94 us at the
// bench shape) and retired.  The text below describes the scheme the bsize-16 kernel still follows (window, slabs, swizzles, ring).
//
// Why: with one workgroup per weight block (bsmm_updat_tr.h) every block streams its own X and DY column slabs:
// 2 * N * 64 B per block = 3.4 GB at 4096^2 / 20% / N = 8192, and the kernel runs at the fabric bandwidth (5.6 TB/s,
// 608 us).  Here a workgroup owns a UW x UW window of the block grid.  Per 32-row minibatch chunk it DMA-stages ONE
// X slab (32 rows x UW*64 B: 512 contiguous bytes per row -> full-line L2 requests) and ONE DY slab into LDS, and
// every nonzero block of the window -- statically owned by one of the 8 waves, <= UP_MAXB per wave -- builds its MFMA
// fragments from the shared slabs with the transposing read ds_read_b64_tr_b16.  Traffic per block drops from
// 2 slabs to (2*UW)/(UW*UW*density) slabs (~0.6 at 20%).
//
//   LDS slab image: row-major, 512 B per row, the 16-byte pieces of row r XOR-swizzled with 4*(r & 3) so that the four
//   rows a transposing read touches fall on disjoint bank ranges (a 512-byte stride alone maps them to the same banks).
//   Ring of UWN_D slots (slot = X slab + DY slab): the DMA of chunk q+UWN_D-1 is issued right after the barrier of
//   chunk q; each wave issues a constant 2*UWN_NI DMA instructions per chunk (chunks past the end re-fetch the last
//   rows), so a counted `vmcnt` = "my share of chunk q has landed".  One barrier per chunk, 512 threads, 1 WG per CU.
//   Minibatch split: gridDim.y workgroups share an item; with a scratch buffer (always when gridDim.y > 1; also the
//   bsize-8 super-block path) partial tiles are added (fp32 atomics) into the zeroed fp32 scratch and a second kernel
//   applies alpha/beta and rounds once; without one (gridDim.y == 1) the workgroup
//   stores directly.
#pragma once
#include <type_traits>

#include "bsmm_common.h"
#include "bsmm_plan.h"
#include "bsmm_updat_tr.h"

namespace bsmm {

constexpr int UWN_D = 2;
constexpr int UWN_CH = 64;                    // minibatch rows per chunk
constexpr int UWN_ROWB = UW * 64;             // bytes per slab row (UW blocks x 64 B)
constexpr int UWN_SLAB = UWN_CH * UWN_ROWB;   // bytes of one operand slab
constexpr int UWN_NI = UWN_SLAB / 1024 / UP_WAVES;   // DMA instructions per wave per slab
constexpr int UWN_SLOT = 2 * UWN_SLAB;
constexpr int UWN_LDS = UWN_D * UWN_SLOT;

// ------------------------------------------------------------------------------------------------------------------
// feature_axis = 0 variant: activations are (C, N) / (K, N), the contraction index n is CONTIGUOUS for both operands, so
// the slabs are [UW*32 feature rows][64 n] (128 B per row, 8-piece XOR swizzle as in bsmm_xcol.h) and the fragments are
// plain ds_read_b128 -- no transposing reads.  Requires N % 8 == 0 (16-byte aligned row pieces); the launcher falls back to the
// per-block kernel otherwise.  (Geometry constants of the bsize-16 kernel's axis-0 form.)
// ------------------------------------------------------------------------------------------------------------------
constexpr int UW0_ROWS = UW * 32;
constexpr int UW0_SLAB = UW0_ROWS * 128;                 // 32 KiB
constexpr int UW0_SLOT = 2 * UW0_SLAB;
constexpr int UW0_LDS = 2 * UW0_SLOT;                    // ring depth 2
constexpr int UW0_NI = UW0_SLAB / 1024 / UP_WAVES;       // DMA instructions per wave per slab (8 rows each)

// ------------------------------------------------------------------------------------------------------------------
// bsize 16: a 16x16-block window is the same 256x256 features, so slabs, DMA and swizzles are those of the bsize-32
// kernels above; only the per-block part differs: v_mfma_f32_16x16x32 (K = 32 minibatch rows / columns per
// instruction, 2 per 64-wide chunk), 4 accumulator registers per block, up to UP16_MAXB = 8 blocks per wave.
//   axis 1: lane (f = lane & 15, q = lane >> 4) builds K = 8q..8q+7 with two transposing reads per operand and K-step;
//   axis 0: plain ds_read_b128 (piece 4*ks + q of the feature row).
// ------------------------------------------------------------------------------------------------------------------
template <class DT, int AXIS>
__global__ void __launch_bounds__(512, 2)
updat16_win_kernel(PtrList8 Xs, PtrList8 Es, typename DT::T* __restrict__ DW, float* __restrict__ scratch,
                   const int32_t* __restrict__ plan, int N, int Cf, int Kf, int pcount, float alpha, float beta) {
    typedef typename DT::T T;
    static_assert(DT::is16, "windowed updat: 16-bit storage types");
    static_assert(UWN_SLOT == UW0_SLOT && UWN_CH == 64, "both axis variants use 64-wide chunks and 64 KiB slots");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if (plan[0] != UPLAN_MAGIC || plan[1] != UPLAN_VERSION || plan[2] != UW16 || plan[3] != UP16_MAXB || plan[7] != UP_WAVES) return;
    const int32_t* item = plan + plan[6] + (size_t)blockIdx.x * UP16_ITEM;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c0 = item[0], k0 = item[1];   // window origin in 16-feature blocks
    if (item[2] == 0) return;
    const int nslots = item[3] & 0xffff;
    const uint32_t cmask = (uint32_t)item[2] >> 16, kmask = (uint32_t)item[3] >> 16;   // 16-feature blocks of the window in use
    int meta[UP16_MAXB], wid[UP16_MAXB];
#pragma unroll
    for (int j = 0; j < UP16_MAXB; ++j) {
        meta[j] = item[4 + (wave * UP16_MAXB + j) * 2];
        wid[j] = item[4 + (wave * UP16_MAXB + j) * 2 + 1];
    }
    const int nchunks = (N + 63) >> 6;
    const int per = (nchunks + gridDim.y - 1) / gridDim.y;
    const int q_beg = blockIdx.y * per, q_end = min(nchunks, q_beg + per);
    const uint32_t base_addr = lds_addr_of(smem);
    const int f = lane & 15, q = lane >> 4;

    // ---- DMA addressing (identical to the bsize-32 kernels, window origin in units of 16 features) ----
    constexpr int NI = (AXIS == 1) ? UWN_NI : UW0_NI;
    int xcol[NI], ecol[NI];           // axis 1: source element column; axis 0: piece offset (elements) inside the row
    size_t xrow[NI], erow[NI];        // axis 0: element offset of the source row
    uint64_t xon[NI], eon[NI];        // lanes of the instruction whose part belongs to a block the item uses (others are not staged)
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        if constexpr (AXIS == 1) {
            constexpr int PPR = UWN_ROWB / 16, RPI = 1024 / UWN_ROWB;
            const int row = RPI * (NI * wave + i) + lane / PPR;
            const int piece = (lane % PPR) ^ (4 * (row & 3));
            xcol[i] = min(c0 * 16 + piece * 8, Cf - 8);
            ecol[i] = min(k0 * 16 + piece * 8, Kf - 8);
            xrow[i] = erow[i] = 0;
            xon[i] = eon[i] = ~0ull;      // (partial 512-byte rows save no cache lines: everything is staged, see the bsize-32 kernel)
        } else {
            const int row = 8 * (NI * wave + i) + (lane >> 3);
            xcol[i] = ecol[i] = ((lane & 7) ^ ((row >> 1) & 7)) * 8;
            xrow[i] = (size_t)min(c0 * 16 + row, Cf - 1) * N;
            erow[i] = (size_t)min(k0 * 16 + row, Kf - 1) * N;
            xon[i] = __ballot((cmask >> (row >> 4)) & 1);
            eon[i] = __ballot((kmask >> (row >> 4)) & 1);
        }
    }
    // ---- fragment addressing ----
    int aoff[UP16_MAXB], boff[UP16_MAXB];
    const int t16 = lane & 15, trow = t16 >> 2;
#pragma unroll
    for (int j = 0; j < UP16_MAXB; ++j) {
        const int cidx = meta[j] & 15, kidx = (meta[j] >> 4) & 15;
        if constexpr (AXIS == 1) {   // byte offset of this lane's 8 bytes inside a 4-row band (row = trow)
            aoff[j] = trow * UWN_ROWB + (((cidx * 2 + ((t16 & 3) >> 1)) ^ (4 * trow)) << 4) + 8 * (t16 & 1);
            boff[j] = UWN_SLAB + trow * UWN_ROWB + (((kidx * 2 + ((t16 & 3) >> 1)) ^ (4 * trow)) << 4) + 8 * (t16 & 1);
        } else {                     // byte offset of this lane's feature row
            aoff[j] = (cidx * 16 + f) * 128;
            boff[j] = UW0_SLAB + (kidx * 16 + f) * 128;
        }
    }
    const int fsw = (f >> 1) & 7;    // axis 0: piece swizzle of row (cidx*16 + f): ((row >> 1) & 7) == (f >> 1) & 7 (16 | cidx*16)

    f32x4 acc[UP16_MAXB];
#pragma unroll
    for (int j = 0; j < UP16_MAXB; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto run = [&](auto ns_tag) {
        constexpr int NS = decltype(ns_tag)::value;
        for (int p = 0; p < pcount; ++p) {
            const T* X = static_cast<const T*>(Xs.p[p]);
            const T* E = static_cast<const T*>(Es.p[p]);
            auto issue = [&](int qq, int pos) {
                const int n0 = qq * 64;
                const uint32_t slot = base_addr + pos * UWN_SLOT;
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    const uint32_t dst = __builtin_amdgcn_readfirstlane(slot + (NI * wave + i) * 1024);
                    if constexpr (AXIS == 1) {
                        constexpr int PPR = UWN_ROWB / 16, RPI = 1024 / UWN_ROWB;
                        const int row = min(n0 + RPI * (NI * wave + i) + lane / PPR, N - 1);
                        glds16_asm(X + (size_t)row * Cf + xcol[i], dst);
                        glds16_asm(E + (size_t)row * Kf + ecol[i], dst + UWN_SLAB);
                    } else {
                        const int col = min(n0 + xcol[i], N - 8);
                        glds16_asm_masked(X + xrow[i] + col, dst, xon[i]);
                        glds16_asm_masked(E + erow[i] + col, dst + UW0_SLAB, eon[i]);
                    }
                }
            };
            if (q_beg >= q_end) break;
            issue(q_beg, 0);
            int pos = 0;
            for (int qq = q_beg; qq < q_end; ++qq) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (qq + 1 < q_end) issue(qq + 1, pos ^ 1);
                const unsigned char* slot = smem + pos * UWN_SLOT;
                pos ^= 1;
                const int n0 = qq * 64;
                const bool tail = n0 + 64 > N;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    uint4 a[NS], b[NS];
#pragma unroll
                    for (int j = 0; j < NS; ++j) {
                        if constexpr (AXIS == 1) {
                            const unsigned char* sa = slot + (32 * ks + 8 * q) * UWN_ROWB + aoff[j];
                            const unsigned char* sb = slot + (32 * ks + 8 * q) * UWN_ROWB + boff[j];
                            const uint2 a0 = ds_tr16(sa), a1 = ds_tr16(sa + 4 * UWN_ROWB);
                            const uint2 b0 = ds_tr16(sb), b1 = ds_tr16(sb + 4 * UWN_ROWB);
                            a[j] = make_uint4(a0.x, a0.y, a1.x, a1.y);
                            b[j] = make_uint4(b0.x, b0.y, b1.x, b1.y);
                        } else {
                            const int po = ((4 * ks + q) ^ fsw) << 4;
                            a[j] = *reinterpret_cast<const uint4*>(slot + aoff[j] + po);
                            b[j] = *reinterpret_cast<const uint4*>(slot + boff[j] + po);
                        }
                    }
                    if (tail) {
                        const int nb = n0 + 32 * ks + 8 * q;
#pragma unroll
                        for (int j = 0; j < NS; ++j) {
                            uint32_t* u = reinterpret_cast<uint32_t*>(&a[j]);
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const uint32_t lo = (nb + 2 * e < N) ? 0xffffu : 0u, hi = (nb + 2 * e + 1 < N) ? 0xffff0000u : 0u;
                                u[e] &= (lo | hi);
                            }
                        }
                    }
#pragma unroll
                    for (int j = 0; j < NS; ++j) acc[j] = DT::mfma16(a[j], b[j], acc[j]);
                }
            }
            __syncthreads();
        }
    };
    switch (nslots) {
        case 1: run(std::integral_constant<int, 1>{}); break;
        case 2: run(std::integral_constant<int, 2>{}); break;
        case 3: run(std::integral_constant<int, 3>{}); break;
        case 4: run(std::integral_constant<int, 4>{}); break;
        case 5: run(std::integral_constant<int, 5>{}); break;
        case 6: run(std::integral_constant<int, 6>{}); break;
        case 7: run(std::integral_constant<int, 7>{}); break;
        default: run(std::integral_constant<int, 8>{}); break;
    }

    // D[ci][ko]: col = ko = lane & 15, row ci = 4 * (lane >> 4) + reg
#pragma unroll
    for (int j = 0; j < UP16_MAXB; ++j) {
        if (!(meta[j] & 256)) continue;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const size_t idx = (size_t)wid[j] * 256 + (4 * q + reg) * 16 + f;
            if (scratch == nullptr) {
                float out = alpha * acc[j][reg];
                if (beta != 0.f) out += beta * DT::to_f32(DW[idx]);
                DW[idx] = DT::from_f32(out);
            } else {
                __hip_atomic_fetch_add(scratch + idx, acc[j][reg], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

// DW = alpha * scratch + beta * DW, rounded once (second pass of the split-minibatch path)
template <class DT>
__global__ void __launch_bounds__(256)
updat_finalize_kernel(const float* __restrict__ scratch, typename DT::T* __restrict__ DW, size_t n, float alpha, float beta) {
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    const float4 s = *reinterpret_cast<const float4*>(scratch + i);
    float v[4] = {alpha * s.x, alpha * s.y, alpha * s.z, alpha * s.w};
    if (beta != 0.f) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += beta * DT::to_f32(DW[i + e]);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) DW[i + e] = DT::from_f32(v[e]);
}

}  // namespace bsmm
