// bsmm_xcol_v3.h -- pipelined xprop kernel ('BSX3' plans), feature_axis = 1, bsize 32, 16-bit storage types.
//
// The staged kernel (bsmm_xcol_v2.h) waits out one memory round trip per phase: everything a phase needs is requested at the
// top of the phase before it and drained at its own top, with nothing in flight behind (its DMA stream alone takes 68 of the
// kernel's 81 us at the bench shape).  Hiding that round trip needs a third set of buffers, which 160 KiB does not hold for
// 128-row tiles.  This kernel works on 64-ROW tiles: slabs of 8 KiB, a ring of three rows of two slabs (48 KiB) and a pool of
// 56 weight slots handed out by the plan builder (bsmm_plan.h), so that in iteration i every wave issues one piece of the
// slabs and NW half weight blocks of the row that runs in iteration i + 2 -- a constant 1 + NW DMA instructions, hence
// `s_waitcnt vmcnt(1 + NW)` = "what I requested two iterations ago has landed" while the previous iteration's requests stay in
// flight.  The price: every weight block is fetched by twice as many workgroups (64 instead of 128 rows each).
//   workgroup = 16 output blocks x 64 minibatch rows, 16 waves, wave v owns output block v (2 row tiles x 16 accumulators);
//   one barrier per iteration (a row of up to two pair steps);
//   activation slab: 64 rows of 128 B, the eight 16-byte pieces of row r XOR-swizzled with (r >> 1) & 7;
//   weight block: rows of 64 B, four pieces XOR-swizzled with (r >> 2) & 3 (bprop) / natural + transposing reads (fprop).
#pragma once
#include "bsmm_common.h"
#include "bsmm_plan.h"
#include "bsmm_updat_v2.h"   // glds16_saddr, uniform_ptr, ds_tr16
#include "bsmm_xprop.h"      // XMap

namespace bsmm {

constexpr int X3_R = 64;                            // minibatch rows per workgroup
constexpr int X3_SLAB = X3_R * 128;                 // 8 KiB
constexpr int X3_WBASE = 6 * X3_SLAB;               // weight pool behind the ring of 3 x 2 slabs
constexpr int X3_LDS = X3_WBASE + (X3_POOL + 1) * 2048;
static_assert(X3_LDS <= 163840 && X3_R * X3_G * 64 <= X3_LDS && X3_AHEAD == 2, "ring, pool and epilogue tile must fit the LDS");

template <class DT, bool TRANSW, int NW>
__global__ void __launch_bounds__(64 * X3_G, 4)
xcol32_v3_kernel(const typename DT::T* __restrict__ X, const typename DT::T* __restrict__ Wsel,
                 typename DT::T* __restrict__ Y, const int32_t* __restrict__ plan, XMap map, int N, int Cin, int Kout) {
    typedef typename DT::T T;
    static_assert(DT::is16 && NW >= 1 && NW <= 4, "xcol v3 kernel: 16-bit storage types");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int tile, grp;
    if (!xmap_decode(map, blockIdx.x, tile, grp)) return;
    const int4 gh = *reinterpret_cast<const int4*>(plan + plan[5] + 4 * grp);
    const int it_off = __builtin_amdgcn_readfirstlane(gh.x), niters = __builtin_amdgcn_readfirstlane(gh.y);
    const int ob0 = __builtin_amdgcn_readfirstlane(gh.z), nob = __builtin_amdgcn_readfirstlane(gh.w);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int32_t* pxt = plan + plan[6] + it_off;
    const int32_t* cwt = plan + plan[7] + (size_t)it_off * X3_G + wave;
    const int32_t* dut = plan + plan[10] + ((size_t)it_off * X3_G + wave) * (2 * NW);
    const int r = lane & 31, h = lane >> 5;
    const int n_tile = tile * X3_R;
    const uint32_t base_addr = lds_addr_of(smem);
    const int npairs_full = Cin / 64;

    // activation DMA: a row's two slabs are 16 instructions of 1 KiB (8 rows of 128 B); wave v issues instruction v & 7 of slab v >> 3
    const int xu = wave >> 3;
    uint32_t xvoff, xvoff_tail;
    {
        const int row = 8 * (wave & 7) + (lane >> 3);
        const int xr = min(n_tile + row, N - 1) - n_tile;            // rows past N are clamped (never stored)
        const int piece = (lane & 7) ^ ((row >> 1) & 7);
        xvoff = (uint32_t)xr * (uint32_t)Cin * 2u + piece * 16;
        xvoff_tail = xvoff - ((piece & 4) ? 64 : 0);                  // last pair of an odd block count: re-read its even half
    }
    const unsigned char* xtile = static_cast<const unsigned char*>(uniform_ptr(reinterpret_cast<const unsigned char*>(X) + (size_t)n_tile * Cin * 2));
    const unsigned char* wsel = static_cast<const unsigned char*>(uniform_ptr(Wsel));
    const uint32_t wvoff = TRANSW ? (uint32_t)lane * 16u : (uint32_t)((lane >> 2) * 64 + (((lane & 3) ^ ((lane >> 4) & 3)) << 4));
    const uint32_t xdst0 = base_addr + xu * X3_SLAB + (wave & 7) * 1024, wdst0 = base_addr + X3_WBASE;

    // fragment read offsets
    const int xsw = (r >> 1) & 7;
    uint32_t xrd[2][2];      // [half][kk], inside a 32-row band of a slab
#pragma unroll
    for (int half = 0; half < 2; ++half)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) xrd[half][kk] = r * 128 + (((4 * half + 2 * kk + h) ^ xsw) << 4);
    uint32_t wrd[2];         // [kk], inside slot 0 of the pool
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        if constexpr (TRANSW) {
            const int g16 = lane >> 4, t16 = lane & 15;
            wrd[kk] = X3_WBASE + (16 * kk + 8 * h + (t16 >> 2)) * 64 + (16 * (g16 & 1) + 4 * (t16 & 3)) * 2;
        } else {
            wrd[kk] = X3_WBASE + r * 64 + (((2 * kk + h) ^ ((r >> 2) & 3)) << 4);
        }
    }

    f32x16 acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

    // requests of one iteration: my piece of the slabs of the pairs in px_ into ring row (it_ % 3), my NW half weight blocks
#define X3_ISSUE(px_, it_, dsrc_, ddst_)                                                                                    \
    do {                                                                                                                    \
        const int pu = (int)(((uint32_t)(px_) >> (16 * xu)) & 0xffffu);                                                     \
        glds16_saddr(xtile + (size_t)pu * 128, pu < npairs_full ? xvoff : xvoff_tail, xdst0 + (uint32_t)((it_) % 3) * (2 * X3_SLAB)); \
        _Pragma("unroll") for (int k_ = 0; k_ < NW; ++k_)                                                                   \
            glds16_saddr(wsel + (uint32_t)(dsrc_[k_]), wvoff, wdst0 + (uint32_t)(ddst_[k_]));                               \
    } while (0)

    // one block: weight fragment from its slot, the two row tiles' activation fragments, 4 MFMAs
    auto block = [&](uint32_t xoff, uint32_t slot, int half) {
        uint4 wq[2], xf[2][2];
        const uint32_t woff = slot << 11;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            if constexpr (TRANSW) {
                const uint2 lo = ds_tr16(smem + wrd[kk] + woff), hi = ds_tr16(smem + wrd[kk] + woff + 4 * 64);
                wq[kk] = make_uint4(lo.x, lo.y, hi.x, hi.y);
            } else {
                wq[kk] = *reinterpret_cast<const uint4*>(smem + wrd[kk] + woff);
            }
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int t = 0; t < 2; ++t) xf[t][kk] = *reinterpret_cast<const uint4*>(smem + xrd[half][kk] + xoff + t * 4096);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int t = 0; t < 2; ++t) acc[t] = DT::mfma32(wq[kk], xf[t][kk], acc[t]);
    };

    if (niters > X3_AHEAD) {
        for (int tb = X3_AHEAD; tb < niters; tb += 64) {     // lane-indexed tables of iterations [tb, tb + 64)
            const int idx = min(tb + lane, niters - 1);
            int pxv = pxt[idx], cwv = cwt[(size_t)idx * X3_G];
            int dv[2 * NW];
#pragma unroll
            for (int k = 0; k < 2 * NW; ++k) dv[k] = dut[(size_t)idx * X3_G * 2 * NW + k];
            // the table loads must have landed before the loop: a wait the compiler places INSIDE it would drain the DMA queue
            asm volatile("" : "+v"(pxv), "+v"(cwv));
#pragma unroll
            for (int k = 0; k < 2 * NW; ++k) asm volatile("" : "+v"(dv[k]));
            if (tb == X3_AHEAD) {    // prologue: iterations 0 and 1 only request
#pragma unroll
                for (int it = 0; it < X3_AHEAD; ++it) {
                    const int p = __builtin_amdgcn_readfirstlane(pxt[it]);
                    int src[NW], dst[NW];
#pragma unroll
                    for (int k = 0; k < NW; ++k) {
                        src[k] = __builtin_amdgcn_readfirstlane(dut[(size_t)it * X3_G * 2 * NW + 2 * k]);
                        dst[k] = __builtin_amdgcn_readfirstlane(dut[(size_t)it * X3_G * 2 * NW + 2 * k + 1]);
                    }
                    X3_ISSUE(p, it, src, dst);
                }
            }
            const int tend = min(64, niters - tb);
            for (int qi = 0; qi < tend; ++qi) {
                const int it = tb + qi;
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(1 + NW) : "memory");   // my requests of iteration it - 2 have landed
                __builtin_amdgcn_s_barrier();                                    // everyone's have; everyone left iteration it - 1
                {
                    const int p = __builtin_amdgcn_readlane(pxv, qi);
                    int src[NW], dst[NW];
#pragma unroll
                    for (int k = 0; k < NW; ++k) { src[k] = __builtin_amdgcn_readlane(dv[2 * k], qi); dst[k] = __builtin_amdgcn_readlane(dv[2 * k + 1], qi); }
                    X3_ISSUE(p, it, src, dst);
                }
                const uint32_t cw = (uint32_t)__builtin_amdgcn_readlane(cwv, qi);
                const uint32_t xo = (uint32_t)((it + 1) % 3) * (2 * X3_SLAB);          // ring row of iteration it - 2
                if ((cw & 0xff) != 0xff)         block(xo, cw & 0xff, 0);
                if (((cw >> 8) & 0xff) != 0xff)  block(xo, (cw >> 8) & 0xff, 1);
                if (((cw >> 16) & 0xff) != 0xff) block(xo + X3_SLAB, (cw >> 16) & 0xff, 0);
                if ((cw >> 24) != 0xff)          block(xo + X3_SLAB, cw >> 24, 1);
            }
        }
    }
#undef X3_ISSUE

    // Epilogue: D[o][n]: col = n = r (lane), rows o = (reg & 3) + 8 * (reg >> 2) + 4h.  The 16 waves own 16 ADJACENT output blocks =
    // 1024 contiguous bytes per minibatch row: staged through the idle ring as [64 rows][1024 B] (16-byte pieces of row n
    // XOR-swizzled with n & 31) and stored as full rows.
    constexpr int ROWB = X3_G * 64;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the trailing requests must not land in the staging tile
    __syncthreads();
    if (wave < nob) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int n = t * 32 + r;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t lo = (uint32_t)DT::from_f32(acc[t][4 * q + 0]) | ((uint32_t)DT::from_f32(acc[t][4 * q + 1]) << 16);
                const uint32_t hi = (uint32_t)DT::from_f32(acc[t][4 * q + 2]) | ((uint32_t)DT::from_f32(acc[t][4 * q + 3]) << 16);
                const int piece = wave * 4 + q;
                *reinterpret_cast<uint2*>(smem + n * ROWB + ((piece ^ (n & 31)) << 4) + 8 * h) = make_uint2(lo, hi);
            }
        }
    }
    __syncthreads();
    {
        const int rowbytes = nob * 64;
        T* ybase = Y + (size_t)ob0 * 32;
        constexpr int PPR = ROWB / 16;
        for (int i = threadIdx.x; i < X3_R * PPR; i += 64 * X3_G) {
            const int n = i / PPR, piece = i % PPR;
            if (n_tile + n < N && piece * 16 < rowbytes) {
                const uint4 v = *reinterpret_cast<const uint4*>(smem + n * ROWB + ((piece ^ (n & 31)) << 4));
                *reinterpret_cast<uint4*>(reinterpret_cast<unsigned char*>(ybase + (size_t)(n_tile + n) * Kout) + piece * 16) = v;
            }
        }
    }
}

}  // namespace bsmm
