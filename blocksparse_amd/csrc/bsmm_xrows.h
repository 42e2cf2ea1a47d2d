// bsmm_xrows.h -- xprop kernel "a SIMD owns a row quarter" ('BSX5' plans, round 5): feature_axis = 1, bsize 32, 16-bit storage types.
//
// What bounded bsmm_xflow.h (profiles/r04_headline_ab.md): a wave owned an output COLUMN, and the blocks of a column cluster -- inside
// the ring's window of five steps the busiest of the 16 columns has 5-6 blocks where the mean has 2, each block is a serial chain of
// ~1 300 cycles on its wave (256 of them matrix work), and the whole workgroup advances at that wave's pace: matrix pipe 29 % busy,
// L2 -> LDS delivery at 0.38 of its ceiling.  No order of the steps and no dealing of columns to SIMDs moves it (both measured, round 3 / 4).
//
// Here the work is cut the other way.  The unit is the same (128 minibatch rows x 16 output blocks, one 16 KiB activation slab per PAIR
// of input blocks, every weight block once through the LDS), but a workgroup is 8 waves of 256 registers:
//     wave = (row quarter q = wave & 3, column half hc = wave >> 2); it owns rows 32 q .. 32 q + 31 of the tile and output blocks
//     8 hc .. 8 hc + 7 of the group: 8 accumulators of 32 x 32 (128 registers).
// Waves w and w + 4 share a SIMD, so every SIMD multiplies EVERY block of the group by its own 32 rows: the four matrix pipes carry exactly
// the same work in every step, whatever the layout -- balance by construction instead of by scheduling.  A step is a pair of input blocks
// (split when it holds more than X5_CAP blocks); per step a wave reads its four activation fragments (32 rows x 2 blocks x 2 K halves) once
// and, for each of its blocks in the step (a 16-bit mask: bit 2 kl + half), the two weight fragments and two MFMAs.  The weight fragments
// of the next block are requested right behind the MFMAs of the current one, into the same registers (an MFMA reads A / B at issue).
// Ring: X5_D = 5 activation slabs + X5_NW = 39 weight slots of 2 KiB (all 160 KiB).  The plan (bsmm_plan.h, 'BSX5') is a list of RECORDS per
// group, one per step, preceded by X5_P duty-only records (the prologue).  A record names
//     * the DMA DUTIES of the step: at most one activation slab (16 instructions of 1 KiB: wave w issues 2 w and 2 w + 1) and up to 16 weight
//       blocks (entry e: wave pair e & 3 issues the block's two halves) -- requested as far ahead as the ring allows (slabs 4 steps, weights
//       until the 39 slots are full), every slot's previous occupant being a step all waves have left (they passed this step's barrier);
//     * per wave pair the vmcnt to wait with in front of the step's barrier: the number of DMA instructions the pair issued AFTER the last one
//       this step reads (counted by the builder: the order of a wave's vector-memory operations is fixed by the plan);
//     * the step's slab slot, block masks and first weight slot per column half.
// One s_barrier per step and nothing else: no counters, no polling, no per-event bookkeeping.  The kernel is persistent (one workgroup per CU
// walks its units); the prologue of the NEXT unit is issued behind the last step's barrier and lands while the waves write their output
// (through the one slab slot the prologue does not touch).
// Fragment layouts, swizzles, MFMA operand roles and the order in which a column sums its blocks are those of bsmm_xcol_v2.h / bsmm_xflow.h:
// bit-identical outputs.
#pragma once
#include "bsmm_common.h"
#include "bsmm_plan.h"
#include "bsmm_updat_v2.h"   // glds16_saddr, uniform_ptr
#include "bsmm_updat_tr.h"   // ds_tr16
#include "bsmm_xprop.h"      // XMap

namespace bsmm {

#ifndef X5_NO_XDMA
#define X5_NO_XDMA 0          // ablation switches: wrong results by construction
#endif
#ifndef X5_NO_WDMA
#define X5_NO_WDMA 0
#endif
#ifndef X5_NO_MATH
#define X5_NO_MATH 0
#endif
#ifndef X5_LATE_DUTIES
#define X5_LATE_DUTIES 1      // 1: the waves of column half 1 issue their duties AFTER their blocks (each SIMD then has one wave requesting
#endif                        //    while the other multiplies); 0: every wave right behind the barrier
#ifdef X5_STAMPS
// cycle accounting of the first 64 workgroups (debug builds): per wave [0] vmcnt waits, [1] barrier, [2] duties, [3] blocks, [4] epilogue,
// [5] whole kernel, [6] steps, [7] blocks; read back with bsmm_debug_x5_trace_copy()
__device__ unsigned long long g_x5_trace[64 * 8 * 8];
#define X5_T0() const unsigned long long t0_ = __builtin_readcyclecounter()
#define X5_T1(k) tacc[k] += __builtin_readcyclecounter() - t0_
#else
#define X5_T0() do { } while (0)
#define X5_T1(k) do { } while (0)
#endif

constexpr int X5_R = 128;                              // minibatch rows per unit
constexpr int X5_SLAB = X5_R * 128;                    // 16 KiB
constexpr int X5_WBASE = X5_D * X5_SLAB;               // weight slots behind the slabs
constexpr int X5_LDS = 163840;                         // slots 0 .. X5_NW - 1, and one guard slot the fragment prefetch may read
constexpr int X5_STAGE = (X5_D - 1) * X5_SLAB;         // epilogue staging: the slab slot the prologue never requests, 2 KiB per wave
static_assert(X5_WBASE + (X5_NW + 1) * 2048 <= X5_LDS, "rows kernel: ring must fit the LDS");

// s_waitcnt vmcnt(n), n wave-uniform at run time (0 .. 31): a computed jump into a table of 32 waits
__device__ __forceinline__ void x5_wait_vmcnt(uint32_t n) {
#define X5_W1(k) "s_waitcnt vmcnt(" #k ")\n\ts_branch 99f\n\t"
    asm volatile("s_getpc_b64 s[20:21]\n\t"
                 "s_lshl_b32 s22, %0, 3\n\t"
                 "s_add_u32 s22, s22, 20\n\t"
                 "s_add_u32 s20, s20, s22\n\t"
                 "s_addc_u32 s21, s21, 0\n\t"
                 "s_setpc_b64 s[20:21]\n\t"
                 X5_W1(0) X5_W1(1) X5_W1(2) X5_W1(3) X5_W1(4) X5_W1(5) X5_W1(6) X5_W1(7)
                 X5_W1(8) X5_W1(9) X5_W1(10) X5_W1(11) X5_W1(12) X5_W1(13) X5_W1(14) X5_W1(15)
                 X5_W1(16) X5_W1(17) X5_W1(18) X5_W1(19) X5_W1(20) X5_W1(21) X5_W1(22) X5_W1(23)
                 X5_W1(24) X5_W1(25) X5_W1(26) X5_W1(27) X5_W1(28) X5_W1(29) X5_W1(30) X5_W1(31)
                 "99:"
                 ::"s"(n) : "memory", "scc", "s20", "s21", "s22");
#undef X5_W1
}

// lgkmcnt(0) + s_barrier.  The wait is the BUILTIN (encoding: vmcnt 63, expcnt 7, lgkmcnt 0): the compiler's own counter model must know that
// no LDS read is pending behind it -- a fragment prefetch nobody consumed would otherwise make it wait (lgkmcnt(0), scalar loads included)
// in front of the first register it reuses
__device__ __forceinline__ void x5_barrier() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

template <class DT, bool TRANSW>
__global__ void __launch_bounds__(512, 2)
xrows32_kernel(const typename DT::T* __restrict__ X, const typename DT::T* __restrict__ Wsel, typename DT::T* __restrict__ Y,
               const int32_t* __restrict__ plan, XMap map, int N, int Cin, int Kout) {
    typedef typename DT::T T;
    static_assert(DT::is16, "rows kernel: 16-bit storage types");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int q = wave & 3, hc = wave >> 2, wp = wave >> 1, wh = wave & 1;
    const uint32_t base_addr = lds_addr_of(smem);

    const int npairs_full = Cin / 64;
    const unsigned char* xt = reinterpret_cast<const unsigned char*>(X);
    const unsigned char* wsel = static_cast<const unsigned char*>(uniform_ptr(Wsel));
    // weight DMA (bsmm_xcol_v2.h): lane i of an instruction writes piece i of a 1 KiB half block; it fetches the piece that the read
    // swizzle expects there (none for the transposing reads of fprop).  This wave moves half `wh` of the blocks of its pair's entries.
    const uint32_t wvoff = (TRANSW ? (uint32_t)lane * 16u : (uint32_t)((lane >> 2) * 64 + (((lane & 3) ^ ((lane >> 4) & 3)) << 4))) + (uint32_t)wh * 1024u;
    // fragment read offsets.  Activations: row 32 q + r of the slab, 16-byte piece (2 kk + h + 4 half) ^ ((row >> 1) & 7); weights: relative
    // to the block's slot
    const int xsw = (r >> 1) & 7;
    uint32_t xo[2][2], wrd[2];
#pragma unroll
    for (int ab = 0; ab < 2; ++ab)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) xo[ab][kk] = (uint32_t)((32 * q + r) * 128 + (((2 * kk + h + 4 * ab) ^ xsw) << 4));
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        if constexpr (TRANSW) {
            const int g16 = lane >> 4, t16 = lane & 15;
            wrd[kk] = X5_WBASE + (16 * kk + 8 * h + (t16 >> 2)) * 64 + (16 * (g16 & 1) + 4 * (t16 & 3)) * 2;
        } else {
            wrd[kk] = X5_WBASE + r * 64 + (((2 * kk + h) ^ ((r >> 2) & 3)) << 4);
        }
    }
    // slab request offsets of a full tile: DMA instruction ii covers rows 8 ii + (lane >> 3); the 16-byte piece a lane fetches is
    // (lane & 7) ^ ((row >> 1) & 7) = (lane & 7) ^ (lane >> 4) ^ (4 (ii & 1)).  This wave issues ii = 2 wave (even) and 2 wave + 1 (odd).
    const uint32_t stride16 = (uint32_t)Cin * 16u;                       // bytes between the first rows of consecutive instructions
    const uint32_t pc_e = (uint32_t)((lane & 7) ^ (lane >> 4));
    const uint32_t vo_e0 = (uint32_t)(lane >> 3) * (uint32_t)Cin * 2u + pc_e * 16u;
    const uint32_t vx_e = vo_e0 + (uint32_t)(2 * wave) * stride16;
    const uint32_t vx_o = ((pc_e & 4u) ? vo_e0 - 64u : vo_e0 + 64u) + (uint32_t)(2 * wave + 1) * stride16;
    const uint32_t xdst_w = (uint32_t)(2 * wave) * 1024u;                // my two instructions' place inside a slab

#ifdef X5_STAMPS
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long tstart = __builtin_readcyclecounter();
#endif
    const int nunits = map.grid();
    const int32_t* const groups = plan + plan[5];
    const int32_t* const recs0 = plan + plan[6];

    // the words of a record a wave needs: [0..3] pair / slab slot / masks / first weight slots, [4..7] waits / slab duty (pair, slot) / -,
    // and its wave pair's four fetch entries -- three scalar loads, issued together one step ahead
#define X5_LOAD_REC(A, B, E, rc_)                                                                                                \
    do {                                                                                                                         \
        A = *reinterpret_cast<const int4*>(rc_);                                                                                 \
        B = *reinterpret_cast<const int4*>((rc_) + 4);                                                                           \
        E = *reinterpret_cast<const int4*>((rc_) + 16 + 4 * wp);                                                                 \
    } while (0)
    // the duties of one record, for the unit whose rows start at n_tile
    auto duties = [&](const int4 rb, const int4 re, const unsigned char* xtile, int n_tile) {
        const int xp = __builtin_amdgcn_readfirstlane(rb.y);
        if (xp >= 0) {
            const uint32_t dst = base_addr + (uint32_t)__builtin_amdgcn_readfirstlane(rb.z) * (uint32_t)X5_SLAB + xdst_w;
            if (!X5_NO_XDMA) {
                if (n_tile + X5_R <= N && xp < npairs_full) {
                    const uint32_t po = (uint32_t)xp * 128u;
                    glds16_saddr_x2(xtile, vx_e + po, vx_o + po, dst);
                } else {
                    const bool tail = xp >= npairs_full;
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const int row = 8 * (2 * wave + k) + (lane >> 3);
                        const int xr = min(n_tile + row, N - 1) - n_tile;    // rows past N are clamped (never stored)
                        const int piece = (lane & 7) ^ ((row >> 1) & 7);
                        uint32_t voff = (uint32_t)xr * (uint32_t)Cin * 2u + piece * 16 + (uint32_t)xp * 128u;
                        if (tail && (piece & 4)) voff -= 64;                 // last pair of an odd block count: re-read its even half
                        glds16_saddr(xtile, voff, dst + k * 1024);
                    }
                }
            }
        }
        const int e[4] = {__builtin_amdgcn_readfirstlane(re.x), __builtin_amdgcn_readfirstlane(re.y), __builtin_amdgcn_readfirstlane(re.z),
                          __builtin_amdgcn_readfirstlane(re.w)};
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (e[i] >= 0) {
                if (!X5_NO_WDMA) {
                    const uint32_t fo = ((uint32_t)e[i] & 0x1fffffu) << 11, slot = ((uint32_t)e[i] >> 21) & 63u;
                    glds16_saddr(wsel, wvoff + fo, base_addr + (uint32_t)X5_WBASE + slot * 2048u + (uint32_t)wh * 1024u);
                }
            }
    };
    static_assert(X5_P == 4, "the prologue is written out for four records");
#define X5_PROLOGUE(rc_, xtile_, n_tile_)                                                                                        \
    do {                                                                                                                         \
        int4 pa, pb0, pe0, pb1, pe1, pb2, pe2, pb3, pe3;                                                                         \
        X5_LOAD_REC(pa, pb0, pe0, rc_);                                                                                          \
        X5_LOAD_REC(pa, pb1, pe1, (rc_) + X5_REC);                                                                               \
        X5_LOAD_REC(pa, pb2, pe2, (rc_) + 2 * X5_REC);                                                                           \
        X5_LOAD_REC(pa, pb3, pe3, (rc_) + 3 * X5_REC);                                                                           \
        (void)pa;                                                                                                                \
        duties(pb0, pe0, xtile_, n_tile_);                                                                                       \
        duties(pb1, pe1, xtile_, n_tile_);                                                                                       \
        duties(pb2, pe2, xtile_, n_tile_);                                                                                       \
        duties(pb3, pe3, xtile_, n_tile_);                                                                                       \
    } while (0)

    auto decode = [&](int u, int& tile, int& grp) -> int {        // first valid unit at or behind u (nunits: none)
        for (; u < nunits; u += gridDim.x)
            if (xmap_decode(map, u, tile, grp)) return u;
        return nunits;
    };

    int tile = 0, grp = 0;
    int unit = decode(blockIdx.x, tile, grp);
    if (unit < nunits) {
        const int32_t* gh = groups + X5_GROUP * grp;
        const int32_t* rc = recs0 + (size_t)__builtin_amdgcn_readfirstlane(gh[0]) * X5_REC;
        const unsigned char* xtile = static_cast<const unsigned char*>(uniform_ptr(xt + (size_t)tile * X5_R * Cin * 2));
        X5_PROLOGUE(rc, xtile, tile * X5_R);
    }
    f32x16 acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

    while (unit < nunits) {
        const int32_t* gh = groups + X5_GROUP * grp;
        const int nsteps = __builtin_amdgcn_readfirstlane(gh[1]);
        const int ob0 = __builtin_amdgcn_readfirstlane(gh[2]), nob = __builtin_amdgcn_readfirstlane(gh[3]);
        const int32_t* recs = recs0 + ((size_t)__builtin_amdgcn_readfirstlane(gh[0]) + X5_P) * X5_REC;
        const int n_tile = tile * X5_R;
        const unsigned char* xtile = static_cast<const unsigned char*>(uniform_ptr(xt + (size_t)n_tile * Cin * 2));

        int4 cur_a, cur_b, cur_e;                                        // (a group without blocks has no steps: the record behind its prologue is
        X5_LOAD_REC(cur_a, cur_b, cur_e, recs);                          //  the next group's, or the plan's padding record: read, not used)
        for (int s = 0; s < nsteps; ++s) {
            const int4 hd = cur_a, rb = cur_b, re = cur_e;               // hd: pair, slab slot, masks, first weight slots
            const uint32_t wn = ((uint32_t)__builtin_amdgcn_readfirstlane(rb.x) >> (8 * wp)) & 31u;
            { X5_T0(); x5_wait_vmcnt(wn); X5_T1(0); }                    // my requests that this step reads have landed
            { X5_T0(); x5_barrier(); X5_T1(1); }   // everyone's have; everyone left the previous step
            X5_LOAD_REC(cur_a, cur_b, cur_e, recs + (size_t)(s + 1) * X5_REC);   // the next step's words, in flight under this step's work
            const bool late = X5_LATE_DUTIES && hc == 1;
            if (!late) { X5_T0(); duties(rb, re, xtile, n_tile); X5_T1(2); }
#ifdef X5_STAMPS
            tacc[6] += 1;
#endif
            const uint32_t m = ((uint32_t)__builtin_amdgcn_readfirstlane(hd.z) >> (16 * hc)) & 0xffffu;
            if (m != 0 && !X5_NO_MATH) {
                X5_T0();
                const uint32_t xs = (uint32_t)__builtin_amdgcn_readfirstlane(hd.y) * (uint32_t)X5_SLAB;
                const uint32_t ws = (((uint32_t)__builtin_amdgcn_readfirstlane(hd.w) >> (16 * hc)) & 0xffffu) * 2048u;
                uint32_t vp0 = wrd[0] + ws, vp1 = wrd[1] + ws;
                uint4 wq[2];
#define X5_READW()                                                                                                               \
    do {                                                                                                                         \
        if constexpr (TRANSW) {                                                                                                  \
            const uint2 l0 = ds_tr16(smem + vp0), h0 = ds_tr16(smem + vp0 + 4 * 64);                                             \
            const uint2 l1 = ds_tr16(smem + vp1), h1 = ds_tr16(smem + vp1 + 4 * 64);                                             \
            wq[0] = make_uint4(l0.x, l0.y, h0.x, h0.y);                                                                          \
            wq[1] = make_uint4(l1.x, l1.y, h1.x, h1.y);                                                                          \
        } else {                                                                                                                 \
            wq[0] = *reinterpret_cast<const uint4*>(smem + vp0);                                                                 \
            wq[1] = *reinterpret_cast<const uint4*>(smem + vp1);                                                                 \
        }                                                                                                                        \
        vp0 += 2048u; vp1 += 2048u;                                                                                              \
    } while (0)
                X5_READW();
                uint4 xf[2][2];
#pragma unroll
                for (int ab = 0; ab < 2; ++ab)
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk) xf[ab][kk] = *reinterpret_cast<const uint4*>(smem + xs + xo[ab][kk]);
                // the next record's words (scalar loads issued behind the barrier) are pinned HERE: with a scalar load pending the compiler
                // could count none of the fragment reads below and would wait for all of them in front of every MFMA
                // (every loaded word is named: a destination register the compiler may reuse earlier would bring its wait forward with it)
                asm volatile("" ::"s"(cur_a.x), "s"(cur_a.y), "s"(cur_a.z), "s"(cur_a.w), "s"(cur_b.x), "s"(cur_b.y), "s"(cur_b.z), "s"(cur_b.w),
                             "s"(cur_e.x), "s"(cur_e.y), "s"(cur_e.z), "s"(cur_e.w));
                // position p = 2 kl + half.  ONE register set of weight fragments: the matrix instruction reads its A / B operands when it
                // issues (only the accumulator operand is read later), so the next block's fragments are requested into the same registers
                // right behind the block's two MFMAs and land under them and under the partner wave's work
#ifdef X5_STAMPS
#define X5_COUNT() tacc[7] += 1
#else
#define X5_COUNT() do { } while (0)
#endif
#define X5_BLK(p)                                                                                                                \
    if (m & (1u << (p))) {                                                                                                       \
        acc[(p) >> 1] = DT::mfma32(wq[0], xf[(p) & 1][0], acc[(p) >> 1]);                                                        \
        acc[(p) >> 1] = DT::mfma32(wq[1], xf[(p) & 1][1], acc[(p) >> 1]);                                                        \
        X5_READW();                                                                                                              \
        X5_COUNT();                                                                                                              \
    }
                X5_BLK(0) X5_BLK(1) X5_BLK(2) X5_BLK(3) X5_BLK(4) X5_BLK(5) X5_BLK(6) X5_BLK(7)
                X5_BLK(8) X5_BLK(9) X5_BLK(10) X5_BLK(11) X5_BLK(12) X5_BLK(13) X5_BLK(14) X5_BLK(15)
#undef X5_BLK
#undef X5_READW
                X5_T1(3);
            }
            if (late) { X5_T0(); duties(rb, re, xtile, n_tile); X5_T1(2); }
        }

        // ---- unit end: everyone has left the last step -> the ring is free; the next unit's prologue flies while the output is written ----
        x5_barrier();
        int ntile = 0, ngrp = 0;
        const int nxt = decode(unit + gridDim.x, ntile, ngrp);
        if (nxt < nunits) {
            const int32_t* ngh = groups + X5_GROUP * ngrp;
            const int32_t* rc = recs0 + (size_t)__builtin_amdgcn_readfirstlane(ngh[0]) * X5_REC;
            const unsigned char* nxtile = static_cast<const unsigned char*>(uniform_ptr(xt + (size_t)ntile * X5_R * Cin * 2));
            X5_PROLOGUE(rc, nxtile, ntile * X5_R);
        }
        // Epilogue, per wave, through its 2 KiB of the free slab slot: D[o][n] with col n = r (minibatch row 32 q + r), rows o = (reg & 3) +
        // 8 (reg >> 2) + 4 h.  Per output block: [32 rows n][64 B], the four 16-byte pieces of row n XOR-swizzled with (n >> 2) & 3; read back
        // as full 64-byte rows and stored (16 rows per instruction).
        {
            X5_T0();
            unsigned char* stage = smem + X5_STAGE + wave * 2048;
#pragma unroll
            for (int kl = 0; kl < 8; ++kl) {
                if (8 * hc + kl < nob) {
                    unsigned char* ybase = reinterpret_cast<unsigned char*>(Y + (size_t)(ob0 + 8 * hc + kl) * 32);
#pragma unroll
                    for (int qd = 0; qd < 4; ++qd) {
                        const uint32_t lo = (uint32_t)DT::from_f32(acc[kl][4 * qd + 0]) | ((uint32_t)DT::from_f32(acc[kl][4 * qd + 1]) << 16);
                        const uint32_t hi = (uint32_t)DT::from_f32(acc[kl][4 * qd + 2]) | ((uint32_t)DT::from_f32(acc[kl][4 * qd + 3]) << 16);
                        *reinterpret_cast<uint2*>(stage + r * 64 + ((qd ^ ((r >> 2) & 3)) << 4) + 8 * h) = make_uint2(lo, hi);
                    }
                    asm volatile("" ::: "memory");
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const int n = 16 * i + (lane >> 2), pc = lane & 3;
                        const uint4 v = *reinterpret_cast<const uint4*>(stage + n * 64 + ((pc ^ ((n >> 2) & 3)) << 4));
                        const int gn = n_tile + 32 * q + n;
                        if (gn < N) *reinterpret_cast<uint4*>(ybase + (size_t)gn * Kout * 2 + pc * 16) = v;
                    }
                    asm volatile("" ::: "memory");
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[kl][i] = 0.f;
            }
            X5_T1(4);
        }
        unit = nxt; tile = ntile; grp = ngrp;
    }
#ifdef X5_STAMPS
    tacc[5] = __builtin_readcyclecounter() - tstart;
    if (blockIdx.x < 64 && lane == 0)
        for (int k = 0; k < 8; ++k) g_x5_trace[(blockIdx.x * 8 + wave) * 8 + k] = tacc[k];
#endif
}

}  // namespace bsmm
