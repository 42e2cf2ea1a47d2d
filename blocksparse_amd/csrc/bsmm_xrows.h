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
#include <type_traits>

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

// LDS-DMA of 1 KiB (2 x 1 KiB: dst, dst + 1 KiB) from a scalar base + a 32-bit lane offset.  M0 is NOT saved / restored: nothing else in this
// kernel reads it (no LDS-DMA builtin, no GWS / sendmsg; the build script's audit greps the kernel's ISA for other m0 uses), and the two
// scalar moves per request are worth having -- the CU's scalar unit is what this kernel runs out of first (profiles/r05_xrows_v1_*.log)
__device__ __forceinline__ void x5_dma1(const void* sbase, uint32_t voff, uint32_t lds_byte_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_byte_addr) : "memory");
}
__device__ __forceinline__ void x5_dma2(const void* sbase, uint32_t voff0, uint32_t voff1, uint32_t lds_byte_addr) {
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2\n\t"
                 "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                 ::"v"(voff0), "v"(voff1), "s"(sbase), "s"(lds_byte_addr) : "memory", "scc");
}

// ---- the blocks of a step, hand-written: what the compiler makes of a chain of 16 `if (bit) { 2 MFMAs }` costs ~10 scalar
// instructions and two taken branches per block, executed by all four row-quarter waves of a column half -- and the CU has ONE scalar unit.
// Here a block is dispatched by a computed jump: s_ff1 on the mask (bit 16 = sentinel: the exit), clear the bit, jump to body 128 B x index.
// Two register sets of weight fragments, chosen by CODE COPY (copy A multiplies from set A and prefetches the next block into set B, then
// jumps into copy B, and vice versa); activation fragments and both sets live in v[224:255], named literally and listed as clobbers.
//   v[224:227] / v[228:231]  activation fragments of the pair's even block (K half 0 / 1),  v[232:235] / v[236:239] of its odd block
//   v[240:247] set A, v[248:255] set B (fragment of K half 0, then of K half 1)
// v[224:255] are RESERVED from the compiler for the whole kernel (amdgpu_num_vgpr(224)): the fragments a step's exit requests for the next
// step stay in them across the barrier and the compiler's code in between.
#define X5S_(x) #x
#define X5S(x) X5S_(x)
// request the weight fragments of the block at vp0 (/ vp1) into the set that starts at register `a`, advance the pointers
#define X5_RD_N(a)                                                                                                               \
    "ds_read_b128 v[" X5S(a) ":" X5S(a) "+3], %[vp0]\n\t"                                                                          \
    "ds_read_b128 v[" X5S(a) "+4:" X5S(a) "+7], %[vp1]\n\t"                                                                        \
    "v_add_u32 %[vp0], 0x800, %[vp0]\n\t"                                                                                        \
    "v_add_u32 %[vp1], 0x800, %[vp1]\n\t"
#define X5_RD_T(a)                                                                                                               \
    "ds_read_b64_tr_b16 v[" X5S(a) ":" X5S(a) "+1], %[vp0]\n\t"                                                                    \
    "ds_read_b64_tr_b16 v[" X5S(a) "+2:" X5S(a) "+3], %[vp0] offset:256\n\t"                                                       \
    "ds_read_b64_tr_b16 v[" X5S(a) "+4:" X5S(a) "+5], %[vp0] offset:1024\n\t"                                                      \
    "ds_read_b64_tr_b16 v[" X5S(a) "+6:" X5S(a) "+7], %[vp0] offset:1280\n\t"                                                      \
    "v_add_u32 %[vp0], 0x800, %[vp0]\n\t"
// next block: index of the lowest mask bit -> its body in the copy whose base is s[lo:lo+1]
#define X5_DISPATCH(lo, hi)                                                                                                          \
    "s_ff1_i32_b32 s26, %[m]\n\t"                                                                                                \
    "s_bitset0_b32 %[m], s26\n\t"                                                                                                \
    "s_lshl_b32 s26, s26, 7\n\t"                                                                                                 \
    "s_add_u32 s20, s" X5S(lo) ", s26\n\t"                                                                                       \
    "s_addc_u32 s21, s" X5S(hi) ", 0\n\t"                                                                                              \
    "s_setpc_b64 s[20:21]\n\t"
// body of position p in copy `cp` (cur = first register of the set it multiplies from, nxt = of the set it prefetches into, nlo = base of
// the other copy, RD = the request macro, NW = LDS reads per request): x0 = first register of the position's activation fragments
#define X5_BODY(cp, p, accn, x0, cur, nxt, nlo, nhi, RD, NW, MF)                                                                          \
    ".p2align 7\n"                                                                                                               \
    "LX5" cp X5S(p) "_%=:\n\t"                                                                                                    \
    RD(nxt)                                                                                                                      \
    "s_waitcnt lgkmcnt(" X5S(NW) ")\n\t"                                                                                         \
    MF " %[" accn "], v[" X5S(cur) ":" X5S(cur) "+3], v[" X5S(x0) ":" X5S(x0) "+3], %[" accn "]\n\t"                                 \
    MF " %[" accn "], v[" X5S(cur) "+4:" X5S(cur) "+7], v[" X5S(x0) "+4:" X5S(x0) "+7], %[" accn "]\n\t"                             \
    X5_DISPATCH(nlo, nhi)
// the four activation fragments of a step and the weight fragments of its first block (into set A); the pointers move on to the second block
#define X5_FIRST(RD, xa0, xa1, xb0, xb1)                                                                                         \
    "ds_read_b128 v[224:227], %[" xa0 "]\n\t"                                                                                    \
    "ds_read_b128 v[228:231], %[" xa1 "]\n\t"                                                                                    \
    "ds_read_b128 v[232:235], %[" xb0 "]\n\t"                                                                                    \
    "ds_read_b128 v[236:239], %[" xb1 "]\n\t"                                                                                    \
    RD(240)
#define X5_COPY(cp, cur, nxt, nlo, nhi, RD, NW, MF)                                                                              \
    X5_BODY(cp, 0, "a0", 224, cur, nxt, nlo, nhi, RD, NW, MF) X5_BODY(cp, 1, "a0", 232, cur, nxt, nlo, nhi, RD, NW, MF)          \
    X5_BODY(cp, 2, "a1", 224, cur, nxt, nlo, nhi, RD, NW, MF) X5_BODY(cp, 3, "a1", 232, cur, nxt, nlo, nhi, RD, NW, MF)          \
    X5_BODY(cp, 4, "a2", 224, cur, nxt, nlo, nhi, RD, NW, MF) X5_BODY(cp, 5, "a2", 232, cur, nxt, nlo, nhi, RD, NW, MF)          \
    X5_BODY(cp, 6, "a3", 224, cur, nxt, nlo, nhi, RD, NW, MF) X5_BODY(cp, 7, "a3", 232, cur, nxt, nlo, nhi, RD, NW, MF)          \
    X5_BODY(cp, 8, "a4", 224, cur, nxt, nlo, nhi, RD, NW, MF) X5_BODY(cp, 9, "a4", 232, cur, nxt, nlo, nhi, RD, NW, MF)          \
    X5_BODY(cp, 10, "a5", 224, cur, nxt, nlo, nhi, RD, NW, MF) X5_BODY(cp, 11, "a5", 232, cur, nxt, nlo, nhi, RD, NW, MF)        \
    X5_BODY(cp, 12, "a6", 224, cur, nxt, nlo, nhi, RD, NW, MF) X5_BODY(cp, 13, "a6", 232, cur, nxt, nlo, nhi, RD, NW, MF)        \
    X5_BODY(cp, 14, "a7", 224, cur, nxt, nlo, nhi, RD, NW, MF) X5_BODY(cp, 15, "a7", 232, cur, nxt, nlo, nhi, RD, NW, MF)        \
    /* exit: the step's blocks are issued -- request the NEXT step's first fragments (its data landed at this step's barrier; behind the   \
       unit's last step: addresses of slot 0, read and never used).  A prefetch of the last block may still be in flight into set A:      \
       LDS reads of a wave return in order, the later one wins. */                                                                          \
    ".p2align 7\n"                                                                                                               \
    "LX5" cp "16_%=:\n\t"                                                                                                        \
    "v_mov_b32 %[vp0], %[nvp0]\n\t"                                                                                              \
    "v_mov_b32 %[vp1], %[nvp1]\n\t"                                                                                              \
    X5_FIRST(RD, "nxa0", "nxa1", "nxb0", "nxb1")                                                                                 \
    "s_branch LX5END_%=\n"
#define X5_STEP_ASM(RD, NW, MF)                                                                                                  \
    "s_getpc_b64 s[20:21]\n"                                                                                                     \
    "LX5REF_%=:\n\t"                                                                                                             \
    "s_add_u32 s22, s20, LX5A0_%=-LX5REF_%=\n\t"                                                                                 \
    "s_addc_u32 s23, s21, 0\n\t"                                                                                                 \
    "s_add_u32 s24, s20, LX5B0_%=-LX5REF_%=\n\t"                                                                                 \
    "s_addc_u32 s25, s21, 0\n\t"                                                                                                 \
    X5_DISPATCH(22, 23)                                                                                                          \
    X5_COPY("A", 240, 248, 24, 25, RD, NW, MF)                                                                                   \
    X5_COPY("B", 248, 240, 22, 23, RD, NW, MF)                                                                                   \
    "LX5END_%=:\n\t"

#define X5_REGS_CLOBBERED                                                                                                        \
    "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239", "v240",  \
    "v241", "v242", "v243", "v244", "v245", "v246", "v247", "v248", "v249", "v250", "v251", "v252", "v253", "v254", "v255"

// the blocks of one step (mask m, fragments of its first block already requested: by the previous step's exit or by x5_first), then the
// request for the next step's first fragments (nx*: their LDS addresses).  vp0 / vp1 enter pointing at the step's SECOND block and leave
// pointing at the next step's second block.
template <class DT, bool TRANSW>
__device__ __forceinline__ void x5_step_blocks(f32x16 (&acc)[8], uint32_t m, uint32_t& vp0, uint32_t& vp1, uint32_t nxa0, uint32_t nxa1, uint32_t nxb0,
                                               uint32_t nxb1, uint32_t nvp0, uint32_t nvp1) {
    m |= 0x10000u;                                          // the sentinel: position 16 = the exit body
#define X5_ASM_OPERANDS                                                                                                          \
    : [a0] "+v"(acc[0]), [a1] "+v"(acc[1]), [a2] "+v"(acc[2]), [a3] "+v"(acc[3]), [a4] "+v"(acc[4]), [a5] "+v"(acc[5]), [a6] "+v"(acc[6]),  \
      [a7] "+v"(acc[7]), [m] "+s"(m), [vp0] "+v"(vp0), [vp1] "+v"(vp1)                                                            \
    : [nxa0] "v"(nxa0), [nxa1] "v"(nxa1), [nxb0] "v"(nxb0), [nxb1] "v"(nxb1), [nvp0] "v"(nvp0), [nvp1] "v"(nvp1)                 \
    : "memory", "scc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", X5_REGS_CLOBBERED
    if constexpr (std::is_same<DT, DTbf16>::value) {
        if constexpr (TRANSW) asm volatile(X5_STEP_ASM(X5_RD_T, 4, "v_mfma_f32_32x32x16_bf16") X5_ASM_OPERANDS);
        else                  asm volatile(X5_STEP_ASM(X5_RD_N, 2, "v_mfma_f32_32x32x16_bf16") X5_ASM_OPERANDS);
    } else {
        if constexpr (TRANSW) asm volatile(X5_STEP_ASM(X5_RD_T, 4, "v_mfma_f32_32x32x16_f16") X5_ASM_OPERANDS);
        else                  asm volatile(X5_STEP_ASM(X5_RD_N, 2, "v_mfma_f32_32x32x16_f16") X5_ASM_OPERANDS);
    }
#undef X5_ASM_OPERANDS
}
// the first fragments of a unit's first step (nobody ran before it): the same requests as an exit body
template <bool TRANSW>
__device__ __forceinline__ void x5_first(uint32_t& vp0, uint32_t& vp1, uint32_t xa0, uint32_t xa1, uint32_t xb0, uint32_t xb1) {
    if constexpr (TRANSW)
        asm volatile(X5_FIRST(X5_RD_T, "xa0", "xa1", "xb0", "xb1") : [vp0] "+v"(vp0), [vp1] "+v"(vp1)
                     : [xa0] "v"(xa0), [xa1] "v"(xa1), [xb0] "v"(xb0), [xb1] "v"(xb1) : "memory", X5_REGS_CLOBBERED);
    else
        asm volatile(X5_FIRST(X5_RD_N, "xa0", "xa1", "xb0", "xb1") : [vp0] "+v"(vp0), [vp1] "+v"(vp1)
                     : [xa0] "v"(xa0), [xa1] "v"(xa1), [xb0] "v"(xb0), [xb1] "v"(xb1) : "memory", X5_REGS_CLOBBERED);
}

template <class DT, bool TRANSW>
__global__ void __launch_bounds__(512, 2) __attribute__((amdgpu_num_vgpr(224)))
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
    const uint32_t wdst = base_addr + (uint32_t)wh * 1024u;              // + the entry's slot offset: where my half of a weight block lands

#ifdef X5_STAMPS
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long tstart = __builtin_readcyclecounter();
#endif
    const int nunits = map.grid();
    const int32_t* const groups = plan + plan[5];
    const int32_t* const recs0 = plan + plan[6];

    // the words of a record a wave needs: [0..3] pair / slab offset / masks / first weight slots, [4..7] waits / slab duty (pair, offset) / -,
    // [8..11] the next step's slab offset / masks / first weight slots, and its wave pair's four fetch entries (two words each)
#define X5_LOAD_REC(A, B, C, E0, E1, rc_)                                                                                        \
    do {                                                                                                                         \
        A = *reinterpret_cast<const int4*>(rc_);                                                                                 \
        B = *reinterpret_cast<const int4*>((rc_) + 4);                                                                           \
        C = *reinterpret_cast<const int4*>((rc_) + 8);                                                                           \
        E0 = *reinterpret_cast<const int4*>((rc_) + 16 + 8 * wp);                                                                \
        E1 = *reinterpret_cast<const int4*>((rc_) + 20 + 8 * wp);                                                                \
    } while (0)
    // the duties of one record, for the unit whose rows start at n_tile
    // (xp_fast: pairs below it take the regular request offsets -- the whole tile lies inside the minibatch and the pair is a full one;
    //  a unit whose tile is ragged has xp_fast = 0)
    auto duties = [&](const int4 rb, const int4 re0, const int4 re1, const unsigned char* xtile, int n_tile, uint32_t xp_fast) {
        const int xp = __builtin_amdgcn_readfirstlane(rb.y);
        if (!X5_NO_XDMA) {
            if ((uint32_t)xp < xp_fast) {
                const uint32_t dst = base_addr + (uint32_t)__builtin_amdgcn_readfirstlane(rb.z) + xdst_w;
                const uint32_t po = (uint32_t)xp * 128u;
                x5_dma2(xtile, vx_e + po, vx_o + po, dst);
            } else if (xp >= 0) {
                const uint32_t dst = base_addr + (uint32_t)__builtin_amdgcn_readfirstlane(rb.z) + xdst_w;
                const bool tail = xp >= npairs_full;
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int row = 8 * (2 * wave + k) + (lane >> 3);
                    const int xr = min(n_tile + row, N - 1) - n_tile;    // rows past N are clamped (never stored)
                    const int piece = (lane & 7) ^ ((row >> 1) & 7);
                    uint32_t voff = (uint32_t)xr * (uint32_t)Cin * 2u + piece * 16 + (uint32_t)xp * 128u;
                    if (tail && (piece & 4)) voff -= 64;                 // last pair of an odd block count: re-read its even half
                    x5_dma1(xtile, voff, dst + k * 1024);
                }
            }
        }
        if (!X5_NO_WDMA) {
            // (entries are packed: the first empty one ends the list)
            const int d0 = __builtin_amdgcn_readfirstlane(re0.x), d1 = __builtin_amdgcn_readfirstlane(re0.z);
            const int d2 = __builtin_amdgcn_readfirstlane(re1.x), d3 = __builtin_amdgcn_readfirstlane(re1.z);
            if (d0 >= 0) {
                x5_dma1(wsel, wvoff + (uint32_t)__builtin_amdgcn_readfirstlane(re0.y), wdst + (uint32_t)d0);
                if (d1 >= 0) {
                    x5_dma1(wsel, wvoff + (uint32_t)__builtin_amdgcn_readfirstlane(re0.w), wdst + (uint32_t)d1);
                    if (d2 >= 0) {
                        x5_dma1(wsel, wvoff + (uint32_t)__builtin_amdgcn_readfirstlane(re1.y), wdst + (uint32_t)d2);
                        if (d3 >= 0) x5_dma1(wsel, wvoff + (uint32_t)__builtin_amdgcn_readfirstlane(re1.w), wdst + (uint32_t)d3);
                    }
                }
            }
        }
    };
    static_assert(X5_P == 4, "the prologue is written out for four records");
#define X5_PROLOGUE(rc_, xtile_, n_tile_, xpf_)                                                                                  \
    do {                                                                                                                         \
        int4 pa, pc, pb0, pe0, pf0, pb1, pe1, pf1, pb2, pe2, pf2, pb3, pe3, pf3;                                                   \
        X5_LOAD_REC(pa, pb0, pc, pe0, pf0, rc_);                                                                                     \
        X5_LOAD_REC(pa, pb1, pc, pe1, pf1, (rc_) + X5_REC);                                                                          \
        X5_LOAD_REC(pa, pb2, pc, pe2, pf2, (rc_) + 2 * X5_REC);                                                                      \
        X5_LOAD_REC(pa, pb3, pc, pe3, pf3, (rc_) + 3 * X5_REC);                                                                      \
        (void)pa; (void)pc;                                                                                                         \
        duties(pb0, pe0, pf0, xtile_, n_tile_, xpf_);                                                                            \
        duties(pb1, pe1, pf1, xtile_, n_tile_, xpf_);                                                                            \
        duties(pb2, pe2, pf2, xtile_, n_tile_, xpf_);                                                                            \
        duties(pb3, pe3, pf3, xtile_, n_tile_, xpf_);                                                                            \
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
        X5_PROLOGUE(rc, xtile, tile * X5_R, (tile * X5_R + X5_R <= N) ? (uint32_t)npairs_full : 0u);
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

        const uint32_t xp_fast = (n_tile + X5_R <= N) ? (uint32_t)npairs_full : 0u;
        int4 cur_a, cur_b, cur_c, cur_e, cur_f;                          // (a group without blocks has no steps: the record behind its prologue is
        X5_LOAD_REC(cur_a, cur_b, cur_c, cur_e, cur_f, recs);            //  the next group's, or the plan's padding record: read, not used)
        uint32_t vp0 = 0, vp1 = 0;                                       // weight fragment pointers: the second block of the step about to run
        const bool late = X5_LATE_DUTIES && hc == 1;
        for (int s = 0; s < nsteps; ++s) {
            const int4 hd = cur_a, rb = cur_b, rn = cur_c, re = cur_e, rf = cur_f;   // hd: pair, slab offset, masks, first weight slots; rn: the next step's
            const uint32_t wn = ((uint32_t)__builtin_amdgcn_readfirstlane(rb.x) >> (8 * wp)) & 31u;
            { X5_T0(); x5_wait_vmcnt(wn); X5_T1(0); }                    // my requests that the NEXT step reads have landed (step 0: and its own)
            { X5_T0(); x5_barrier(); X5_T1(1); }                         // everyone's have; everyone left the previous step
            if (s == 0) {                                                // nobody requested this step's first fragments: do it now
                const uint32_t xs = base_addr + (uint32_t)__builtin_amdgcn_readfirstlane(hd.y);
                const uint32_t ws = (((uint32_t)__builtin_amdgcn_readfirstlane(hd.w) >> (16 * hc)) & 0xffffu) * 2048u;
                vp0 = base_addr + wrd[0] + ws; vp1 = base_addr + wrd[1] + ws;
                x5_first<TRANSW>(vp0, vp1, xs + xo[0][0], xs + xo[0][1], xs + xo[1][0], xs + xo[1][1]);
            }
            if (!late) { X5_T0(); duties(rb, re, rf, xtile, n_tile, xp_fast); X5_T1(2); }
#ifdef X5_STAMPS
            tacc[6] += 1;
#endif
            if (!X5_NO_MATH) {
                X5_T0();
                const uint32_t m = ((uint32_t)__builtin_amdgcn_readfirstlane(hd.z) >> (16 * hc)) & 0xffffu;
                const uint32_t nxs = base_addr + (uint32_t)__builtin_amdgcn_readfirstlane(rn.x);
                const uint32_t nws = (((uint32_t)__builtin_amdgcn_readfirstlane(rn.z) >> (16 * hc)) & 0xffffu) * 2048u;
                x5_step_blocks<DT, TRANSW>(acc, m, vp0, vp1, nxs + xo[0][0], nxs + xo[0][1], nxs + xo[1][0], nxs + xo[1][1],
                                           base_addr + wrd[0] + nws, base_addr + wrd[1] + nws);
                X5_T1(3);
            }
            // the next step's words: requested BEHIND the blocks (a scalar load pending inside them would sit in the counter their waits
            // count with), in flight under the late waves' duties and the way to the barrier
            X5_LOAD_REC(cur_a, cur_b, cur_c, cur_e, cur_f, recs + (size_t)(s + 1) * X5_REC);
            if (late) { X5_T0(); duties(rb, re, rf, xtile, n_tile, xp_fast); X5_T1(2); }
        }

        // ---- unit end: everyone has left the last step -> the ring is free; the next unit's prologue flies while the output is written ----
        x5_barrier();
        int ntile = 0, ngrp = 0;
        const int nxt = decode(unit + gridDim.x, ntile, ngrp);
        if (nxt < nunits) {
            const int32_t* ngh = groups + X5_GROUP * ngrp;
            const int32_t* rc = recs0 + (size_t)__builtin_amdgcn_readfirstlane(ngh[0]) * X5_REC;
            const unsigned char* nxtile = static_cast<const unsigned char*>(uniform_ptr(xt + (size_t)ntile * X5_R * Cin * 2));
            X5_PROLOGUE(rc, nxtile, ntile * X5_R, (ntile * X5_R + X5_R <= N) ? (uint32_t)npairs_full : 0u);
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
