// bsmm_xrows.h -- xprop kernel "a SIMD owns a row quarter" ('BSX5' plans, round 5): feature_axis = 1, bsize 32, 16-bit storage types.
//
// What bounded bsmm_xflow.h (profiles/r04_headline_ab.md): a wave owned an output COLUMN, and the blocks of a column cluster -- inside
// the ring's window of five steps the busiest of the 16 columns has 5-6 blocks where the mean has 2, each block is a serial chain of
// ~1 300 cycles on its wave (256 of them matrix work), and the whole workgroup advances at that wave's pace.
//
// Here the work is cut the other way.  The unit is the same (128 minibatch rows x 16 output blocks, one 16 KiB activation slab per PAIR
// of input blocks, every weight block once through the LDS), but a workgroup is FOUR waves of 512 registers, one per SIMD:
//     wave q owns rows 32 q .. 32 q + 31 of the tile and ALL 16 output blocks of the group: 16 accumulators of 32 x 32 (256 AGPRs).
// Every SIMD multiplies every block of the group by its own 32 rows: the four matrix pipes carry exactly the same work in every step,
// whatever the layout -- balance by construction -- and the four waves run the same instruction stream in step.  A step is a pair of input
// blocks (split when it holds more than X5_CAP blocks); per step a wave reads its four activation fragments once and, for each block of the
// step (a 32-bit mask: bit 2 col + half), the two weight fragments and two MFMAs.
// First version (8 waves = row quarter x column half; source kept as profiles/r05_xrows_v3_kernel_8waves.h.txt, measurements
// profiles/r05_xrows_v1..v3_*): bit-identical, and SLOWER than the flow kernel (105-125 against 77-90 us): all eight waves walk all steps, and
// the two waves of a SIMD do their fixed parts at the same time behind every barrier -- matrix pipe 21 % busy.  This version halves the scalar
// work per step and has no partner to collide with:
//   * the blocks of a step are dispatched by a COMPUTED JUMP (hand-written: s_ff1 on the mask, clear the bit, jump to body 128 B x index;
//     bit 32 = sentinel = the exit) -- 6 scalar instructions per block instead of a chain of 32 tests and two taken branches;
//   * THREE register sets of weight fragments, chosen by code copy (A multiplies from A while B is in flight and C is requested, then
//     jumps into copy B ...): the fragments of a block are requested two blocks ahead -- a lone wave has nobody to hide an LDS round trip;
//   * everything a step reads has landed one barrier EARLY (the plan's waits are for the NEXT step), so the exit of a step requests the next
//     step's activation fragments and first two weight sets: behind the barrier the first MFMA issues at once;
//   * duties (DMA requests) are issued BEHIND the blocks of the step.
// Ring: X5_D = 5 activation slabs + X5_NW = 39 weight slots of 2 KiB (all 160 KiB).  The plan (bsmm_plan.h, 'BSX5') is a list of RECORDS per
// group, one per step, preceded by X5_P duty-only records (the prologue).  One s_barrier per step and nothing else: no counters, no polling.
// The kernel is persistent; the prologue of the NEXT unit is issued behind the last step's barrier and lands while the waves write their
// output (through the one slab slot the prologue does not touch).
// Fragment layouts, swizzles, MFMA operand roles and the order in which a column sums its blocks are those of bsmm_xcol_v2.h / bsmm_xflow.h:
// bit-identical outputs.
#pragma once
#include <type_traits>

#include "bsmm_common.h"
#include "bsmm_plan.h"
#include "bsmm_updat_v2.h"   // uniform_ptr
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
#ifndef X5_LATE_MASK
#define X5_LATE_MASK 15       // waves (bit q) that issue their duties BEHIND the blocks of the step (0: every wave right behind the barrier)
#endif

#ifdef X5_STAMPS
// cycle accounting of the first 64 workgroups (debug builds; s_memtime): per wave [0] from the record request to the wait count (the scalar
// load's latency + loop), [1] vmcnt wait, [2] barrier, [3] first-step fragment requests + early duties, [4] blocks, [5] record request +
// late duties, [6] unit end (barrier, next prologue, epilogue), [7] whole kernel; read back with bsmm_debug_x5_trace_copy()
__device__ unsigned long long g_x5_trace[64 * 4 * 8];
#define X5_STAMP(k) do { const unsigned long long t_ = __builtin_readcyclecounter(); tacc[k] += t_ - tlast; tlast = t_; } while (0)
#else
#define X5_STAMP(k) do { } while (0)
#endif
constexpr int X5_R = 128;                              // minibatch rows per unit
constexpr int X5_SLAB = X5_R * 128;                    // 16 KiB
constexpr int X5_WBASE = X5_D * X5_SLAB;               // weight slots behind the slabs
constexpr int X5_LDS = 163840;                         // slots 0 .. X5_NW - 1, and one guard slot the fragment prefetch may read
constexpr int X5_STAGE = (X5_D - 1) * X5_SLAB;         // epilogue staging: the slab slot the prologue never requests, 4 KiB per wave
static_assert(X5_WBASE + (X5_NW + 1) * 2048 <= X5_LDS, "rows kernel: ring must fit the LDS");

// s_waitcnt vmcnt(n), n wave-uniform at run time (0 .. 63): a computed jump into a table of 64 waits
__device__ __forceinline__ void x5_wait_vmcnt(uint32_t n) {
#define X5_W1(k) "s_waitcnt vmcnt(" #k ")\n\ts_branch 99f\n\t"
#define X5_W8(a, b, c, d, e, f, g, h) X5_W1(a) X5_W1(b) X5_W1(c) X5_W1(d) X5_W1(e) X5_W1(f) X5_W1(g) X5_W1(h)
    asm volatile("s_getpc_b64 s[20:21]\n\t"
                 "s_lshl_b32 s22, %0, 3\n\t"
                 "s_add_u32 s22, s22, 20\n\t"
                 "s_add_u32 s20, s20, s22\n\t"
                 "s_addc_u32 s21, s21, 0\n\t"
                 "s_setpc_b64 s[20:21]\n\t"
                 X5_W8(0, 1, 2, 3, 4, 5, 6, 7) X5_W8(8, 9, 10, 11, 12, 13, 14, 15) X5_W8(16, 17, 18, 19, 20, 21, 22, 23) X5_W8(24, 25, 26, 27, 28, 29, 30, 31)
                 X5_W8(32, 33, 34, 35, 36, 37, 38, 39) X5_W8(40, 41, 42, 43, 44, 45, 46, 47) X5_W8(48, 49, 50, 51, 52, 53, 54, 55)
                 X5_W8(56, 57, 58, 59, 60, 61, 62, 63)
                 "99:"
                 ::"s"(n) : "memory", "scc", "s20", "s21", "s22");
#undef X5_W8
#undef X5_W1
}

// lgkmcnt(0) + s_barrier.  The wait is the BUILTIN (encoding: vmcnt 63, expcnt 7, lgkmcnt 0): the compiler's own counter model must know that
// no LDS read is pending behind it
__device__ __forceinline__ void x5_barrier() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// LDS-DMA of 1 / 2 / 4 consecutive KiB from a scalar base + 32-bit lane offsets.  M0 is NOT saved / restored: nothing else in this kernel reads
// it (no LDS-DMA builtin, no GWS / sendmsg; tests/test_xrows_plan.py audits the kernel's ISA), and the scalar moves are worth having
__device__ __forceinline__ void x5_dma1(const void* sbase, uint32_t voff, uint32_t lds_byte_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_byte_addr) : "memory");
}
__device__ __forceinline__ void x5_dma2(const void* sbase, uint32_t voff0, uint32_t voff1, uint32_t lds_byte_addr) {
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2\n\t"
                 "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                 ::"v"(voff0), "v"(voff1), "s"(sbase), "s"(lds_byte_addr) : "memory", "scc");
}
__device__ __forceinline__ void x5_dma4(const void* sbase, uint32_t v0, uint32_t v1, uint32_t v2, uint32_t v3, uint32_t lds_byte_addr) {
    asm volatile("s_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %4\n\t"
                 "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %4\n\t"
                 "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %4\n\t"
                 "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %4"
                 ::"v"(v0), "v"(v1), "v"(v2), "v"(v3), "s"(sbase), "s"(lds_byte_addr) : "memory", "scc");
}

// ---- the blocks of a step, hand-written.  v[216:255] are named literally and listed as clobbers; the compiler's own code stays far below them
// (the accumulators live in AGPRs; tests/test_xrows_plan.py audits the ISA), so what a step's exit requests for the next step stays in them
// across the barrier and the compiler's code in between:
//   v[216:219] / v[220:223]  activation fragments of the pair's even block (K half 0 / 1),  v[224:227] / v[228:231] of its odd block
//   v[232:239] set A, v[240:247] set B, v[248:255] set C (fragment of K half 0, then of K half 1)
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
// next block: index of the lowest mask bit -> its body in the copy whose base is s[lo:hi]
#define X5_DISPATCH(lo, hi)                                                                                                      \
    "s_ff1_i32_b64 s28, %[m]\n\t"                                                                                                \
    "s_bitset0_b64 %[m], s28\n\t"                                                                                                \
    "s_lshl_b32 s28, s28, 7\n\t"                                                                                                 \
    "s_add_u32 s20, s" X5S(lo) ", s28\n\t"                                                                                       \
    "s_addc_u32 s21, s" X5S(hi) ", 0\n\t"                                                                                        \
    "s_setpc_b64 s[20:21]\n\t"
// body of position p in copy `cp`: cur = first register of the set it multiplies from, pre = of the set it requests the block TWO ahead into,
// (nlo, nhi) = base of the copy that runs next, x0 = first register of the position's activation fragments, W2 = LDS reads of two requests
#define X5_BODY(cp, p, accn, x0, cur, pre, nlo, nhi, RD, W2, MF)                                                                 \
    ".p2align 7\n"                                                                                                               \
    "LX5" cp X5S(p) "_%=:\n\t"                                                                                                    \
    RD(pre)                                                                                                                      \
    "s_waitcnt lgkmcnt(" X5S(W2) ")\n\t"                                                                                         \
    MF " %[" accn "], v[" X5S(cur) ":" X5S(cur) "+3], v[" X5S(x0) ":" X5S(x0) "+3], %[" accn "]\n\t"                                 \
    MF " %[" accn "], v[" X5S(cur) "+4:" X5S(cur) "+7], v[" X5S(x0) "+4:" X5S(x0) "+7], %[" accn "]\n\t"                             \
    X5_DISPATCH(nlo, nhi)
// the four activation fragments of a step and the weight fragments of its first TWO blocks (sets A, B); the pointers move on to the third
#define X5_FIRST(RD, xa0, xa1, xb0, xb1)                                                                                         \
    "ds_read_b128 v[216:219], %[" xa0 "]\n\t"                                                                                    \
    "ds_read_b128 v[220:223], %[" xa1 "]\n\t"                                                                                    \
    "ds_read_b128 v[224:227], %[" xb0 "]\n\t"                                                                                    \
    "ds_read_b128 v[228:231], %[" xb1 "]\n\t"                                                                                    \
    RD(232) RD(240)
#define X5_COPY(cp, cur, pre, nlo, nhi, RD, W2, MF)                                                                              \
    X5_BODY(cp, 0, "a0", 216, cur, pre, nlo, nhi, RD, W2, MF) X5_BODY(cp, 1, "a0", 224, cur, pre, nlo, nhi, RD, W2, MF)          \
    X5_BODY(cp, 2, "a1", 216, cur, pre, nlo, nhi, RD, W2, MF) X5_BODY(cp, 3, "a1", 224, cur, pre, nlo, nhi, RD, W2, MF)          \
    X5_BODY(cp, 4, "a2", 216, cur, pre, nlo, nhi, RD, W2, MF) X5_BODY(cp, 5, "a2", 224, cur, pre, nlo, nhi, RD, W2, MF)          \
    X5_BODY(cp, 6, "a3", 216, cur, pre, nlo, nhi, RD, W2, MF) X5_BODY(cp, 7, "a3", 224, cur, pre, nlo, nhi, RD, W2, MF)          \
    X5_BODY(cp, 8, "a4", 216, cur, pre, nlo, nhi, RD, W2, MF) X5_BODY(cp, 9, "a4", 224, cur, pre, nlo, nhi, RD, W2, MF)          \
    X5_BODY(cp, 10, "a5", 216, cur, pre, nlo, nhi, RD, W2, MF) X5_BODY(cp, 11, "a5", 224, cur, pre, nlo, nhi, RD, W2, MF)        \
    X5_BODY(cp, 12, "a6", 216, cur, pre, nlo, nhi, RD, W2, MF) X5_BODY(cp, 13, "a6", 224, cur, pre, nlo, nhi, RD, W2, MF)        \
    X5_BODY(cp, 14, "a7", 216, cur, pre, nlo, nhi, RD, W2, MF) X5_BODY(cp, 15, "a7", 224, cur, pre, nlo, nhi, RD, W2, MF)        \
    X5_BODY(cp, 16, "a8", 216, cur, pre, nlo, nhi, RD, W2, MF) X5_BODY(cp, 17, "a8", 224, cur, pre, nlo, nhi, RD, W2, MF)        \
    X5_BODY(cp, 18, "a9", 216, cur, pre, nlo, nhi, RD, W2, MF) X5_BODY(cp, 19, "a9", 224, cur, pre, nlo, nhi, RD, W2, MF)        \
    X5_BODY(cp, 20, "a10", 216, cur, pre, nlo, nhi, RD, W2, MF) X5_BODY(cp, 21, "a10", 224, cur, pre, nlo, nhi, RD, W2, MF)      \
    X5_BODY(cp, 22, "a11", 216, cur, pre, nlo, nhi, RD, W2, MF) X5_BODY(cp, 23, "a11", 224, cur, pre, nlo, nhi, RD, W2, MF)      \
    X5_BODY(cp, 24, "a12", 216, cur, pre, nlo, nhi, RD, W2, MF) X5_BODY(cp, 25, "a12", 224, cur, pre, nlo, nhi, RD, W2, MF)      \
    X5_BODY(cp, 26, "a13", 216, cur, pre, nlo, nhi, RD, W2, MF) X5_BODY(cp, 27, "a13", 224, cur, pre, nlo, nhi, RD, W2, MF)      \
    X5_BODY(cp, 28, "a14", 216, cur, pre, nlo, nhi, RD, W2, MF) X5_BODY(cp, 29, "a14", 224, cur, pre, nlo, nhi, RD, W2, MF)      \
    X5_BODY(cp, 30, "a15", 216, cur, pre, nlo, nhi, RD, W2, MF) X5_BODY(cp, 31, "a15", 224, cur, pre, nlo, nhi, RD, W2, MF)      \
    /* exit (position 32): the step's blocks are issued -- request the NEXT step's first fragments (its data landed at this step's barrier;   \
       behind the unit's last step: addresses of slot 0, read and never used).  Requests of the last blocks may still be in flight into the   \
       same sets: LDS reads of a wave return in order, the later one wins. */                                                                \
    ".p2align 7\n"                                                                                                               \
    "LX5" cp "32_%=:\n\t"                                                                                                        \
    "v_mov_b32 %[vp0], %[nvp0]\n\t"                                                                                              \
    "v_mov_b32 %[vp1], %[nvp1]\n\t"                                                                                              \
    X5_FIRST(RD, "nxa0", "nxa1", "nxb0", "nxb1")                                                                                 \
    "s_branch LX5END_%=\n"
#define X5_STEP_ASM(RD, W2, MF)                                                                                                  \
    "s_getpc_b64 s[20:21]\n"                                                                                                     \
    "LX5REF_%=:\n\t"                                                                                                             \
    "s_add_u32 s22, s20, LX5A0_%=-LX5REF_%=\n\t"                                                                                 \
    "s_addc_u32 s23, s21, 0\n\t"                                                                                                 \
    "s_add_u32 s24, s20, LX5B0_%=-LX5REF_%=\n\t"                                                                                 \
    "s_addc_u32 s25, s21, 0\n\t"                                                                                                 \
    "s_add_u32 s26, s20, LX5C0_%=-LX5REF_%=\n\t"                                                                                 \
    "s_addc_u32 s27, s21, 0\n\t"                                                                                                 \
    X5_DISPATCH(22, 23)                                                                                                          \
    X5_COPY("A", 232, 248, 24, 25, RD, W2, MF)                                                                                   \
    X5_COPY("B", 240, 232, 26, 27, RD, W2, MF)                                                                                   \
    X5_COPY("C", 248, 240, 22, 23, RD, W2, MF)                                                                                   \
    "LX5END_%=:\n\t"

#define X5_REGS_CLOBBERED                                                                                                        \
    "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223", "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231", "v232",  \
    "v233", "v234", "v235", "v236", "v237", "v238", "v239", "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247", "v248", "v249",  \
    "v250", "v251", "v252", "v253", "v254", "v255"

// the blocks of one step (mask m; the fragments of its first two blocks are already requested: by the previous step's exit or by x5_first),
// then the request for the next step's first fragments (nx*: their LDS addresses).  vp0 / vp1 enter pointing at the step's THIRD block and
// leave pointing at the next step's third block.
template <class DT, bool TRANSW>
__device__ __forceinline__ void x5_step_blocks(f32x16 (&acc)[16], uint32_t m32, uint32_t& vp0, uint32_t& vp1, uint32_t nxa0, uint32_t nxa1, uint32_t nxb0,
                                               uint32_t nxb1, uint32_t nvp0, uint32_t nvp1) {
    uint64_t m = (uint64_t)m32 | (1ull << 32);              // the sentinel: position 32 = the exit body
#define X5_ASM_OPERANDS                                                                                                          \
    : [a0] "+a"(acc[0]), [a1] "+a"(acc[1]), [a2] "+a"(acc[2]), [a3] "+a"(acc[3]), [a4] "+a"(acc[4]), [a5] "+a"(acc[5]), [a6] "+a"(acc[6]),  \
      [a7] "+a"(acc[7]), [a8] "+a"(acc[8]), [a9] "+a"(acc[9]), [a10] "+a"(acc[10]), [a11] "+a"(acc[11]), [a12] "+a"(acc[12]),              \
      [a13] "+a"(acc[13]), [a14] "+a"(acc[14]), [a15] "+a"(acc[15]), [m] "+s"(m), [vp0] "+v"(vp0), [vp1] "+v"(vp1)                        \
    : [nxa0] "v"(nxa0), [nxa1] "v"(nxa1), [nxb0] "v"(nxb0), [nxb1] "v"(nxb1), [nvp0] "v"(nvp0), [nvp1] "v"(nvp1)                 \
    : "memory", "scc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", X5_REGS_CLOBBERED
    if constexpr (std::is_same<DT, DTbf16>::value) {
        if constexpr (TRANSW) asm volatile(X5_STEP_ASM(X5_RD_T, 8, "v_mfma_f32_32x32x16_bf16") X5_ASM_OPERANDS);
        else                  asm volatile(X5_STEP_ASM(X5_RD_N, 4, "v_mfma_f32_32x32x16_bf16") X5_ASM_OPERANDS);
    } else {
        if constexpr (TRANSW) asm volatile(X5_STEP_ASM(X5_RD_T, 8, "v_mfma_f32_32x32x16_f16") X5_ASM_OPERANDS);
        else                  asm volatile(X5_STEP_ASM(X5_RD_N, 4, "v_mfma_f32_32x32x16_f16") X5_ASM_OPERANDS);
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
__global__ void __launch_bounds__(256, 1)
xrows32_kernel(const typename DT::T* __restrict__ X, const typename DT::T* __restrict__ Wsel, typename DT::T* __restrict__ Y,
               const int32_t* __restrict__ plan, XMap map, int N, int Cin, int Kout) {
    typedef typename DT::T T;
    static_assert(DT::is16, "rows kernel: 16-bit storage types");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int q = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // wave = row quarter
    const int r = lane & 31, h = lane >> 5;
    const uint32_t base_addr = lds_addr_of(smem);

    const int npairs_full = Cin / 64;
    const unsigned char* xt = reinterpret_cast<const unsigned char*>(X);
    const unsigned char* wsel = static_cast<const unsigned char*>(uniform_ptr(Wsel));
    // weight DMA (bsmm_xcol_v2.h): lane i of an instruction writes piece i of a 1 KiB half block; it fetches the piece that the read
    // swizzle expects there (none for the transposing reads of fprop)
    const uint32_t wvoff = TRANSW ? (uint32_t)lane * 16u : (uint32_t)((lane >> 2) * 64 + (((lane & 3) ^ ((lane >> 4) & 3)) << 4));
    // fragment read addresses.  Activations: row 32 q + r of the slab, 16-byte piece (2 kk + h + 4 half) ^ ((row >> 1) & 7); weights: relative
    // to the block's slot
    const int xsw = (r >> 1) & 7;
    uint32_t xo[2][2], wrd[2];
#pragma unroll
    for (int ab = 0; ab < 2; ++ab)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) xo[ab][kk] = base_addr + (uint32_t)((32 * q + r) * 128 + (((2 * kk + h + 4 * ab) ^ xsw) << 4));
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        if constexpr (TRANSW) {
            const int g16 = lane >> 4, t16 = lane & 15;
            wrd[kk] = base_addr + X5_WBASE + (16 * kk + 8 * h + (t16 >> 2)) * 64 + (16 * (g16 & 1) + 4 * (t16 & 3)) * 2;
        } else {
            wrd[kk] = base_addr + X5_WBASE + r * 64 + (((2 * kk + h) ^ ((r >> 2) & 3)) << 4);
        }
    }
    // slab request offsets of a full tile: DMA instruction ii covers rows 8 ii + (lane >> 3); the 16-byte piece a lane fetches is
    // (lane & 7) ^ ((row >> 1) & 7) = (lane & 7) ^ (lane >> 4) ^ (4 (ii & 1)).  This wave issues ii = 4 q .. 4 q + 3: its OWN 32 rows.
    const uint32_t stride16 = (uint32_t)Cin * 16u;                       // bytes between the first rows of consecutive instructions
    const uint32_t pc_e = (uint32_t)((lane & 7) ^ (lane >> 4));
    const uint32_t vo_e0 = (uint32_t)(lane >> 3) * (uint32_t)Cin * 2u + pc_e * 16u;
    const uint32_t vo_o0 = (pc_e & 4u) ? vo_e0 - 64u : vo_e0 + 64u;
    const uint32_t vx0 = vo_e0 + (uint32_t)(4 * q) * stride16, vx1 = vo_o0 + (uint32_t)(4 * q + 1) * stride16;
    const uint32_t vx2 = vo_e0 + (uint32_t)(4 * q + 2) * stride16, vx3 = vo_o0 + (uint32_t)(4 * q + 3) * stride16;
    const uint32_t xdst_w = base_addr + (uint32_t)(4 * q) * 1024u;       // my four instructions' place inside a slab

    const int nunits = map.grid();
    const int32_t* const groups = plan + plan[5];
    const int32_t* const recs0 = plan + plan[6];

    // the words of a record a wave needs: [0..3] pair / slab offset / mask / first weight slot, [4..7] waits / slab duty (pair, offset) / -,
    // [8..11] the next step's slab offset / mask / first weight slot, and its own four fetch entries (two words each): scalar loads
#define X5_LOAD_REC(A, B, C, E0, E1, rc_)                                                                                        \
    do {                                                                                                                         \
        A = *reinterpret_cast<const int4*>(rc_);                                                                                 \
        B = *reinterpret_cast<const int4*>((rc_) + 4);                                                                           \
        C = *reinterpret_cast<const int4*>((rc_) + 8);                                                                           \
        E0 = *reinterpret_cast<const int4*>((rc_) + 16 + 8 * q);                                                                 \
        E1 = *reinterpret_cast<const int4*>((rc_) + 20 + 8 * q);                                                                 \
    } while (0)
    // the duties of one record, for the unit whose rows start at n_tile (xp_fast: pairs below it take the regular request offsets -- the whole
    // tile lies inside the minibatch and the pair is a full one; a unit whose tile is ragged has xp_fast = 0)
    auto duties = [&](const int4 rb, const int4 re0, const int4 re1, const unsigned char* xtile, int n_tile, uint32_t xp_fast) {
        const int xp = __builtin_amdgcn_readfirstlane(rb.y);
        if (!X5_NO_XDMA) {
            if ((uint32_t)xp < xp_fast) {
                const uint32_t dst = xdst_w + (uint32_t)__builtin_amdgcn_readfirstlane(rb.z);
                const uint32_t po = (uint32_t)xp * 128u;
                x5_dma4(xtile, vx0 + po, vx1 + po, vx2 + po, vx3 + po, dst);
            } else if (xp >= 0) {
                const uint32_t dst = xdst_w + (uint32_t)__builtin_amdgcn_readfirstlane(rb.z);
                const bool tail = xp >= npairs_full;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int row = 8 * (4 * q + k) + (lane >> 3);
                    const int xr = min(n_tile + row, N - 1) - n_tile;    // rows past N are clamped (never stored)
                    const int piece = (lane & 7) ^ ((row >> 1) & 7);
                    uint32_t voff = (uint32_t)xr * (uint32_t)Cin * 2u + piece * 16 + (uint32_t)xp * 128u;
                    if (tail && (piece & 4)) voff -= 64;                 // last pair of an odd block count: re-read its even half
                    x5_dma1(xtile, voff, dst + k * 1024);
                }
            }
        }
        if (!X5_NO_WDMA) {
            // (entries are packed: the first empty one ends the list; an entry = a whole weight block = two instructions)
            const int d0 = __builtin_amdgcn_readfirstlane(re0.x), d1 = __builtin_amdgcn_readfirstlane(re0.z);
            const int d2 = __builtin_amdgcn_readfirstlane(re1.x), d3 = __builtin_amdgcn_readfirstlane(re1.z);
            if (d0 >= 0) {
                const uint32_t o0 = wvoff + (uint32_t)__builtin_amdgcn_readfirstlane(re0.y);
                x5_dma2(wsel, o0, o0 + 1024u, base_addr + (uint32_t)d0);
                if (d1 >= 0) {
                    const uint32_t o1 = wvoff + (uint32_t)__builtin_amdgcn_readfirstlane(re0.w);
                    x5_dma2(wsel, o1, o1 + 1024u, base_addr + (uint32_t)d1);
                    if (d2 >= 0) {
                        const uint32_t o2 = wvoff + (uint32_t)__builtin_amdgcn_readfirstlane(re1.y);
                        x5_dma2(wsel, o2, o2 + 1024u, base_addr + (uint32_t)d2);
                        if (d3 >= 0) {
                            const uint32_t o3 = wvoff + (uint32_t)__builtin_amdgcn_readfirstlane(re1.w);
                            x5_dma2(wsel, o3, o3 + 1024u, base_addr + (uint32_t)d3);
                        }
                    }
                }
            }
        }
    };
    static_assert(X5_P == 4, "the prologue is written out for four records");
#define X5_PROLOGUE(rc_, xtile_, n_tile_, xpf_)                                                                                  \
    do {                                                                                                                         \
        int4 pa, pc, pb0, pe0, pf0, pb1, pe1, pf1, pb2, pe2, pf2, pb3, pe3, pf3;                                                 \
        X5_LOAD_REC(pa, pb0, pc, pe0, pf0, rc_);                                                                                 \
        X5_LOAD_REC(pa, pb1, pc, pe1, pf1, (rc_) + X5_REC);                                                                      \
        X5_LOAD_REC(pa, pb2, pc, pe2, pf2, (rc_) + 2 * X5_REC);                                                                  \
        X5_LOAD_REC(pa, pb3, pc, pe3, pf3, (rc_) + 3 * X5_REC);                                                                  \
        (void)pa; (void)pc;                                                                                                      \
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
    f32x16 acc[16];
#pragma unroll
    for (int t = 0; t < 16; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

    const bool late = ((X5_LATE_MASK >> q) & 1) != 0;
#ifdef X5_STAMPS
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long tstart = __builtin_readcyclecounter();
    unsigned long long tlast = tstart;
#endif
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
        uint32_t vp0 = 0, vp1 = 0;                                       // weight fragment pointers: the third block of the step about to run
        for (int s = 0; s < nsteps; ++s) {
            const int4 hd = cur_a, rb = cur_b, rn = cur_c, re = cur_e, rf = cur_f;   // hd: pair, slab offset, mask, first weight slot; rn: the next step's
            const uint32_t wn = ((uint32_t)__builtin_amdgcn_readfirstlane(rb.x) >> (8 * q)) & 63u;
#ifdef X5_STAMPS
            asm volatile("" ::"s"(wn));
#endif
            X5_STAMP(0);
            x5_wait_vmcnt(wn);                                           // my requests that the NEXT step reads have landed (step 0: and its own)
            X5_STAMP(1);
            x5_barrier();                                                // everyone's have; everyone left the previous step
            X5_STAMP(2);
            if (s == 0) {                                                // nobody requested this step's first fragments: do it now
                const uint32_t xs = (uint32_t)__builtin_amdgcn_readfirstlane(hd.y);
                const uint32_t ws = (uint32_t)__builtin_amdgcn_readfirstlane(hd.w) * 2048u;
                vp0 = wrd[0] + ws; vp1 = wrd[1] + ws;
                x5_first<TRANSW>(vp0, vp1, xs + xo[0][0], xs + xo[0][1], xs + xo[1][0], xs + xo[1][1]);
            }
            if (!late) duties(rb, re, rf, xtile, n_tile, xp_fast);
            X5_STAMP(3);
            if (!X5_NO_MATH) {
                const uint32_t m = (uint32_t)__builtin_amdgcn_readfirstlane(hd.z);
                const uint32_t nxs = (uint32_t)__builtin_amdgcn_readfirstlane(rn.x);
                const uint32_t nws = (uint32_t)__builtin_amdgcn_readfirstlane(rn.z) * 2048u;
                x5_step_blocks<DT, TRANSW>(acc, m, vp0, vp1, nxs + xo[0][0], nxs + xo[0][1], nxs + xo[1][0], nxs + xo[1][1], wrd[0] + nws, wrd[1] + nws);
            }
            // the next step's words: requested BEHIND the blocks (a scalar load pending inside them would sit in the counter their waits
            // count with), in flight under the duties and the way to the barrier
            X5_STAMP(4);
            X5_LOAD_REC(cur_a, cur_b, cur_c, cur_e, cur_f, recs + (size_t)(s + 1) * X5_REC);
            if (late) duties(rb, re, rf, xtile, n_tile, xp_fast);
            X5_STAMP(5);
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
        // Epilogue, per wave, through its 4 KiB of the free slab slot, two output blocks at a time: D[o][n] with col n = r (minibatch row
        // 32 q + r), rows o = (reg & 3) + 8 (reg >> 2) + 4 h.  [32 rows n][128 B], the eight 16-byte pieces of row n XOR-swizzled with
        // (n >> 1) & 7; read back as full 128-byte rows and stored (8 rows per instruction).
        {
            unsigned char* stage = smem + X5_STAGE + q * 4096;
#pragma unroll
            for (int kp = 0; kp < 8; ++kp) {
                if (2 * kp < nob) {
                    unsigned char* ybase = reinterpret_cast<unsigned char*>(Y + (size_t)(ob0 + 2 * kp) * 32);
#pragma unroll
                    for (int cc = 0; cc < 2; ++cc)
#pragma unroll
                        for (int qd = 0; qd < 4; ++qd) {
                            const f32x16& a = acc[2 * kp + cc];
                            const uint32_t lo = (uint32_t)DT::from_f32(a[4 * qd + 0]) | ((uint32_t)DT::from_f32(a[4 * qd + 1]) << 16);
                            const uint32_t hi = (uint32_t)DT::from_f32(a[4 * qd + 2]) | ((uint32_t)DT::from_f32(a[4 * qd + 3]) << 16);
                            *reinterpret_cast<uint2*>(stage + r * 128 + (((4 * cc + qd) ^ ((r >> 1) & 7)) << 4) + 8 * h) = make_uint2(lo, hi);
                        }
                    asm volatile("" ::: "memory");
                    const bool both = 2 * kp + 1 < nob;                  // (the group's last output block may be the first of a pair)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int n = 8 * i + (lane >> 3), pc = lane & 7;
                        const uint4 v = *reinterpret_cast<const uint4*>(stage + n * 128 + ((pc ^ ((n >> 1) & 7)) << 4));
                        const int gn = n_tile + 32 * q + n;
                        if (gn < N && (both || pc < 4)) *reinterpret_cast<uint4*>(ybase + (size_t)gn * Kout * 2 + pc * 16) = v;
                    }
                    asm volatile("" ::: "memory");
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) { acc[2 * kp][i] = 0.f; acc[2 * kp + 1][i] = 0.f; }
            }
        }
        unit = nxt; tile = ntile; grp = ngrp;
        X5_STAMP(6);
    }
#ifdef X5_STAMPS
    tacc[7] = __builtin_readcyclecounter() - tstart;
    if (blockIdx.x < 64 && lane == 0)
        for (int k = 0; k < 8; ++k) g_x5_trace[(blockIdx.x * 4 + q) * 8 + k] = tacc[k];
#endif
}

}  // namespace bsmm
