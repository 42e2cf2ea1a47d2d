// bsmm_xcol.h -- xprop kernel "wave owns an output column", feature_axis = 1, bsize 32, 16-bit storage types.
//
// Lesson from the grouped kernels (profiles/): when every wave carries all G accumulators of a group, each
// 32x32 block costs a dispatch decision, two LDS fragment reads and their latency for only 2 MFMAs -- the control
// skeleton, not a bandwidth, bounds the kernel (~15% of the MFMA rate).  Here the ownership is transposed:
//   workgroup = G (16; narrow variant 8) consecutive output blocks x XC_R (128) minibatch rows, G waves;
//   wave v owns output block v for ALL 128 rows: 4 row tiles x 16 accumulator registers, statically addressed.
// Per step (one PAIR of input blocks = one 128-byte line per row) the X slab [128 rows x 128 B] is DMA'd once into an
// LDS ring and shared by all waves; a wave whose column has a block in either half of the pair loads that
// block's fragment straight from global memory (prefetched one step ahead) and issues 8 MFMAs with it; waves without
// a block skip to the barrier (one per PHASE of PH steps).  One decision per step per wave, ~8 LDS reads per 8 MFMAs.
//   slab image: rows of 128 B, the eight 16-byte pieces of row r XOR-swizzled with (r >> 1) & 7 (conflict-free b128).
//
// Measured (4096^2, 20%, N = 8192, bf16; kernel-trace ablation, profiles/): 133 us total; with the X DMA removed 114,
// W loads removed 117, MFMA work removed 103, barriers removed 114, all four removed 55 (= ~22 us of Y stores +
// ~30 us of per-step bookkeeping): no single resource bounds it, the pieces just overlap poorly.  T(density) is about
// 70 us + 260 us * density; at 100% density it runs 837 TF (hipBLASLt dense GEMM of the same shape: 1090 TF).
#pragma once
#include "bsmm_common.h"
#include "bsmm_plan.h"
#include "bsmm_xprop.h"   // XMap
#include "bsmm_updat_tr.h"

namespace bsmm {

#ifndef BSMM_XC_RT
#define BSMM_XC_RT 4
#endif
#ifndef BSMM_XC_PH
#define BSMM_XC_PH 2
#endif
constexpr int XC_RT = BSMM_XC_RT;         // 32-row tiles per wave
constexpr int XC_PH = BSMM_XC_PH;         // steps per phase (one barrier per phase); ring = 2*XC_PH slabs
constexpr int XC_R = 32 * XC_RT;          // minibatch rows per workgroup
constexpr int XC_SLAB = XC_R * 128;       // one pair step of 16-bit activations: 128 rows x 128 B = 16 KiB
constexpr int XC_RING = 2 * XC_PH;

// TRANSW = true (fprop): Wsel is W in its natural [c-in-block][k-in-block] layout and each lane gathers its fragment
// transposed (16 two-byte loads, stride 64 B, prefetched a step ahead) -- no transposed copy of W, no workspace.
// Measured slower than the transpose pre-pass (fprop 145 vs 131 us), so the launcher passes TRANSW = false today.
// launch_bounds(512, 4) = at most 128 VGPRs: with 64 KiB of LDS that is what lets TWO workgroups share a CU, and the kernel sits
// exactly on that edge (128 registers, no spills).  Variants that need a few registers more (wave priority around the MFMAs:
// 152; in-kernel transposed weight gather: 131; deeper read-ahead) drop to one workgroup per CU and from ~120 to ~160 us --
// most "slower" results of this file's experiments were this cliff, not the idea itself.
#ifndef BSMM_XC_OCC
#define BSMM_XC_OCC 4
#endif
#ifdef BSMM_XC_TRACE
// cycle stamps of the first 8 workgroups: [wg][wave][phase][6] = before wait, after wait, after barrier, after DMA issue,
// after step 0, after step 1 (s_memtime); read back with bsmm_debug_trace_copy().  Debug builds only.
__device__ unsigned long long g_xc_trace[8 * 8 * 40 * 6];
#define XC_STAMP(k) do { if (blockIdx.x < 8 && (s / XC_PH) < 40 && lane == 0) g_xc_trace[((blockIdx.x * 8 + wave) * 40 + (s / XC_PH)) * 6 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define XC_STAMP(k) do { } while (0)
#endif
// G = output blocks (= waves) per workgroup, PH = steps per phase.  <8, 2>: 512 threads, 64 KiB, two workgroups per CU.
// <16, 4> ("wide"): 1024 threads, one workgroup per CU with the same 16 resident waves, but ONE slab feeds 16 columns --
// half the L2->LDS slab traffic per MFMA -- and the 128 KiB ring holds phases of four steps (half the barriers).
constexpr int xc_lds_bytes(int g, int ph) { return (2 * ph * XC_SLAB > XC_R * g * 64) ? 2 * ph * XC_SLAB : XC_R * g * 64; }

template <class DT, bool TRANSW, int G = XC_G, int PH = XC_PH>
__global__ void __launch_bounds__(64 * G, BSMM_XC_OCC)
xcol32_a1_kernel(const typename DT::T* __restrict__ X, const typename DT::T* __restrict__ Wsel,
                 typename DT::T* __restrict__ Y, const int32_t* __restrict__ plan, XMap map, int N, int Cin, int Kout) {
    typedef typename DT::T T;
    static_assert(DT::is16, "xcol kernel: 16-bit storage types");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int tile, grp;
    if (!xmap_decode(map, blockIdx.x, tile, grp)) return;
    if (plan[0] != XCPLAN_MAGIC || plan[1] != XCPLAN_VERSION || plan[2] != G) return;
    const int4 gh = *reinterpret_cast<const int4*>(plan + plan[5] + 4 * grp);
    const int step_off = gh.x, nsteps = gh.y, ob0 = gh.z, nob = gh.w;
    const int32_t* pairs = plan + plan[6] + step_off;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int NI = XC_SLAB / 1024 / G;      // DMA instructions per wave per slab
    constexpr int RING = 2 * PH;
    constexpr int ROWB = G * 64;                // bytes per row of the epilogue staging tile
    static_assert(NI >= 1 && 64 % RING == 0, "slab split / ring must divide the 64-step table batch");
    const int32_t* wt0 = plan + plan[7] + 2 * G * step_off + (2 * wave) * nsteps;   // my column, even half
    const int32_t* wt1 = wt0 + nsteps;                                                  // odd half
    const int r = lane & 31, h = lane >> 5;
    const int n_tile = tile * XC_R;

    // X DMA: slab = 32 instructions of 1 KiB (8 rows each); wave v issues instructions 4v .. 4v+3
    const uint32_t base_addr = lds_addr_of(smem);
    const int npairs_full = Cin / 64;
    const T* xsrc[NI];
    int oddsub[2];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int row = 8 * (NI * wave + i) + (lane >> 3);
        const int xr = min(n_tile + row, N - 1);                     // rows past N are clamped (never stored)
        const int piece = (lane & 7) ^ ((row >> 1) & 7);
        xsrc[i] = X + (size_t)xr * Cin + piece * 8;
        if (i < 2) oddsub[i] = (piece & 4) ? 32 : 0;                 // piece pattern repeats every 2 instructions
    }
    auto issue_x = [&](int p, int pos) {
        const bool full = p < npairs_full;
#pragma unroll
        for (int i = 0; i < NI; ++i)
            glds16_asm(xsrc[i] + (p * 64 - (full ? 0 : oddsub[i & 1])),
                       __builtin_amdgcn_readfirstlane(base_addr + pos * XC_SLAB + (NI * wave + i) * 1024));
    };
    // fragment read offsets inside a 32-row band of the slab: piece = 4*half + 2*kk + h
    const int xsw = (r >> 1) & 7;
    int xrd[2][2];
#pragma unroll
    for (int half = 0; half < 2; ++half)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) xrd[half][kk] = r * 128 + (((4 * half + 2 * kk + h) ^ xsw) << 4);

    f32x16 acc[XC_RT];
#pragma unroll
    for (int t = 0; t < XC_RT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

    auto load_w = [&](int w, Frag32<DT>& f) {
        if (w >= 0) {
            if constexpr (TRANSW) f.load_strided(Wsel + (size_t)w * 1024 + r, 32, h);   // Wop[o][i] = W[i][o]
            else                  f.load_contig(Wsel + (size_t)w * 1024 + r * 32, h);
        }
    };
    auto block = [&](const Frag32<DT>& wf, const unsigned char* slab, int half) {
        // all X fragments first, then the MFMAs with the two K-halves of one accumulator XC_RT instructions apart
        // hipcc schedules this as read-2 / wait / 2 MFMAs.  Forcing all eight reads first (sched_barrier)
        // measured SLOWER (160 vs 122 us): the eight waves then hit the LDS in one burst and nobody has MFMA work meanwhile.
        Frag32<DT> xf[XC_RT];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int t = 0; t < XC_RT; ++t) xf[t].q[kk] = *reinterpret_cast<const uint4*>(slab + t * 4096 + xrd[half][kk]);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int t = 0; t < XC_RT; ++t) acc[t] = DT::mfma32(wf.q[kk], xf[t].q[kk], acc[t]);
    };

    // Phases of PH steps, one barrier per phase, ring of 2*PH slabs.  At the phase barrier both slabs of the phase have
    // landed (each wave waited for its DMA share, issued a whole phase earlier) and everyone has left the previous
    // phase, so the two slabs of the NEXT phase are requested right away (prefetch distance = one phase); a wave's W
    // fragments are still fetched one step ahead.  Half the barriers, and the per-step imbalance between waves
    // (0, 1 or 2 blocks) averages over two steps.  (Deeper rings with counted waits were measured slower: registers.)
    // Cycle stamps (-DBSMM_XC_TRACE, scripts/gpu_xc_trace.py, profiles/r01_xcol_phase_stamps.txt): of a 3956-cycle phase a wave
    // spends 361 in the vmcnt(0), 757 at the barrier, 635 issuing its 4 slab DMAs and ~1030 in each step.  Requesting the
    // second step's weight fragments BEFORE the DMAs (asm loads into a second register set, vmcnt(8) instead of a drain)
    // was measured slower, 141 vs 120 us.
    const bool owner = wave < nob;
    if (nsteps > 0) {
        for (int tb = 0; tb < nsteps; tb += 64) {     // lane-indexed tables for steps [tb, tb+64)
            const int idx = min(tb + lane, nsteps - 1);
            const int pv = pairs[idx];
            const int w0v = owner ? wt0[idx] : -1, w1v = owner ? wt1[idx] : -1;
            const int tend = min(64, nsteps - tb);    // steps in this batch (64 % RING == 0: slot = step % RING)
            Frag32<DT> wc0, wc1, wn0, wn1;
            wc0.zero(); wc1.zero(); wn0.zero(); wn1.zero();
            int c0 = __builtin_amdgcn_readlane(w0v, 0), c1 = __builtin_amdgcn_readlane(w1v, 0);
            load_w(c0, wc0);
            load_w(c1, wc1);
#pragma unroll
            for (int u = 0; u < PH; ++u)
                if (u < tend) issue_x(__builtin_amdgcn_readlane(pv, u), u);
            for (int s = 0; s < tend; s += PH) {
                XC_STAMP(0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // my shares of this phase's slabs (+ my W fragments) landed
                XC_STAMP(1);
                __syncthreads();                                  // everyone's did; everyone left the previous phase
                XC_STAMP(2);
#pragma unroll
                for (int u = 0; u < PH; ++u)
                    if (s + PH + u < tend) issue_x(__builtin_amdgcn_readlane(pv, s + PH + u), (s + PH + u) % RING);
                XC_STAMP(3);
#pragma unroll
                for (int u = 0; u < PH; ++u) {
                    const int ss = s + u;
                    if (ss >= tend) break;
                    if (u >= 1) { XC_STAMP(4); }
                    if (u >= 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // W fragments of this step (slab is already in)
                    int n0 = -1, n1 = -1;
                    if (ss + 1 < tend) {
                        n0 = __builtin_amdgcn_readlane(w0v, ss + 1);
                        n1 = __builtin_amdgcn_readlane(w1v, ss + 1);
                        load_w(n0, wn0);
                        load_w(n1, wn1);
                    }
                    const unsigned char* slab = smem + (ss % RING) * XC_SLAB;
                    if (c0 >= 0) block(wc0, slab, 0);
                    if (c1 >= 0) block(wc1, slab, 1);
                    wc0 = wn0; wc1 = wn1; c0 = n0; c1 = n1;
                }
                XC_STAMP(5);
            }
            __syncthreads();   // the next batch re-primes slots 0/1
        }
    }

    // Epilogue.  D[o][n]: col = n = r (lane), rows o = (reg & 3) + 8 * (reg >> 2) + 4h, i.e. 4 consecutive o (8 bytes)
    // per register quad.  The 8 waves own 8 ADJACENT output blocks = 512 contiguous bytes per minibatch row, so the
    // tile goes through the (now idle) LDS ring and is stored as full rows: 16 bytes per lane, a wave covers 2 rows x
    // 512 B per instruction (direct 8-byte stores at an 8 KiB stride cost ~22 us for the 67 MB; this costs ~13).
    // staging image: [XC_R rows][512 B], 16-byte pieces of row n XOR-swizzled with (n & 31) (bank-conflict free both
    // ways: writers walk n across lanes, readers walk pieces across lanes).
    static_assert(XC_R * ROWB <= xc_lds_bytes(G, PH), "staging tile must fit");
    __syncthreads();   // everyone is done with the slabs
    if (owner) {
#pragma unroll
        for (int t = 0; t < XC_RT; ++t) {
            const int n = t * 32 + r;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                uint32_t lo = (uint32_t)DT::from_f32(acc[t][4 * q + 0]) | ((uint32_t)DT::from_f32(acc[t][4 * q + 1]) << 16);
                uint32_t hi = (uint32_t)DT::from_f32(acc[t][4 * q + 2]) | ((uint32_t)DT::from_f32(acc[t][4 * q + 3]) << 16);
                const int piece = wave * 4 + q;                     // 16-byte piece inside the 512-byte row: o = 8q + 4h + ...
                *reinterpret_cast<uint2*>(smem + n * ROWB + ((piece ^ (n & 31)) << 4) + 8 * h) = make_uint2(lo, hi);
            }
        }
    }
    __syncthreads();
    {
        const int rowbytes = nob * 64;                              // partial last group: fewer valid columns
        T* ybase = Y + (size_t)ob0 * 32;
        constexpr int PPR = ROWB / 16;                              // 16-byte pieces per row
        for (int i = threadIdx.x; i < XC_R * PPR; i += 64 * G) {
            const int n = i / PPR, piece = i % PPR;
            if (n_tile + n < N && piece * 16 < rowbytes) {
                const uint4 v = *reinterpret_cast<const uint4*>(smem + n * ROWB + ((piece ^ (n & 31)) << 4));
                *reinterpret_cast<uint4*>(reinterpret_cast<unsigned char*>(ybase + (size_t)(n_tile + n) * Kout) + piece * 16) = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// feature_axis = 0 variant: activations (C, N), minibatch contiguous.  The X slab of a pair step is [64 feature rows] x
// [XC_R minibatch columns] (256 B per row, 16-byte pieces XOR-swizzled with 4*(row & 3)); the MFMA B operand wants, per
// lane, 8 consecutive FEATURES of one minibatch column, i.e. a column of the slab: built with ds_read_b64_tr_b16.
// Output rows are features, so stores are 64-byte row segments.  Requires N % 8 == 0 (16-byte aligned row pieces).
// ------------------------------------------------------------------------------------------------------------------
constexpr int XC0_ROWB = XC_R * 2;                 // bytes per slab row
constexpr int XC0_SLAB = 64 * XC0_ROWB;            // 16 KiB
constexpr int XC0_PPR = XC0_ROWB / 16;             // 16-byte pieces per row
constexpr int XC0_RPI = 1024 / XC0_ROWB;           // rows per DMA instruction

template <class DT, bool TRANSW, int G = XC_G, int PH = XC_PH>      // <8, 2> or the wide <16, 4> (see xcol32_a1_kernel)
__global__ void __launch_bounds__(64 * G, BSMM_XC_OCC)
xcol32_a0_kernel(const typename DT::T* __restrict__ X, const typename DT::T* __restrict__ Wsel,
                 typename DT::T* __restrict__ Y, const int32_t* __restrict__ plan, XMap map, int N, int Cin, int Kout) {
    typedef typename DT::T T;
    static_assert(DT::is16, "xcol kernel: 16-bit storage types");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int tile, grp;
    if (!xmap_decode(map, blockIdx.x, tile, grp)) return;
    if (plan[0] != XCPLAN_MAGIC || plan[1] != XCPLAN_VERSION || plan[2] != G) return;
    const int4 gh = *reinterpret_cast<const int4*>(plan + plan[5] + 4 * grp);
    const int step_off = gh.x, nsteps = gh.y, ob0 = gh.z, nob = gh.w;
    const int32_t* pairs = plan + plan[6] + step_off;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int NI = XC0_SLAB / 1024 / G;     // DMA instructions per wave per slab
    constexpr int RING = 2 * PH;
    static_assert(NI >= 1 && 64 % RING == 0, "slab split / ring must divide the 64-step table batch");
    const int32_t* wt0 = plan + plan[7] + 2 * G * step_off + (2 * wave) * nsteps;
    const int32_t* wt1 = wt0 + nsteps;
    const int r = lane & 31, h = lane >> 5;
    const int n_tile = tile * XC_R;

    const uint32_t base_addr = lds_addr_of(smem);
    // DMA: instruction i covers rows XC0_RPI*i ..; lane -> (row + lane / PPR, stored piece lane % PPR)
    int drow[NI], dcol[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int row = XC0_RPI * (NI * wave + i) + lane / XC0_PPR;
        const int piece = (lane % XC0_PPR) ^ (4 * (row & 3));
        drow[i] = row;
        dcol[i] = min(n_tile + piece * 8, N - 8);          // columns past N are clamped re-reads (never stored)
    }
    auto issue_x = [&](int p, int pos) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int frow = min(p * 64 + drow[i], Cin - 1);   // an odd half that does not exist re-reads the last row
            glds16_asm(X + (size_t)frow * N + dcol[i], __builtin_amdgcn_readfirstlane(base_addr + pos * XC0_SLAB + (NI * wave + i) * 1024));
        }
    };
    // transposing-read addressing: 16-lane group g16 -> minibatch columns 16*(g16&1) .. +15 of a 32-column tile, K half
    // g16 >> 1; lane t16 points at row (t16 >> 2) of a 4-row band, 8 bytes at column 4*(t16 & 3)
    const int g16 = lane >> 4, t16 = lane & 15;
    const int trow = t16 >> 2;
    const int tcolb = (16 * (g16 & 1) + 4 * (t16 & 3)) * 2;   // byte offset of this lane's 8 bytes inside a 64-byte tile row

    f32x16 acc[XC_RT];
#pragma unroll
    for (int t = 0; t < XC_RT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

    auto load_w = [&](int w, Frag32<DT>& f) {
        if (w >= 0) {
            if constexpr (TRANSW) f.load_strided(Wsel + (size_t)w * 1024 + r, 32, h);   // Wop[o][i] = W[i][o]
            else                  f.load_contig(Wsel + (size_t)w * 1024 + r * 32, h);
        }
    };
    auto block = [&](const Frag32<DT>& wf, const unsigned char* slab, int half) {
        uint4 xf[XC_RT][2];
#pragma unroll
        for (int t = 0; t < XC_RT; ++t)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                // rows (features) 32*half + 16*kk + 8*(g16>>1) + {0..3 | 4..7}; row & 3 == trow for both bands
                const int row0 = 32 * half + 16 * kk + 8 * (g16 >> 1) + trow;
                const int byte = 64 * t + tcolb;                                  // byte inside the row (before swizzle)
                const int sw = (((byte >> 4) ^ (4 * trow)) << 4) | (byte & 15);
                const uint2 lo = ds_tr16(slab + row0 * XC0_ROWB + sw), hi = ds_tr16(slab + (row0 + 4) * XC0_ROWB + sw);
                xf[t][kk] = make_uint4(lo.x, lo.y, hi.x, hi.y);
            }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int t = 0; t < XC_RT; ++t) acc[t] = DT::mfma32(wf.q[kk], xf[t][kk], acc[t]);
    };

    const bool owner = wave < nob;
    if (nsteps > 0) {   // phases of PH steps, one barrier per phase (see xcol32_a1_kernel)
        for (int tb = 0; tb < nsteps; tb += 64) {
            const int idx = min(tb + lane, nsteps - 1);
            const int pv = pairs[idx];
            const int w0v = owner ? wt0[idx] : -1, w1v = owner ? wt1[idx] : -1;
            const int tend = min(64, nsteps - tb);
            Frag32<DT> wc0, wc1, wn0, wn1;
            wc0.zero(); wc1.zero(); wn0.zero(); wn1.zero();
            int c0 = __builtin_amdgcn_readlane(w0v, 0), c1 = __builtin_amdgcn_readlane(w1v, 0);
            load_w(c0, wc0);
            load_w(c1, wc1);
#pragma unroll
            for (int u = 0; u < PH; ++u)
                if (u < tend) issue_x(__builtin_amdgcn_readlane(pv, u), u);
            for (int s = 0; s < tend; s += PH) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
#pragma unroll
                for (int u = 0; u < PH; ++u)
                    if (s + PH + u < tend) issue_x(__builtin_amdgcn_readlane(pv, s + PH + u), (s + PH + u) % RING);
#pragma unroll
                for (int u = 0; u < PH; ++u) {
                    const int ss = s + u;
                    if (ss >= tend) break;
                    if (u >= 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    int n0 = -1, n1 = -1;
                    if (ss + 1 < tend) {
                        n0 = __builtin_amdgcn_readlane(w0v, ss + 1);
                        n1 = __builtin_amdgcn_readlane(w1v, ss + 1);
                        load_w(n0, wn0);
                        load_w(n1, wn1);
                    }
                    const unsigned char* slab = smem + (ss % RING) * XC0_SLAB;
                    if (c0 >= 0) block(wc0, slab, 0);
                    if (c1 >= 0) block(wc1, slab, 1);
                    wc0 = wn0; wc1 = wn1; c0 = n0; c1 = n1;
                }
            }
            __syncthreads();
        }
    }
    if (wave >= nob) return;

    // D[o][n]: col = n = r, rows o = (reg & 3) + 8 * (reg >> 2) + 4h  ->  Y[(ob*32 + o) * N + n]
#pragma unroll
    for (int t = 0; t < XC_RT; ++t) {
        const int n = n_tile + t * 32 + r;
        if (n >= N) continue;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int o = (reg & 3) + 8 * (reg >> 2) + 4 * h;
            Y[(size_t)((ob0 + wave) * 32 + o) * N + n] = DT::from_f32(acc[t][reg]);
        }
    }
}

// (The fp32 matrix-core version of this kernel, xcol32f_kernel on v_mfma_f32_32x32x2_f32 with its 'BSXF' plans -- 0.65 / 0.61 ms at the bench
// shape against 0.48 / 0.47 for the exact bf16 three-piece split of bsmm_xcols.h -- was kept for A/B through round 3 and retired in round 4.)

}  // namespace bsmm
