// bsmm_xflow.h -- xprop kernel "wave owns an output column", barrier-free and persistent ('BSX4' plans, round 4): feature_axis = 1,
// bsize 32, 16-bit storage types.
//
// What bounded bsmm_xcol_v2.h (profiles/r02_xcol_v2_ablation.md, r03_xcol_ab.md): its request stream alone and its matrix work alone
// each take ~65 us of a ~85 us pass -- the stream because every phase ends in a full drain (vmcnt(0) + barrier: ~0.5 us with nothing
// in flight), the matrix work because a phase lasts as long as its busiest SIMD (1.5x the mean at 20 % density) and every wave pays
// every barrier.  Here there is NO workgroup barrier in the main loop and a wave executes only the EVENTS the plan lists for it:
//   * one STEP = one pair of input blocks = one 16 KiB activation slab [128 rows][128 B] in a ring of X4_D slabs;
//   * BLOCK event: the wave's weight block of a step (private to it: it owns the output column) sits in one of its own two 2 KiB
//     slots, fetched by the wave itself two BLOCK events earlier and awaited with its own vmcnt; the wave then waits until the
//     slab's parts have been announced (full[slot][part] = global step number + 1), multiplies, and publishes its PROGRESS word =
//     the step of its next block ("I need no slab before that one");
//   * REQ event: request one part of a slab X4_DX steps ahead, once every wave's progress word has passed the slab that occupied
//     the ring slot; ANN event, one step later in the wave's order: wait for those requests, announce the part.  The plan deals
//     these duties to waves that have no blocks around that point (10 of 16 at 20 % density);
//   * the plan fixes the order of a wave's vector-memory operations, so every wait is a counted vmcnt the plan supplies;
//   * a wave never visits a step it has nothing to do in (first version: every wave walked every step with two polls and two
//     counter updates: 136 us of pure bookkeeping per pass at the bench shape -- the scalar unit is shared by the CU's 16 waves);
//   * the kernel is PERSISTENT: one workgroup per CU walks its (row tile, group) units back to back; ring slots and step numbers
//     run through the unit boundaries, the waves that are done with a unit start the next one's duties while the others finish;
//     the epilogue of a wave goes through its own (then idle) weight slots, so it needs no barrier either.
// Fragment layouts, swizzles, MFMA order per column: those of bsmm_xcol_v2.h (a column sums its blocks in the same order with the
// same instructions: bit-identical outputs).
#pragma once
#include "bsmm_common.h"
#include "bsmm_plan.h"
#include "bsmm_updat_v2.h"   // glds16_saddr, uniform_ptr
#include "bsmm_updat_tr.h"   // ds_tr16
#include "bsmm_xprop.h"      // XMap

namespace bsmm {

#ifndef X4_NO_XDMA
#define X4_NO_XDMA 0          // ablation switches: wrong results by construction
#endif
#ifndef X4_NO_WDMA
#define X4_NO_WDMA 0
#endif
#ifndef X4_NO_MATH
#define X4_NO_MATH 0
#endif
#ifndef X4_READS_FIRST
#define X4_READS_FIRST 1
#endif
#ifndef X4_PUBLISH_FIRST
#define X4_PUBLISH_FIRST 1
#endif
#ifndef X4_FETCH_EARLY
#define X4_FETCH_EARLY 0      // 1: a BLOCK event requests its wave's weight block two events ahead right behind its first MFMA instead of behind the
                              //    block: bit-identical, measured 91.4 / 81.7 against 88.9 / 78.7 us (profiles/r05_flow_fetch_early.txt) -- not taken
#endif
#ifndef X4_STAGGER
#define X4_STAGGER 0
#endif
#ifndef X4_PRIO
#define X4_PRIO 0             // 1: a wave multiplying a block runs at raised issue priority (the pollers at 0)
#endif
#ifdef X4_TIMELINE
// time line of workgroup 0 (debug builds): per global step [0..255]: [0..3] REQ start (after its progress poll) of part 0 / 1, then REQ
// issued of part 0 / 1; [4..5] ANN of part 0 / 1; [6 + 2 w] BLOCK of wave w has its slab, [7 + 2 w] BLOCK done; bsmm_debug_x4_timeline_copy()
__device__ unsigned long long g_x4_tl[256 * 40];
#define X4_TL(step, k) do { if (blockIdx.x == 0 && (step) < 256) { const unsigned long long t_ = __builtin_readcyclecounter(); if (lane == 0) g_x4_tl[(step) * 40 + (k)] = t_; } } while (0)
#else
#define X4_TL(step, k) do { } while (0)
#endif
#ifdef X4_STAMPS
// cycle accounting of the first 64 workgroups (debug builds; s_memtime): per wave [0] waits for my fetch / requests, [1] slab polls,
// [2] multiplies, [3] progress polls of REQ, [4] request issue, [5] fetch issue + progress write, [6] whole kernel, [7] events;
// read back with bsmm_debug_x4_trace_copy()
__device__ unsigned long long g_x4_trace[64 * 16 * 8];
#define X4_T0() const unsigned long long t0_ = __builtin_readcyclecounter()
#define X4_T1(k) tacc[k] += __builtin_readcyclecounter() - t0_
#else
#define X4_T0() do { } while (0)
#define X4_T1(k) do { } while (0)
#endif
constexpr int X4_R = 128;                              // minibatch rows per unit
constexpr int X4_SLAB = X4_R * 128;                    // 16 KiB
constexpr int X4_S = 2;                                // private weight slots per wave
constexpr int X4_WBASE = X4_D * X4_SLAB;
constexpr int X4_WWAVE = X4_S * 2048;                  // bytes of weight slots per wave
constexpr int X4_FLAGS = X4_WBASE + X4_G * X4_WWAVE;   // prog[16] (uint32), then full[8]
constexpr int X4_LDS = X4_FLAGS + 64 + 64;             // (full[]: up to 16 slab counters)
constexpr int X4_DI = 16 / X4_PARTS;                   // DMA instructions of 1 KiB per slab part of a 128-row slab (what the PLAN counts per REQ)
static_assert(X4_LDS <= 163840, "flow kernel: ring must fit the LDS");
static_assert(X4_WWAVE >= 64 * 64, "the epilogue stages 64 rows x 64 B per pass in a wave's weight slots");
static_assert((X4_WWAVE & (X4_WWAVE - 1)) == 0 && X4_WBASE % X4_WWAVE == 0, "weight slot toggling by XOR needs aligned slots");

// The CU's 16 waves share ONE scalar unit: a scalar instruction per event and wave costs 16 cycles per event round.  The per-event
// fields therefore live in lane-indexed vector registers (lane j = event j of the chunk), are derived once per unit with vector
// instructions, and are used in place under EXEC = that lane: the helpers below take the lane mask `em` (1 << event).
// counter / word update by the event's own lane: [addr] (+)= value, both lane-resident
__device__ __forceinline__ void x4_lane_write(uint64_t em, uint32_t v_addr, uint32_t v_value) {
    asm volatile("s_mov_b64 exec, %0\n\tds_write_b32 %1, %2\n\ts_mov_b64 exec, -1" ::"s"(em), "v"(v_addr), "v"(v_value) : "memory");
}
__device__ __forceinline__ void x4_lane_add(uint64_t em, uint32_t v_addr, uint32_t v_value) {
    asm volatile("s_mov_b64 exec, %0\n\tds_add_u32 %1, %2\n\ts_mov_b64 exec, -1" ::"s"(em), "v"(v_addr), "v"(v_value) : "memory");
}
// spin until [addr] >= target, compared in the event's own lane
__device__ __forceinline__ void x4_lane_poll_ge(uint64_t em, uint32_t v_addr, uint32_t v_target) {
    uint32_t t;
    asm volatile("s_mov_b64 exec, %1\n"
                 "1:\n\tds_read_b32 %0, %2\n\ts_waitcnt lgkmcnt(0)\n\tv_cmp_ge_u32 vcc, %0, %3\n\ts_cbranch_vccnz 2f\n\ts_sleep 1\n\ts_branch 1b\n"
                 "2:\n\ts_mov_b64 exec, -1"
                 : "=&v"(t) : "s"(em), "v"(v_addr), "v"(v_target) : "memory", "vcc");
}
// spin until every lane's word [addr] >= need (all lanes active; lane l reads the progress word of wave l & 15)
__device__ __forceinline__ void x4_poll_all_ge(uint32_t v_addr, uint32_t s_need) {
    uint32_t t;
    asm volatile("1:\n\tds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)\n\tv_cmp_gt_u32 vcc, %2, %0\n\ts_cbranch_vccz 2f\n\ts_sleep 1\n\ts_branch 1b\n2:"
                 : "=&v"(t) : "v"(v_addr), "s"(s_need) : "memory", "vcc");
}

// the wait of a BLOCK / ANN event, chosen by the event's control word: bit 3 vmcnt(2), bit 4 vmcnt(0), else vmcnt(DI) with DI = the requests a REQ
// of THIS kernel issues (the plan counts X4_DI per REQ; a kernel with smaller slabs issues fewer, and the wait must not assume more)
// the activation slabs' requests with a cache policy of their own (X4_XPOL: "" default, " nt" = streaming: measured in profiles/r05_flow_cache_policy.txt)
#ifndef X4_XPOL
#define X4_XPOL ""
#endif
__device__ __forceinline__ void x4_glds_x(const void* sbase, uint32_t voff, uint32_t lds_byte_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" X4_XPOL "\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_byte_addr) : "memory");
}
__device__ __forceinline__ void x4_glds_x4(const void* sbase, uint32_t v0, uint32_t v1, uint32_t v2, uint32_t v3, uint32_t lds_byte_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %5" X4_XPOL "\n\t"
                 "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %5" X4_XPOL "\n\t"
                 "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %5" X4_XPOL "\n\t"
                 "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %5" X4_XPOL "\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(v0), "v"(v1), "v"(v2), "v"(v3), "s"(sbase), "s"(lds_byte_addr) : "memory", "scc");
}

template <int DI>
__device__ __forceinline__ void x4_wait_ctl(uint32_t ctl) {
    asm volatile("s_bitcmp1_b32 %0, 3\n\ts_cbranch_scc0 1f\n\ts_waitcnt vmcnt(2)\n\ts_branch 3f\n"
                 "1:\n\ts_bitcmp1_b32 %0, 4\n\ts_cbranch_scc0 2f\n\ts_waitcnt vmcnt(0)\n\ts_branch 3f\n"
                 "2:\n\ts_waitcnt vmcnt(%1)\n3:" ::"s"(ctl), "n"(DI) : "memory", "scc");
}

// RT = 32-row tiles per unit: 4 (128 rows, slabs of 16 KiB, ring of X4_D) or 2 (64 rows, slabs of 8 KiB, ring of 2 X4_D: twice the units for
// minibatches that do not fill the chip with 128-row units; same plans, same LDS layout).
#ifdef X4_ENDSTAMPS
__device__ unsigned long long g_x4_ends[512 * 16 * 8];
#endif
template <class DT, bool TRANSW, int RT = 4>
__global__ void __launch_bounds__(64 * X4_G, 4)
xflow32_kernel(const typename DT::T* __restrict__ X, const typename DT::T* __restrict__ Wsel, typename DT::T* __restrict__ Y,
               const int32_t* __restrict__ plan, XMap map, int N, int Cin, int Kout) {
    typedef typename DT::T T;
    static_assert(DT::is16, "flow kernel: 16-bit storage types");
    static_assert(RT == 4 || RT == 2, "flow kernel: units of 128 or 64 rows");
    constexpr int R = 32 * RT, SLAB = R * 128, D = X4_D * 4 / RT, DI = X4_DI * RT / 4;     // rows, slab bytes, ring depth, requests per REQ
    static_assert(D * SLAB == X4_WBASE && DI % 4 == 0 && D <= 16, "flow kernel: ring geometry");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, h = lane >> 5;
    const uint32_t base_addr = lds_addr_of(smem);
    const uint32_t prog_addr = base_addr + X4_FLAGS, full_addr = base_addr + X4_FLAGS + 64;
    if (threadIdx.x < 32) reinterpret_cast<uint32_t*>(smem + X4_FLAGS)[threadIdx.x] = 0u;
    __syncthreads();                                   // the only workgroup barrier of the kernel
#if X4_STAGGER
    // experiment: the workgroups of an XCD start X4_STAGGER x 64 cycles apart, so that the tiles that share a weight block (and the groups that
    // share a slab) do not ask for it in the same few hundred cycles -- the followers should find it in the L2 instead of waiting for the fill
    for (int i = ((blockIdx.x >> 3) & 31) * X4_STAGGER; i > 0; --i) __builtin_amdgcn_s_sleep(1);
#endif

    const int npairs_full = Cin / 64;
    const unsigned char* xt = reinterpret_cast<const unsigned char*>(X);
    const unsigned char* wsel = static_cast<const unsigned char*>(uniform_ptr(Wsel));
    // weight DMA (bsmm_xcol_v2.h): lane i of an instruction writes piece i of a 1 KiB half block; it fetches the piece that the read
    // swizzle expects there (none for the transposing reads of fprop)
    const uint32_t wvoff = TRANSW ? (uint32_t)lane * 16u : (uint32_t)((lane >> 2) * 64 + (((lane & 3) ^ ((lane >> 4) & 3)) << 4));
    const uint32_t wslot0 = X4_WBASE + wave * X4_WWAVE;                  // my weight slots (byte offset inside smem)
    // fragment read offsets: activations (slab 0, half 0; the event's word XORs in slot << 14 | half << 6), weights (my slot 0; XOR 2048)
    const int xsw = (r >> 1) & 7;
    uint32_t xrd[2], wrd[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        xrd[kk] = r * 128 + (((2 * kk + h) ^ xsw) << 4);
        if constexpr (TRANSW) {
            const int g16 = lane >> 4, t16 = lane & 15;
            wrd[kk] = wslot0 + (16 * kk + 8 * h + (t16 >> 2)) * 64 + (16 * (g16 & 1) + 4 * (t16 & 3)) * 2;
        } else {
            wrd[kk] = wslot0 + r * 64 + (((2 * kk + h) ^ ((r >> 2) & 3)) << 4);
        }
    }
    const uint32_t my_prog_s = prog_addr + 4 * wave;                     // (uniform: materialised in a vector register only where it is used)
    // slab request offsets of a full tile: DMA instruction ii covers rows 8 ii + (lane >> 3); the 16-byte piece a lane fetches is
    // (lane & 7) ^ ((row >> 1) & 7) = (lane & 7) ^ (lane >> 4) ^ (4 (ii & 1)): one pattern for even, one for odd instructions
    const uint32_t stride16 = (uint32_t)Cin * 16u;                       // bytes between the first rows of consecutive instructions
    const uint32_t pc_e = (uint32_t)((lane & 7) ^ (lane >> 4));
    const uint32_t vo_e = (uint32_t)(lane >> 3) * (uint32_t)Cin * 2u + pc_e * 16u;     // (odd instructions: the piece index XOR 4 = +- 64 bytes)

#ifdef X4_STAMPS
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long tstart = __builtin_readcyclecounter();
#endif
#ifdef X4_ENDSTAMPS
    // wall clock (s_memrealtime, 100 MHz) of every wave of every workgroup: [0] start, [1 + u] its unit u stored (u < 6), [7] units; debug builds
    int x4e_u = 0;
    if (lane == 0 && blockIdx.x < 512) g_x4_ends[(blockIdx.x * 16 + wave) * 8] = __builtin_amdgcn_s_memrealtime();
#endif
    uint32_t gs = 0;                 // global step number of the current unit's step 0 (ring slot = global step % X4_D)
    const int nunits = map.grid();
    for (int unit = blockIdx.x; unit < nunits; unit += gridDim.x) {
        int tile, grp;
        if (!xmap_decode(map, unit, tile, grp)) continue;
        const int32_t* gh = plan + plan[5] + X4_GROUP * grp;
        const int step_off = __builtin_amdgcn_readfirstlane(gh[0]), nsteps = __builtin_amdgcn_readfirstlane(gh[1]);
        const int list_off = __builtin_amdgcn_readfirstlane(gh[4]), lcap = __builtin_amdgcn_readfirstlane(gh[5]);
        const int32_t* pairs = plan + plan[6] + step_off;
        const int32_t* lists = plan + plan[7] + list_off;
        const int nev = __builtin_amdgcn_readfirstlane(lists[wave]);
        const int my_ob = __builtin_amdgcn_readfirstlane(lists[X4_G + wave]);      // the output block this wave owns in this group (-1: none)
        const int32_t* mine = lists + 2 * X4_G + (size_t)2 * lcap * wave;
        const int n_tile = tile * R;
        // (a group without any block has no steps: its waves run their two NOPs and the epilogue writes the zeros the output must hold)
        const unsigned char* xtile = static_cast<const unsigned char*>(uniform_ptr(xt + (size_t)n_tile * Cin * 2));
        const bool fast_tile = n_tile + R <= N;                       // no row of the tile lies past N: the request offsets are regular

        f32x16 acc[RT];
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

        uint32_t s_w = 0, s_wf = 0;  // XOR offset of the weight slot to multiply from / to fetch into (0 or 2048)
        for (int eb = 0; eb < nev; eb += 64) {
            // ---- my events [eb, eb + 64), lane-indexed; the per-event fields are derived here, once, with vector instructions ----
            const int ei = min(eb + lane, nev - 1);
            const uint32_t w0 = (uint32_t)mine[2 * ei], w1 = (uint32_t)mine[2 * ei + 1];
            const uint32_t ty = w0 & 3, hp = (w0 >> 2) & 3, step = (w0 >> 4) & 0xfff, nxt = (w0 >> 16) & 0xfff;
            const uint32_t gstep = gs + step, sm = gstep % (uint32_t)D, guse = gstep / (uint32_t)D + 1u;   // ring slot, its uses so far
            uint32_t p128 = 0;
            if (ty == 2) p128 = (uint32_t)pairs[step];                   // REQ lanes: the pair of their step
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // (drains the wave's memory queue: once per unit and 64 events)
            const uint32_t wn = w1 >> 27;                                // vmcnt the plan computed for this event's wait
            // control word: type (bits 0..1) | wait: bit 3 vmcnt(2), bit 4 vmcnt(0), neither vmcnt(X4_DI) | bit 6 has a fetch | bit 7 request needs the slow path
            const uint32_t f27 = w1 & X4_NOFETCH;
            uint32_t v_ctl = ty | (wn >= (uint32_t)X4_DI ? 0u : (wn >= 2u ? 8u : 16u)) | (f27 != X4_NOFETCH ? 64u : 0u) |
                             ((ty == 2 && (!fast_tile || (int)p128 >= npairs_full)) ? 128u : 0u);
            // BLOCK: XOR word of the activation fragment addresses; REQ: byte offset of the part inside the ring
            uint32_t v_x = ty == 1 ? ((sm * (uint32_t)SLAB) | (hp << 6)) : (sm * (uint32_t)SLAB + hp * (uint32_t)(DI * 1024));
            // BLOCK / ANN: address of the slab's counter; REQ: byte offset of the pair inside an activation row
            uint32_t v_fa = ty == 2 ? p128 * 128u : full_addr + 4 * sm;
            // BLOCK: the counter value that says "every part of this step's slab is in" (the slot's uses so far, this one included, times
            // the parts); REQ: global number of the step + 1 (the progress every wave must have passed, plus X4_D)
            uint32_t v_g = ty == 2 ? gstep + 1u : guse * (uint32_t)X4_PARTS;
            uint32_t v_pv = gs + nxt;                                    // my progress after the event: the step of my next block
            uint32_t v_fw = f27 << 11;                                   // byte offset of the weight block to fetch after the event
            asm volatile("" : "+v"(v_ctl), "+v"(v_x), "+v"(v_fa), "+v"(v_g), "+v"(v_pv), "+v"(v_fw));
#ifdef X4_TIMELINE
            const uint32_t v_dbg = gstep | (hp << 16);
#endif
            const int nchunk = min(64, nev - eb);
            for (int idx = 0; idx < nchunk; ++idx) {
                const uint64_t em = 1ull << idx;
                const uint32_t ctl = (uint32_t)__builtin_amdgcn_readlane(v_ctl, idx);
                const uint32_t ty_s = ctl & 3;
#ifdef X4_STAMPS
                tacc[7] += 1;
#endif
                bool fetched = false;
                if (ty_s & 1) { X4_T0(); x4_wait_ctl<DI>(ctl); X4_T1(0); }     // BLOCK (1) or ANN (3): wait for my fetch / my requests
                if (ty_s == 1) {
                    // ---- BLOCK: my weight block (fetched two BLOCK events ago) x the slab of its step ----
                    { X4_T0(); x4_lane_poll_ge(em, v_fa, v_g); X4_T1(1); }  // every part of the slab has been announced
#ifdef X4_TIMELINE
                    const uint32_t dbg_s = (uint32_t)__builtin_amdgcn_readlane(v_dbg, idx) & 0xffff;
                    X4_TL(dbg_s, 6 + 2 * wave);
#endif
                    X4_T0();
                    if (X4_PRIO) __builtin_amdgcn_s_setprio(3);
                    if (!X4_NO_MATH) {
                        const uint32_t sx = (uint32_t)__builtin_amdgcn_readlane(v_x, idx);
                        uint4 wq[2];
#pragma unroll
                        for (int kk = 0; kk < 2; ++kk) {
                            if constexpr (TRANSW) {
                                const uint2 lo = ds_tr16(smem + (wrd[kk] ^ s_w)), hi = ds_tr16(smem + (wrd[kk] ^ s_w) + 4 * 64);
                                wq[kk] = make_uint4(lo.x, lo.y, hi.x, hi.y);
                            } else {
                                wq[kk] = *reinterpret_cast<const uint4*>(smem + (wrd[kk] ^ s_w));
                            }
                        }
#if X4_READS_FIRST
                        // a block is on its wave's critical path: its LATENCY counts.  Both weight fragments and the four K-half-0
                        // activation fragments are in flight before the first MFMA; each K-half-1 fragment is requested right behind
                        // the MFMA that consumed its K-half-0 sibling (into the same registers) and lands under the other three
                        uint4 xf[RT];
#pragma unroll
                        for (int t = 0; t < RT; ++t) xf[t] = *reinterpret_cast<const uint4*>(smem + (xrd[0] ^ sx) + t * 4096);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int t = 0; t < RT; ++t) {
                            acc[t] = DT::mfma32(wq[0], xf[t], acc[t]);
                            xf[t] = *reinterpret_cast<const uint4*>(smem + (xrd[1] ^ sx) + t * 4096);
                            __builtin_amdgcn_sched_barrier(0);
#if X4_FETCH_EARLY
                            // my weight block two BLOCK events ahead goes into the slot THIS block's fragments came from: they are in registers
                            // once the first MFMA could issue (LDS reads return in order: its activation fragment was requested behind them), so
                            // the two requests are issued here, under the matrix work, instead of behind the block on the wave's critical path
                            // (the order of the wave's vector-memory operations, hence every vmcnt of the plan, is unchanged)
                            if (t == 0 && (ctl & 64u)) {
                                if (!X4_NO_WDMA) {
                                    const uint32_t fo = (uint32_t)__builtin_amdgcn_readlane(v_fw, idx);
                                    glds16_saddr_x2(wsel, wvoff + fo, wvoff + fo + 1024u, base_addr + (wslot0 ^ s_wf));
                                }
                                s_wf ^= 2048u;
                                fetched = true;
                            }
#endif
                        }
#pragma unroll
                        for (int t = 0; t < RT; ++t) acc[t] = DT::mfma32(wq[1], xf[t], acc[t]);
#else
                        uint4 xf[RT][2];
#pragma unroll
                        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                            for (int t = 0; t < RT; ++t) xf[t][kk] = *reinterpret_cast<const uint4*>(smem + (xrd[kk] ^ sx) + t * 4096);
#pragma unroll
                        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                            for (int t = 0; t < RT; ++t) acc[t] = DT::mfma32(wq[kk], xf[t][kk], acc[t]);
#endif
                    }
                    s_w ^= 2048u;
#ifdef X4_STAMPS
                    asm volatile("s_nop 0" ::"v"(acc[0][0]), "v"(acc[RT - 1][0]));
#endif
                    X4_T1(2);
#ifdef X4_TIMELINE
                    asm volatile("s_nop 0" ::"v"(acc[0][0]), "v"(acc[RT - 1][0]));
                    X4_TL(dbg_s, 7 + 2 * wave);
#endif
                } else if (ty_s == 2) {
                    // ---- REQ: request one part of a slab, once every wave's next block lies beyond the slab that occupied the slot ----
                    const uint32_t g1 = (uint32_t)__builtin_amdgcn_readlane(v_g, idx);
                    { X4_T0(); if (g1 > (uint32_t)D) x4_poll_all_ge(prog_addr + 4 * (lane & 15), g1 - (uint32_t)D); X4_T1(3); }
                    X4_T0();
#ifdef X4_TIMELINE
                    const uint32_t dbg_r = (uint32_t)__builtin_amdgcn_readlane(v_dbg, idx);
                    X4_TL(dbg_r & 0xffff, dbg_r >> 16);
#endif
                    if (!X4_NO_XDMA) {
                        const uint32_t dst = base_addr + (uint32_t)__builtin_amdgcn_readlane(v_x, idx);
                        const uint32_t poff = (uint32_t)__builtin_amdgcn_readlane(v_fa, idx);
                        const uint32_t part_i = (uint32_t)__builtin_amdgcn_readlane(v_x, idx) % (uint32_t)SLAB / 1024u;   // first instruction of the part
                        if (!(ctl & 128u)) {
                            // instruction ii covers rows 8 ii ..: lane offset = ii * 16 Cin + the even / odd pattern (+ the pair's offset)
                            const uint32_t k0 = part_i * stride16 + poff;
                            const uint32_t vo_o = (pc_e & 4u) ? vo_e - 64u : vo_e + 64u;
#pragma unroll
                            for (int k = 0; k < DI; k += 4)
                                x4_glds_x4(xtile, vo_e + (k0 + (k + 0) * stride16), vo_o + (k0 + (k + 1) * stride16), vo_e + (k0 + (k + 2) * stride16),
                                                vo_o + (k0 + (k + 3) * stride16), dst + k * 1024);
                        } else {
                            const bool tail = (int)(poff >> 7) >= npairs_full;
#pragma unroll
                            for (int k = 0; k < DI; ++k) {
                                const int row = 8 * ((int)part_i + k) + (lane >> 3);
                                const int xr = min(n_tile + row, N - 1) - n_tile;    // rows past N are clamped (never stored)
                                const int piece = (lane & 7) ^ ((row >> 1) & 7);
                                uint32_t voff = (uint32_t)xr * (uint32_t)Cin * 2u + piece * 16 + poff;
                                if (tail && (piece & 4)) voff -= 64;                 // last pair of an odd block count: re-read its even half
                                x4_glds_x(xtile, voff, dst + k * 1024);
                            }
                        }
                    }
                    X4_T1(4);
#ifdef X4_TIMELINE
                    X4_TL(dbg_r & 0xffff, 2 + (dbg_r >> 16));
#endif
                } else if (ty_s == 3) {
                    // ---- ANN: my requests for that part have landed -> count it in ----
                    x4_lane_add(em, v_fa, 1u);
#ifdef X4_TIMELINE
                    { const uint32_t dbg_a = (uint32_t)__builtin_amdgcn_readlane(v_dbg, idx); X4_TL(dbg_a & 0xffff, 4 + (dbg_a >> 16)); }
#endif
                }
                if (ty_s < 2) {
                    // ---- after a BLOCK / NOP: my next weight fetch (into the slot the block just freed), my progress ----
                    X4_T0();
                    if (X4_PUBLISH_FIRST) x4_lane_write(em, my_prog_s, v_pv);   // (first: the requesters of the slot I just left are waiting for this)
                    if ((ctl & 64u) && !fetched) {
                        if (!X4_NO_WDMA) {
                            const uint32_t fo = (uint32_t)__builtin_amdgcn_readlane(v_fw, idx);
                            glds16_saddr_x2(wsel, wvoff + fo, wvoff + fo + 1024u, base_addr + (wslot0 ^ s_wf));
                        }
                        s_wf ^= 2048u;
                    }
                    if (!X4_PUBLISH_FIRST) x4_lane_write(em, my_prog_s, v_pv);
                    if (X4_PRIO) __builtin_amdgcn_s_setprio(0);
                    X4_T1(5);
                }
            }
        }
        gs += (uint32_t)nsteps;

        // Epilogue, per wave, through its own weight slots (idle now): D[o][n] with col n = r, rows o = (reg & 3) + 8 (reg >> 2) + 4h.
        // Two passes of two row tiles: [64 rows][64 B], the four 16-byte pieces of row n XOR-swizzled with (n >> 2) & 3; read back as
        // full 64-byte rows and stored (16 rows per instruction).
        if (my_ob >= 0) {
            unsigned char* stage = smem + wslot0;
            unsigned char* ybase = reinterpret_cast<unsigned char*>(Y + (size_t)my_ob * 32);
#pragma unroll
            for (int pass = 0; pass < RT / 2; ++pass) {
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    const int t = 2 * pass + tt;
                    const int n = tt * 32 + r;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const uint32_t lo = (uint32_t)DT::from_f32(acc[t][4 * q + 0]) | ((uint32_t)DT::from_f32(acc[t][4 * q + 1]) << 16);
                        const uint32_t hi = (uint32_t)DT::from_f32(acc[t][4 * q + 2]) | ((uint32_t)DT::from_f32(acc[t][4 * q + 3]) << 16);
                        *reinterpret_cast<uint2*>(stage + n * 64 + ((q ^ ((n >> 2) & 3)) << 4) + 8 * h) = make_uint2(lo, hi);
                    }
                }
                asm volatile("" ::: "memory");
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int n = 16 * i + (lane >> 2), pc = lane & 3;
                    const uint4 v = *reinterpret_cast<const uint4*>(stage + n * 64 + ((pc ^ ((n >> 2) & 3)) << 4));
                    const int gn = n_tile + 64 * pass + n;
                    if (gn < N) *reinterpret_cast<uint4*>(ybase + (size_t)gn * Kout * 2 + pc * 16) = v;
                }
                asm volatile("" ::: "memory");
            }
        }
#ifdef X4_ENDSTAMPS
        if (lane == 0 && blockIdx.x < 512 && x4e_u < 6) g_x4_ends[(blockIdx.x * 16 + wave) * 8 + 1 + x4e_u] = __builtin_amdgcn_s_memrealtime();
        ++x4e_u;
#endif
    }
#ifdef X4_ENDSTAMPS
    if (lane == 0 && blockIdx.x < 512) g_x4_ends[(blockIdx.x * 16 + wave) * 8 + 7] = (unsigned long long)x4e_u;
#endif
#ifdef X4_STAMPS
    tacc[6] = __builtin_readcyclecounter() - tstart;
    if (blockIdx.x < 64 && lane == 0)
        for (int k = 0; k < 8; ++k) g_x4_trace[(blockIdx.x * 16 + wave) * 8 + k] = tacc[k];
#endif
}

}  // namespace bsmm
