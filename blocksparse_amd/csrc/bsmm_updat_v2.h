// bsmm_updat_v2.h -- streaming weight-gradient kernel, feature_axis = 1, bsize 32, 16-bit storage types ('BSU2' plans).
//
//   DW[w][ci][ko] = alpha * sum_p sum_n X_p[n][c*32+ci] * DY_p[n][k*32+ko] + beta * DW[w][ci][ko],  (c,k) = updat_lut[w]
//
// What bounded the windowed kernel of round 1 (bsmm_updat_win.h; 122 us = 18 % of the bf16 matrix-core peak at the bench
// shape), from its counters: 2.1 GB through the L2 -> LDS path per pass (8x8 windows: 16 slab rows for ~13 blocks), 0.4 GB
// of it missing the XCD's L2 (a window PATCH per XCD reads 4/16 of X and 8/16 of DY), 29 % of the MFMAs on empty slots,
// 4 transposing reads per MFMA, and a drained DMA queue at every chunk barrier.  This kernel changes all five:
//   * windows of WS x WS blocks with WS = 16 (<= 64 blocks: 16 waves x 4 accumulator slots) halve the staged bytes per
//     block; every slot of an item is a real block (per-wave slot counts, no padding MFMAs);
//   * a wave's blocks come from at most two block ROWS of the window, whose X^T fragment is read once per row
//     (2 + 2 n transposing reads for n blocks instead of 4 n);
//   * 16-row chunks in a ring of four slots, the DMA of chunk i+2 requested in interval i and `s_waitcnt vmcnt(NI)`:
//     two chunks are always in flight and the queue never drains;
//   * two wave sets per SIMD half an interval out of phase (one multiplies while the other's fragment reads are in flight);
//   * the schedule of the plan (bsmm_plan.h): XCD x owns an item set and a part of the minibatch, its workgroups walk the
//     set's items in lockstep through that part, so an XCD's L2 sees a quarter of half of X and of all of DY;
//   * the partial sums of a workgroup's (item, minibatch range) leave as plain 16-byte stores in register order into that
//     (round, workgroup)'s region of the workspace; updat2_reduce_kernel walks the same schedule, sums the regions of every
//     block and applies alpha / beta (and the optional gate) with ONE rounding.  Items that are one workgroup's are stored directly.
// LDS image of a chunk: X slab [16 rows][WS*64 B] then DY slab, 16-byte pieces of row r XOR-swizzled with 4*(r & 3)
// (bank-conflict free for ds_read_b64_tr_b16, as in bsmm_updat_win.h).  1024 threads, 128 KiB (WS = 16): one workgroup per CU.
//
// Measured on the way (profiles/r02_updat_ablation.md): with everything but the loop skeleton compiled out (no DMA, no
// fragment reads, no MFMA, no epilogue) the first version still took 53 of its 128 us -- ~600 cycles per 16-row interval of
// scalar bookkeeping, branches and the barrier, against ~400 cycles of matrix work.  Hence the shape of the loop below: one
// specialisation per (slots of row group 0, slots of row group 1, wave set), nothing but pointer increments in the regular
// interval, and every irregularity (ragged last chunk of a pair, pair boundary, end of the range) inside `issue`, which
// also ZEROES the rows past N in LDS so that the fragment reads never mask.
#pragma once
#include <type_traits>

#include "bsmm_common.h"
#include "bsmm_plan.h"
#include "bsmm_updat_tr.h"

namespace bsmm {

#ifndef U2_NT_PARTS
#define U2_NT_PARTS 1       // non-temporal stores / loads of the partial sums: 50-80 MB that are written once and read once must not
#endif                      // push the activations out of the Infinity Cache (bench: the fprop that follows runs 91 instead of 99 us)
// slices per item of a set's last, incomplete round: every slice leaves a partial sum that the reduce pass reads back -- with 16 or 32
// slices of two left-over items the blocks of those items made that pass 22 us long (a serial chain of loads), for 2 us of kernel time
#ifndef U2_MAX_SLICES
#define U2_MAX_SLICES 8
#endif
#ifndef U2_CH_ROWS
#define U2_CH_ROWS 16
#endif
#ifndef U2_SETB_FROM
#define U2_SETB_FROM 0      // > 0: waves U2_SETB_FROM .. 15 form wave set B whatever the window (measurement builds); 0: the kernel's own choice
#endif
constexpr int U2_CH = U2_CH_ROWS;          // minibatch rows per chunk (one interval = one barrier): 16 or 32
constexpr int U2_KS = U2_CH / 16;          // MFMA K-steps per chunk
constexpr int U2_D = 64 / U2_CH;           // ring slots (128 KiB with 16x16 windows)
constexpr int U2_AHEAD = U2_D / 2;         // prefetch distance in chunks (see the ring protocol at the chunk loop)
static_assert(U2_CH == 16 || U2_CH == 32, "chunk = 1 or 2 K-steps");
constexpr int u2_ring(int ws) { return ws == 32 ? 2 : U2_D; }          // 32 x 32 windows (sparse layouts): two slots of 64 KiB
constexpr int u2_lds_bytes(int ws) { return u2_ring(ws) * 2 * U2_CH * ws * 64; }
// feature_axis 0 (activations (C, N): slab rows are FEATURES, a chunk is 32 minibatch columns = 64 B per row -- a piece length the
// L2 -> LDS path delivers at the rate of 128-byte pieces, scripts/micro/seg_bw.hip; 16 columns = 32 B deliver at half of it):
constexpr int U2_CH0 = 32;                 // minibatch columns per chunk
// Ring: two slots of 64 KiB (16x16 windows) / four of 32 KiB.  Measured at 4096^2, N = 8192 (profiles/r03_updat_a0_stream*.txt): 111 / 125 / 180 us
// at 10 / 20 / 50 % against 120 / 129 / 270 for the windowed kernel of round 1 -- and 71 / 95 / 174 on feature axis 1 for the same bytes.  The
// counters say why (profiles/r03_pmc_updat_a0_a1.txt): the L1 sends one L2 request per 64 contiguous bytes, 17.7 M requests against 8.7 M, and
// the per-CU request rate is what bounds these kernels; 128-byte row pieces (64 columns) would need 128 KiB per chunk of a 512 x 512 window.
// -DU2_RING5=1: FIVE half slots (X of chunk k in half slot 2k mod 5, DY in 2k + 1 mod 5; X requested two chunks ahead, DY one: 1.5 chunks
// always in flight) -- measured 109 / 129 / 192: more data in flight does not help a request-rate bound.
#ifndef U2_RING5
#define U2_RING5 0
#endif
constexpr int u2_lds_bytes0(int ws) { return U2_RING5 ? 5 * ws * 32 * (U2_CH0 * 2) : 131072; }

#ifdef U2_STAMPS
// wall-clock stamps (s_memrealtime, 100 MHz, one clock for the chip) of wave 0 of every workgroup: [0] start, [1] chunk loop of its first range
// done, [2] that range's sums stored, [3] end, [4] ranges walked, [5] chunks of the first range; bsmm_debug_u2_trace_copy(), scripts/gpu_updat_stamps.py.
// Debug builds only (profiles/r06_updat_loop.md).
__device__ unsigned long long g_u2_trace[1024 * 8];
#define U2_STAMP(k, v) do { if (wave == 0 && lane == 0 && blockIdx.x < 1024) g_u2_trace[blockIdx.x * 8 + (k)] = (v); } while (0)
#else
#define U2_STAMP(k, v) do { } while (0)
#endif

// a wave-uniform pointer, provably so for the compiler (an "s" asm operand fed from a value it regards as divergent is
// emitted as a VGPR and does not assemble)
__device__ __forceinline__ const void* uniform_ptr(const void* p) {
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return reinterpret_cast<const void*>(((uint64_t)hi << 32) | lo);
}

// LDS-DMA with a scalar base and a 32-bit per-lane byte offset (saddr form): no 64-bit address VGPRs in the loop.
__device__ __forceinline__ void glds16_saddr(const void* sbase, uint32_t voff, uint32_t lds_byte_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_byte_addr)
                 : "memory");
}
// NI consecutive 1 KiB pieces (LDS dst, dst + 1 KiB) from one scalar base: M0 is saved / restored once and stepped with a
// scalar add; 5 scalar instructions for two DMAs.  (The CU has ONE scalar unit for its 16 waves: the ~40 scalar instructions
// per wave and interval of the first version cost ~600 cycles per interval, more than the matrix work.)
__device__ __forceinline__ void glds16_saddr_x2(const void* sbase, uint32_t voff0, uint32_t voff1, uint32_t lds_byte_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\t"
                 "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff0), "v"(voff1), "s"(sbase), "s"(lds_byte_addr)
                 : "memory", "scc");      // (s_add_u32 writes SCC: without the clobber a compare hoisted above the block loses its result)
}

__device__ __forceinline__ void glds16_saddr_x4(const void* sbase, uint32_t v0, uint32_t v1, uint32_t v2, uint32_t v3, uint32_t lds_byte_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %5\n\t"
                 "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %5\n\t"
                 "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %5\n\t"
                 "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %5\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(v0), "v"(v1), "v"(v2), "v"(v3), "s"(sbase), "s"(lds_byte_addr)
                 : "memory", "scc");
}

// entry `idx` (wave-uniform, runtime) of a pointer list that lives in kernel-argument SGPRs: a chain of scalar selects
// (a dynamic index would make the compiler copy the list to scratch memory and read it back with vector loads)
__device__ __forceinline__ const unsigned char* pick_ptr(const PtrList8& l, int idx) {
    const void* p = l.p[0];
#pragma unroll
    for (int k = 1; k < 8; ++k) p = (idx == k) ? l.p[k] : p;
    return static_cast<const unsigned char*>(p);
}

// DIRECT blocks (bsmm_plan.h, 'BSU2' version 3; feature axis 1): workgroup `e` behind the schedule's multiplies quarter e % U2_DIRECT_PARTS of the
// minibatch for direct block e / U2_DIRECT_PARTS -- the block's own 64-byte row pieces only (the per-block scheme of bsmm_updat_tr.h): wave v takes
// 16-row chunks v, v + 16, ... of the quarter through a private ring of LDSB / 16 bytes (2 KiB per chunk: X rows | DY rows, the plain LDS-DMA image,
// fragments by transposing reads), one MFMA per chunk; the 16 partial tiles meet in LDS and leave as ONE partial sum in accumulator slot 0 of this
// workgroup's region, in the register order updat2_reduce_kernel expects.  ~3 us for a quarter of 2048 rows; nothing is shared, nothing is waited for.
template <class DT, int LDSB>
__device__ __forceinline__ void u2_direct_block(const PtrList8& Xs, const PtrList8& Es, float* __restrict__ scratch, const int32_t* __restrict__ plan,
                                                int N, int Cf, int Kf, int pcount, int e, long region0, unsigned char* smem) {
    typedef typename DT::T T;
    constexpr int SLOT = 2048, D = LDSB / U2_WAVES / SLOT;       // ring slots per wave: 4 (128 KiB) or 2 (64 KiB)
    static_assert(D >= 2 && LDSB >= U2_WAVES * 4096, "direct blocks: a ring of two chunks per wave and 64 KiB for the reduction");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int DP = plan[30];
    const int d = e / DP, q = e - d * DP;
    const int32_t* de = plan + plan[29] + 4 * d;
    const int c = __builtin_amdgcn_readfirstlane(de[1]), k = __builtin_amdgcn_readfirstlane(de[2]);
    unsigned char* ring = smem + wave * (D * SLOT);
    const uint32_t ring_addr = lds_addr_of(ring);
    const int drow = lane >> 2, dpiece = lane & 3;                // DMA: lane -> (row of the chunk, 16-byte piece of the 64-byte row piece)
    const int g16 = lane >> 4, t16 = lane & 15, h = g16 >> 1;     // fragments: as updat32_a1_tr_kernel
    const int rd_base = (8 * h + (t16 >> 2)) * 64 + (16 * (g16 & 1) + 4 * (t16 & 3)) * 2;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const int nch = (N + 15) >> 4;
    const int lo = (int)((long)nch * q / DP), hi = (int)((long)nch * (q + 1) / DP);
    const int first = lo + wave;
    const int myq = hi > first ? (hi - first + U2_WAVES - 1) / U2_WAVES : 0;      // chunks of a pair this wave owns
    for (int p = 0; p < pcount; ++p) {
        const T* X = reinterpret_cast<const T*>(pick_ptr(Xs, p)) + c * 32 + dpiece * 8;
        const T* E = reinterpret_cast<const T*>(pick_ptr(Es, p)) + k * 32 + dpiece * 8;
        auto issue = [&](int j, int pos) {                       // (chunks past the end: clamped re-reads of row N - 1, never multiplied)
            const int r = min((first + U2_WAVES * j) * 16 + drow, N - 1);
            const uint32_t slot = __builtin_amdgcn_readfirstlane(ring_addr + pos * SLOT);
            glds16_asm(X + (size_t)r * Cf, slot);
            glds16_asm(E + (size_t)r * Kf, slot + 1024);
        };
#pragma unroll
        for (int j = 0; j < D - 1; ++j) issue(j, j);
        int rd_pos = 0, wr_pos = D - 1;
        for (int j = 0; j < myq; ++j) {
            issue(j + D - 1, wr_pos);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (D - 1)) : "memory");     // chunk j has landed
            const unsigned char* sp = ring + rd_pos * SLOT + rd_base;
            rd_pos = (rd_pos + 1 == D) ? 0 : rd_pos + 1;
            wr_pos = (wr_pos + 1 == D) ? 0 : wr_pos + 1;
            const uint2 a0 = ds_tr16(sp), a1 = ds_tr16(sp + 4 * 64), b0 = ds_tr16(sp + 1024), b1 = ds_tr16(sp + 1024 + 4 * 64);
            uint4 a = make_uint4(a0.x, a0.y, a1.x, a1.y);
            const uint4 b = make_uint4(b0.x, b0.y, b1.x, b1.y);
            const int nb = (first + U2_WAVES * j) * 16 + 8 * h;      // K index i of this lane's fragment is row nb + i
            if (nb + 8 > N) {                                        // ragged tail: rows >= N were clamped re-reads
                uint32_t* u = reinterpret_cast<uint32_t*>(&a);
#pragma unroll
                for (int i = 0; i < 4; ++i) u[i] &= ((nb + 2 * i < N) ? 0xffffu : 0u) | ((nb + 2 * i + 1 < N) ? 0xffff0000u : 0u);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // (the next request may overwrite this slot only after the reads returned)
            acc = DT::mfma32(a, b, acc);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // the clamped requests past the end, before the ring is reused
    }
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) red[wave * 1024 + reg * 64 + lane] = acc[reg];
    __syncthreads();
    float sum = 0.f;
#pragma unroll
    for (int v = 0; v < U2_WAVES; ++v) sum += red[v * 1024 + threadIdx.x];         // element (reg = tid >> 6, lane = tid & 63), waves in order
    const int reg = threadIdx.x >> 6, ln = threadIdx.x & 63;
    // ONE partial sum of 4 KiB per direct workgroup, packed behind the regions of the schedule's workgroups of EVERY round (rounds * main grid regions)
    float* region = scratch + ((size_t)region0 * (U2_WAVES * U2_SLOTS) + (size_t)e) * 1024;
    region[((reg >> 2) * 64 + ln) * 4 + (reg & 3)] = sum;
}

// AXIS = 0: the same schedule, ring protocol, partial-sum regions and epilogue over slabs whose ROWS are the window's features
// ([WS*32 rows][64 B] per operand and chunk; the 16-byte pieces of row r XOR-swizzled with (r >> 2) & 3, conflict-free for the plain
// ds_read_b128 fragment reads: lane (row, k-half) takes 8 consecutive minibatch columns of its feature).  Needs N % 8 == 0.
template <class DT, int WS, int AXIS = 1>
__global__ void __launch_bounds__(64 * U2_WAVES, 4)
updat32_a1_v2_kernel(PtrList8 Xs, PtrList8 Es, typename DT::T* __restrict__ DW, float* __restrict__ scratch,
                     const int32_t* __restrict__ plan, int N, int Cf, int Kf, int pcount, float alpha, float beta, int flat, int main_grid, int rounds) {
    typedef typename DT::T T;
    static_assert(DT::is16 && (WS == 8 || WS == 16 || (WS == 32 && AXIS == 1)), "updat v2: 16-bit storage types, 8x8 / 16x16 windows (32x32: feature axis 1)");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int CH = AXIS == 1 ? U2_CH : U2_CH0;                // minibatch entries per chunk
    constexpr int KS = CH / 16;                                   // MFMA K-steps per chunk
    constexpr bool FIVE = AXIS == 0 && U2_RING5;                  // axis 0, experiment: five HALF slots instead (see u2_lds_bytes0)
    constexpr int D = AXIS == 1 ? u2_ring(WS) : (WS == 16 ? 2 : 4);   // ring slots (axis 0: 64 / 32 KiB each)
    constexpr int AHEAD = D / 2;                                  // prefetch distance in chunks
    constexpr int ROWB = AXIS == 1 ? WS * 64 : CH * 2;            // bytes per slab row
    constexpr int SLAB = AXIS == 1 ? CH * ROWB : WS * 32 * ROWB;  // one operand, one chunk
    constexpr int SLOT = 2 * SLAB;
    constexpr int PPR = ROWB / 16;                // 16-byte pieces per row
    constexpr int RPI = ROWB <= 1024 ? 1024 / ROWB : 1;   // rows per DMA instruction ...
    constexpr int IPR = ROWB <= 1024 ? 1 : ROWB / 1024;   // ... or instructions per row (32 x 32 windows: a row is 2 KiB)
    constexpr int IPO = SLAB / 1024;              // DMA instructions per operand and chunk
    constexpr int NI = 2 * IPO / U2_WAVES;        // DMA instructions per wave and chunk (consecutive pieces of ONE operand)
    static_assert((NI == 1 || NI == 2 || NI == 4) && NI * U2_WAVES == 2 * IPO && IPO % NI == 0, "the chunk must split evenly over the waves");

    // workgroups behind the schedule's `main_grid`: direct blocks (feature axis 1; the launcher adds them only there and only with partial sums)
    const int mg = main_grid > 0 ? main_grid : (int)gridDim.x;
    if ((int)blockIdx.x >= mg) {
        if constexpr (AXIS == 1) u2_direct_block<DT, u2_lds_bytes(WS)>(Xs, Es, scratch, plan, N, Cf, Kf, pcount, (int)blockIdx.x - mg, (long)rounds * mg, smem);
        return;
    }
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int32_t* items = plan + plan[6];
    const int nchunks = (N + CH - 1) / CH;
    const int nfull = N / CH;                                                 // chunks of a pair with all CH rows / columns
    const int CPI = pcount * nchunks;                                         // chunks per item (pairs back to back)
    // ---- schedule (see bsmm_plan.h): XCD x = blockIdx.x % 8 owns item set x / nparts and minibatch part x % nparts; its
    //      U = gridDim.x / 8 workgroups take the set's items in rounds of U, one item each over the whole part; a last
    //      incomplete round of m items is cut into floor(U / m) slices of the part per item so that every workgroup works.
    //      flat != 0: one set of all items, one part, the same rounds over all workgroups (grid = items x slices).
    const bool xcd_mode = flat == 0 && (mg & 7) == 0;
    const int nsets = xcd_mode ? plan[8] : 1;
    const int nparts = xcd_mode ? 8 / nsets : 1;
    const int U = xcd_mode ? mg >> 3 : mg;
    const int xcd = xcd_mode ? (blockIdx.x & 7) : 0, uj = xcd_mode ? (blockIdx.x >> 3) : blockIdx.x;
    const int set = xcd / nparts, part = xcd - set * nparts;
    const int set_first = xcd_mode ? plan[9 + 2 * set] : 0, set_count = xcd_mode ? plan[10 + 2 * set] : plan[4];
    const int part_lo = (int)((long)part * CPI / nparts), part_hi = (int)((long)(part + 1) * CPI / nparts);
    const int full_rounds = set_count / U, m_last = set_count - full_rounds * U;
    const int k_last = m_last > 0 ? min(U / m_last, U2_MAX_SLICES) : 0;      // slices per item of the last round
    const int total_rounds = full_rounds + ((m_last > 0 && uj < m_last * k_last) ? 1 : 0);

    const uint32_t base_addr = lds_addr_of(smem);
    // ---- DMA geometry of this wave: instructions NI*wave .. NI*wave + NI-1 of a chunk = consecutive 1 KiB pieces of one
    //      operand's slab; instruction q covers slab rows d_row0 + q*RPI (+ lane / PPR), this lane's 16-byte piece lane % PPR ----
    const int opE = (NI * wave) / IPO;                                        // 0: X slab, 1: DY slab (wave-uniform)
    const int ii0 = (NI * wave) % IPO;                                        // my first instruction of the operand's slab
    const int d_row0 = ROWB <= 1024 ? ii0 * RPI + lane / PPR : ii0 / IPR;     // its (this lane's) slab row
    auto row_of = [&](int q) -> int { return ROWB <= 1024 ? d_row0 + q * RPI : (ii0 + q) / IPR; };          // slab row of my instruction q
    auto pos_of = [&](int q) -> int { return ROWB <= 1024 ? lane % PPR : ((ii0 + q) % IPR) * 64 + lane; };  // LDS piece of the row this lane fills
    const uint32_t sub_off = ii0 * 1024;                                      // my first piece inside my operand's slab
    const uint32_t wave_off = (!FIVE ? opE * SLAB : 0) + sub_off;         // ... inside a ring slot (five half slots: the half slot carries the operand)
    constexpr uint32_t HS = SLAB, RING0 = 5 * SLAB;                           // axis 0: half slot, ring
    const int F = opE ? Kf : Cf;                                              // row length (axis 1) / row count (axis 0) of my operand
    const int my_piece0 = (lane % PPR) ^ ((d_row0 >> 2) & 3);                 // axis 0: the source piece of my 16 bytes (the same for all my instructions: RPI = 16)
    // ---- fragment geometry.  axis 1 (see bsmm_updat_tr.h): 16-lane group g16 -> features 16*(g16&1).., K half h.
    //      axis 0: lane (row r = lane & 31, K half h = lane >> 5) reads 16 bytes = 8 columns of feature row r of the block ----
    const int g16 = lane >> 4, t16 = lane & 15;
    const int h = AXIS == 1 ? g16 >> 1 : lane >> 5;
    const int trow = t16 >> 2;
    const int tsub = (2 * (g16 & 1) + ((t16 & 3) >> 1)) * 16 + (t16 & 1) * 8;
    const int frag_row = AXIS == 1 ? (8 * h + trow) * ROWB + tsub : (lane & 31) * ROWB + ((h ^ (((lane & 31) >> 2) & 3)) << 4);
#ifdef U2_NO_PINGPONG
    const bool setb = false;
#else
    // waves SB .. 15 form wave set B.  8 x 8 windows (dense layouts, feature axis 1): FOUR waves in set A, twelve in set B -- measured at 4096^2 50 %
    // (profiles/r06_updat_setb.txt): 170.8 against 176.9 us per call, every item 152 against 159 us; 16 x 16 windows: 4 / 8 / 10 / 12 within a microsecond
    constexpr int SB = U2_SETB_FROM > 0 ? U2_SETB_FROM : ((WS == 8 && AXIS == 1) ? 4 : U2_WAVES / 2);
    const bool setb = wave >= SB;
#endif

    U2_STAMP(0, __builtin_amdgcn_s_memrealtime());
    U2_STAMP(4, (unsigned long long)total_rounds);
    for (int round = 0; round < total_rounds; ++round) {
        int item, r0, cnt;
        if (round < full_rounds) {
            item = set_first + round * U + uj;
            r0 = part_lo; cnt = part_hi - part_lo;
        } else {
            const int which = uj / k_last, slice = uj - which * k_last, L = part_hi - part_lo;
            item = set_first + full_rounds * U + which;
            r0 = part_lo + (int)((long)slice * L / k_last);
            cnt = part_lo + (int)((long)(slice + 1) * L / k_last) - r0;
        }
        if (cnt <= 0) continue;
        item = __builtin_amdgcn_readfirstlane(item);           // (wave-uniform by construction; say so to the compiler)
        r0 = __builtin_amdgcn_readfirstlane(r0);
        cnt = __builtin_amdgcn_readfirstlane(cnt);
        const int32_t* it = items + (size_t)item * U2_ITEM;
        const int c0 = it[0], k0 = it[1];
        const int32_t* wd = it + 4 + wave * U2_WWORDS;
        const uint32_t meta = (uint32_t)__builtin_amdgcn_readfirstlane(wd[0]);
        const int n0 = meta & 15, n1 = (meta >> 4) & 15;

        // per-lane byte offsets of my pieces inside a regular chunk (source columns clamped inside the matrix: clamped
        // pieces belong to window columns no block of the item uses)
        // (plain scalars, not arrays: with NI == 1 hipcc left a one-element array that lambdas capture by reference in scratch memory)
        auto piece_off = [&](int q) {
            const int row = row_of(q);
            if constexpr (AXIS == 1) {
                const int piece = pos_of(q) ^ (4 * (row & 3));                    // source piece of the LDS piece this lane fills
                const int col = min((opE ? k0 : c0) * 32 + piece * 8, F - 8);
                return (uint32_t)(row * F + col) * 2u;
            } else {                                                          // row = feature row of the window (clamped inside the matrix)
                const int piece = (lane % PPR) ^ ((row >> 2) & 3);
                const int feat = min((opE ? k0 : c0) * 32 + row, F - 1);
                return ((uint32_t)feat * (uint32_t)N + (uint32_t)piece * 8u) * 2u;
            }
        };
        const uint32_t voff0 = piece_off(0), voff1 = NI > 1 ? piece_off(1) : 0u, voff2 = NI > 2 ? piece_off(2) : 0u, voff3 = NI > 2 ? piece_off(3) : 0u;
        // fragment offsets inside a ring slot
        int aoff[2], boff[U2_SLOTS];
        if constexpr (AXIS == 1) {
            // (32 x 32 windows: the fifth bit of an index rides in the id words, see the plan format)
            int c0i = (meta >> 8) & 15, c1i = (meta >> 12) & 15;
            if constexpr (WS == 32) {
                c0i |= (__builtin_amdgcn_readfirstlane(wd[1]) >> 29 & 1) << 4;
                c1i |= n1 > 0 ? (__builtin_amdgcn_readfirstlane(wd[1 + n0]) >> 29 & 1) << 4 : 0;
            }
            aoff[0] = frag_row + ((c0i ^ trow) << 6);
            aoff[1] = frag_row + ((c1i ^ trow) << 6);
#pragma unroll
            for (int j = 0; j < U2_SLOTS; ++j) {
                int ki = (meta >> (16 + 4 * j)) & 15;
                if constexpr (WS == 32) ki |= (__builtin_amdgcn_readfirstlane(wd[1 + j]) >> 30 & 1) << 4;
                boff[j] = SLAB + frag_row + ((ki ^ trow) << 6);
            }
        } else {                                  // block index * 32 rows
            aoff[0] = frag_row + (int)((meta >> 8) & 15) * (32 * ROWB);
            aoff[1] = frag_row + (int)((meta >> 12) & 15) * (32 * ROWB);
#pragma unroll
            for (int j = 0; j < U2_SLOTS; ++j) boff[j] = (FIVE ? 0 : SLAB) + frag_row + (int)((meta >> (16 + 4 * j)) & 15) * (32 * ROWB);
        }

        f32x16 acc[U2_SLOTS];
#pragma unroll
        for (int j = 0; j < U2_SLOTS; ++j)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;

        // ---- DMA issue.  Issue k (k = 0 .. cnt + U2_AHEAD - 1; the first U2_AHEAD prime the ring, one more per interval) fetches
        //      chunk min(k, cnt - 1) of the range into ring slot k % U2_D: exactly NI instructions per wave every time, so one
        //      constant vmcnt tells that a chunk has landed (the issues past the end re-fetch the last chunk into a slot nobody
        //      reads).  REGULAR issue: the chunk follows the previous one in the same pair and has all its rows -> the per-lane
        //      offsets just advance by one chunk (vector ALU), `reg_left` counts how many such issues lie ahead.  Everything
        //      else (first chunk, ragged last chunk of a pair: rows past N re-read row N - 1 and are zeroed in LDS before use,
        //      pair boundary, past the end) goes through issue_slow, which recomputes the cursor from k.
        const uint32_t fstride = AXIS == 1 ? (uint32_t)(CH * F) * 2u : (uint32_t)CH * 2u;
        // Two wave sets per SIMD (waves v, v+4 | v+8, v+12), half an interval out of phase: set A reads the fragments of
        // chunk i and multiplies them; set B multiplies the fragments it read during the PREVIOUS interval and then reads
        // chunk i's.  While one set's transposing reads are in flight the other set feeds the matrix pipe, with one barrier
        // per chunk and no second fragment set in registers.  Ring protocol: at the top of interval i chunk i has landed
        // (my share: vmcnt, everyone's: the barrier); the DMA issued in interval i fills slot (i + 2) % 4, last read in
        // interval i - 2, and set B consumed those reads (its MFMAs of interval i - 1) before it passed barrier i.
        // Scalar instructions per regular interval: the DMA block (5), its ring offset (3), reg_left and the loop (4) -- the CU
        // has ONE scalar unit for its 16 waves; the fragment-read ring offset and the source offsets advance on the vector ALU.
        auto run = [&](auto n0_tag, auto n1_tag, auto setb_tag) {
            constexpr int N0 = decltype(n0_tag)::value, N1 = decltype(n1_tag)::value;
            constexpr bool SETB = decltype(setb_tag)::value;
            uint4 a0 = zero_u4(), a1 = zero_u4(), b[(N0 + N1) ? (N0 + N1) : 1];
#pragma unroll
            for (int j = 0; j < N0 + N1; ++j) b[j] = zero_u4();   // set B multiplies "the previous fragments" from interval 0 on
            auto read_frags = [&](const unsigned char* slot, const unsigned char* slot_y, int ks) {
#ifdef U2_NO_READS
                asm volatile("" ::"s"(slot));
                return;
#endif
                if constexpr (AXIS == 1) {
                    slot += ks * 16 * ROWB;
                    {
                        const uint2 lo = ds_tr16(slot + aoff[0]), hi = ds_tr16(slot + aoff[0] + 4 * ROWB);
                        a0 = make_uint4(lo.x, lo.y, hi.x, hi.y);
                    }
                    if constexpr (N1 > 0) {
                        const uint2 lo = ds_tr16(slot + aoff[1]), hi = ds_tr16(slot + aoff[1] + 4 * ROWB);
                        a1 = make_uint4(lo.x, lo.y, hi.x, hi.y);
                    }
#pragma unroll
                    for (int j = 0; j < N0 + N1; ++j) {
                        const uint2 lo = ds_tr16(slot + boff[j]), hi = ds_tr16(slot + boff[j] + 4 * ROWB);
                        b[j] = make_uint4(lo.x, lo.y, hi.x, hi.y);
                    }
                } else {                          // K-step ks = pieces 2 ks + h of the row: the swizzled offset XOR 32 ks
                    a0 = *reinterpret_cast<const uint4*>(slot + (aoff[0] ^ (ks << 5)));
                    if constexpr (N1 > 0) a1 = *reinterpret_cast<const uint4*>(slot + (aoff[1] ^ (ks << 5)));
#pragma unroll
                    for (int j = 0; j < N0 + N1; ++j) b[j] = *reinterpret_cast<const uint4*>(slot_y + (boff[j] ^ (ks << 5)));
                }
            };
            auto multiply = [&]() {
#ifndef U2_NO_MFMA
#pragma unroll
                for (int j = 0; j < N0; ++j) acc[j] = DT::mfma32(a0, b[j], acc[j]);
                if constexpr (N1 > 0) {
#pragma unroll
                    for (int j = 0; j < N1; ++j) acc[N0 + j] = DT::mfma32(a1, b[N0 + j], acc[N0 + j]);
                }
#else
#pragma unroll
                for (int j = 0; j < N0 + N1; ++j) asm volatile("" ::"v"(b[j].x), "v"(b[j].y), "v"(b[j].z), "v"(b[j].w));
                asm volatile("" ::"v"(a0.x), "v"(a0.y), "v"(a0.z), "v"(a0.w));
                if constexpr (N1 > 0) asm volatile("" ::"v"(a1.x), "v"(a1.y), "v"(a1.z), "v"(a1.w));
#endif
            };
            // issue state: plain locals of this scope, updated only by the two macros below (lambdas that captured them by
            // reference made hipcc keep them in scratch memory, with vmcnt(0) waits around every access)
            const unsigned char* fb = nullptr;      // base pointer of the current pair in my operand
            uint32_t rv0 = 0, rv1 = 0, rv2 = 0, rv3 = 0;   // per-lane byte offsets of the NEXT regular issue
            int reg_left = 0, k_issue = 0;
#define U2_ISSUE_SLOW()                                                                                                      \
    do {                                                                                                                     \
        const int kk_ = min(k_issue, cnt - 1);                                                                               \
        const int g_ = r0 + kk_, p_ = __builtin_amdgcn_readfirstlane(g_ / nchunks), q_ = g_ - p_ * nchunks;                  \
        fb = static_cast<const unsigned char*>(uniform_ptr(opE ? pick_ptr(Es, p_) : pick_ptr(Xs, p_)));                      \
        const uint32_t dst_ = __builtin_amdgcn_readfirstlane(base_addr + wave_off + (!FIVE ? (uint32_t)(k_issue & (D - 1)) * SLOT : (uint32_t)((2 * k_issue + opE) % 5) * HS)); \
        uint32_t row_off_, sub0_, sub1_, sub2_, sub3_;                                                                       \
        if constexpr (AXIS == 1) {      /* rows past N re-read row N - 1 */                                                  \
            row_off_ = (uint32_t)(q_ * CH * F) * 2u;                                                                         \
            sub0_ = (uint32_t)(max(0, q_ * CH + row_of(0) - (N - 1)) * F) * 2u;                                              \
            sub1_ = (uint32_t)(max(0, q_ * CH + row_of(1) - (N - 1)) * F) * 2u;                                              \
            sub2_ = (uint32_t)(max(0, q_ * CH + row_of(2) - (N - 1)) * F) * 2u;                                              \
            sub3_ = (uint32_t)(max(0, q_ * CH + row_of(3) - (N - 1)) * F) * 2u;                                              \
        } else {                        /* 8-column pieces past N re-read the row's last piece (N % 8 == 0) */               \
            row_off_ = (uint32_t)(q_ * CH) * 2u;                                                                             \
            sub0_ = sub1_ = sub2_ = sub3_ = (uint32_t)max(0, q_ * CH + my_piece0 * 8 + 8 - N) * 2u;                          \
        }                                                                                                                    \
        U2_DMA1(fb, voff0 + row_off_ - sub0_, dst_);                                                                         \
        if constexpr (NI >= 2) U2_DMA1(fb, voff1 + row_off_ - sub1_, dst_ + 1024);                                           \
        if constexpr (NI == 4) U2_DMA1(fb, voff2 + row_off_ - sub2_, dst_ + 2048);                                           \
        if constexpr (NI == 4) U2_DMA1(fb, voff3 + row_off_ - sub3_, dst_ + 3072);                                           \
        rv0 = voff0 + row_off_ + fstride;                                                                                    \
        rv1 = voff1 + row_off_ + fstride;                                                                                    \
        rv2 = voff2 + row_off_ + fstride;                                                                                    \
        rv3 = voff3 + row_off_ + fstride;                                                                                    \
        reg_left = __builtin_amdgcn_readfirstlane(max(0, min(nfull - (q_ + 1), cnt - 1 - kk_)));                             \
        ++k_issue;                                                                                                           \
    } while (0)
#ifndef U2_NO_DMA
#define U2_DMA1(b_, v_, d_) glds16_saddr(b_, v_, d_)
#define U2_DMA2(b_, v0_, v1_, d_) glds16_saddr_x2(b_, v0_, v1_, d_)
#define U2_DMA4(b_, v0_, v1_, v2_, v3_, d_) glds16_saddr_x4(b_, v0_, v1_, v2_, v3_, d_)
#else
#define U2_DMA1(b_, v_, d_) asm volatile("" ::"v"(v_), "s"(d_), "s"(b_))
#define U2_DMA2(b_, v0_, v1_, d_) asm volatile("" ::"v"(v0_), "v"(v1_), "s"(d_), "s"(b_))
#define U2_DMA4(b_, v0_, v1_, v2_, v3_, d_) asm volatile("" ::"v"(v0_), "v"(v1_), "v"(v2_), "v"(v3_), "s"(d_), "s"(b_))
#endif
// one issue per interval: regular (pointer arithmetic only) or slow
#define U2_ISSUE()                                                                                                           \
    do {                                                                                                                     \
        if (reg_left > 0) {                                                                                                  \
            --reg_left; ++k_issue;                                                                                           \
            if constexpr (NI == 4)      U2_DMA4(fb, rv0, rv1, rv2, rv3, ring_base + doff);                                       \
            else if constexpr (NI == 2) U2_DMA2(fb, rv0, rv1, ring_base + doff);                                                 \
            else                        U2_DMA1(fb, rv0, ring_base + doff);                                                      \
            rv0 += fstride;                                                                                                  \
            rv1 += fstride;                                                                                                  \
            if constexpr (NI == 4) { rv2 += fstride; rv3 += fstride; }                                                       \
        } else {                                                                                                             \
            U2_ISSUE_SLOW();                                                                                                 \
        }                                                                                                                    \
        if constexpr (!FIVE) doff = (doff + SLOT) & (D * SLOT - 1);                                                      \
        else { doff += 2 * HS; if (doff >= RING0) doff -= RING0; }                                                           \
    } while (0)
            k_issue = 0; reg_left = 0;
            if constexpr (!FIVE) {
#pragma unroll
                for (int d = 0; d < AHEAD; ++d) U2_ISSUE_SLOW();
            } else {                                                // X runs two chunks ahead, DY one
                U2_ISSUE_SLOW();
                if (!opE) U2_ISSUE_SLOW();
            }
            uint32_t vslot = !FIVE ? 0u : (uint32_t)HS;         // axis 1: ring offset of the chunk; axis 0: of its DY slab (vslot0: its X slab)
            uint32_t vslot0 = 0;
            asm volatile("" : "+v"(vslot), "+v"(vslot0));           // the ring offsets of the fragment reads live on the vector ALU
            uint32_t doff = !FIVE ? (uint32_t)AHEAD * SLOT : (uint32_t)((2 * k_issue + opE) % 5) * HS;   // ring offset of the next DMA destination
            const uint32_t ring_base = __builtin_amdgcn_readfirstlane(base_addr + wave_off);
            int cq = __builtin_amdgcn_readfirstlane(r0 - (r0 / nchunks) * nchunks);   // chunk-in-pair index of the compute cursor (ragged N only)
#pragma unroll 1
            for (int i = 0; i < cnt; ++i) {
                if constexpr (!FIVE) {
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI * AHEAD - NI) : "memory");  // my share of chunk i has landed
                } else {                           // (an X wave's latest request is chunk i + 1's)
                    if (opE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    else     asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI) : "memory");
                }
                if (nfull != nchunks) {            // ragged N: is chunk i the last chunk of its pair?  zero its rows / columns >= N
                    if (cq == nfull) {
#pragma unroll
                        for (int e = 0; e < NI; ++e)
                            if (AXIS == 1 ? (cq * CH + row_of(e) >= N) : (cq * CH + my_piece0 * 8 >= N))
                                *reinterpret_cast<uint4*>(smem + (!FIVE ? (uint32_t)(i & (D - 1)) * SLOT : (uint32_t)((2 * i + opE) % 5) * HS) + wave_off + e * 1024 + lane * 16) = zero_u4();
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    }
                    if (++cq == nchunks) cq = 0;
                }
                // (two-slot ring: the slot refilled in this interval is the one set B read at the END of the previous interval --
                //  those reads must have returned before anyone may request the refill)
                if constexpr ((D == 2 || FIVE) && SETB) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifndef U2_NO_BARRIER
                __builtin_amdgcn_s_barrier();                                               // everyone's has
#endif
                const unsigned char* slot = smem + (!FIVE ? vslot : vslot0);      // X fragments
                const unsigned char* slot_y = smem + vslot;                           // DY fragments (axis 1: boff holds the slab offset)
                if constexpr (N0 == 0) {
                    U2_ISSUE();
                } else if constexpr (!SETB) {
                    read_frags(slot, slot_y, 0);
                    U2_ISSUE();
                    multiply();
#pragma unroll
                    for (int ks = 1; ks < KS; ++ks) { read_frags(slot, slot_y, ks); multiply(); }
                } else {
                    multiply();
                    __builtin_amdgcn_sched_barrier(0);
                    U2_ISSUE();
#pragma unroll
                    for (int ks = 0; ks < KS - 1; ++ks) { read_frags(slot, slot_y, ks); multiply(); }
                    read_frags(slot, slot_y, KS - 1);
                }
                if constexpr (!FIVE) {
                    vslot = (vslot + SLOT) & (D * SLOT - 1);
                } else {
                    vslot += 2 * HS;  vslot = vslot >= RING0 ? vslot - RING0 : vslot;
                    vslot0 += 2 * HS; vslot0 = vslot0 >= RING0 ? vslot0 - RING0 : vslot0;
                }
            }
#undef U2_ISSUE
#undef U2_ISSUE_SLOW
#undef U2_DMA1
#undef U2_DMA2
#undef U2_DMA4
            if constexpr (SETB && N0 > 0) multiply();           // the fragments of the last chunk
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the re-fetches past the end
            __builtin_amdgcn_s_barrier();                       // everyone has read the ring: the next range may re-prime it
        };
        using std::integral_constant;
#define RUN(A, B)                                                                                                          \
    do {                                                                                                                   \
        if (setb) run(integral_constant<int, A>{}, integral_constant<int, B>{}, integral_constant<bool, true>{});          \
        else      run(integral_constant<int, A>{}, integral_constant<int, B>{}, integral_constant<bool, false>{});         \
    } while (0)
        switch (n0 * 8 + n1) {
            case 0 * 8 + 0: RUN(0, 0); break;
            case 1 * 8 + 0: RUN(1, 0); break;
            case 1 * 8 + 1: RUN(1, 1); break;
            case 1 * 8 + 2: RUN(1, 2); break;
            case 1 * 8 + 3: RUN(1, 3); break;
            case 2 * 8 + 0: RUN(2, 0); break;
            case 2 * 8 + 1: RUN(2, 1); break;
            case 2 * 8 + 2: RUN(2, 2); break;
            case 3 * 8 + 0: RUN(3, 0); break;
            case 3 * 8 + 1: RUN(3, 1); break;
            default:        RUN(4, 0); break;
        }
#undef RUN
        if (round == 0) { U2_STAMP(1, __builtin_amdgcn_s_memrealtime()); U2_STAMP(5, (unsigned long long)cnt); }

        // D[ci][ko]: col = ko = lane & 31, row ci = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
        if (scratch != nullptr) {
            // partial sums of this workgroup's (item, minibatch range): plain 16-byte stores in REGISTER order into the region of
            // this (round, workgroup) -- [slot = wave * 4 + j][quad q][lane] float4, 1 KiB per instruction; updat2_reduce_kernel
            // knows the schedule, sums the regions of a block and undoes the order.  (Round 2 first used fp32 atomics into
            // one zeroed image: 52 MiB of cross-XCD atomics per pass, ~30 us of the 112.)
            float4* reg_base = reinterpret_cast<float4*>(scratch) + ((size_t)(round * mg + blockIdx.x) * (U2_WAVES * U2_SLOTS) + wave * U2_SLOTS) * 256 + lane;
#pragma unroll
            for (int j = 0; j < U2_SLOTS; ++j) {
                if (j >= n0 + n1) break;
#ifdef U2_NO_EPILOGUE
                if (alpha != 12345.f) continue;
#endif
#pragma unroll
                for (int q = 0; q < 4; ++q)
                {
#if U2_NT_PARTS
                    typedef float f4v __attribute__((ext_vector_type(4)));
                    const f4v v = {acc[j][4 * q + 0], acc[j][4 * q + 1], acc[j][4 * q + 2], acc[j][4 * q + 3]};
                    __builtin_nontemporal_store(v, reinterpret_cast<f4v*>(reg_base + (j * 4 + q) * 64));
#else
                    reg_base[(j * 4 + q) * 64] = make_float4(acc[j][4 * q + 0], acc[j][4 * q + 1], acc[j][4 * q + 2], acc[j][4 * q + 3]);
#endif
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < U2_SLOTS; ++j) {
                if (j >= n0 + n1) break;
                const int wid = __builtin_amdgcn_readfirstlane(wd[1 + j]) & 0x1fffffff;
                const size_t base = (size_t)wid * 1024 + (lane & 31);
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int ci = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
                    const size_t idx = base + ci * 32;
#ifdef U2_NO_EPILOGUE
                    if (alpha == 12345.f) DW[idx] = DT::from_f32(acc[j][reg]);
                    continue;
#endif
                    float out = alpha * acc[j][reg];
                    if (beta != 0.f) out += beta * DT::to_f32(DW[idx]);
                    DW[idx] = DT::from_f32(out);
                }
            }
        }
        if (round == 0) U2_STAMP(2, __builtin_amdgcn_s_memrealtime());
    }
    U2_STAMP(3, __builtin_amdgcn_s_memrealtime());
}

__device__ __forceinline__ float4 u2_ld(const float4* p) {
#if U2_NT_PARTS
    typedef float f4v __attribute__((ext_vector_type(4)));
    const f4v v = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
#else
    return *p;
#endif
}

// Second pass of the partial-sum path: one workgroup per block w sums the regions that hold a partial sum of w -- it walks
// the SAME schedule as updat32_a1_v2_kernel (keep the two in step) -- and writes DW[w] = alpha * [gate[w] *] sum + beta * DW[w]
// rounded once, or (SUMS) the raw fp32 sums [blocks][32][32] for the data-parallel all-reduce.
// thread = (quad q = tid >> 6, lane l = tid & 63): elements ci = 8 q + 4 (l >> 5) + e, e = 0..3, ko = l & 31.
template <class DT, bool SUMS>
__global__ void __launch_bounds__(128)
updat2_reduce_kernel(const float* __restrict__ parts, typename DT::T* __restrict__ DW, float* __restrict__ sums, const int32_t* __restrict__ plan,
                     const int32_t* __restrict__ bmap, const float* __restrict__ gate, int direct_region0, int main_grid, int flat, int CPI, float alpha, float beta, int q64 = 0) {
    // 128 threads = (quad pair qq = tid >> 6, lane l): quads qq and qq + 2 -- all workgroups of the bench shape are resident at once,
    // and the block map is addressed from an argument so that its load does not wait for the plan header
    const int w = blockIdx.x;
    const int qq = threadIdx.x >> 6, l = threadIdx.x & 63;
    // (`main_grid`: the schedule's workgroups -- the region index of (round, workgroup) is round * main_grid + workgroup; `direct_region0` = rounds *
    //  main_grid: behind those regions the partial sums of the direct blocks' workgroups, 4 KiB each, packed)
    const int32_t bm = bmap[w];
    const bool direct = bm <= -2;
    const int item = direct ? 0 : bm >> 8, slot = direct ? 0 : bm & 255;
    const bool xcd_mode = flat == 0 && (main_grid & 7) == 0;
    const int nsets = xcd_mode ? plan[8] : 1;
    const int nparts = xcd_mode ? 8 / nsets : 1;
    const int U = xcd_mode ? main_grid >> 3 : main_grid;
    int set = 0;
    if (xcd_mode)
        for (int s = 1; s < nsets; ++s) if (item >= plan[9 + 2 * s]) set = s;
    const int set_first = xcd_mode ? plan[9 + 2 * set] : 0, set_count = xcd_mode ? plan[10 + 2 * set] : plan[4];
    const int pos = item - set_first;
    const int full_rounds = set_count / U, m_last = set_count - full_rounds * U;
    const int k_last = m_last > 0 ? min(U / m_last, U2_MAX_SLICES) : 0;
    const float4* base = reinterpret_cast<const float4*>(parts) + ((size_t)slot * 4 + qq) * 64 + l;    // quad qq; quad qq + 2 is 128 float4 further
    constexpr size_t REGION = (size_t)U2_WAVES * U2_SLOTS * 256;       // float4 per region
    // the partial sums of this block: one region per (part, slice); independent loads in flight, no 64-bit division in the
    // walk (a first version spent 40 us at 20 % density on index arithmetic and serial loads)
    const bool sliced = pos >= full_rounds * U;
    const int which = pos - full_rounds * U;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 acc0 = zero4, acc1 = zero4;
    auto part_len = [&](int part) -> int {
        const unsigned part_lo = (unsigned)part * (unsigned)CPI / (unsigned)nparts, part_hi = (unsigned)(part + 1) * (unsigned)CPI / (unsigned)nparts;
        return (int)(part_hi - part_lo);
    };
    auto add = [](float4& a, const float4& v) { a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; };
    if (direct) {
        // a direct block: one partial sum per quarter of the minibatch, slot 0 of the regions of its U2_DIRECT_PARTS workgroups (always written)
        const int dp = plan[30];
        const float4* p0 = base + (size_t)direct_region0 * REGION + (size_t)((-2 - bm) * dp) * 256;
        float4 v0[8], v1[8];
#pragma unroll
        for (int part = 0; part < 8; ++part) {
            v0[part] = v1[part] = zero4;
            if (part < dp) { v0[part] = u2_ld(p0 + (size_t)part * 256); v1[part] = u2_ld(p0 + (size_t)part * 256 + 128); }
        }
#pragma unroll
        for (int part = 0; part < 8; ++part) { add(acc0, v0[part]); add(acc1, v1[part]); }
    } else if (!sliced) {
        // one region per minibatch part (<= 8): all loads first
        const int round = pos / U, uj = pos - round * U;
        float4 v0[8], v1[8];
#pragma unroll
        for (int part = 0; part < 8; ++part) {
            v0[part] = v1[part] = zero4;
            if (part < nparts && part_len(part) > 0) {
                const int xcd = set * nparts + part;
                const float4* p = base + ((size_t)round * main_grid + (xcd_mode ? uj * 8 + xcd : uj)) * REGION;
                v0[part] = u2_ld(p); v1[part] = u2_ld(p + 128);
            }
        }
#pragma unroll
        for (int part = 0; part < 8; ++part) { add(acc0, v0[part]); add(acc1, v1[part]); }
    } else {
        for (int part = 0; part < nparts; ++part) {
            const int L = part_len(part);
            if (L <= 0) continue;
            const int xcd = set * nparts + part;
            const float4* rb = base + (size_t)full_rounds * main_grid * REGION;
            auto where = [&](int slice) -> const float4* {
                if (slice >= k_last) return nullptr;
                if (L < k_last) {      // fewer chunks than slices: some slices are empty and wrote nothing
                    const unsigned lo = (unsigned)slice * (unsigned)L / (unsigned)k_last, hi = (unsigned)(slice + 1) * (unsigned)L / (unsigned)k_last;
                    if (hi == lo) return nullptr;
                }
                const int uj = which * k_last + slice;
                return rb + (size_t)(xcd_mode ? uj * 8 + xcd : uj) * REGION;
            };
            for (int sl = 0; sl < k_last; sl += 8) {
                float4 a[8], b[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float4* p = where(sl + e);
                    a[e] = b[e] = zero4;
                    if (p) { a[e] = u2_ld(p); b[e] = u2_ld(p + 128); }
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) { add(acc0, a[e]); add(acc1, b[e]); }
            }
        }
    }
#pragma unroll
    for (int hq = 0; hq < 2; ++hq) {
        const int q = qq + 2 * hq;
        const float4 acc = hq ? acc1 : acc0;
        const float r[4] = {acc.x, acc.y, acc.z, acc.w};
        // (q64: the blocks are the quadrants of 64 x 64 blocks, 4 w64 + 2 (row half) + (column half), bsmm_api.hip::updat64 -- the element goes to
        //  its place in the 64 x 64 block, the gate is the 64-block's: no separate pass puts the quadrants together)
        const int row = 8 * q + 4 * (l >> 5), col = l & 31;
        const size_t o = q64 ? (size_t)(w >> 2) * 4096 + (size_t)(32 * ((w >> 1) & 1) + row) * 64 + 32 * (w & 1) + col : (size_t)w * 1024 + (size_t)row * 32 + col;
        const int ostep = q64 ? 64 : 32;
        if constexpr (SUMS) {
#pragma unroll
            for (int e = 0; e < 4; ++e) sums[o + e * ostep] = r[e];
        } else {
            const float a = gate ? alpha * gate[q64 ? w >> 2 : w] : alpha;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = a * r[e];
                if (beta != 0.f) v += beta * DT::to_f32(DW[o + e * ostep]);
                DW[o + e * ostep] = DT::from_f32(v);
            }
        }
    }
}

// DW[w] = alpha * gate[w] * scratch[w] + beta * DW[w], rounded once (second pass of the scratch path; gate may be NULL)
template <class DT>
__global__ void __launch_bounds__(256)
updat_finalize_gated_kernel(const float* __restrict__ scratch, typename DT::T* __restrict__ DW, size_t n, int bsq, float alpha, float beta,
                            const float* __restrict__ gate, const int32_t* __restrict__ skip_if = nullptr) {
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n || (skip_if && skip_if[0] != 0)) return;      // (skip_if: the fp32 split paths' non-finite flag -- their repair pass writes DW then)
    const float4 s = *reinterpret_cast<const float4*>(scratch + i);
    const float a = gate ? alpha * gate[i / bsq] : alpha;
    float v[4] = {a * s.x, a * s.y, a * s.z, a * s.w};
    if (beta != 0.f) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += beta * DT::to_f32(DW[i + e]);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) DW[i + e] = DT::from_f32(v[e]);
}

}  // namespace bsmm
