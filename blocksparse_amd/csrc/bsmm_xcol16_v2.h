// bsmm_xcol16_v2.h -- bsize 16 xprop kernel with the weight blocks staged through LDS ('BSX7' plans): the scheme of
// bsmm_xcol_v2.h for 16x16 blocks, 16-bit storage types, both feature axes.
//
// bsmm_xcol16.h fetched a wave's weight fragments into registers one step ahead with ordinary loads; the vector-memory counter
// is in order, so the wait in front of every step also drained the activation slabs of the next phase.  Here, as in
// bsmm_xcol_v2.h, everything a phase needs comes by LDS-DMA one phase ahead and there is one wait per phase:
//   workgroup = 32 output blocks (512 features) x 128 minibatch rows, 16 waves, wave v owns output blocks 2v, 2v+1 for all rows
//   (2 x 8 row tiles of 16 x 4 accumulator registers); step = QUAD of input blocks (64 features), the activation slab is
//   byte-for-byte the slab of a bsize-32 pair step; phase = up to two steps and up to X7_WCAP weight blocks;
//   LDS = 2 halves x (2 slabs of 16 KiB + 96 slots of 512 B) = 160 KiB.
//   A K-step is a PAIR of input blocks (32 features).  When a column has both blocks of the pair, one v_mfma_f32_16x16x32
//   K-concatenates them (lane (o, q) takes its 8 weights from block 2 * ks + (q >> 1), its 8 activations from the same 16
//   features); when it has only one -- 18 % of the pairs against 1 % at 10 % density -- the block is multiplied on its own by
//   v_mfma_f32_16x16x16 (same issue time, lane (o, q) holds k = 4q .. 4q+3) from 8-byte fragment reads of exactly its 16
//   features: half the LDS fragment bytes of a zero-padded K = 32 instruction, and the activations of the ABSENT partner are
//   never touched (round 2 multiplied a zero weight fragment against them: 0 * Inf = NaN where the reference never reads
//   that feature, blocksparse/matmul.py:353-392).
//   Weight DMA: one instruction = two blocks (lanes 0..31 / 32..63), slots 2j and 2j+1.
#pragma once
#include "bsmm_common.h"
#include "bsmm_plan.h"
#include "bsmm_updat_v2.h"   // glds16_saddr, uniform_ptr
#include "bsmm_xcol.h"       // slab geometry
#include "bsmm_xcol.h"
// row tiles whose X fragments the K = 32 pair loop holds in registers at once (measured on the retired round-1 kernel and kept: 2 on
// feature axis 1, 1 on axis 0 where the transposing LDS reads take longer; all 8 tiles = 156 VGPRs halved the occupancy)
#ifndef BSMM_XC16_TH_A1
#define BSMM_XC16_TH_A1 2
#endif
#ifndef BSMM_XC16_TH_A0
#define BSMM_XC16_TH_A0 1
#endif

namespace bsmm {

constexpr int X7_SLAB = XC_SLAB;                       // 16 KiB
constexpr int X7_XHALF = 2 * X7_SLAB;
constexpr int X7_WHALF = (X7_WCAP + 2) * 512;          // the phase's slots, the zero slot, one spare
constexpr int X7_WBASE = 2 * X7_XHALF;
constexpr int X7_LDS = X7_WBASE + 2 * X7_WHALF;        // 160 KiB
static_assert(X7_LDS <= 163840 && XC_R * X7_G * 32 <= X7_LDS && X7_WCAP % 2 == 0, "ring and epilogue tile must fit the LDS");

// D[o][n] of a wave's 2 x 8 accumulator tiles: col = n = lane & 15 (row of tile tt), rows o = 4q + reg
template <class DT, int AXIS>
__device__ __forceinline__ void x7_epilogue(f32x4 (&acc)[2][8], unsigned char* smem, typename DT::T* __restrict__ Y, int lane, int wave, int n_tile,
                                            const int32_t* __restrict__ cols, int N, int Kout) {
    // (`cols`: the group's 32 output blocks by column position, -1 = none -- 'BSX7' version 3, round 6: consecutive blocks, or the builder's pick for
    //  an unbalanced layout; adjacent pairs stay together, so a wave's two columns are neighbours either way)
    typedef typename DT::T T;
    const int o16 = lane & 15, q = lane >> 4;
    const int ob_0 = __builtin_amdgcn_readfirstlane(cols[2 * wave]), ob_1 = __builtin_amdgcn_readfirstlane(cols[2 * wave + 1]);
    const bool own0 = ob_0 >= 0, own1 = ob_1 >= 0;
    if constexpr (AXIS == 1) {
        constexpr int ROWB = X7_G * 32;       // staged through LDS and stored as full rows (see xcol32_a1_kernel)
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            if (!((c == 0) ? own0 : own1)) continue;
#pragma unroll
            for (int tt = 0; tt < 8; ++tt) {
                const int n = 16 * tt + o16;
                const uint32_t lo = (uint32_t)DT::from_f32(acc[c][tt][0]) | ((uint32_t)DT::from_f32(acc[c][tt][1]) << 16);
                const uint32_t hi = (uint32_t)DT::from_f32(acc[c][tt][2]) | ((uint32_t)DT::from_f32(acc[c][tt][3]) << 16);
                const int piece = (2 * wave + c) * 2 + (q >> 1);
                *reinterpret_cast<uint2*>(smem + n * ROWB + ((piece ^ (n & 31)) << 4) + 8 * (q & 1)) = make_uint2(lo, hi);
            }
        }
        __syncthreads();
        constexpr int PPR = ROWB / 16;
        for (int i = threadIdx.x; i < XC_R * PPR; i += 1024) {
            const int n = i / PPR, piece = i % PPR;
            const int ob = cols[piece >> 1];                    // (a 16-wide output block is two 16-byte pieces of the row)
            if (n_tile + n < N && ob >= 0) {
                const uint4 v = *reinterpret_cast<const uint4*>(smem + n * ROWB + ((piece ^ (n & 31)) << 4));
                *reinterpret_cast<uint4*>(reinterpret_cast<unsigned char*>(Y + (size_t)(n_tile + n) * Kout + (size_t)ob * 16) + (piece & 1) * 16) = v;
            }
        }
    } else {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            if (!((c == 0) ? own0 : own1)) continue;
#pragma unroll
            for (int tt = 0; tt < 8; ++tt) {
                const int n = n_tile + 16 * tt + o16;
                if (n >= N) continue;
#pragma unroll
                for (int reg = 0; reg < 4; ++reg)
                    Y[(size_t)((c ? ob_1 : ob_0) * 16 + 4 * q + reg) * N + n] = DT::from_f32(acc[c][tt][reg]);
            }
        }
    }
}

// GATED (NOT INSTANTIATED since round 6: it spilled 22 registers and ran 3.3x slower than the ungated list kernel; gated bsize-16 calls are
// bsmm_gate_weights + the ungated call, or the per-segment kernel): per-block fp32 gates (the reference gates all three tensor-core block sizes,
// src/blocksparse_hgemm_cn_64_op_gpu.cu:256-717).
// The wave that requests a pair of weight blocks also fetches their gates into the ring half's gate table (the two spare slots);
// a block with gate 0 is treated as absent, gate 1 takes the plain path, otherwise g * w is formed per fragment element in fp32 and
// split into TWO 16-bit pieces that are both multiplied (exact to ~2^-17; see bsmm_xcol_v2.h).
template <class DT, int AXIS, bool GATED = false>
__global__ void __launch_bounds__(1024, 4)
xcol16_v2_kernel(const typename DT::T* __restrict__ X, const typename DT::T* __restrict__ Wsel,
                 typename DT::T* __restrict__ Y, const int32_t* __restrict__ plan, XMap map, int N, int Cin, int Kout,
                 const float* __restrict__ gate = nullptr) {
    typedef typename DT::T T;
    static_assert(DT::is16, "xcol16 v2 kernel: 16-bit storage types");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int tile, grp;
    if (!xmap_decode(map, blockIdx.x, tile, grp)) return;
    const int4 gh = *reinterpret_cast<const int4*>(plan + plan[5] + 4 * grp);
    const int ph_off = __builtin_amdgcn_readfirstlane(gh.x), nph = __builtin_amdgcn_readfirstlane(gh.y);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int32_t* pxt = plan + plan[6] + ph_off;
    const int4* tab = reinterpret_cast<const int4*>(plan + plan[7]) + ((size_t)ph_off * 16 + wave) * 3;   // X7_ROW = 12 words
    const int o16 = lane & 15, q = lane >> 4;
    const int n_tile = tile * XC_R;
    const uint32_t base_addr = lds_addr_of(smem);
    const int nquads_full = Cin / 64;

    // activation DMA: wave v issues instruction v of each slab (16 x 1 KiB), geometry of bsmm_xcol_v2.h
    uint32_t xvoff, xvoff_tail;
    if constexpr (AXIS == 1) {
        const int row = 8 * wave + (lane >> 3);
        const int xr = min(n_tile + row, N - 1) - n_tile;
        const int piece = (lane & 7) ^ ((row >> 1) & 7);
        xvoff = (uint32_t)xr * (uint32_t)Cin * 2u + piece * 16;
        // a trailing quad may lack blocks (Cin % 64 != 0): pieces past the row end re-read the row's last 16 bytes
        xvoff_tail = xvoff - 2u * (uint32_t)max(0, nquads_full * 64 + piece * 8 + 8 - Cin);
    } else {
        const int row = 4 * wave + (lane >> 4);
        const int piece = (lane & 15) ^ (4 * (row & 3));
        const int col = min(n_tile + piece * 8, N - 8) - n_tile;
        xvoff = (uint32_t)row * (uint32_t)N * 2u + (uint32_t)col * 2u;
        xvoff_tail = (uint32_t)min(row, max(0, Cin - nquads_full * 64 - 1)) * (uint32_t)N * 2u + (uint32_t)col * 2u;
    }
    const size_t xstep = AXIS == 1 ? (size_t)128 : (size_t)N * 128;
    const unsigned char* xtile = static_cast<const unsigned char*>(
        uniform_ptr(reinterpret_cast<const unsigned char*>(X) + (AXIS == 1 ? (size_t)n_tile * Cin * 2 : (size_t)n_tile * 2)));
    const unsigned char* wsel = static_cast<const unsigned char*>(uniform_ptr(Wsel));
    const uint32_t wlane = (uint32_t)(lane & 31) * 16u;     // my piece of the 512-byte block (lanes 0..31: block A, 32..63: block B)

    // B (X) fragment of row tile tt (16 minibatch rows), K-step ks (32 features) -- as xcol16_kernel
    const int t16 = lane & 15, trow = t16 >> 2;
    auto xfrag = [&](const unsigned char* slab, int tt, int ks) -> uint4 {
        if constexpr (AXIS == 1) {
            const int row = 16 * tt + o16;
            return *reinterpret_cast<const uint4*>(slab + row * 128 + (((4 * ks + q) ^ ((row >> 1) & 7)) << 4));
        } else {
            const int row0 = 32 * ks + 8 * q + trow;
            const int byte = (16 * tt + 4 * (t16 & 3)) * 2;
            const int sw = (((byte >> 4) ^ (4 * trow)) << 4) | (byte & 15);
            const uint2 lo = ds_tr16(slab + row0 * XC0_ROWB + sw), hi = ds_tr16(slab + (row0 + 4) * XC0_ROWB + sw);
            return make_uint4(lo.x, lo.y, hi.x, hi.y);
        }
    };
    // ... of ONE block of the K-step (hf = 0 / 1: features 32 ks + 16 hf .. + 16) for the K = 16 instruction: lane (n, q) holds
    // features 4q .. 4q+3 of the block
    auto xfrag16 = [&](const unsigned char* slab, int tt, int ks, int hf) -> uint2 {
        if constexpr (AXIS == 1) {
            const int row = 16 * tt + o16;
            return *reinterpret_cast<const uint2*>(slab + row * 128 + (((4 * ks + 2 * hf + (q >> 1)) ^ ((row >> 1) & 7)) << 4) + (q & 1) * 8);
        } else {
            const int row0 = 32 * ks + 16 * hf + 4 * q + trow;
            const int byte = (16 * tt + 4 * (t16 & 3)) * 2;
            const int sw = (((byte >> 4) ^ (4 * (row0 & 3))) << 4) | (byte & 15);
            return ds_tr16(slab + row0 * XC0_ROWB + sw);
        }
    };
    // A (W) fragment: row o16 of the block.  K = 32: 8 weights at 8 * (q & 1) of slot s_lo (q < 2) or s_hi (q >= 2); K = 16: 4 weights at 4q
    const uint32_t wfrag_lane = (uint32_t)(o16 * 32 + (q & 1) * 16);
    const uint32_t wfrag16_lane = (uint32_t)(o16 * 32 + q * 8);

    f32x4 acc[2][8];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int t = 0; t < 8; ++t) acc[c][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    // gated calls: lanes 0..5 fetch the gates of the (up to) six blocks my three duties request; the values are written into the
    // ring half's gate table (fp32 per slot, in the two spare slots behind the weight slots) at the top of the next phase, behind
    // the same wait as the DMAs
    constexpr int GTAB = X7_WCAP * 512;                // byte offset of the gate table inside a weight ring half
    float gpend = 0.f;
    uint32_t gaddr = X7_WBASE + GTAB + 255 * 4;        // entry 255: nobody's
    // (a macro over scalars: an array passed to a lambda by reference ends up in scratch memory)
#define X7_FETCH_GATES(d0_, d1_, d2_, d3_, d4_, d5_, hbn_)                                                                          \
    do {                                                                                                                             \
        if constexpr (GATED) {                                                                                                       \
            const int i_ = lane >> 1;                                            /* duty of this lane (lanes 0..5) */                \
            const int da_ = i_ == 0 ? (d0_) : (i_ == 1 ? (d2_) : (d4_)), db_ = i_ == 0 ? (d1_) : (i_ == 1 ? (d3_) : (d5_));          \
            const bool valid_ = lane < 6 && da_ != -1;                                                                               \
            const int blk_ = (lane & 1) ? db_ : (da_ & 0x3ffffff);                                                                   \
            const uint32_t slot_ = 2u * ((uint32_t)da_ >> 26) + (lane & 1);                                                          \
            gpend = valid_ ? gate[blk_] : 0.f;                                                                                       \
            gaddr = X7_WBASE + (hbn_) * X7_WHALF + GTAB + (valid_ ? slot_ : 255u) * 4;                                               \
        }                                                                                                                            \
    } while (0)
    auto gate_of = [&](int hbc, uint32_t slot) -> float {
        return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(*reinterpret_cast<const uint32_t*>(smem + X7_WBASE + hbc * X7_WHALF + GTAB + slot * 4)));
    };
    // (hi, lo) 16-bit pieces of g * w per element: hi = round(g w), lo = round(g w - hi)
    auto split2 = [&](uint32_t src, float g, uint32_t& hi, uint32_t& lo) {
        const float p0f = g * DT::to_f32((uint16_t)(src & 0xffffu)), p1f = g * DT::to_f32((uint16_t)(src >> 16));
        const uint16_t h0 = DT::from_f32(p0f), h1 = DT::from_f32(p1f);
        const uint16_t l0 = DT::from_f32(p0f - DT::to_f32(h0)), l1 = DT::from_f32(p1f - DT::to_f32(h1));
        hi = (uint32_t)h0 | ((uint32_t)h1 << 16);
        lo = (uint32_t)l0 | ((uint32_t)l1 << 16);
    };

#define X7_WDUTY(a_, b_)                                                                                                   \
    if ((a_) != -1) {                                                                                                       \
        const uint32_t oa = ((uint32_t)(a_) & 0x3ffffffu) << 9, ob = (uint32_t)(b_) << 9;                                   \
        X7_DMA(wsel, (lane < 32 ? oa : ob) + wlane, wdst + (((uint32_t)(a_) >> 26) << 10));                           \
    }
#ifdef X7_NO_DMA        // ablation builds (scripts/build_variants.py): what is left without the DMA / without the matrix work
#define X7_DMA(b_, v_, d_) asm volatile("" ::"s"(b_), "v"(v_), "s"(d_))
#else
#define X7_DMA(b_, v_, d_) glds16_saddr(b_, v_, d_)
#endif
#define X7_ISSUE(px_, d_, hb_)                                                                                              \
    do {                                                                                                                    \
        const uint32_t xdst = base_addr + (hb_) * X7_XHALF + wave * 1024;                                                   \
        const uint32_t wdst = base_addr + X7_WBASE + (hb_) * X7_WHALF;                                                      \
        const int p0 = (px_) & 0xffff, p1 = (int)((uint32_t)(px_) >> 16);                                                   \
        X7_DMA(xtile + (size_t)p0 * xstep, p0 < nquads_full ? xvoff : xvoff_tail, xdst);                              \
        if (p1 != 0xffff) X7_DMA(xtile + (size_t)p1 * xstep, p1 < nquads_full ? xvoff : xvoff_tail, xdst + X7_SLAB);  \
        X7_WDUTY(d_[0], d_[1]) X7_WDUTY(d_[2], d_[3]) X7_WDUTY(d_[4], d_[5])                                                \
    } while (0)

    if (nph > 0) {
        {   // prologue: phase 0 into ring half 0
            const int4 da = tab[1], db = tab[2];
            const int px0 = __builtin_amdgcn_readfirstlane(pxt[0]);
            const int d[6] = {__builtin_amdgcn_readfirstlane(da.x), __builtin_amdgcn_readfirstlane(da.y), __builtin_amdgcn_readfirstlane(da.z),
                              __builtin_amdgcn_readfirstlane(da.w), __builtin_amdgcn_readfirstlane(db.x), __builtin_amdgcn_readfirstlane(db.y)};
            X7_ISSUE(px0, d, 0);
            X7_FETCH_GATES(d[0], d[1], d[2], d[3], d[4], d[5], 0);
        }
        int hb = 0;
        for (int tb = 0; tb < nph; tb += 64) {       // lane-indexed tables for phases [tb, tb + 64)
            const int idx = min(tb + lane, nph - 1), idn = min(tb + lane + 1, nph - 1);
            const int4 c4 = tab[(size_t)idx * 48];
            int cw0 = c4.x, cw1 = c4.y, cw2 = c4.z, cw3 = c4.w;
            const int4 da = tab[(size_t)idn * 48 + 1], db = tab[(size_t)idn * 48 + 2];
            int d0 = da.x, d1 = da.y, d2 = da.z, d3 = da.w, d4 = db.x, d5 = db.y, pxv = pxt[idn];
            // the table loads must have landed before the loop: a wait the compiler places INSIDE it would drain the DMA queue
            asm volatile("" : "+v"(cw0), "+v"(cw1), "+v"(cw2), "+v"(cw3), "+v"(pxv));
            asm volatile("" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5));
            const int tend = min(64, nph - tb);
            for (int qi = 0; qi < tend; ++qi) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // my DMA shares of this phase have landed
                if constexpr (GATED) {                               // ... and the gates I fetched with them: into this half's table
                    *reinterpret_cast<float*>(smem + gaddr) = gpend;
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
                __builtin_amdgcn_s_barrier();                        // everyone's have; everyone left the previous phase
                if (tb + qi + 1 < nph) {
                    const int px1 = __builtin_amdgcn_readlane(pxv, qi);
                    const int d[6] = {__builtin_amdgcn_readlane(d0, qi), __builtin_amdgcn_readlane(d1, qi), __builtin_amdgcn_readlane(d2, qi),
                                      __builtin_amdgcn_readlane(d3, qi), __builtin_amdgcn_readlane(d4, qi), __builtin_amdgcn_readlane(d5, qi)};
                    X7_ISSUE(px1, d, hb ^ 1);
                    X7_FETCH_GATES(d[0], d[1], d[2], d[3], d[4], d[5], hb ^ 1);
                } else if constexpr (GATED) {
                    gaddr = X7_WBASE + GTAB + 255 * 4;
                }
                const uint32_t cw[4] = {(uint32_t)__builtin_amdgcn_readlane(cw0, qi), (uint32_t)__builtin_amdgcn_readlane(cw1, qi),
                                        (uint32_t)__builtin_amdgcn_readlane(cw2, qi), (uint32_t)__builtin_amdgcn_readlane(cw3, qi)};
                const unsigned char* wring = smem + X7_WBASE + hb * X7_WHALF;
#pragma unroll
                for (int u = 0; u < 2; ++u) {
#ifdef X7_NO_COMPUTE
                    if (N != 12345) continue;
#endif
                    if ((cw[2 * u] & cw[2 * u + 1]) == 0xffffffffu) continue;        // nothing of mine in this step
                    const unsigned char* slab = smem + hb * X7_XHALF + u * X7_SLAB;
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        // One column at a time.  Per column: 0 = no block in this pair, 1 = only the first, 2 = only the second, 3 = both.
                        // A lone block is multiplied by the K = 16 instruction from 8-byte fragments of exactly its features: TH16 row
                        // tiles' reads are issued together (one LDS latency per block, not per pair of MFMAs).
                        constexpr int TH32 = GATED ? 1 : (AXIS == 1 ? BSMM_XC16_TH_A1 : BSMM_XC16_TH_A0);
                        constexpr int TH16 = GATED ? (AXIS == 1 ? 4 : 2) : (AXIS == 1 ? 8 : 4);
#pragma unroll
                        for (int c = 0; c < 2; ++c) {
                            const uint32_t pairb = (cw[2 * u + c] >> (16 * ks)) & 0xffffu;    // slots of sub-blocks 2ks, 2ks+1
                            if (pairb == 0xffffu) continue;
                            const uint32_t lo = pairb & 0xff, hi = pairb >> 8;
                            int pat = (lo != 0xff ? 1 : 0) | (hi != 0xff ? 2 : 0);
                            float g_lo = 1.f, g_hi = 1.f;
                            if constexpr (GATED) {
                                if (pat & 1) { g_lo = gate_of(hb, lo); if (g_lo == 0.f) pat &= ~1; }   // gate 0: the block is skipped
                                if (pat & 2) { g_hi = gate_of(hb, hi); if (g_hi == 0.f) pat &= ~2; }
                                if (pat == 0) continue;
                            }
                            if (pat == 3) {
                                uint4 w4 = *reinterpret_cast<const uint4*>(wring + wfrag_lane + (((q >> 1) ? hi : lo) << 9)), l4 = zero_u4();
                                bool two = false;
                                if constexpr (GATED) {
                                    two = g_lo != 1.f || g_hi != 1.f;
                                    if (two) {
                                        const float g = (q >> 1) ? g_hi : g_lo;
                                        uint4 h4;
                                        split2(w4.x, g, h4.x, l4.x); split2(w4.y, g, h4.y, l4.y); split2(w4.z, g, h4.z, l4.z); split2(w4.w, g, h4.w, l4.w);
                                        w4 = h4;
                                    }
                                }
#pragma unroll
                                for (int t0 = 0; t0 < 8; t0 += TH32) {
                                    uint4 xf[TH32];
#pragma unroll
                                    for (int tt = 0; tt < TH32; ++tt) xf[tt] = xfrag(slab, t0 + tt, ks);
#pragma unroll
                                    for (int tt = 0; tt < TH32; ++tt) acc[c][t0 + tt] = DT::mfma16(w4, xf[tt], acc[c][t0 + tt]);
                                    if constexpr (GATED) {
                                        if (two) {
#pragma unroll
                                            for (int tt = 0; tt < TH32; ++tt) acc[c][t0 + tt] = DT::mfma16(l4, xf[tt], acc[c][t0 + tt]);
                                        }
                                    }
                                }
                            } else {
                                const int hf = pat - 1;
                                uint2 w2 = *reinterpret_cast<const uint2*>(wring + wfrag16_lane + ((hf ? hi : lo) << 9)), l2 = make_uint2(0u, 0u);
                                bool two = false;
                                if constexpr (GATED) {
                                    const float g = hf ? g_hi : g_lo;
                                    two = g != 1.f;
                                    if (two) {
                                        uint2 h2;
                                        split2(w2.x, g, h2.x, l2.x); split2(w2.y, g, h2.y, l2.y);
                                        w2 = h2;
                                    }
                                }
#pragma unroll
                                for (int t0 = 0; t0 < 8; t0 += TH16) {
                                    uint2 xh[TH16];
#pragma unroll
                                    for (int tt = 0; tt < TH16; ++tt) xh[tt] = xfrag16(slab, t0 + tt, ks, hf);
#pragma unroll
                                    for (int tt = 0; tt < TH16; ++tt) acc[c][t0 + tt] = DT::mfma16k16(w2, xh[tt], acc[c][t0 + tt]);
                                    if constexpr (GATED) {
                                        if (two) {
#pragma unroll
                                            for (int tt = 0; tt < TH16; ++tt) acc[c][t0 + tt] = DT::mfma16k16(l2, xh[tt], acc[c][t0 + tt]);
                                        }
                                    }
                                }
                            }
                        }
                    }
                }
                hb ^= 1;
            }
        }
    }
    x7_epilogue<DT, AXIS>(acc, smem, Y, lane, wave, n_tile, plan + plan[12] + X7_G * grp, N, Kout);
}

// ------------------------------------------------------------------------------------------------------------------
// List-driven variant ('BSX7' plans of version 2 carry, per phase and wave, the blocks of the wave's two columns as a LIST, and per
// phase ONE request table).  What the ablation builds and cycle stamps of the kernel above said (profiles/r03_x7_*.txt, 4096^2 /
// 10 % / N = 8192): without the DMAs the pass still takes 106-117 of its 114-128 us, without the matrix work 62-71 -- the pass
// is bound by how a wave gets through a phase, not by delivery:
//   * per block one dependent chain  scalar position test -> weight fragment read -> 8 fragment reads -> 8 MFMAs, nothing of the
//     next block in flight, 16 position tests per phase for 1.6 blocks;
//   * the request section in front of it: every wave issues 2-5 LDS-DMA instructions behind ~60 scalar instructions; with 16 waves
//     issuing at once an instruction holds its wave ~120 cycles (scripts/micro/dma_issue2.hip) and a SIMD issues one scalar
//     instruction per four cycles: 700-1900 cycles per phase in which the wave multiplies nothing.
// Here:
//   * a wave that issues requests is HELD by them: the memory pipeline takes them only as fast as it delivers (a phase's 45 KB at
//     ~30 B/clk = ~1500 cycles).  So FOUR waves issue ALL requests of the next phase from one lane-indexed table -- the four with
//     the fewest blocks in the current phase, named by the plan (one in five (wave, phase) pairs has no block at all at 10 %) --
//     and the other twelve go from the barrier straight to their blocks;
//   * the plan lists a wave's blocks per column (words: activation-slab bits, weight slot offset), the row of the NEXT phase is
//     fetched lane-indexed with one load during the current one, a visit costs two v_readlane and a handful of vector ops;
//   * a visit is software-pipelined in halves of 4 row tiles: [reads tiles 4-7 of e] [4 MFMAs tiles 0-3] [reads weights and
//     tiles 0-3 of e + 1] [4 MFMAs tiles 4-7];
//   * every block is multiplied on its own by the K = 16 instruction (a pair of blocks = two visits).
// Activation fragment address: lane part XOR (position bits | slab bit | ring-half bit) -- the swizzles of both slab layouts are
// XORs of disjoint bit fields, see xfrag16 above; tile tt: + 2048 tt (axis 1) / XOR 32 tt (axis 0).
// ------------------------------------------------------------------------------------------------------------------
#ifdef X7L_TRACE
// cycle stamps of the first 8 workgroups: [wg][wave][phase < 40][5] = top, after the wait, after the barrier, after the requests,
// after the list (s_memtime); read back with bsmm_debug_x7_trace_copy().  Debug builds only.
__device__ unsigned long long g_x7_trace[8 * 16 * 40 * 5];
#define X7L_STAMP(k) do { if (blockIdx.x < 8 && ph < 40 && lane == 0) g_x7_trace[((blockIdx.x * 16 + wave) * 40 + ph) * 5 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define X7L_STAMP(k) do { } while (0)
#endif
#ifdef X7_NO_DMA
#define X7L_DMA4(b_, v0_, v1_, v2_, v3_, d_) asm volatile("" ::"s"(b_), "v"(v0_), "v"(v1_), "v"(v2_), "v"(v3_), "s"(d_))
#else
#define X7L_DMA4(b_, v0_, v1_, v2_, v3_, d_) glds16_saddr_x4(b_, v0_, v1_, v2_, v3_, d_)
#endif
// TRANSW: the staged blocks are W's own (fprop multiplies by the transpose): the weight fragment is then read with the transposing
// read -- 16-lane group q points at rows 4q .. 4q+3 of the 16 x 16 block, lane t receives column t -- instead of from a transposed
// copy of W that a pre-pass wrote (5.5 us and a launch per fprop at BASELINE configs[2]).
template <class DT, int AXIS, bool TRANSW>
__global__ void __launch_bounds__(1024, 4)
xcol16_list_kernel(const typename DT::T* __restrict__ X, const typename DT::T* __restrict__ Wsel,
                   typename DT::T* __restrict__ Y, const int32_t* __restrict__ plan, XMap map, int N, int Cin, int Kout) {
    typedef typename DT::T T;
    static_assert(DT::is16, "xcol16 list kernel: 16-bit storage types");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int tile, grp;
    if (!xmap_decode(map, blockIdx.x, tile, grp)) return;
    const int4 gh = *reinterpret_cast<const int4*>(plan + plan[5] + 4 * grp);
    const int ph_off = __builtin_amdgcn_readfirstlane(gh.x), nph = __builtin_amdgcn_readfirstlane(gh.y);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // list section of my group: per phase X7_PHW words = [16 waves][X7_LIST] block lists, [64][2] request table
    const unsigned char* sect = static_cast<const unsigned char*>(uniform_ptr(plan + plan[11] + (size_t)ph_off * X7_PHW));
    const uint32_t llane = (uint32_t)(wave * X7_LIST + min(lane, X7_LIST - 1)) * 4u;    // my word of my list row (lanes 0..39 are read)
    const uint32_t rlane = (uint32_t)(16 * X7_LIST + 2 * lane) * 4u;                     // my entry of the request table
    const int o16 = lane & 15, q = lane >> 4;
    const int n_tile = tile * XC_R;
    const uint32_t base_addr = lds_addr_of(smem);
    const int nquads_full = Cin / 64;

    // activation requests: issuer iss fetches pieces 4 iss .. 4 iss + 3 (1 KiB each) of every slab of the phase; geometry of a piece
    // as in xcol16_v2_kernel (there: piece = wave)
    const uint32_t xrow_bytes = AXIS == 1 ? (uint32_t)Cin * 2u : (uint32_t)N * 2u;
    auto x_offset = [&](int iss, int e, bool tail) -> uint32_t {
        const int j = 4 * iss + e;
        if constexpr (AXIS == 1) {
            const int row = 8 * j + (lane >> 3);
            const int xr = min(n_tile + row, N - 1) - n_tile;
            const int piece = (lane & 7) ^ ((row >> 1) & 7);
            const uint32_t v = (uint32_t)xr * xrow_bytes + piece * 16;
            // a trailing quad may lack blocks (Cin % 64 != 0): pieces past the row end re-read the row's last 16 bytes
            return tail ? v - 2u * (uint32_t)max(0, nquads_full * 64 + piece * 8 + 8 - Cin) : v;
        } else {
            const int row = 4 * j + (lane >> 4);
            const int piece = (lane & 15) ^ (4 * (row & 3));
            const int col = min(n_tile + piece * 8, N - 8) - n_tile;
            return (uint32_t)(tail ? min(row, max(0, Cin - nquads_full * 64 - 1)) : row) * xrow_bytes + (uint32_t)col * 2u;
        }
    };
    const size_t xstep = AXIS == 1 ? (size_t)128 : (size_t)N * 128;
    const unsigned char* xtile = static_cast<const unsigned char*>(
        uniform_ptr(reinterpret_cast<const unsigned char*>(X) + (AXIS == 1 ? (size_t)n_tile * Cin * 2 : (size_t)n_tile * 2)));
    const unsigned char* wsel = static_cast<const unsigned char*>(uniform_ptr(Wsel));
    const uint32_t wlane = (uint32_t)(lane & 31) * 16u;

    // fragment addressing (LDS byte addresses; the ring half is folded into the lane parts once per phase)
    const int t16 = lane & 15, trow = t16 >> 2;
    uint32_t xl;          // lane part of the activation fragment address, tile 0
    if constexpr (AXIS == 1) xl = (uint32_t)(o16 * 128 + ((((q >> 1) ^ ((o16 >> 1) & 7))) << 4) + (q & 1) * 8);
    else                     xl = (uint32_t)((4 * q + trow) * XC0_ROWB + (((((t16 & 3) >> 1) ^ (4 * trow))) << 4) + 8 * (t16 & 1));
    const uint32_t wl = (uint32_t)(X7_WBASE + (TRANSW ? (4 * q + trow) * 32 + (t16 & 3) * 8 : o16 * 32 + q * 8));
    auto wread = [&](uint32_t wa) -> uint2 {
        if constexpr (TRANSW) return ds_tr16(smem + wa);
        else                  return *reinterpret_cast<const uint2*>(smem + wa);
    };
    constexpr uint32_t XMASK = AXIS == 1 ? ((3u << 5) | (1u << 14)) : ((3u << 12) | (1u << 14));
    auto xread = [&](uint32_t xa, int tt) -> uint2 {
        if constexpr (AXIS == 1) return *reinterpret_cast<const uint2*>(smem + xa + 2048 * tt);
        else                     return ds_tr16(smem + (xa ^ (uint32_t)(tt << 5)));
    };

    f32x4 acc[2][8];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int t = 0; t < 8; ++t) acc[c][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    // all requests of one phase (table row in rA / rB: lane k = byte offsets of the blocks of slots 2k, 2k+1; lane 48 = px, npairs)
    // into ring half hb_: my four pieces of each activation slab, and every fourth pair of weight blocks
#define X7L_REQUESTS(rA_, rB_, hb_, iss_)                                                                                        \
    do {                                                                                                                         \
        const uint32_t px_ = (uint32_t)__builtin_amdgcn_readlane(rA_, 48);                                                       \
        const int np_ = __builtin_amdgcn_readlane(rB_, 48);                                                                      \
        const int p0_ = px_ & 0xffff, p1_ = (int)(px_ >> 16);                                                                    \
        const uint32_t xdst_ = base_addr + (hb_) * X7_XHALF + (iss_) * 4096;                                                     \
        {                                                                                                                        \
            const bool t_ = p0_ >= nquads_full;                                                                                  \
            X7L_DMA4(xtile + (size_t)p0_ * xstep, x_offset(iss_, 0, t_), x_offset(iss_, 1, t_), x_offset(iss_, 2, t_), x_offset(iss_, 3, t_), xdst_); \
        }                                                                                                                        \
        if (p1_ != 0xffff) {                                                                                                     \
            const bool t_ = p1_ >= nquads_full;                                                                                  \
            X7L_DMA4(xtile + (size_t)p1_ * xstep, x_offset(iss_, 0, t_), x_offset(iss_, 1, t_), x_offset(iss_, 2, t_), x_offset(iss_, 3, t_), xdst_ + X7_SLAB); \
        }                                                                                                                        \
        const uint32_t wdst_ = base_addr + X7_WBASE + (hb_) * X7_WHALF;                                                          \
        for (int k_ = (iss_); k_ < np_; k_ += 4) {                                                                               \
            const uint32_t oa_ = (uint32_t)__builtin_amdgcn_readlane(rA_, k_), ob_ = (uint32_t)__builtin_amdgcn_readlane(rB_, k_); \
            X7_DMA(wsel, (lane < 32 ? oa_ : ob_) + wlane, wdst_ + (uint32_t)k_ * 1024u);                                         \
        }                                                                                                                        \
    } while (0)
    // (plain loads between asm statements with memory clobbers: they stay where they are written, and their first use is the copy at
    //  the loop's back edge, i.e. the compiler's own wait lands next to the vmcnt(0) at the top of the next phase.  Loading through
    //  inline asm instead lets the compiler copy the destination register while the data is still in flight.
    //  The pointers are cast to the global address space: a FLAT load would also count on lgkmcnt and stall the fragment reads.)
    typedef const uint32_t __attribute__((address_space(1))) * gptr1_t;
    typedef const unsigned long long __attribute__((address_space(1))) * gptr2_t;
#define X7L_LOAD1(dst_, voff_, base_) dst_ = *reinterpret_cast<gptr1_t>(reinterpret_cast<uintptr_t>((base_) + (voff_)))
#define X7L_LOAD2(dst_, voff_, base_)                                                                                             \
    do {                                                                                                                         \
        const unsigned long long t_ = *reinterpret_cast<gptr2_t>(reinterpret_cast<uintptr_t>((base_) + (voff_)));                \
        dst_ = make_uint2((uint32_t)t_, (uint32_t)(t_ >> 32));                                                                   \
    } while (0)
#ifdef X7L_NO_PHASES
    if (nph > 0 && N == 12345) {
#else
    if (nph > 0) {
#endif
        uint32_t lnext;                 // my words of my list row of the next phase (lane-indexed)
        uint2 rnext = make_uint2(0u, 0u);   // request table of the phase after the next, if its requests will be mine
        {   // prologue: the requests of phase 0 (by waves 12..15) into ring half 0; the issuers of phase 0 fetch the table of phase 1
            if (wave >= 12) {
                uint2 r0;
                X7L_LOAD2(r0, rlane, sect);
                X7L_REQUESTS(r0.x, r0.y, 0, wave & 3);
            }
            if (((uint32_t)reinterpret_cast<const int32_t*>(sect)[wave * X7_LIST + 38] >> 16) != 0 && nph > 1) {
                const unsigned char* nrow = static_cast<const unsigned char*>(uniform_ptr(sect + (size_t)X7_PHW * 4));
                X7L_LOAD2(rnext, rlane, nrow);
            }
            X7L_LOAD1(lnext, llane, sect);
        }
        int hb = 0;
#pragma unroll 1
        for (int ph = 0; ph < nph; ++ph) {
            X7L_STAMP(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // my requests for this phase have landed (my list row and request table too)
            const uint32_t lcur = lnext;
            X7L_STAMP(1);
            __builtin_amdgcn_s_barrier();                        // everyone's have; everyone left the previous phase
            X7L_STAMP(2);
            const uint32_t cnt = (uint32_t)__builtin_amdgcn_readlane(lcur, 38);   // blocks of my columns | my role in this phase
            if (ph + 1 < nph) {
                const unsigned char* nrow = static_cast<const unsigned char*>(uniform_ptr(sect + (size_t)(ph + 1) * (X7_PHW * 4)));
                const int my_iss = (int)((cnt >> 16) & 0xff);
                if (my_iss != 0) X7L_REQUESTS(rnext.x, rnext.y, hb ^ 1, my_iss - 1);
                // (after my requests: a wave that issues in two consecutive phases reads the old table first)
                if ((((uint32_t)__builtin_amdgcn_readlane(lcur, 39) >> 16) & 0xff) != 0 && ph + 2 < nph) X7L_LOAD2(rnext, rlane, nrow + X7_PHW * 4);
                X7L_LOAD1(lnext, llane, nrow);
            }
                X7L_STAMP(3);
#ifdef X7_NO_COMPUTE
                const int n0 = N != 12345 ? 0 : (cnt & 0xff), n1 = N != 12345 ? 0 : ((cnt >> 8) & 0xff);
#else
                const int n0 = cnt & 0xff, n1 = (cnt >> 8) & 0xff;
#endif
                if (n0 + n1 > 0) {
#ifdef X7L_FULL_AHEAD
                    // one block ahead: the 9 fragment reads of block e + 1 are issued before the 8 MFMAs of block e, two register sets
                    const uint32_t xlh = xl ^ (uint32_t)(hb << 15), wlh = wl + (uint32_t)(hb * X7_WHALF);
                    uint2 wA, wB, fA[8], fB[8];
#define X7_FETCH(W_, F_, e_)                                                                                                     \
    do {                                                                                                                         \
        const uint32_t sx_ = (uint32_t)__builtin_amdgcn_readlane(lcur, 2 * (e_)) & XMASK, sw_ = (uint32_t)__builtin_amdgcn_readlane(lcur, 2 * (e_) + 1); \
        const uint32_t xa_ = xlh ^ sx_;                                                                                          \
        W_ = wread(wlh + sw_);                                                                  \
        _Pragma("unroll") for (int t = 0; t < 8; ++t) F_[t] = xread(xa_, t);                                                     \
    } while (0)
#define X7_VISIT(C)                                                                                                              \
    do {                                                                                                                         \
        ++e;                                                                                                                     \
        X7_FETCH(wB, fB, e);                  /* (past the end: zero words of the row's tail -- a valid address, not used) */     \
        __builtin_amdgcn_sched_barrier(0);                                                                                       \
        _Pragma("unroll") for (int t = 0; t < 8; ++t) acc[C][t] = DT::mfma16k16(wA, fA[t], acc[C][t]);                          \
        __builtin_amdgcn_sched_barrier(0);                                                                                       \
        wA = wB;                                                                                                                 \
        _Pragma("unroll") for (int t = 0; t < 8; ++t) fA[t] = fB[t];                                                             \
    } while (0)
                    X7_FETCH(wA, fA, 0);
                    int e = 0;
#pragma unroll 1
                    for (int k = 0; k < n0; ++k) X7_VISIT(0);
#pragma unroll 1
                    for (int k = 0; k < n1; ++k) X7_VISIT(1);
#undef X7_VISIT
#undef X7_FETCH
#else
                    // software pipeline in halves of 4 row tiles: [reads tiles 4-7 of e] [4 MFMAs tiles 0-3] [reads weights and tiles 0-3 of
                    // e + 1] [4 MFMAs tiles 4-7].  (Measured: fetching the whole next block ahead -- X7L_FULL_AHEAD, two register sets and 9
                    // 64-bit moves per block -- is slower, 91 against 81 us.)
                    const uint32_t xlh = xl ^ (uint32_t)(hb << 15), wlh = wl + (uint32_t)(hb * X7_WHALF);
                    int e = 0;
                    uint2 w, xA[4], xB[4];
                    uint32_t xa;
                    {
                        const uint32_t sx = (uint32_t)__builtin_amdgcn_readlane(lcur, 0) & XMASK, sw = (uint32_t)__builtin_amdgcn_readlane(lcur, 1);
                        xa = xlh ^ sx;
                        w = wread(wlh + sw);
#ifndef X7L_NO_READS
#pragma unroll
                        for (int t = 0; t < 4; ++t) xA[t] = xread(xa, t);
#else
#pragma unroll
                        for (int t = 0; t < 4; ++t) xA[t] = make_uint2(xa, xa + t);
#endif
                    }
#ifndef X7L_NO_READS
#define X7L_XREAD(a_, t_) xread(a_, t_)
#else
#define X7L_XREAD(a_, t_) make_uint2(a_, a_ + t_)
#endif
#ifndef X7L_NO_MFMA
#define X7L_MFMA(w_, x_, c_) c_ = DT::mfma16k16(w_, x_, c_)
#else
#define X7L_MFMA(w_, x_, c_) asm volatile("" : "+v"(c_) : "v"(w_), "v"(x_))
#endif
#define X7_VISIT(C)                                                                                                              \
    do {                                                                                                                         \
        _Pragma("unroll") for (int t = 0; t < 4; ++t) xB[t] = X7L_XREAD(xa, 4 + t);                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                                       \
        _Pragma("unroll") for (int t = 0; t < 4; ++t) X7L_MFMA(w, xA[t], acc[C][t]);                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                                       \
        ++e;                                                                                                                     \
        const uint32_t sx_ = (uint32_t)__builtin_amdgcn_readlane(lcur, 2 * e) & XMASK, sw_ = (uint32_t)__builtin_amdgcn_readlane(lcur, 2 * e + 1); \
        xa = xlh ^ sx_;                                                                                                          \
        const uint2 wn_ = wread(wlh + sw_);                                                     \
        _Pragma("unroll") for (int t = 0; t < 4; ++t) xA[t] = X7L_XREAD(xa, t);                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                                       \
        _Pragma("unroll") for (int t = 0; t < 4; ++t) X7L_MFMA(w, xB[t], acc[C][4 + t]);                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                                       \
        w = wn_;                                                                                                                 \
    } while (0)
#pragma unroll 1
                    for (int k = 0; k < n0; ++k) X7_VISIT(0);
#pragma unroll 1
                    for (int k = 0; k < n1; ++k) X7_VISIT(1);
#undef X7_VISIT
#undef X7L_XREAD
#undef X7L_MFMA
#endif
                }
            X7L_STAMP(4);
            hb ^= 1;
        }
    }
#undef X7L_REQUESTS
#undef X7L_LOAD1
#undef X7L_LOAD2
    x7_epilogue<DT, AXIS>(acc, smem, Y, lane, wave, n_tile, plan + plan[12] + X7_G * grp, N, Kout);
}

#undef X7L_DMA4
#undef X7L_STAMP
#undef X7_ISSUE
#undef X7_DMA
#undef X7_FETCH_GATES
#undef X7_WDUTY

}  // namespace bsmm
