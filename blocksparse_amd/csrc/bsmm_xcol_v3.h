// bsmm_xcol_v3.h -- xprop kernel "wave owns an output column", activations AND weights staged through LDS, the activation slabs
// requested TWO phases ahead ('BSX3' plans, bsmm_plan.h).  bsize 32, 16-bit storage types, both feature axes.
//
// Round 2 (bsmm_xcol_v2.h, 'BSX2': two ring halves, `vmcnt(0)` + barrier per phase) was bound by the request -> landed round trip:
// the requests for phase p+1 could only be issued behind barrier p and had to have landed at barrier p+1, and they take ~1500
// cycles from the last request (57-80 KiB per phase through a 52-64 B/clk path, first touches served by HBM / the Infinity
// Cache): a phase lasted ~2600 cycles against ~820 of matrix work per SIMD (profiles/r03_xcol_ab.md; dealing the requests to the
// idle waves or balancing the columns over the SIMDs changed nothing, concentrating the requests made it worse).  Here
//   * the LDS holds a ring of THREE phases of activation slabs (3 * PH * 16 KiB) and one POOL of weight slots in which the plan
//     gives consecutive phases disjoint slot ranges (PH = 2: 96 KiB + 31 slots, PH = 1: 48 KiB + 55 slots; + 1 slot for gates);
//   * behind barrier p a wave requests its share of the weight blocks of phase p+1, multiplies its blocks of phase p, then
//     requests its share (piece `wave` of each slab: PH instructions, always) of the slabs of phase p+2, which overwrite phase p-1's;
//   * its wait at the top of phase p+1 is `vmcnt(PH)`: the slab requests, the youngest, stay in flight across the barrier and
//     have a whole phase more to land; only the weight blocks (a third of the bytes, mostly L2 hits) are on the round trip.
//   workgroup = 16 output blocks x 128 minibatch rows, 16 waves, one output block per wave (4 row tiles x 16 accumulators);
//   per block 2 + 8 ds_read_b128 and 8 MFMAs;
//   activation slab: rows of 128 B, the eight 16-byte pieces of row r XOR-swizzled with (r >> 1) & 7;
//   weight block:   rows of 64 B, the four pieces of row r XOR-swizzled with (r >> 2) & 3 (both conflict-free for b128).
// A column accumulates its blocks in the same order with the same MFMAs as every earlier kernel: bit-identical outputs.
#pragma once
#include "bsmm_common.h"
#include "bsmm_plan.h"
#include "bsmm_updat_v2.h"   // glds16_saddr, uniform_ptr
#include "bsmm_xprop.h"      // XMap

namespace bsmm {

// measurement switches (ablation builds, scripts/build_variants.py): wrong results by construction
#ifndef X3_NO_XDMA
#define X3_NO_XDMA 0
#endif
#ifndef X3_NO_WDMA
#define X3_NO_WDMA 0
#endif
#ifndef X3_NO_READS
#define X3_NO_READS 0
#endif
#ifndef X3_NO_MFMA
#define X3_NO_MFMA 0
#endif
#ifndef X3_NO_EPILOGUE
#define X3_NO_EPILOGUE 0
#endif
#ifndef X3_SLABS_FIRST
#define X3_SLABS_FIRST 0      // 1: request the slabs of phase p+2 BEFORE multiplying phase p's blocks
#endif
// cycle stamps of the first 8 workgroups (-DBSMM_X3_TRACE, scripts/gpu_x3_trace.py): [workgroup][wave][phase < 48][6]
#ifdef BSMM_X3_TRACE
__device__ unsigned long long g_x3_trace[8 * 16 * 48 * 6];
#define X3_STAMP(k_) do { if (blockIdx.x < 8 && p < 48 && lane == 0) g_x3_trace[((blockIdx.x * 16 + wave) * 48 + p) * 6 + (k_)] = __builtin_readcyclecounter(); } while (0)
#else
#define X3_STAMP(k_) do { } while (0)
#endif
constexpr int X3_R = 128;                          // minibatch rows per workgroup
constexpr int X3_SLAB = X3_R * 128;                // 16 KiB
constexpr int X3_LDS = 163840;                     // 160 KiB
static_assert(X3_R * X3_G * 64 <= X3_LDS && x3_pool(2) == 31 && x3_pool(1) == 55, "ring, pool and epilogue tile must fit the LDS");

// TRANSW = true (fprop): Wsel is W in its natural [c-in-block][k-in-block] layout; the blocks are staged unswizzled and the
// fragment (8 consecutive c for one k per lane) is built with four transposing 8-byte reads -- no transposed copy of W, no
// pre-pass, no workspace.
// AXIS = 0: activations (C, N), minibatch contiguous: a slab is [64 feature rows] x [128 minibatch columns] (256 B per row,
// 16-byte pieces XOR-swizzled with 4 * (row & 3)), the B operand (8 consecutive FEATURES of one minibatch column per lane) is
// built with transposing reads, output rows are features.  Requires N % 8 == 0.  Same plans.
// GATED: per-block fp32 gates (hgemm_blocksparse_*_sdd's `Gate`, src/blocksparse_hgemm_cn_64_op_gpu.cu:54-66,96-124).  The wave that
// requests the first half of a weight block also fetches its gate into the gate table (one fp32 per pool slot); a block with
// gate 0 is skipped, gate 1 takes the plain path, otherwise g * w is formed in fp32 per fragment element and split into TWO
// 16-bit pieces (hi = round(g w), lo = round(g w - hi)) that are both multiplied: exact to ~2^-17 (a single bf16 rounding of
// g * w measured 2.2e-3 against the oracle, above the 1e-3 bar; the reference gates the fp32 block product,
// blocksparse/matmul.py:367-373).
// PH = steps per phase (1 or 2; the plan's choice).
template <class DT, bool TRANSW, int AXIS = 1, bool GATED = false, int PH = 2>
__global__ void __launch_bounds__(64 * X3_G, 4)
xcol32_v3_kernel(const typename DT::T* __restrict__ X, const typename DT::T* __restrict__ Wsel,
                 typename DT::T* __restrict__ Y, const int32_t* __restrict__ plan, XMap map, int N, int Cin, int Kout,
                 const float* __restrict__ gate = nullptr) {
    typedef typename DT::T T;
    static_assert(DT::is16 && (PH == 1 || PH == 2), "xcol v3 kernel: 16-bit storage types, 1 or 2 steps per phase");
    constexpr int XPHASE = PH * X3_SLAB;                // activation bytes per phase
    constexpr int WBASE = 3 * XPHASE;                   // the weight pool behind the slab ring
    constexpr int GTAB = WBASE + x3_pool(PH) * 2048;    // the gate table: fp32 per pool slot (gated calls)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int tile, grp;
    if (!xmap_decode(map, blockIdx.x, tile, grp)) return;
    const int4* ghp = reinterpret_cast<const int4*>(plan + plan[5] + X3_GROUPW * grp);
    const int4 gh = ghp[0], gc = ghp[1];
    const int ph_off = __builtin_amdgcn_readfirstlane(gh.x), nph = __builtin_amdgcn_readfirstlane(gh.y);
    const int ob0 = __builtin_amdgcn_readfirstlane(gh.z), nob = __builtin_amdgcn_readfirstlane(gh.w);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // the column (output block of the group) this wave owns
    const int col = __builtin_amdgcn_readfirstlane((int)(((uint32_t)(wave < 8 ? gc.x : gc.y) >> (4 * (wave & 7))) & 15u));
    const int32_t* pxt = plan + plan[6] + ph_off;                                                          // one word per phase
    const int4* tab = reinterpret_cast<const int4*>(plan + plan[7]) + ((size_t)ph_off * X3_G + wave) * 2;  // X3_ROW = 8 words
    const int r = lane & 31, h = lane >> 5;
    const int n_tile = tile * X3_R;
    const uint32_t base_addr = lds_addr_of(smem);
    const int npairs_full = Cin / 64;

    // activation DMA: a slab is 16 instructions of 1 KiB (8 rows of 128 B); wave v issues instruction v of each slab
    const unsigned char* xt = reinterpret_cast<const unsigned char*>(X);
    uint32_t xvoff, xvoff_tail;
    if constexpr (AXIS == 1) {
        const int row = 8 * wave + (lane >> 3);
        const int xr = min(n_tile + row, N - 1) - n_tile;            // rows past N are clamped (never stored)
        const int piece = (lane & 7) ^ ((row >> 1) & 7);
        xvoff = (uint32_t)xr * (uint32_t)Cin * 2u + piece * 16;
        xvoff_tail = xvoff - ((piece & 4) ? 64 : 0);                  // last pair of an odd block count: re-read its even half
    } else {
        // instruction v = slab rows 4v .. 4v+3 (256 B each): lane -> (row, stored piece lane & 15); columns past N are clamped
        const int row = 4 * wave + (lane >> 4);
        const int piece = (lane & 15) ^ (4 * (row & 3));
        const int colx = min(n_tile + piece * 8, N - 8) - n_tile;
        xvoff = (uint32_t)row * (uint32_t)N * 2u + (uint32_t)colx * 2u;
        xvoff_tail = (uint32_t)min(row, 31) * (uint32_t)N * 2u + (uint32_t)colx * 2u;   // missing odd block: re-read row 31 of the even one
    }
    // per pair step the source moves by 128 B (axis 1: 64 features of a row) / by 64 rows of N elements (axis 0)
    const size_t xstep = AXIS == 1 ? (size_t)128 : (size_t)N * 128;
    const unsigned char* xtile = static_cast<const unsigned char*>(uniform_ptr(xt + (AXIS == 1 ? (size_t)n_tile * Cin * 2 : (size_t)n_tile * 2)));
    const unsigned char* wsel = static_cast<const unsigned char*>(uniform_ptr(Wsel));
    // weight DMA: lane i of an instruction writes piece i of a 1 KiB half block (rows 16*hb + (i >> 2)); it fetches the
    // piece that the swizzle puts there: (i & 3) ^ ((row >> 2) & 3) = (i & 3) ^ ((i >> 4) & 3)
    const uint32_t wvoff = TRANSW ? (uint32_t)lane * 16u : (uint32_t)((lane >> 2) * 64 + (((lane & 3) ^ ((lane >> 4) & 3)) << 4));

    // fragment read offsets
    const int xsw = (r >> 1) & 7;
    uint32_t xrd[2][2];      // axis 1: [half][kk], inside a 32-row band of slab 0 of ring position 0
#pragma unroll
    for (int half = 0; half < 2; ++half)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) xrd[half][kk] = r * 128 + (((4 * half + 2 * kk + h) ^ xsw) << 4);
    // axis 0: 16-lane group g16 -> minibatch columns 16 * (g16 & 1) .. of a 32-column tile, K half g16 >> 1;
    // lane t16 points at row (t16 >> 2) of a 4-row band, 8 bytes at column 4 * (t16 & 3)
    const int g16x = lane >> 4, t16x = lane & 15, trowx = t16x >> 2;
    const int tcolbx = (16 * (g16x & 1) + 4 * (t16x & 3)) * 2;
    uint32_t wrd[2];         // [kk], inside pool slot 0
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        if constexpr (TRANSW) {   // rows 16kk + 8h + (t16 >> 2) (+4), 8 bytes at column 16 * (g16 & 1) + 4 * (t16 & 3)  (bsmm_updat_tr.h)
            wrd[kk] = WBASE + (16 * kk + 8 * h + (t16x >> 2)) * 64 + (16 * (g16x & 1) + 4 * (t16x & 3)) * 2;
        } else {
            wrd[kk] = WBASE + r * 64 + (((2 * kk + h) ^ ((r >> 2) & 3)) << 4);
        }
    }

    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

    // gated calls: lanes 0..3 fetch the gates of my weight requests (first half block of a block only); the values are written
    // into the gate table at the top of the next phase, behind the same wait as the DMAs
    float gpend = 0.f;
    uint32_t gaddr = GTAB + 63 * 4;                    // LDS address my pending gate goes to (entry 63: nobody's)
    auto fetch_gates = [&](int d0, int d1, int d2, int d3) {
        if constexpr (GATED) {
            const int dsel = lane == 0 ? d0 : (lane == 1 ? d1 : (lane == 2 ? d2 : d3));
            const bool valid = lane < 4 && dsel != -1 && !(dsel & 1);
            gpend = valid ? gate[(dsel & 0x1ffffff) >> 1] : 0.f;
            gaddr = GTAB + (valid ? (((uint32_t)dsel >> 25) >> 1) : 63u) * 4;
        }
    };

    // weight requests of one phase (absolute pool slots)
#define X3_ISSUE_W(d0_, d1_, d2_, d3_)                                                                                               \
    do {                                                                                                                             \
        if (X3_NO_WDMA) break;                                                                                                       \
        const uint32_t wdst = base_addr + WBASE;                                                                                     \
        if ((d0_) != -1) glds16_saddr(wsel + ((size_t)((d0_) & 0x1ffffff) << 10), wvoff, wdst + (((uint32_t)(d0_) >> 25) << 10));    \
        if ((d1_) != -1) glds16_saddr(wsel + ((size_t)((d1_) & 0x1ffffff) << 10), wvoff, wdst + (((uint32_t)(d1_) >> 25) << 10));    \
        if ((d2_) != -1) glds16_saddr(wsel + ((size_t)((d2_) & 0x1ffffff) << 10), wvoff, wdst + (((uint32_t)(d2_) >> 25) << 10));    \
        if ((d3_) != -1) glds16_saddr(wsel + ((size_t)((d3_) & 0x1ffffff) << 10), wvoff, wdst + (((uint32_t)(d3_) >> 25) << 10));    \
    } while (0)
    // slab requests of one phase into ring position rp_: px = pair of step 0 | pair of step 1 << 16.  ALWAYS PH instructions (a
    // phase without a second step re-requests its first pair): the waits count on it.
#define X3_ISSUE_X(px_, rp_)                                                                                                         \
    do {                                                                                                                             \
        const uint32_t xdst = base_addr + (rp_) * XPHASE + wave * 1024;                                                              \
        _Pragma("unroll") for (int u_ = 0; u_ < PH; ++u_) {                                                                          \
            int pu = (int)(((uint32_t)(px_) >> (16 * u_)) & 0xffffu);                                                                \
            if (pu == 0xffff) pu = (int)((uint32_t)(px_) & 0xffffu);                                                                 \
            if (!X3_NO_XDMA) glds16_saddr(xtile + (size_t)pu * xstep, pu < npairs_full ? xvoff : xvoff_tail, xdst + u_ * X3_SLAB);   \
        }                                                                                                                            \
    } while (0)

    // one block: weight fragment from its slot, the four row tiles' activation fragments, 8 MFMAs
    auto block = [&](uint32_t xoff, uint32_t slot, int half) {
        if (X3_NO_READS) return;
        const uint32_t woff = slot << 11;
        float g = 1.f;
        if constexpr (GATED) {
            g = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(*reinterpret_cast<const uint32_t*>(smem + GTAB + slot * 4)));
            if (g == 0.f) return;
        }
        auto xread = [&](int t, int kk) -> uint4 {
            if constexpr (AXIS == 1) {
                return *reinterpret_cast<const uint4*>(smem + xrd[half][kk] + xoff + t * 4096);
            } else {
                // rows (features) 32 * half + 16 * kk + 8 * (g16 >> 1) + {0..3 | 4..7}; row & 3 == trow for both bands
                const int row0 = 32 * half + 16 * kk + 8 * (g16x >> 1) + trowx;
                const int byte = 64 * t + tcolbx;                             // byte inside the 256-byte row (before swizzle)
                const int sw = (((byte >> 4) ^ (4 * trowx)) << 4) | (byte & 15);
                const uint2 lo = ds_tr16(smem + xoff + row0 * 256 + sw), hi = ds_tr16(smem + xoff + (row0 + 4) * 256 + sw);
                return make_uint4(lo.x, lo.y, hi.x, hi.y);
            }
        };
        uint4 wq[2];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            if constexpr (TRANSW) {
                const uint2 lo = ds_tr16(smem + wrd[kk] + woff), hi = ds_tr16(smem + wrd[kk] + woff + 4 * 64);
                wq[kk] = make_uint4(lo.x, lo.y, hi.x, hi.y);
            } else {
                wq[kk] = *reinterpret_cast<const uint4*>(smem + wrd[kk] + woff);
            }
        }
        if constexpr (GATED) {
            // one K half at a time (4 activation fragments live instead of 8: the split pieces need the registers)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                uint4 xg[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) xg[t] = xread(t, kk);
                uint4 whi = wq[kk], wlo = zero_u4();
                if (g != 1.f) {              // (gate 1 -- the usual value of a pruning mask -- needs no arithmetic: hi = w, lo = 0)
                    uint32_t hi[4], lo[4];
                    const uint32_t src[4] = {wq[kk].x, wq[kk].y, wq[kk].z, wq[kk].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float p0f = g * DT::to_f32((uint16_t)(src[e] & 0xffffu)), p1f = g * DT::to_f32((uint16_t)(src[e] >> 16));
                        const uint16_t h0 = DT::from_f32(p0f), h1 = DT::from_f32(p1f);
                        const uint16_t l0 = DT::from_f32(p0f - DT::to_f32(h0)), l1 = DT::from_f32(p1f - DT::to_f32(h1));
                        hi[e] = (uint32_t)h0 | ((uint32_t)h1 << 16);
                        lo[e] = (uint32_t)l0 | ((uint32_t)l1 << 16);
                    }
                    whi = make_uint4(hi[0], hi[1], hi[2], hi[3]); wlo = make_uint4(lo[0], lo[1], lo[2], lo[3]);
                }
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t] = DT::mfma32(whi, xg[t], acc[t]);
                if (g != 1.f) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[t] = DT::mfma32(wlo, xg[t], acc[t]);
                }
            }
            return;
        }
        uint4 xf[4][2];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int t = 0; t < 4; ++t) xf[t][kk] = xread(t, kk);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (X3_NO_MFMA) asm volatile("" ::"v"(wq[kk].x), "v"(wq[kk].w), "v"(xf[t][kk].x), "v"(xf[t][kk].w));
                else acc[t] = DT::mfma32(wq[kk], xf[t][kk], acc[t]);
            }
    };

    if (nph > 0) {
        {   // prologue: slabs and weights of phase 0, then the slabs of phase 1
            const int4 d = tab[0], e = tab[1];
            const int px0 = __builtin_amdgcn_readfirstlane(pxt[0]);
            const int d0 = __builtin_amdgcn_readfirstlane(d.z), d1 = __builtin_amdgcn_readfirstlane(d.w);
            const int d2 = __builtin_amdgcn_readfirstlane(e.x), d3 = __builtin_amdgcn_readfirstlane(e.y);
            X3_ISSUE_X(px0, 0);
            X3_ISSUE_W(d0, d1, d2, d3);
            fetch_gates(d0, d1, d2, d3);
            if (nph > 1) {
                const int px1 = __builtin_amdgcn_readfirstlane(pxt[1]);
                X3_ISSUE_X(px1, 1);
            }
        }
        int rp = 0;                                  // ring position of the current phase (phase % 3)
        for (int tb = 0; tb < nph; tb += 64) {       // lane-indexed tables for phases [tb, tb + 64)
            const int idx = min(tb + lane, nph - 1), idn = min(tb + lane + 1, nph - 1), idnn = min(tb + lane + 2, nph - 1);
            int cwv = tab[(size_t)idx * X3_G * 2].x;
            const int4 dn = tab[(size_t)idn * X3_G * 2], en = tab[(size_t)idn * X3_G * 2 + 1];
            int d0v = dn.z, d1v = dn.w, d2v = en.x, d3v = en.y, pxv = pxt[idnn];
            // the table loads must have landed before the loop: a wait the compiler places INSIDE it would drain the DMA queue
            asm volatile("" : "+v"(cwv), "+v"(d0v), "+v"(d1v), "+v"(d2v), "+v"(d3v), "+v"(pxv));
            const int tend = min(64, nph - tb);
            for (int qi = 0; qi < tend; ++qi) {
                const int p = tb + qi;
                X3_STAMP(0);
                // my shares of this phase have landed; the slab requests of phase p+1 (the youngest PH) may still be in flight
                if (p + 1 < nph) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PH) : "memory");
                else             asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                X3_STAMP(1);
                if constexpr (GATED) {                               // ... and the gates I fetched with them: into the table
                    *reinterpret_cast<float*>(smem + gaddr) = gpend;
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
                __builtin_amdgcn_s_barrier();                        // everyone's have; everyone left the previous phase
                X3_STAMP(2);
                if (p + 1 < nph) {
                    const int d0 = __builtin_amdgcn_readlane(d0v, qi), d1 = __builtin_amdgcn_readlane(d1v, qi);
                    const int d2 = __builtin_amdgcn_readlane(d2v, qi), d3 = __builtin_amdgcn_readlane(d3v, qi);
                    X3_ISSUE_W(d0, d1, d2, d3);
                    fetch_gates(d0, d1, d2, d3);
                }
                const int rpn = rp == 0 ? 2 : rp - 1;                // ring position of phase p+2 = that of phase p-1
                X3_STAMP(3);
                if (X3_SLABS_FIRST && p + 2 < nph) {
                    const int px2 = __builtin_amdgcn_readlane(pxv, qi);
                    X3_ISSUE_X(px2, rpn);
                }
                const uint32_t cw = (uint32_t)__builtin_amdgcn_readlane(cwv, qi);
                const uint32_t xo = rp * XPHASE;
                if ((cw & 0xff) != 0xff)         block(xo, cw & 0xff, 0);
                if (((cw >> 8) & 0xff) != 0xff)  block(xo, (cw >> 8) & 0xff, 1);
                if constexpr (PH > 1) {
                    if (((cw >> 16) & 0xff) != 0xff) block(xo + X3_SLAB, (cw >> 16) & 0xff, 0);
                    if ((cw >> 24) != 0xff)          block(xo + X3_SLAB, cw >> 24, 1);
                }
                X3_STAMP(4);
                if (!X3_SLABS_FIRST && p + 2 < nph) {
                    const int px2 = __builtin_amdgcn_readlane(pxv, qi);
                    X3_ISSUE_X(px2, rpn);
                }
                X3_STAMP(5);
                rp = rp == 2 ? 0 : rp + 1;
            }
        }
    }
#undef X3_ISSUE_W
#undef X3_ISSUE_X

    if constexpr (AXIS == 0) {
        // D[o][n]: col = n = r, rows o = (reg & 3) + 8 * (reg >> 2) + 4h  ->  Y[(ob * 32 + o) * N + n]: 64-byte row segments
        if (X3_NO_EPILOGUE) { if (acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] == 12345.f) Y[0] = DT::from_f32(1.f); return; }
        if (col >= nob) return;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int n = n_tile + t * 32 + r;
            if (n >= N) continue;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int o = (reg & 3) + 8 * (reg >> 2) + 4 * h;
                Y[(size_t)((ob0 + col) * 32 + o) * N + n] = DT::from_f32(acc[t][reg]);
            }
        }
        return;
    }
    // Epilogue: D[o][n]: col = n = r (lane), rows o = (reg & 3) + 8 * (reg >> 2) + 4h.  The 16 waves own the group's 16 ADJACENT
    // output blocks = 1024 contiguous bytes per minibatch row: staged through the idle LDS as [128 rows][1024 B]
    // (16-byte pieces of row n XOR-swizzled with n & 31) and stored as full rows.
    constexpr int ROWB = X3_G * 64;
    if (X3_NO_EPILOGUE) { if (acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] == 12345.f) Y[0] = DT::from_f32(1.f); return; }
    __syncthreads();
    if (col < nob) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int n = t * 32 + r;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t lo = (uint32_t)DT::from_f32(acc[t][4 * q + 0]) | ((uint32_t)DT::from_f32(acc[t][4 * q + 1]) << 16);
                const uint32_t hi = (uint32_t)DT::from_f32(acc[t][4 * q + 2]) | ((uint32_t)DT::from_f32(acc[t][4 * q + 3]) << 16);
                const int piece = col * 4 + q;
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
