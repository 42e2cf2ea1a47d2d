// bsmm_xcol16.h -- bsize 16 version of the "wave owns output columns" xprop kernel (see bsmm_xcol.h), 16-bit types.
//   workgroup = 16 consecutive 16-wide output blocks (256 features) x XC_R = 128 minibatch rows, 8 waves;
//   wave v owns output blocks 2v, 2v+1 for all rows: 2 x 8 row tiles of 16 x 4 accumulator registers.
//   step = QUAD of input blocks (64 features): the X slab is byte-for-byte the slab of a bsize-32 pair step
//   (axis 1: [128 rows][128 B], axis 0: [64 feature rows][256 B]), same DMA, swizzles, phases and ring.
//   MFMA v_mfma_f32_16x16x32: K = 32 = TWO input blocks; lane (o = lane & 15, q = lane >> 4) takes its 8 weights from
//   block (2*ks + (q >> 1)) of the quad -- a zero fragment when that block does not exist -- so one instruction covers
//   up to two 16x16 blocks (K-concatenation; the reference does this on Volta for its 8-wide blocks,
//   src/blocksparse_hgemm_cn_64_op_gpu.cu:541-624).
#pragma once
#include "bsmm_common.h"
#include "bsmm_plan.h"
#include "bsmm_xcol.h"

namespace bsmm {

// Register budget: the kernel takes 64 KB of LDS, so two workgroups fit a CU only when it stays within 128 VGPRs
// (512 threads x 2 = 4 waves per SIMD).  Holding all 8 row-tile X fragments at once costs 156 VGPRs and halves the
// occupancy (0.174 -> 0.129 ms on the 4096^2 10 % bprop); the compute loop therefore holds TH tiles at a time.
// Measured best: TH = 2 on axis 1, TH = 1 on axis 0 (the transposed LDS reads are longer-latency there).
#ifndef BSMM_XC16_OCC
#define BSMM_XC16_OCC 4
#endif
#ifndef BSMM_XC16_TH_A1
#define BSMM_XC16_TH_A1 2
#endif
#ifndef BSMM_XC16_TH_A0
#define BSMM_XC16_TH_A0 1
#endif
// NW = waves per workgroup (each owns 2 output blocks), PH = steps per phase.  <8, 2>: 512 threads, 64 KiB, two workgroups
// per CU; <16, 4> ("wide", see xcol32_a1_kernel): 1024 threads, 32 output blocks share a slab, phases of four steps.
template <class DT, int AXIS, int NW = 8, int PH = XC_PH>
__global__ void __launch_bounds__(64 * NW, BSMM_XC16_OCC)
xcol16_kernel(const typename DT::T* __restrict__ X, const typename DT::T* __restrict__ Wsel,
              typename DT::T* __restrict__ Y, const int32_t* __restrict__ plan, XMap map, int N, int Cin, int Kout) {
    typedef typename DT::T T;
    static_assert(DT::is16, "xcol16 kernel: 16-bit storage types");
    static_assert(XC_R == 128 && XC0_SLAB == XC_SLAB, "slab geometry shared with the bsize-32 kernels");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int tile, grp;
    if (!xmap_decode(map, blockIdx.x, tile, grp)) return;
    if (plan[0] != XC16PLAN_MAGIC || plan[1] != XC16PLAN_VERSION || plan[2] != 2 * NW) return;
    const int4 gh = *reinterpret_cast<const int4*>(plan + plan[5] + 4 * grp);
    const int step_off = gh.x, nsteps = gh.y, ob0 = gh.z, nob = gh.w;
    const int32_t* quads = plan + plan[6] + step_off;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int G16 = 2 * NW;                       // output blocks per workgroup
    constexpr int NI1 = XC_SLAB / 1024 / NW, NI0 = XC0_SLAB / 1024 / NW;   // DMA instructions per wave per slab (axis 1 / 0)
    constexpr int RING = 2 * PH;
    constexpr int ROWB = G16 * 32;                    // bytes per row of the axis-1 epilogue staging tile
    static_assert(NI1 >= 1 && NI0 >= 1 && 64 % RING == 0 && XC_R * ROWB <= xc_lds_bytes(NW, PH), "geometry");
    const int32_t* wt = plan + plan[7] + 4 * G16 * step_off + (size_t)(8 * wave) * nsteps;   // my 8 slots: (col, sub)
    const int o16 = lane & 15, q = lane >> 4;
    const int n_tile = tile * XC_R;
    const uint32_t base_addr = lds_addr_of(smem);

    // ---- X DMA: identical to xcol32 (a quad of 16-blocks = the 64 features of a bsize-32 pair) ----
    const int nquads_full = Cin / 64;
    const T* xsrc[NI1];
    int oddsub[2];
    int drow[NI0], dcol[NI0];
    if constexpr (AXIS == 1) {
#pragma unroll
        for (int i = 0; i < NI1; ++i) {
            const int row = 8 * (NI1 * wave + i) + (lane >> 3);
            const int xr = min(n_tile + row, N - 1);
            const int piece = (lane & 7) ^ ((row >> 1) & 7);
            xsrc[i] = X + (size_t)xr * Cin + piece * 8;
            if (i < 2) oddsub[i] = piece * 8;      // element offset of this lane's piece inside the quad's 64 features
        }
    } else {
#pragma unroll
        for (int i = 0; i < NI0; ++i) {
            const int row = XC0_RPI * (NI0 * wave + i) + lane / XC0_PPR;
            const int piece = (lane % XC0_PPR) ^ (4 * (row & 3));
            drow[i] = row;
            dcol[i] = min(n_tile + piece * 8, N - 8);
        }
    }
    auto issue_x = [&](int p, int pos) {
        if constexpr (AXIS == 1) {
            // a trailing quad may lack some blocks (Cin % 64 != 0): pieces past the row end re-read the row's last 16 bytes
#pragma unroll
            for (int i = 0; i < NI1; ++i) {
                const int over = (p < nquads_full) ? 0 : max(0, p * 64 + oddsub[i & 1] + 8 - Cin);
                glds16_asm(xsrc[i] + (p * 64 - over), __builtin_amdgcn_readfirstlane(base_addr + pos * XC_SLAB + (NI1 * wave + i) * 1024));
            }
        } else {
#pragma unroll
            for (int i = 0; i < NI0; ++i) {
                const int frow = min(p * 64 + drow[i], Cin - 1);
                glds16_asm(X + (size_t)frow * N + dcol[i], __builtin_amdgcn_readfirstlane(base_addr + pos * XC0_SLAB + (NI0 * wave + i) * 1024));
            }
        }
    };

    // ---- B (X) fragment addressing: row tile tt (16 minibatch rows), K-step ks (32 features) ----
    const int t16 = lane & 15, trow = t16 >> 2;
    auto xfrag = [&](const unsigned char* slab, int tt, int ks) -> uint4 {
        if constexpr (AXIS == 1) {   // row = 16*tt + n (n = lane & 15), piece 4*ks + q of the 128-byte row
            const int row = 16 * tt + o16;
            return *reinterpret_cast<const uint4*>(slab + row * 128 + (((4 * ks + q) ^ ((row >> 1) & 7)) << 4));
        } else {                     // transposing reads: feature rows 32*ks + 8*q + {0..3 | 4..7}, columns 16*tt + ...
            const int row0 = 32 * ks + 8 * q + trow;
            const int byte = (16 * tt + 4 * (t16 & 3)) * 2;
            const int sw = (((byte >> 4) ^ (4 * trow)) << 4) | (byte & 15);
            const uint2 lo = ds_tr16(slab + row0 * XC0_ROWB + sw), hi = ds_tr16(slab + (row0 + 4) * XC0_ROWB + sw);
            return make_uint4(lo.x, lo.y, hi.x, hi.y);
        }
    };

    f32x4 acc[2][8];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int t = 0; t < 8; ++t) acc[c][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    // W fragment of (column c, K-step ks): this lane's 8 weights come from block ids[2*ks + (q >> 1)] (or zero)
    auto load_w = [&](int w_lo, int w_hi, uint4& f) {
        const int w = (q >> 1) ? w_hi : w_lo;
        f = zero_u4();
        if (w >= 0) f = *reinterpret_cast<const uint4*>(Wsel + (size_t)w * 256 + o16 * 16 + 8 * (q & 1));
    };

    const bool own0 = 2 * wave < nob, own1 = 2 * wave + 1 < nob;
    if (nsteps > 0) {
        for (int tb = 0; tb < nsteps; tb += 64) {
            const int idx = min(tb + lane, nsteps - 1);
            const int pv = quads[idx];
            int wv[8];
#pragma unroll
            for (int s = 0; s < 8; ++s) wv[s] = ((s < 4) ? own0 : own1) ? wt[(size_t)s * nsteps + idx] : -1;
            const int tend = min(64, nsteps - tb);
            uint4 wc[2][2], wn[2][2];
            int act_c[2], act_n[2];    // bit ks set: column has a block in that K-step
            auto fetch = [&](int l, uint4 (&wf)[2][2], int (&act)[2]) {
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    act[c] = 0;
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        const int w_lo = __builtin_amdgcn_readlane(wv[4 * c + 2 * ks], l), w_hi = __builtin_amdgcn_readlane(wv[4 * c + 2 * ks + 1], l);
                        if ((w_lo & w_hi) >= 0) { load_w(w_lo, w_hi, wf[c][ks]); act[c] |= 1 << ks; }   // at least one id >= 0
                    }
                }
            };
            fetch(0, wc, act_c);
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
                    act_n[0] = act_n[1] = 0;
                    if (ss + 1 < tend) fetch(ss + 1, wn, act_n);
                    const unsigned char* slab = smem + (ss % RING) * XC_SLAB;
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        if (((act_c[0] | act_c[1]) >> ks) & 1) {
                            constexpr int TH = AXIS == 1 ? BSMM_XC16_TH_A1 : BSMM_XC16_TH_A0;      // row tiles whose X fragments are in registers at once
#pragma unroll
                            for (int t0 = 0; t0 < 8; t0 += TH) {
                                uint4 xf[TH];
#pragma unroll
                                for (int tt = 0; tt < TH; ++tt) xf[tt] = xfrag(slab, t0 + tt, ks);
#pragma unroll
                                for (int c = 0; c < 2; ++c)
                                    if ((act_c[c] >> ks) & 1) {
#pragma unroll
                                        for (int tt = 0; tt < TH; ++tt) acc[c][t0 + tt] = DT::mfma16(wc[c][ks], xf[tt], acc[c][t0 + tt]);
                                    }
                            }
                        }
                    }
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        act_c[c] = act_n[c];
#pragma unroll
                        for (int ks = 0; ks < 2; ++ks) wc[c][ks] = wn[c][ks];
                    }
                }
            }
            __syncthreads();
        }
    }

    // D[o][n]: col = n = lane & 15 (row of tile tt), rows o = 4q + reg
    if constexpr (AXIS == 1) {
        // staged through LDS and stored as full rows (see xcol32_a1_kernel)
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            if (!((c == 0) ? own0 : own1)) continue;
#pragma unroll
            for (int tt = 0; tt < 8; ++tt) {
                const int n = 16 * tt + o16;
                const uint32_t lo = (uint32_t)DT::from_f32(acc[c][tt][0]) | ((uint32_t)DT::from_f32(acc[c][tt][1]) << 16);
                const uint32_t hi = (uint32_t)DT::from_f32(acc[c][tt][2]) | ((uint32_t)DT::from_f32(acc[c][tt][3]) << 16);
                const int piece = (2 * wave + c) * 2 + (q >> 1);          // 16 features x 2 B = 2 pieces per block
                *reinterpret_cast<uint2*>(smem + n * ROWB + ((piece ^ (n & 31)) << 4) + 8 * (q & 1)) = make_uint2(lo, hi);
            }
        }
        __syncthreads();
        const int rowbytes = nob * 32;
        T* ybase = Y + (size_t)ob0 * 16;
        constexpr int PPR = ROWB / 16;
        for (int i = threadIdx.x; i < XC_R * PPR; i += 64 * NW) {
            const int n = i / PPR, piece = i % PPR;
            if (n_tile + n < N && piece * 16 < rowbytes) {
                const uint4 v = *reinterpret_cast<const uint4*>(smem + n * ROWB + ((piece ^ (n & 31)) << 4));
                *reinterpret_cast<uint4*>(reinterpret_cast<unsigned char*>(ybase + (size_t)(n_tile + n) * Kout) + piece * 16) = v;
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
                    Y[(size_t)((ob0 + 2 * wave + c) * 16 + 4 * q + reg) * N + n] = DT::from_f32(acc[c][tt][reg]);
            }
        }
    }
}

}  // namespace bsmm
