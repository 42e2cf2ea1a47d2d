// bsmm_xgroup.h -- grouped ("panel-stationary") xprop kernel for bsize 32, 16-bit types.
//
// Why: in the one-segment-per-workgroup kernel (bsmm_xprop.h) every 32x32 block product fetches its own
// X fragment from L2, so the kernel is bound by L2->CU bandwidth at <10% of the MFMA rate.  Here a
// workgroup owns G consecutive output blocks and one minibatch tile, walks the union of their input blocks
// ONCE (host-built plan, bsmm_plan.h) and re-uses each X fragment -- held in registers -- for every nonzero
// block (c, ob) of the group: L2->CU bytes per MFMA drop by the average multiplicity m = G*d/(1-(1-d)^G).
// The weight blocks of a stage are streamed into LDS with global_load_lds (LDS-DMA, no VGPR round trip)
// and shared by the 4 waves (which split the minibatch tile, so all waves do identical work per step).
//
// LDS image of one weight block: 32 rows (o) x 64 B, the four 16-byte slots of a row XOR-swizzled with
// (row >> 2) & 3.  LDS-DMA writes lane-linearly, so the swizzle is applied to the per-lane SOURCE address and
// again to the ds_read_b128 address (guide rule 21); with it the fragment read is bank-conflict free.
#pragma once
#include "bsmm_common.h"
#include "bsmm_plan.h"
#include "bsmm_xprop.h"

namespace bsmm {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

struct PlanView {
    const int32_t* groups;   // [ngroups][4]
    const int32_t* stages;   // [nstages][4]
    const int32_t* steps;    // [nsteps][2]
    const int32_t* wlist;    // [nblocks]
};

constexpr int XG_G = 8;      // output blocks per group of the axis-0 kernel (plan[2])
constexpr int XG_SB = 16;    // weight blocks per LDS stage (plan[3]): 32 KiB per buffer

// ---- axis 1: grouped kernel "S3" (pair steps) -------------------------------------------------------
// Measured on MI355X (profiles/): the grouped kernels are bound by the L2 REQUEST rate (~11 of 16 requests/clk/XCD),
// and a 32-feature bf16 row piece is only a 64-byte half line.  S3 therefore walks PAIRS of adjacent input blocks:
// one step fetches, per minibatch row, the full 128-byte line holding blocks (2p, 2p+1), and multiplies it with every
// nonzero block of the group that lives in either of the two input-block rows.
//   256 threads = 4 waves, two workgroups per CU; wave v owns rows [tile*128 + 32v, +32) and all G accumulators.
//   X: wave-private LDS-DMA ring of XS_D slots; slot = 32 rows x 128 B, the eight 16-byte pieces of a row XOR-swizzled
//      with (row >> 1) & 7 (conflict-free ds_read_b128 at a 128-byte row stride); DMA for step s+XS_D-1 is issued at
//      the top of step s, `vmcnt(4*(XS_D-1))` then guarantees step s's slot has landed (in-order completion).
//   W: one stage (<= XG_SB blocks, 2 KiB images, 4-slot swizzle) at a time, DMA'd by all waves between two barriers.
//   Tables (pair index and member mask per step, weight ids) are loaded one stage ahead as lane-indexed vectors and read
//   back with v_readlane.  All DMAs come from inline asm (guide 5.7): hipcc neither counts nor drains them.
constexpr int XS_D = 3;

__device__ __forceinline__ void glds16_asm(const void* gsrc, uint32_t lds_byte_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_byte_addr)
                 : "memory");
}
__device__ __forceinline__ uint32_t lds_addr_of(const void* p) {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}

struct XS3 {
    static constexpr int NT = 128;          // minibatch rows per workgroup
    static constexpr int SLOT = 32 * 128;   // bytes per ring slot
    static constexpr int LDS = XG_SB * 2048 + 4 * XS_D * SLOT;
};

template <class DT, int G>
__global__ void __launch_bounds__(256, 2)
xs3_a1_kernel(const typename DT::T* __restrict__ X, const typename DT::T* __restrict__ Wsel,
              typename DT::T* __restrict__ Y, const int32_t* __restrict__ plan, XMap map, int N, int Cin, int Kout) {
    typedef typename DT::T T;
    static_assert(DT::is16 && G <= 16, "grouped kernel: 16-bit storage types, <= 16 members");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // [XG_SB*2048 weight stage][4 waves][XS_D][SLOT]
    int tile, grp;
    if (!xmap_decode(map, blockIdx.x, tile, grp)) return;
    if (plan[0] != PLAN_MAGIC || plan[1] != PLAN_VERSION || plan[2] != G || plan[3] > XG_SB || plan[14] != 1) return;
    const int32_t* groups = plan + plan[8];
    const int32_t* stages = plan + plan[9];
    const int32_t* steps = plan + plan[10];
    const int32_t* wlist = plan + plan[11];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int4 gh = *reinterpret_cast<const int4*>(groups + 4 * grp);
    const int stage_beg = gh.x, nstages = gh.y, ob0 = gh.z, nob = gh.w;
    const int n_wave = tile * XS3::NT + wave * 32;
    const int n = n_wave + r;

    // weight block image: 32 rows x 64 B, 4 pieces per row swizzled with (row >> 2) & 3
    const int wsw = (r >> 2) & 3;
    const int wrd0 = r * 64 + ((h ^ wsw) << 4);
    const int wrd1 = r * 64 + (((2 + h) ^ wsw) << 4);
    // X slot image: 32 rows x 128 B, 8 pieces per row swizzled with (row >> 1) & 7; piece = 4*half + 2*khalf + h
    const int xsw = (r >> 1) & 7;
    int xrd[2][2];
#pragma unroll
    for (int half = 0; half < 2; ++half)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) xrd[half][kk] = r * 128 + (((4 * half + 2 * kk + h) ^ xsw) << 4);

    unsigned char* xring = smem + XG_SB * 2048 + wave * (XS_D * XS3::SLOT);
    const uint32_t xring_addr = lds_addr_of(xring);
    const uint32_t wbuf_addr = lds_addr_of(smem);
    // X DMA: instruction i covers rows 8i .. 8i+7; lane -> (row 8i + (lane >> 3), stored piece lane & 7)
    const int npairs_full = Cin / 64;          // pairs whose odd block exists
    const T* xsrc[4];
    int oddsub[2];                             // a trailing pair without an odd block: odd pieces re-read the even block
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = 8 * i + (lane >> 3);
        const int xr = min(n_wave + row, N - 1);                     // rows past N are clamped (never stored)
        const int piece = (lane & 7) ^ ((row >> 1) & 7);             // source piece that lands in stored piece lane & 7
        xsrc[i] = X + (size_t)xr * Cin + piece * 8;
        if (i < 2) oddsub[i] = (piece & 4) ? 32 : 0;                 // piece(i+2) == piece(i)
    }
    const int dslot = lane & 3, drow = lane >> 2;
    const int wsrc0 = drow * 32 + ((dslot ^ ((drow >> 2) & 3)) << 3);                 // element offsets inside a block
    const int wsrc1 = (16 + drow) * 32 + ((dslot ^ (((16 + drow) >> 2) & 3)) << 3);

    f32x16 acc[G];
#pragma clang loop unroll(full)
    for (int g = 0; g < G; ++g)
#pragma clang loop unroll(full)
        for (int i = 0; i < 16; ++i) acc[g][i] = 0.f;

    if (nstages > 0) {
        const int first_step = stages[4 * stage_beg];
        const int4 last_stage = *reinterpret_cast<const int4*>(stages + 4 * (stage_beg + nstages - 1));
        const int end_step = last_stage.x + last_stage.y;   // one past the group's last step

        auto issue_x = [&](int pos, int p) {                 // pos = ring slot index
            const uint32_t slot = __builtin_amdgcn_readfirstlane(xring_addr + pos * XS3::SLOT);
            const bool full = p < npairs_full;
#pragma unroll
            for (int i = 0; i < 4; ++i) glds16_asm(xsrc[i] + (p * 64 - (full ? 0 : oddsub[i & 1])), slot + i * 1024);
        };

        int4 sg = *reinterpret_cast<const int4*>(stages + 4 * stage_beg);
        int cv = steps[2 * min(sg.x + lane, end_step - 1)];        // pair index of steps sg.x .. sg.x+63 (clamped)
        int mv = steps[2 * min(sg.x + lane, end_step - 1) + 1];    // member mask of those steps
        int wv = (lane < sg.w) ? wlist[sg.z + lane] : 0;           // weight ids of the stage
#pragma unroll
        for (int d = 0; d < XS_D - 1; ++d) issue_x(d, __builtin_amdgcn_readlane(cv, min(d, end_step - 1 - first_step)));
        int rd_pos = 0, wr_pos = XS_D - 1;   // ring slot of the current step / of the step being prefetched

        for (int st = 0; st < nstages; ++st) {
            const int step_beg = sg.x, nsteps = sg.y, nw = sg.w;
            int4 sgn = sg;
            int cvn = 0, mvn = 0, wvn = 0;
            if (st + 1 < nstages) {   // next stage's tables: plain loads, complete by the vmcnt(0) below
                sgn = *reinterpret_cast<const int4*>(stages + 4 * (stage_beg + st + 1));
                cvn = steps[2 * min(sgn.x + lane, end_step - 1)];
                mvn = steps[2 * min(sgn.x + lane, end_step - 1) + 1];
                wvn = (lane < sgn.w) ? wlist[sgn.z + lane] : 0;
            }
            __syncthreads();   // every wave has left the previous stage's weight blocks
            for (int q = wave; q < 2 * nw; q += 4) {
                const int w = __builtin_amdgcn_readlane(wv, q >> 1);
                glds16_asm(Wsel + (size_t)w * 1024 + ((q & 1) ? wsrc1 : wsrc0), __builtin_amdgcn_readfirstlane(wbuf_addr + q * 1024));
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            asm volatile("" ::"v"(cvn), "v"(mvn), "v"(wvn));   // any compiler wait for the table loads goes HERE (free)

            int lds_off = 0;
            for (int s = 0; s < nsteps; ++s) {
                const uint32_t mask = (uint32_t)__builtin_amdgcn_readlane(mv, s);
                issue_x(wr_pos, __builtin_amdgcn_readlane(cv, min(s + XS_D - 1, end_step - 1 - step_beg)));
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (XS_D - 1)) : "memory");   // this step's slot has landed
                const unsigned char* slot = xring + rd_pos * XS3::SLOT;
                rd_pos = (rd_pos + 1 == XS_D) ? 0 : rd_pos + 1;
                wr_pos = (wr_pos + 1 == XS_D) ? 0 : wr_pos + 1;
                Frag32<DT> xf[2];
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    xf[half].q[0] = *reinterpret_cast<const uint4*>(slot + xrd[half][0]);
                    xf[half].q[1] = *reinterpret_cast<const uint4*>(slot + xrd[half][1]);
                }
#pragma clang loop unroll(full)
                for (int g = 0; g < G; ++g) {
#pragma clang loop unroll(full)
                    for (int half = 0; half < 2; ++half) {
                        if ((mask >> (2 * g + half)) & 1) {
                            Frag32<DT> wf;
                            wf.q[0] = *reinterpret_cast<const uint4*>(smem + lds_off + wrd0);
                            wf.q[1] = *reinterpret_cast<const uint4*>(smem + lds_off + wrd1);
                            lds_off += 2048;
                            mma32<DT>(wf, xf[half], acc[g]);
                        }
                    }
                }
                // (the slot is re-filled two steps from now; our reads of it were consumed by the MFMAs above)
            }
            sg = sgn;
            cv = cvn;
            mv = mvn;
            wv = wvn;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // trailing (clamped) prefetches must not outlive the workgroup
    }

    if (n >= N) return;
#pragma clang loop unroll(full)
    for (int g = 0; g < G; ++g) {
        if (g < nob) {
            T* yrow = Y + (size_t)n * Kout + (ob0 + g) * 32 + 4 * h;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                uint32_t lo = (uint32_t)DT::from_f32(acc[g][4 * q + 0]) | ((uint32_t)DT::from_f32(acc[g][4 * q + 1]) << 16);
                uint32_t hi = (uint32_t)DT::from_f32(acc[g][4 * q + 2]) | ((uint32_t)DT::from_f32(acc[g][4 * q + 3]) << 16);
                *reinterpret_cast<uint2*>(yrow + 8 * q) = make_uint2(lo, hi);
            }
        }
    }
}

// ---- axis 0 (and the first, register-direct version): X fragments by ordinary loads ---------------
template <class DT, int AXIS, int G>
__global__ void __launch_bounds__(256, 2)
xgroup32_kernel(const typename DT::T* __restrict__ X, const typename DT::T* __restrict__ Wsel,
                typename DT::T* __restrict__ Y, const int32_t* __restrict__ plan, XMap map, int N, int Cin, int Kout) {
    typedef typename DT::T T;
    static_assert(DT::is16, "grouped kernel is for 16-bit storage types");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int tile, grp;
    if (!xmap_decode(map, blockIdx.x, tile, grp)) return;
    if (plan[0] != PLAN_MAGIC || plan[1] != PLAN_VERSION || plan[2] != G || plan[3] > XG_SB || plan[14] != 0) return;   // not our plan
    PlanView pv;
    pv.groups = plan + plan[8];
    pv.stages = plan + plan[9];
    pv.steps = plan + plan[10];
    pv.wlist = plan + plan[11];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int4 gh = *reinterpret_cast<const int4*>(pv.groups + 4 * grp);
    const int stage_beg = gh.x, nstages = gh.y, ob0 = gh.z, nob = gh.w;
    const int n = tile * 128 + wave * 32 + r;
    const bool valid = n < N;

    // fragment-read swizzle for this lane's row of a weight block
    const int sw = (r >> 2) & 3;
    const int rd0 = r * 64 + ((h ^ sw) << 4);          // K-half 0: slot h
    const int rd1 = r * 64 + (((2 + h) ^ sw) << 4);    // K-half 1: slot 2+h
    // DMA source: lane L of a 1 KiB chunk fills LDS row (L>>2) (+16 for the second half), stored slot L&3
    const int drow = lane >> 2, dslot = lane & 3;

    f32x16 acc[G];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[g][i] = 0.f;

    for (int st = stage_beg; st < stage_beg + nstages; ++st) {
        const int4 sg = *reinterpret_cast<const int4*>(pv.stages + 4 * st);
        const int step_beg = sg.x, nsteps = sg.y, w_beg = sg.z, nw = sg.w;
        __syncthreads();   // every wave is done reading the previous stage's blocks
        for (int q = wave; q < 2 * nw; q += 4) {
            const int w = pv.wlist[w_beg + (q >> 1)];
            const int rr = 16 * (q & 1) + drow;
            const int s = dslot ^ ((rr >> 2) & 3);
            const T* src = Wsel + (size_t)w * 1024 + rr * 32 + s * 8;
            __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)(smem + q * 1024), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();

        int lds_off = 0;
        Frag32<DT> xf;
        {
            const int c0 = pv.steps[2 * step_beg];
            if (valid) {
                if constexpr (AXIS == 1) xf.load_contig(X + (size_t)n * Cin + c0 * 32, h);
                else                     xf.load_strided(X + (size_t)c0 * 32 * N + n, (size_t)N, h);
            } else xf.zero();
        }
        for (int s = 0; s < nsteps; ++s) {
            const int mask = pv.steps[2 * (step_beg + s) + 1];
            Frag32<DT> xn;
            if (s + 1 < nsteps) {   // prefetch the next step's X fragment under this step's MFMAs
                const int c1 = pv.steps[2 * (step_beg + s + 1)];
                if (valid) {
                    if constexpr (AXIS == 1) xn.load_contig(X + (size_t)n * Cin + c1 * 32, h);
                    else                     xn.load_strided(X + (size_t)c1 * 32 * N + n, (size_t)N, h);
                } else xn.zero();
            } else xn.zero();
#pragma unroll
            for (int g = 0; g < G; ++g) {
                if ((mask >> g) & 1) {
                    Frag32<DT> wf;
                    wf.q[0] = *reinterpret_cast<const uint4*>(smem + lds_off + rd0);
                    wf.q[1] = *reinterpret_cast<const uint4*>(smem + lds_off + rd1);
                    lds_off += 2048;
                    mma32<DT>(wf, xf, acc[g]);
                }
            }
            xf = xn;
        }
    }

    if (!valid) return;
#pragma unroll
    for (int g = 0; g < G; ++g) {
        if (g >= nob) break;
        const int ob = ob0 + g;
        if constexpr (AXIS == 1) {
            T* yrow = Y + (size_t)n * Kout + ob * 32 + 4 * h;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                uint32_t lo = (uint32_t)DT::from_f32(acc[g][4 * q + 0]) | ((uint32_t)DT::from_f32(acc[g][4 * q + 1]) << 16);
                uint32_t hi = (uint32_t)DT::from_f32(acc[g][4 * q + 2]) | ((uint32_t)DT::from_f32(acc[g][4 * q + 3]) << 16);
                *reinterpret_cast<uint2*>(yrow + 8 * q) = make_uint2(lo, hi);
            }
        } else {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int o = (reg & 3) + 8 * (reg >> 2) + 4 * h;
                Y[(size_t)(ob * 32 + o) * N + n] = DT::from_f32(acc[g][reg]);
            }
        }
    }
}

}  // namespace bsmm
