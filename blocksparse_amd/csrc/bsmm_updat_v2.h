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
//   * 16-row chunks in a ring of four slots, the DMA of chunk i+3 requested right after the barrier of chunk i and
//     `s_waitcnt vmcnt(2*NI)`: three chunks are always in flight and the queue never drains;
//   * a FIXED grid walks the flattened (item, chunk) sequence, every workgroup a contiguous range of it: any item count
//     fills the chip evenly, and the plan orders the items so that an XCD's workgroups read one minibatch quarter of half
//     of X and all of DY (bsmm_plan.h);
//   * partial sums of an item meet in the fp32 scratch (atomics; zeroed by the launcher) and updat_finalize_kernel applies
//     alpha / beta (and the optional gate) with ONE rounding.  With one workgroup per item the tile is stored directly.
// LDS image of a chunk: X slab [16 rows][WS*64 B] then DY slab, 16-byte pieces of row r XOR-swizzled with 4*(r & 3)
// (bank-conflict free for ds_read_b64_tr_b16, as in bsmm_updat_win.h).  1024 threads, 128 KiB (WS = 16): one workgroup per CU.
#pragma once
#include <type_traits>

#include "bsmm_common.h"
#include "bsmm_plan.h"
#include "bsmm_updat_tr.h"

namespace bsmm {

constexpr int U2_CH = 16;      // minibatch rows per chunk = one MFMA K-step
constexpr int U2_D = 4;        // ring slots; prefetch distance U2_D - 1
constexpr int u2_lds_bytes(int ws) { return U2_D * 2 * U2_CH * ws * 64; }

// LDS-DMA with a scalar base and a 32-bit per-lane byte offset (saddr form): no 64-bit address VGPRs in the loop.
__device__ __forceinline__ void glds16_saddr(const void* sbase, uint32_t voff, uint32_t lds_byte_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_byte_addr)
                 : "memory");
}

template <class DT, int WS>
__global__ void __launch_bounds__(64 * U2_WAVES, 4)
updat32_a1_v2_kernel(PtrList8 Xs, PtrList8 Es, typename DT::T* __restrict__ DW, float* __restrict__ scratch,
                     const int32_t* __restrict__ plan, int N, int Cf, int Kf, int pcount, float alpha, float beta) {
    typedef typename DT::T T;
    static_assert(DT::is16 && (WS == 8 || WS == 16), "updat v2: 16-bit storage types, 8x8 or 16x16 windows");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int ROWB = WS * 64;                 // bytes per slab row
    constexpr int SLAB = U2_CH * ROWB;            // one operand, one chunk
    constexpr int SLOT = 2 * SLAB;
    constexpr int PPR = ROWB / 16;                // 16-byte pieces per row
    constexpr int RPI = 1024 / ROWB;              // rows per DMA instruction
    constexpr int IPO = U2_CH / RPI;              // DMA instructions per operand and chunk
    constexpr int NI = 2 * IPO / U2_WAVES;        // DMA instructions per wave and chunk
    static_assert(NI >= 1 && NI * U2_WAVES == 2 * IPO, "the chunk must split evenly over the waves");

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nitems = plan[4];
    const int32_t* items = plan + plan[6];
    const int nchunks = (N + U2_CH - 1) / U2_CH;
    const long CPI = (long)pcount * nchunks;                                  // chunks per item
    const long TOT = (long)nitems * CPI;
    long beg = (long)blockIdx.x * TOT / gridDim.x;
    const long end = (long)(blockIdx.x + 1) * TOT / gridDim.x;

    const uint32_t base_addr = lds_addr_of(smem);
    // ---- DMA geometry of this wave: instruction NI*wave + i -> operand, rows, and this lane's 16-byte piece ----
    int d_isE[NI], d_row[NI], d_piece[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int ii = NI * wave + i;
        d_isE[i] = ii / IPO;                                                  // wave-uniform
        const int row = (ii % IPO) * RPI + lane / PPR;
        d_row[i] = row;
        d_piece[i] = (lane % PPR) ^ (4 * (row & 3));                          // source piece of the LDS piece lane % PPR
    }
    // ---- fragment geometry (see bsmm_updat_tr.h): 16-lane group g16 -> features 16*(g16&1).., K half h ----
    const int g16 = lane >> 4, t16 = lane & 15;
    const int h = g16 >> 1;
    const int trow = t16 >> 2;
    const int tsub = (2 * (g16 & 1) + ((t16 & 3) >> 1)) * 16 + (t16 & 1) * 8;
    const int frag_row = (8 * h + trow) * ROWB + tsub;

    while (beg < end) {
        const int item = (int)(beg / CPI);
        const long r0 = beg - (long)item * CPI;
        const int cnt = (int)((CPI - r0 < end - beg) ? (CPI - r0) : (end - beg));
        beg += cnt;
        const int32_t* it = items + (size_t)item * U2_ITEM;
        const int c0 = it[0], k0 = it[1];
        const int32_t* wd = it + 4 + wave * U2_WWORDS;
        const uint32_t meta = (uint32_t)__builtin_amdgcn_readfirstlane(wd[0]);
        const int n0 = meta & 15, n1 = (meta >> 4) & 15;
        int wid[U2_SLOTS];
#pragma unroll
        for (int j = 0; j < U2_SLOTS; ++j) wid[j] = __builtin_amdgcn_readfirstlane(wd[1 + j]);

        // per-lane source offsets (elements) inside a row of X / DY, clamped inside the matrix (clamped pieces belong to
        // window columns no block of the item uses)
        int d_col[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i)
            d_col[i] = d_isE[i] ? min(k0 * 32 + d_piece[i] * 8, Kf - 8) : min(c0 * 32 + d_piece[i] * 8, Cf - 8);
        // fragment offsets inside a ring slot
        int aoff[2], boff[U2_SLOTS];
        aoff[0] = frag_row + ((((meta >> 8) & 15) ^ trow) << 6);
        aoff[1] = frag_row + ((((meta >> 12) & 15) ^ trow) << 6);
#pragma unroll
        for (int j = 0; j < U2_SLOTS; ++j) boff[j] = SLAB + frag_row + ((((meta >> (16 + 4 * j)) & 15) ^ trow) << 6);

        f32x16 acc[U2_SLOTS];
#pragma unroll
        for (int j = 0; j < U2_SLOTS; ++j)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;

        // the range's chunks are (pair, chunk-in-pair) = (ip, iq), (ip, iq + 1), ... wrapping into the next pair; the issue
        // cursor runs U2_D - 1 chunks ahead of the compute cursor and stops at the range's last chunk (re-fetched, never read)
        const int p_first = (int)(r0 / nchunks), q_first = (int)(r0 - (long)p_first * nchunks);
        int ip = p_first, iq = q_first, issued = 0;
        auto issue = [&](int pos) {
            const int n_first = iq * U2_CH;
            const T* Xp = static_cast<const T*>(Xs.p[ip]);
            const T* Ep = static_cast<const T*>(Es.p[ip]);
            if (++issued < cnt) {
                if (++iq == nchunks) { iq = 0; ++ip; }
            }
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int row = min(n_first + d_row[i], N - 1);               // rows past N: clamped re-reads, masked below
                const uint32_t dst = __builtin_amdgcn_readfirstlane(base_addr + pos * SLOT + d_isE[i] * SLAB + ((NI * wave + i) % IPO) * 1024);
                const uint32_t voff = (uint32_t)(row * (d_isE[i] ? Kf : Cf) + d_col[i]) * 2u;
                glds16_saddr(d_isE[i] ? (const void*)Ep : (const void*)Xp, voff, dst);
            }
        };

        auto run = [&](auto n0_tag, auto n1_tag) {
            constexpr int N0 = decltype(n0_tag)::value, N1 = decltype(n1_tag)::value;
#pragma unroll
            for (int d = 0; d < U2_D - 1; ++d) issue(d);
            int cq = q_first;
            for (int i = 0; i < cnt; ++i) {
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI * (U2_D - 2)) : "memory");   // my share of chunk i has landed
                __builtin_amdgcn_s_barrier();                                          // everyone's has; everyone left chunk i-1
                issue((i + U2_D - 1) & (U2_D - 1));                                    // refills the slot chunk i-1 used
                const int n_first = cq * U2_CH;
                if (++cq == nchunks) cq = 0;
                if constexpr (N0 > 0) {
                    const unsigned char* slot = smem + (i & (U2_D - 1)) * SLOT;
                    uint4 a0, a1;
                    {
                        const uint2 lo = ds_tr16(slot + aoff[0]), hi = ds_tr16(slot + aoff[0] + 4 * ROWB);
                        a0 = make_uint4(lo.x, lo.y, hi.x, hi.y);
                    }
                    if constexpr (N1 > 0) {
                        const uint2 lo = ds_tr16(slot + aoff[1]), hi = ds_tr16(slot + aoff[1] + 4 * ROWB);
                        a1 = make_uint4(lo.x, lo.y, hi.x, hi.y);
                    }
                    if (n_first + U2_CH > N) {      // ragged tail: rows >= N were clamped re-reads -> zero them (X side suffices)
                        const int nb = n_first + 8 * h;
                        uint32_t* u0 = reinterpret_cast<uint32_t*>(&a0);
                        uint32_t* u1 = reinterpret_cast<uint32_t*>(&a1);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const uint32_t lo = (nb + 2 * e < N) ? 0xffffu : 0u, hi = (nb + 2 * e + 1 < N) ? 0xffff0000u : 0u;
                            u0[e] &= (lo | hi);
                            if constexpr (N1 > 0) u1[e] &= (lo | hi);
                        }
                    }
                    uint4 b[N0 + N1];
#pragma unroll
                    for (int j = 0; j < N0 + N1; ++j) {
                        const uint2 lo = ds_tr16(slot + boff[j]), hi = ds_tr16(slot + boff[j] + 4 * ROWB);
                        b[j] = make_uint4(lo.x, lo.y, hi.x, hi.y);
                    }
#pragma unroll
                    for (int j = 0; j < N0; ++j) acc[j] = DT::mfma32(a0, b[j], acc[j]);
                    if constexpr (N1 > 0) {
#pragma unroll
                        for (int j = 0; j < N1; ++j) acc[N0 + j] = DT::mfma32(a1, b[N0 + j], acc[N0 + j]);
                    }
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the run-ahead DMAs: the ring is re-primed by the next range
            __builtin_amdgcn_s_barrier();
        };
        using std::integral_constant;
        switch (n0 * 8 + n1) {
            case 0 * 8 + 0: run(integral_constant<int, 0>{}, integral_constant<int, 0>{}); break;
            case 1 * 8 + 0: run(integral_constant<int, 1>{}, integral_constant<int, 0>{}); break;
            case 1 * 8 + 1: run(integral_constant<int, 1>{}, integral_constant<int, 1>{}); break;
            case 1 * 8 + 2: run(integral_constant<int, 1>{}, integral_constant<int, 2>{}); break;
            case 1 * 8 + 3: run(integral_constant<int, 1>{}, integral_constant<int, 3>{}); break;
            case 2 * 8 + 0: run(integral_constant<int, 2>{}, integral_constant<int, 0>{}); break;
            case 2 * 8 + 1: run(integral_constant<int, 2>{}, integral_constant<int, 1>{}); break;
            case 2 * 8 + 2: run(integral_constant<int, 2>{}, integral_constant<int, 2>{}); break;
            case 3 * 8 + 0: run(integral_constant<int, 3>{}, integral_constant<int, 0>{}); break;
            case 3 * 8 + 1: run(integral_constant<int, 3>{}, integral_constant<int, 1>{}); break;
            default:        run(integral_constant<int, 4>{}, integral_constant<int, 0>{}); break;
        }

        // D[ci][ko]: col = ko = lane & 31, row ci = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
        for (int j = 0; j < U2_SLOTS; ++j) {
            if (j >= n0 + n1) break;
            const size_t base = (size_t)wid[j] * 1024 + (lane & 31);
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int ci = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
                const size_t idx = base + ci * 32;
                if (scratch == nullptr) {
                    float out = alpha * acc[j][reg];
                    if (beta != 0.f) out += beta * DT::to_f32(DW[idx]);
                    DW[idx] = DT::from_f32(out);
                } else {
                    __hip_atomic_fetch_add(scratch + idx, acc[j][reg], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
    }
}

// DW[w] = alpha * gate[w] * scratch[w] + beta * DW[w], rounded once (second pass of the scratch path; gate may be NULL)
template <class DT>
__global__ void __launch_bounds__(256)
updat_finalize_gated_kernel(const float* __restrict__ scratch, typename DT::T* __restrict__ DW, size_t n, int bsq, float alpha, float beta,
                            const float* __restrict__ gate) {
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
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
