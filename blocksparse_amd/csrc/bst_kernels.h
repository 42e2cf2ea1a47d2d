// bst_kernels.h -- block-sparse attention kernels (BlocksparseTransformer path).
//
//   scores (nt)   S[n][h][b]      = Q[n][q rows][h] . K[n][k rows][h]^T           one 32x32 tile per wave
//   values (nn)   C[n][q rows][h] = sum_{b in row q}    S[n][h][b]   . V[n][k rows][h]
//   grads  (tn)   C[n][k rows][h] = sum_{b in column k} S[n][h][b]^T . E[n][q rows][h]
//   softmax / softmax grad over all blocks of a query row-block; partial autoregressive mask (integer).
//
// Replaces bst_sgemm_nt / bst_sgemm_xn / bst_hgemm_* (src/bst_sgemm_op_gpu.cu:13-497, src/bst_hgemm_op_gpu.cu) and
// bst_masked_softmax{,_grad} / bst_partial_autoregressive_mask (src/bst_softmax_op_gpu.cu:11-520).
//
// Arithmetic: the reference's fp32 pathway multiplies fp32 activations in fp32 and stores bf16 scores; MFMA f32
// (v_mfma_f32_32x32x2_f32) keeps exactly that (16-bit operands are widened on load, which is exact), for bsize 32 and
// 64 (a 64-block is 2x2 tiles).  bsize 8 / 16 run plain VALU kernels.  Operand roles are chosen so that every global
// access walks consecutive addresses across the lanes of a wave:
//   nt: M side = key rows, N side = query rows  ->  D[j][i], a lane holds 4 consecutive keys j of query row i = lane:
//       8-byte stores into the row-major [i][j] block.
//   nn/tn: M side = output rows, N side = 32 features -> D[row][c], lane = feature c: 128-byte row segments.
#pragma once
#include "bsmm_common.h"
#include "bsmm_updat_tr.h" // ds_tr16

namespace bsmm {

// 16 consecutive elements -> fp32 (valid while k0 + t < klim; klim is a multiple of 4 here)
template <class DT>
__device__ __forceinline__ void load16_f32(const typename DT::T* p, int k0, int klim, float (&v)[16]) {
    if constexpr (DT::is16) {
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            uint4 x = make_uint4(0, 0, 0, 0);
            if (k0 + 8 * g < klim) x = *reinterpret_cast<const uint4*>(p + 8 * g);
            const uint32_t w[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                v[8 * g + 2 * i] = DT::to_f32((uint16_t)(w[i] & 0xffffu));
                v[8 * g + 2 * i + 1] = DT::to_f32((uint16_t)(w[i] >> 16));
            }
        }
    } else {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
            if (k0 + 4 * g < klim) x = *reinterpret_cast<const float4*>(p + 4 * g);
            v[4 * g] = x.x; v[4 * g + 1] = x.y; v[4 * g + 2] = x.z; v[4 * g + 3] = x.w;
        }
    }
}

// Workgroup -> (batch x head, workgroup within the head) so that one (n, h) pair runs entirely on ONE XCD (workgroups
// are dealt round-robin over the 8 XCDs, each with its own 4 MiB L2): the K / V / Q rows of a head are re-read by every
// block row that attends to them, and with the plain (x = block, y = head, z = batch) grid every XCD pulled every head
// through the fabric (nt: 1.5 GB of L2 fills for 134 MB of activations, 202 us; the fabric sustains ~7 TB/s).
// grid = 8 * ceil(batch * heads / 8) * G workgroups, G = workgroups per head.
__device__ __forceinline__ bool xcd_head_map(int G, int heads, int batch, int& n, int& h, int& wg, int il = 1) {
    const int i = blockIdx.x, xcd = i & 7, j = i >> 3;
    const int grp = j / (il * G), within = j - grp * (il * G);       // il heads of an XCD are interleaved workgroup by workgroup
    const int hp = xcd + 8 * (grp * il + within % il);
    wg = within / il;
    n = hp / heads;
    h = hp - n * heads;
    return hp < heads * batch;
}
inline int xcd_head_grid(int G, int heads, int batch, int il = 1) { const int per = (heads * batch + 7) / 8; return 8 * ((per + il - 1) / il) * il * G; }

// (Raising the wave priority around the MFMA phase, which helps a bare LDS-fed MFMA loop by 10 % in
// scripts/micro/mfma_lds_chain.hip, made nn / tn slower here: 0.24 vs 0.20 ms.)
__device__ __forceinline__ void mma32_f32(const float (&a)[16], const float (&b)[16], f32x16& acc) {
#pragma unroll
    for (int t = 0; t < 16; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b[t], acc, 0, 0, 0);
}

// ------------------------------------------------------------------------------------------------------------------
// nt: one wave per 32x32 tile of a block.  grid (ceil(blocks * SUB^2 / 4), heads, batch), 256 threads.
//
// Operand tiles are [32 rows] x [32 features] chunks of row-major activations.  The MFMA wants lane r to hold row r,
// but a load in that shape touches 32 cache lines per instruction and uses an eighth of each: with ~32 waves per CU
// the lines are evicted from L1 between the instructions that share them and come from L2 four times (measured: 43 TF,
// L2 -> L1 bound).  So the wave reads each chunk fully coalesced (8 rows x 128 B per instruction for fp32), all chunks
// of both operands up front (registers are the in-flight buffer), and transposes through a wave-private LDS tile:
// 16-byte pieces XOR-swizzled with the row, written as loaded, read back as [row = lane][16 features].
// ------------------------------------------------------------------------------------------------------------------
template <class T> struct NtTile {
    static constexpr int ROWB = 32 * (int)sizeof(T);      // bytes per row of a 32-feature chunk
    static constexpr int PPR = ROWB / 16;                 // 16-byte pieces per row (8 fp32, 4 16-bit)
    static constexpr int RPI = 64 / PPR;                  // rows per load instruction
    static constexpr int NI = 32 / RPI;                   // load instructions per tile
    static constexpr int EPP = 16 / (int)sizeof(T);       // elements per piece
    static constexpr int BYTES = 32 * ROWB;
    static __device__ __forceinline__ int sw(int row) { return (row / (128 / ROWB)) & (PPR - 1); }
};

// One wave (= one workgroup) walks NT_NB CONSECUTIVE tiles of the block list.  Blocks are numbered row-major, so
// consecutive tiles nearly always share their query rows: the Q tile stays in LDS until the row changes (1.5 GB of
// L2 -> CU traffic become ~0.9 GB).  Tiles arrive by LDS-DMA (global_load_lds_dwordx4: 1 KiB of whole rows per
// instruction, the swizzle applied on the source address); per tile the wave first pulls every fragment into registers,
// then requests the next tile into the now idle buffers, then runs the 16 * chunks MFMAs with that DMA in flight.
// Ablation of the previous version (coalesced VGPR loads + ds_write, 215 us): without loads 125, without MFMA 142.
constexpr int NT_NB = 8;

// SPLIT (fp32 activations only): both operands are split exactly into three bf16 pieces (see bst_xn_split_kernel) and the six
// largest of the nine piece products run on v_mfma_f32_32x32x16_bf16 (the dropped ones are below 2^-24 of the leading term):
// 12 MFMAs of 32 cycles per 32-feature chunk instead of 16 fp32 MFMAs of 64.  The split Q pieces live in registers while
// the query row does not change.  K labels: MFMA kk of a chunk takes this lane's values 8kk .. 8kk+7 (k = 16hh + 8kk + j) on
// both sides, which is all the hardware needs.
template <class TA, class TS, int BS, int NCH, bool SPLIT = false>
__global__ void __launch_bounds__(64)
bst_nt_mfma_kernel(const typename TA::T* __restrict__ A, const typename TA::T* __restrict__ B, typename TS::T* __restrict__ S,
                   const int32_t* __restrict__ lut, int lut_stride, int blocks, int heads, int batch, int hs, int rows_q, int rows_k, int il) {
    typedef typename TA::T T;
    typedef NtTile<T> TL;
    constexpr int SUB = BS / 32;
    __shared__ __attribute__((aligned(16))) unsigned char lds[2][NCH][TL::BYTES];        // [K | Q][chunk]
    __shared__ __attribute__((aligned(16))) unsigned char olds[32 * 64];                 // finished 32x32 tile of 16-bit scores
    const int lane = threadIdx.x;
    const int r = lane & 31, hh = lane >> 5;
    const int ntiles = blocks * SUB * SUB;
    int n, h, wg;
    if (!xcd_head_map((ntiles + NT_NB - 1) / NT_NB, heads, batch, n, h, wg, il)) return;
    const int tb0 = wg * NT_NB;
    const size_t state = (size_t)heads * hs;
    const int lrow = lane / TL::PPR, lp = lane % TL::PPR;
    const T* abase = A + (size_t)n * rows_q * state + (size_t)h * hs;
    const T* bbase = B + (size_t)n * rows_k * state + (size_t)h * hs;
    const int32_t* hl = lut + (size_t)h * lut_stride;
    const uint32_t lds_k = lds_addr_of(&lds[0][0][0]), lds_q = lds_addr_of(&lds[1][0][0]);

    // source offsets (elements, relative to the tile's first row) of this lane's piece in every DMA instruction
    size_t soff[TL::NI];
#pragma unroll
    for (int i = 0; i < TL::NI; ++i) {
        const int row = TL::RPI * i + lrow;
        soff[i] = (size_t)row * state + (size_t)((lp ^ TL::sw(row)) * TL::EPP);
    }
    auto dma_tile = [&](const T* tile0, uint32_t dst) {       // [32 rows][NCH chunks] -> NCH swizzled tile images
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int i = 0; i < TL::NI; ++i)
                glds16_asm(tile0 + soff[i] + 32 * c, __builtin_amdgcn_readfirstlane(dst + c * TL::BYTES + i * 1024));
    };
    auto frag = [&](const unsigned char* tile, float (&v)[16]) {               // row r, features 16hh .. 16hh+15
        constexpr int NP = 16 / TL::EPP;
#pragma unroll
        for (int g = 0; g < NP; ++g) {
            const uint4 x = *reinterpret_cast<const uint4*>(tile + r * TL::ROWB + (((NP * hh + g) ^ TL::sw(r)) << 4));
            const uint32_t w[4] = {x.x, x.y, x.z, x.w};
            if constexpr (TA::is16) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    v[8 * g + 2 * i] = TA::to_f32((uint16_t)(w[i] & 0xffffu));
                    v[8 * g + 2 * i + 1] = TA::to_f32((uint16_t)(w[i] >> 16));
                }
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) v[4 * g + i] = __builtin_bit_cast(float, w[i]);
            }
        }
    };
    auto tile_of = [&](int tb, int& qrow, int& krow) {
        const int b = tb / (SUB * SUB), ti = (tb / SUB) % SUB, tj = tb % SUB;
        const int2 qk = *reinterpret_cast<const int2*>(hl + 2 * b);
        qrow = qk.x * BS + 32 * ti;
        krow = qk.y * BS + 32 * tj;
    };

    // Output tile: D[j][i] has 4 consecutive keys j per register quad for query row i = lane, i.e. 8-byte pieces at a 64-byte
    // stride -- stored directly that is 192 MB of partial-line writes (62 us on their own).  The tile is parked in a 2 KiB
    // LDS image instead ([32 rows][64 B], 16-byte pieces XOR-swizzled with (row >> 2) & 3) and written one iteration LATER
    // as two fully contiguous 1 KiB stores, so that the write latency is not in front of the next tile's wait either.
    auto park = [&](const f32x16& acc) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const uint32_t lo = (uint32_t)TS::from_f32(acc[4 * g + 0]) | ((uint32_t)TS::from_f32(acc[4 * g + 1]) << 16);
            const uint32_t hi = (uint32_t)TS::from_f32(acc[4 * g + 2]) | ((uint32_t)TS::from_f32(acc[4 * g + 3]) << 16);
            *reinterpret_cast<uint2*>(&olds[r * 64 + ((g ^ ((r >> 2) & 3)) << 4) + 8 * hh]) = make_uint2(lo, hi);
        }
    };
    auto flush = [&](typename TS::T* out) {                    // out = first element of the 32x32 tile (row stride BS)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int P = k * 64 + lane, i = P >> 2, p = P & 3;
            const uint4 v = *reinterpret_cast<const uint4*>(&olds[i * 64 + ((p ^ ((i >> 2) & 3)) << 4)]);
            *reinterpret_cast<uint4*>(out + (size_t)i * BS + p * 8) = v;
        }
    };
    auto out_of = [&](int tb) {
        const int b = tb / (SUB * SUB), ti = (tb / SUB) % SUB, tj = tb % SUB;
        return S + (((size_t)n * heads + h) * blocks + b) * (BS * BS) + (size_t)(32 * ti) * BS + 32 * tj;
    };

    int qrow, krow, qnext = 0, knext = 0;
    tile_of(tb0, qrow, krow);
    if (tb0 + 1 < ntiles) tile_of(tb0 + 1, qnext, knext);
    dma_tile(bbase + (size_t)krow * state, lds_k);
    dma_tile(abase + (size_t)qrow * state, lds_q);
    typename TS::T* pending = nullptr;
    uint4 qp[(SPLIT && !TA::is16) ? NCH : 1][3][2];           // SPLIT: bf16 pieces of the current Q fragments
    int qsplit_row = -1;
    for (int it = 0; it < NT_NB; ++it) {
        const int tb = tb0 + it;
        if (tb >= ntiles) break;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                       // this tile's DMAs have landed
        // fp32 activations: fragments of 16 floats for v_mfma_f32_32x32x2_f32.  16-bit activations: the raw 16-byte pieces ARE the
        // operands of v_mfma_f32_32x32x16_{f16,bf16} (q[0]: k = 8hh + j -> piece hh, q[1]: k = 16 + 8hh + j -> piece 2 + hh)
        float fk[TA::is16 ? 1 : NCH][16], fq[TA::is16 ? 1 : NCH][16];
        uint4 rk[TA::is16 ? NCH : 1][2], rq[TA::is16 ? NCH : 1][2];
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            if constexpr (TA::is16) {
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const int off = r * TL::ROWB + (((2 * kk + hh) ^ TL::sw(r)) << 4);
                    rk[c][kk] = *reinterpret_cast<const uint4*>(&lds[0][c][0] + off);
                    rq[c][kk] = *reinterpret_cast<const uint4*>(&lds[1][c][0] + off);
                }
            } else {
                frag(&lds[0][c][0], fk[c]);
                frag(&lds[1][c][0], fq[c]);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                     // fragments are in registers: the buffers are free
        const bool more = it + 1 < NT_NB && tb + 1 < ntiles;
        const int qcur = qrow;
        if (more) {
            qrow = qnext; krow = knext;
            dma_tile(bbase + (size_t)krow * state, lds_k);
            if (qrow != qcur) dma_tile(abase + (size_t)qrow * state, lds_q);
            if (tb + 2 < ntiles) tile_of(tb + 2, qnext, knext);
        }
        if (pending) flush(pending);                                           // previous tile's scores
        // two independent accumulator chains (even / odd chunks), interleaved instruction by instruction
        f32x16 acc, acc1;
#pragma unroll
        for (int i = 0; i < 16; ++i) { acc[i] = 0.f; acc1[i] = 0.f; }
        if constexpr (SPLIT && !TA::is16) {
            auto split3 = [](const float (&v)[16], uint4 (&p)[3][2]) {          // 16 floats -> 3 pieces x 2 MFMA operands
                auto pack2 = [](float lo, float hi) { return bf16_pack2(lo, hi); };
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    uint32_t a[4], b[4], c3[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float v0 = v[8 * kk + 2 * i], v1 = v[8 * kk + 2 * i + 1];
                        a[i] = pack2(v0, v1);
                        const float r0 = v0 - __builtin_bit_cast(float, a[i] << 16), r1 = v1 - __builtin_bit_cast(float, a[i] & 0xffff0000u);
                        b[i] = pack2(r0, r1);
                        c3[i] = pack2(r0 - __builtin_bit_cast(float, b[i] << 16), r1 - __builtin_bit_cast(float, b[i] & 0xffff0000u));
                    }
                    p[0][kk] = make_uint4(a[0], a[1], a[2], a[3]);
                    p[1][kk] = make_uint4(b[0], b[1], b[2], b[3]);
                    p[2][kk] = make_uint4(c3[0], c3[1], c3[2], c3[3]);
                }
            };
            if (qsplit_row != qcur) {                                          // new query rows: split the Q fragments once
#pragma unroll
                for (int c = 0; c < NCH; ++c) split3(fq[c], qp[c]);
                qsplit_row = qcur;
            }
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                uint4 kp[3][2];
                split3(fk[c], kp);
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {                               // smallest products first; M side = keys, N side = queries
                    acc1 = DTbf16::mfma32(kp[2][kk], qp[c][0][kk], acc1);
                    acc1 = DTbf16::mfma32(kp[0][kk], qp[c][2][kk], acc1);
                    acc1 = DTbf16::mfma32(kp[1][kk], qp[c][1][kk], acc1);
                    acc = DTbf16::mfma32(kp[1][kk], qp[c][0][kk], acc);
                    acc = DTbf16::mfma32(kp[0][kk], qp[c][1][kk], acc);
                    acc = DTbf16::mfma32(kp[0][kk], qp[c][0][kk], acc);
                }
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] += acc1[i];
        } else if constexpr (TA::is16) {
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                acc = TA::mfma32(rk[c][0], rq[c][0], acc);
                acc1 = TA::mfma32(rk[c][1], rq[c][1], acc1);
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] += acc1[i];
        } else if constexpr (NCH >= 2) {
#pragma unroll
            for (int c = 0; c < NCH; c += 2)
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fk[c][t], fq[c][t], acc, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fk[c + 1][t], fq[c + 1][t], acc1, 0, 0, 0);
                }
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] += acc1[i];
        } else {
            mma32_f32(fk[0], fq[0], acc);
        }
        park(acc);
        pending = out_of(tb);
    }
    if (pending) flush(pending);
}

// any head_state (chunks are loaded, transposed and multiplied one at a time: no register-resident prefetch)
template <class TA, class TS, int BS>
__global__ void __launch_bounds__(256)
bst_nt_mfma_direct_kernel(const typename TA::T* __restrict__ A, const typename TA::T* __restrict__ B, typename TS::T* __restrict__ S,
                          const int32_t* __restrict__ lut, int lut_stride, int blocks, int heads, int batch, int hs, int rows_q, int rows_k) {
    constexpr int SUB = BS / 32;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, hh = lane >> 5;
    int n, h, wg;
    if (!xcd_head_map((blocks * SUB * SUB + 3) / 4, heads, batch, n, h, wg)) return;
    const int tb = wg * 4 + wave;
    if (tb >= blocks * SUB * SUB) return;
    const int b = tb / (SUB * SUB), ti = (tb / SUB) % SUB, tj = tb % SUB;
    const int2 qk = *reinterpret_cast<const int2*>(lut + (size_t)h * lut_stride + 2 * b);
    const size_t state = (size_t)heads * hs;
    const typename TA::T* qrow = A + ((size_t)n * rows_q + (size_t)qk.x * BS + 32 * ti + r) * state + (size_t)h * hs + 16 * hh;
    const typename TA::T* krow = B + ((size_t)n * rows_k + (size_t)qk.y * BS + 32 * tj + r) * state + (size_t)h * hs + 16 * hh;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    for (int kc = 0; kc < hs; kc += 32) {
        float fk[16], fq[16];
        load16_f32<TA>(krow + kc, kc + 16 * hh, hs, fk);
        load16_f32<TA>(qrow + kc, kc + 16 * hh, hs, fq);
        mma32_f32(fk, fq, acc);
    }
    typename TS::T* out = S + (((size_t)n * heads + h) * blocks + b) * (BS * BS) + (size_t)(32 * ti + r) * BS + 32 * tj + 4 * hh;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const uint32_t lo = (uint32_t)TS::from_f32(acc[4 * g + 0]) | ((uint32_t)TS::from_f32(acc[4 * g + 1]) << 16);
        const uint32_t hi = (uint32_t)TS::from_f32(acc[4 * g + 2]) | ((uint32_t)TS::from_f32(acc[4 * g + 3]) << 16);
        *reinterpret_cast<uint2*>(out + 8 * g) = make_uint2(lo, hi);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// nn (TRANS = false) / tn (TRANS = true): one WORKGROUP per (output block, 32-row sub tile, 32-feature tile); its four waves
// take every fourth step of the block list (rows of a causal layout have 1 .. nn_max blocks: one wave per tile left the
// longest rows as a serial chain) and reduce through LDS.  lut = nn_lut / tn_lut (header + entries).
// ------------------------------------------------------------------------------------------------------------------
template <class TS, class TB, int BS, bool TRANS>
__global__ void __launch_bounds__(256)
bst_xn_mfma_kernel(const typename TS::T* __restrict__ S, const typename TB::T* __restrict__ Bm, typename TB::T* __restrict__ C,
                   const int32_t* __restrict__ lut, int lut_stride, int blocks, int heads, int batch, int hs, int ctx_c, int rows_b, int rows_c) {
    constexpr int SUB = BS / 32;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int nct = (hs + 31) / 32;
    int n, h, wid;
    if (!xcd_head_map(ctx_c * SUB * nct, heads, batch, n, h, wid)) return;      // one output tile per workgroup, 4 waves split its steps
    const int ct = wid % nct, ts = (wid / nct) % SUB, oc = wid / (nct * SUB);
    const int32_t* hl = lut + (size_t)h * lut_stride;
    const int2 hdr = *reinterpret_cast<const int2*>(hl + 2 * oc);
    const size_t state = (size_t)heads * hs;
    const int c = 32 * ct + r;
    const bool cvalid = c < hs;
    const typename TS::T* sbase = S + ((size_t)n * heads + h) * blocks * (BS * BS);
    const typename TB::T* bcol = Bm + (size_t)n * rows_b * state + (size_t)h * hs + c;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    // Steps q = (entry, 32-wide slice of the contraction index); the operands of step q+1 are requested before the 16 MFMAs
    // of step q, so a wave's chain over its (up to nn_max) entries is bound by max(load latency, MFMA time) per step instead
    // of their sum (the longest rows set the kernel time: causal layouts have rows of 1 .. nn_max blocks).
    constexpr int NRA = TRANS ? 16 : (TS::is16 ? 2 : 4);                          // raw A registers (16-byte units or single elements)
    uint4 ra4[TRANS ? 1 : NRA];
    typename TS::T ra1[TRANS ? 16 : 1];
    typename TB::T rb[16];
    const int nsteps = hdr.y * SUB;
    auto request = [&](int q) {
        const int e = q / SUB, tk = q % SUB;
        const int2 ent = *reinterpret_cast<const int2*>(hl + 2 * (hdr.x + e));     // (block id, other-side block)
        const typename TS::T* sb = sbase + (size_t)ent.x * (BS * BS);
        if constexpr (!TRANS) {       // A[i][j] = S_b[32ts + i][32tk + j], j = 16hh + t: contiguous
            const typename TS::T* p = sb + (size_t)(32 * ts + r) * BS + 32 * tk + 16 * hh;
#pragma unroll
            for (int g = 0; g < NRA; ++g) ra4[g] = *reinterpret_cast<const uint4*>(p + g * (16 / sizeof(typename TS::T)));
        } else {                      // A[j][i] = S_b[32tk + i][32ts + j], i = 16hh + t: stride BS (lanes walk j)
#pragma unroll
            for (int t = 0; t < 16; ++t) ra1[t] = sb[(size_t)(32 * tk + 16 * hh + t) * BS + 32 * ts + r];
        }
        const typename TB::T* bp = bcol + ((size_t)ent.y * BS + 32 * tk + 16 * hh) * state;
#pragma unroll
        for (int t = 0; t < 16; ++t) rb[t] = cvalid ? bp[(size_t)t * state] : (typename TB::T)0;
    };
    if (wave < nsteps) request(wave);
    for (int q = wave; q < nsteps; q += 4) {
        float fa[16], fb[16];
        if constexpr (!TRANS) {
            if constexpr (TS::is16) {
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    const uint32_t w[4] = {ra4[g].x, ra4[g].y, ra4[g].z, ra4[g].w};
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        fa[8 * g + 2 * i] = TS::to_f32((uint16_t)(w[i] & 0xffffu));
                        fa[8 * g + 2 * i + 1] = TS::to_f32((uint16_t)(w[i] >> 16));
                    }
                }
            } else {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    fa[4 * g] = __builtin_bit_cast(float, ra4[g].x); fa[4 * g + 1] = __builtin_bit_cast(float, ra4[g].y);
                    fa[4 * g + 2] = __builtin_bit_cast(float, ra4[g].z); fa[4 * g + 3] = __builtin_bit_cast(float, ra4[g].w);
                }
            }
        } else {
#pragma unroll
            for (int t = 0; t < 16; ++t) fa[t] = TS::to_f32(ra1[t]);
        }
#pragma unroll
        for (int t = 0; t < 16; ++t) fb[t] = TB::to_f32(rb[t]);
        if (q + 4 < nsteps) request(q + 4);
        mma32_f32(fa, fb, acc);
    }
    // the four partial tiles meet in LDS; wave w then owns registers 4w .. 4w+3 (rows 8w + {0..3} + 4hh) of the sum
    __shared__ float red[4][16][64];
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) red[wave][reg][lane] = acc[reg];
    __syncthreads();
    if (!cvalid) return;
    typename TB::T* out = C + ((size_t)n * rows_c + (size_t)oc * BS + 32 * ts) * state + (size_t)h * hs + c;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int reg = 4 * wave + i;
        const float v = red[0][reg][lane] + red[1][reg][lane] + red[2][reg][lane] + red[3][reg][lane];
        const int row = (reg & 3) + 8 * (reg >> 2) + 4 * hh;
        out[(size_t)row * state] = TB::from_f32(v);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// nn / tn with bf16 scores and fp32 activations (BASELINE configs[4]) on the 16-bit matrix core, EXACTLY: every fp32
// activation value v is split into three bf16 pieces b1 = bf16(v), b2 = bf16(v - b1), b3 = bf16(v - b1 - b2) (8 + 8 + 8
// significand bits: v = b1 + b2 + b3, the subtractions are exact), the scores are bf16 already, bf16 x bf16 products are
// exact in fp32 and the accumulation is fp32 as before -- the result differs from the fp32-MFMA kernel only by the
// order of the fp32 additions.  Three v_mfma_f32_32x32x16_bf16 pairs (192 cycles) replace 16 v_mfma_f32_32x32x2_f32
// (1024 cycles) per step; the split costs ~90 VALU operations per step and lane.  Same decomposition as
// bst_xn_mfma_kernel (workgroup = output tile, waves split the steps, next step's operands requested ahead, LDS reduce).
// K labels of the 16-bit MFMA: q[kk] holds k = 16 kk + 8 hh + j, so this lane's 16 activation rows are
// 8hh .. 8hh+7 and 16+8hh .. 16+8hh+7 of the 32-row slice.
// ------------------------------------------------------------------------------------------------------------------
template <int BS, bool TRANS>
__global__ void __launch_bounds__(256)
bst_xn_split_kernel(const uint16_t* __restrict__ S, const float* __restrict__ Bm, float* __restrict__ C,
                    const int32_t* __restrict__ lut, int lut_stride, int blocks, int heads, int batch, int hs, int ctx_c, int rows_b, int rows_c) {
    constexpr int SUB = BS / 32;
    __shared__ float red[4][16][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int nct = (hs + 31) / 32;
    int n, h, wid;
    if (!xcd_head_map(ctx_c * SUB * nct, heads, batch, n, h, wid)) return;
    const int ct = wid % nct, ts = (wid / nct) % SUB, oc = wid / (nct * SUB);
    const int32_t* hl = lut + (size_t)h * lut_stride;
    const int2 hdr = *reinterpret_cast<const int2*>(hl + 2 * oc);
    const size_t state = (size_t)heads * hs;
    const int c = 32 * ct + r;
    const bool cvalid = c < hs;
    const uint16_t* sbase = S + ((size_t)n * heads + h) * blocks * (BS * BS);
    const float* bcol = Bm + (size_t)n * rows_b * state + (size_t)h * hs + c;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    uint4 ra[2];                 // score fragment (nn: two contiguous 16-byte loads)
    uint16_t ra1[TRANS ? 16 : 1];
    float rb[16];
    const int nsteps = hdr.y * SUB;
    auto request = [&](int q) {
        const int e = q / SUB, tk = q % SUB;
        const int2 ent = *reinterpret_cast<const int2*>(hl + 2 * (hdr.x + e));
        const uint16_t* sb = sbase + (size_t)ent.x * (BS * BS);
        if constexpr (!TRANS) {       // A[i][k = j] = S_b[32ts + i][32tk + j]
            const uint16_t* p = sb + (size_t)(32 * ts + r) * BS + 32 * tk + 8 * hh;
            ra[0] = *reinterpret_cast<const uint4*>(p);
            ra[1] = *reinterpret_cast<const uint4*>(p + 16);
        } else {                      // A[j][k = i] = S_b[32tk + i][32ts + j]
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int j = 0; j < 8; ++j) ra1[8 * kk + j] = sb[(size_t)(32 * tk + 16 * kk + 8 * hh + j) * BS + 32 * ts + r];
        }
        const float* bp = bcol + ((size_t)ent.y * BS + 32 * tk + 8 * hh) * state;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int j = 0; j < 8; ++j) rb[8 * kk + j] = cvalid ? bp[(size_t)(16 * kk + j) * state] : 0.f;
    };
    auto pack2 = [](float lo, float hi) { return bf16_pack2(lo, hi); };
    if (wave < nsteps) request(wave);
    for (int q = wave; q < nsteps; q += 4) {
        uint4 fa[2];
        if constexpr (!TRANS) {
            fa[0] = ra[0];
            fa[1] = ra[1];
        } else {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
                fa[kk] = make_uint4((uint32_t)ra1[8 * kk + 0] | ((uint32_t)ra1[8 * kk + 1] << 16), (uint32_t)ra1[8 * kk + 2] | ((uint32_t)ra1[8 * kk + 3] << 16),
                                    (uint32_t)ra1[8 * kk + 4] | ((uint32_t)ra1[8 * kk + 5] << 16), (uint32_t)ra1[8 * kk + 6] | ((uint32_t)ra1[8 * kk + 7] << 16));
        }
        uint32_t p1[2][4], p2[2][4], p3[2][4];         // the three bf16 pieces of this lane's 16 activation values, packed
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float v0 = rb[8 * kk + 2 * i], v1 = rb[8 * kk + 2 * i + 1];
                const uint32_t a = pack2(v0, v1);
                const float r0 = v0 - __builtin_bit_cast(float, a << 16), r1 = v1 - __builtin_bit_cast(float, a & 0xffff0000u);
                const uint32_t b = pack2(r0, r1);
                const float s0 = r0 - __builtin_bit_cast(float, b << 16), s1 = r1 - __builtin_bit_cast(float, b & 0xffff0000u);
                p1[kk][i] = a;
                p2[kk][i] = b;
                p3[kk][i] = pack2(s0, s1);
            }
        if (q + 4 < nsteps) request(q + 4);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            acc = DTbf16::mfma32(fa[kk], make_uint4(p3[kk][0], p3[kk][1], p3[kk][2], p3[kk][3]), acc);     // smallest pieces first
            acc = DTbf16::mfma32(fa[kk], make_uint4(p2[kk][0], p2[kk][1], p2[kk][2], p2[kk][3]), acc);
            acc = DTbf16::mfma32(fa[kk], make_uint4(p1[kk][0], p1[kk][1], p1[kk][2], p1[kk][3]), acc);
        }
    }
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) red[wave][reg][lane] = acc[reg];
    __syncthreads();
    if (!cvalid) return;
    float* out = C + ((size_t)n * rows_c + (size_t)oc * BS + 32 * ts) * state + (size_t)h * hs + c;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int reg = 4 * wave + i;
        const float v = red[0][reg][lane] + red[1][reg][lane] + red[2][reg][lane] + red[3][reg][lane];
        const int row = (reg & 3) + 8 * (reg >> 2) + 4 * hh;
        out[(size_t)row * state] = v;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// nn / tn with 16-bit scores AND 16-bit activations (the reference's fp16 tensor-core pathway, bst_hgemm_xn,
// src/bst_hgemm_op_gpu.cu): v_mfma_f32_32x32x16_{f16,bf16}, no widening.  Same decomposition as bst_xn_mfma_kernel
// (workgroup = output tile, waves split the steps, LDS reduction).  The operand whose contraction index is its ROW
// index (the activation tile always; the score block too for tn) is staged as a plain [32 rows][64 B] LDS image by
// LDS-DMA and read with ds_read_b64_tr_b16 (lane t of a 16-lane group receives column t of the 4 x 16 patch its group
// points at: profiles/r01_tr_probe.log); the nn score fragment is two contiguous 16-byte loads.  Needs head_state % 32 == 0.
// ------------------------------------------------------------------------------------------------------------------
template <class T16, int BS, bool TRANS>
__global__ void __launch_bounds__(256)
bst_xn_mfma16_kernel(const typename T16::T* __restrict__ S, const typename T16::T* __restrict__ Bm, typename T16::T* __restrict__ C,
                     const int32_t* __restrict__ lut, int lut_stride, int blocks, int heads, int batch, int hs, int ctx_c, int rows_b, int rows_c) {
    typedef typename T16::T T;
    constexpr int SUB = BS / 32;
    __shared__ __attribute__((aligned(16))) unsigned char tiles[4][2][2][2048];     // [wave][buffer][B tile | S tile]
    __shared__ float red[4][16][64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const int nct = hs / 32;
    int n, h, wid;
    if (!xcd_head_map(ctx_c * SUB * nct, heads, batch, n, h, wid)) return;
    const int ct = wid % nct, ts = (wid / nct) % SUB, oc = wid / (nct * SUB);
    const int32_t* hl = lut + (size_t)h * lut_stride;
    const int2 hdr = *reinterpret_cast<const int2*>(hl + 2 * oc);
    const size_t state = (size_t)heads * hs;
    const T* sbase = S + ((size_t)n * heads + h) * blocks * (BS * BS);
    const T* bbase = Bm + (size_t)n * rows_b * state + (size_t)h * hs + 32 * ct;
    const int drow = lane >> 2, dpiece = lane & 3;                                   // DMA: 16 rows x 64 B per instruction
    const int g16 = lane >> 4, t16 = lane & 15;
    const int rd_base = (t16 >> 2) * 64 + (16 * (g16 & 1) + 4 * (t16 & 3)) * 2;      // transposing read, see bsmm_updat_tr.h
    const uint32_t my_lds = lds_addr_of(&tiles[wave][0][0][0]);
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const int nsteps = hdr.y * SUB;
    uint4 sa[2];                                                                      // nn: score fragment of the step in flight
    auto request = [&](int q, int buf) {
        const int e = q / SUB, tk = q % SUB;
        const int2 ent = *reinterpret_cast<const int2*>(hl + 2 * (hdr.x + e));      // (block id, other-side block)
        const T* bt = bbase + ((size_t)ent.y * BS + 32 * tk) * state;               // 32 rows (contraction index) x 32 features
        const uint32_t dst = __builtin_amdgcn_readfirstlane(my_lds + buf * 4096);
        glds16_asm(bt + (size_t)drow * state + dpiece * 8, dst);
        glds16_asm(bt + (size_t)(16 + drow) * state + dpiece * 8, dst + 1024);
        const T* sb = sbase + (size_t)ent.x * (BS * BS);
        if constexpr (TRANS) {        // rows i = contraction index 32tk.., features j = 32ts..: [32][32] patch of the score block
            const T* st = sb + (size_t)(32 * tk) * BS + 32 * ts;
            glds16_asm(st + (size_t)drow * BS + dpiece * 8, dst + 2048);
            glds16_asm(st + (size_t)(16 + drow) * BS + dpiece * 8, dst + 3072);
        } else {                      // A[i][j] = S_b[32ts + i][32tk + j]: k = j contiguous per lane
            const T* p = sb + (size_t)(32 * ts + r) * BS + 32 * tk + 8 * hh;
            sa[0] = *reinterpret_cast<const uint4*>(p);
            sa[1] = *reinterpret_cast<const uint4*>(p + 16);
        }
    };
    auto tr_frag = [&](const unsigned char* tile, uint4 (&f)[2]) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const unsigned char* sp = tile + (16 * kk + 8 * hh) * 64 + rd_base;
            const uint2 lo = ds_tr16(sp), hi = ds_tr16(sp + 4 * 64);
            f[kk] = make_uint4(lo.x, lo.y, hi.x, hi.y);
        }
    };
    if (wave < nsteps) request(wave, 0);
    int buf = 0;
    for (int q = wave; q < nsteps; q += 4) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                             // this step's tiles (and score fragment) are here
        uint4 fa[2], fb[2];
        tr_frag(&tiles[wave][buf][0][0], fb);
        if constexpr (TRANS) tr_frag(&tiles[wave][buf][1][0], fa);
        else { fa[0] = sa[0]; fa[1] = sa[1]; }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (q + 4 < nsteps) request(q + 4, buf ^ 1);
        acc = T16::mfma32(fa[0], fb[0], acc);
        acc = T16::mfma32(fa[1], fb[1], acc);
        buf ^= 1;
    }
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) red[wave][reg][lane] = acc[reg];
    __syncthreads();
    T* out = C + ((size_t)n * rows_c + (size_t)oc * BS + 32 * ts) * state + (size_t)h * hs + 32 * ct + r;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int reg = 4 * wave + i;
        const float v = red[0][reg][lane] + red[1][reg][lane] + red[2][reg][lane] + red[3][reg][lane];
        const int row = (reg & 3) + 8 * (reg >> 2) + 4 * hh;
        out[(size_t)row * state] = T16::from_f32(v);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// VALU versions (bsize 8 / 16; any bsize works).  nt: grid (blocks, heads, batch), BS*BS threads (<= 256: one output
// element per thread for BS <= 16).  xn: grid (ctx_c, heads, batch), 256 threads striding over the BS x hs outputs.
// ------------------------------------------------------------------------------------------------------------------
template <class TA, class TS, int BS>
__global__ void bst_nt_valu_kernel(const typename TA::T* __restrict__ A, const typename TA::T* __restrict__ B, typename TS::T* __restrict__ S,
                                   const int32_t* __restrict__ lut, int lut_stride, int blocks, int heads, int hs, int rows_q, int rows_k) {
    const int b = blockIdx.x, h = blockIdx.y, n = blockIdx.z;
    const int2 qk = *reinterpret_cast<const int2*>(lut + (size_t)h * lut_stride + 2 * b);
    const size_t state = (size_t)heads * hs;
    for (int o = threadIdx.x; o < BS * BS; o += blockDim.x) {
        const int i = o / BS, j = o % BS;
        const typename TA::T* q = A + ((size_t)n * rows_q + (size_t)qk.x * BS + i) * state + (size_t)h * hs;
        const typename TA::T* k = B + ((size_t)n * rows_k + (size_t)qk.y * BS + j) * state + (size_t)h * hs;
        float s = 0.f;
        for (int x = 0; x < hs; ++x) s = fmaf(TA::to_f32(q[x]), TA::to_f32(k[x]), s);
        S[(((size_t)n * heads + h) * blocks + b) * (BS * BS) + o] = TS::from_f32(s);
    }
}

template <class TS, class TB, int BS, bool TRANS>
__global__ void bst_xn_valu_kernel(const typename TS::T* __restrict__ S, const typename TB::T* __restrict__ Bm, typename TB::T* __restrict__ C,
                                   const int32_t* __restrict__ lut, int lut_stride, int blocks, int heads, int hs, int rows_b, int rows_c) {
    const int oc = blockIdx.x, h = blockIdx.y, n = blockIdx.z;
    const int32_t* hl = lut + (size_t)h * lut_stride;
    const int2 hdr = *reinterpret_cast<const int2*>(hl + 2 * oc);
    const size_t state = (size_t)heads * hs;
    const typename TS::T* sbase = S + ((size_t)n * heads + h) * blocks * (BS * BS);
    for (int o = threadIdx.x; o < BS * hs; o += blockDim.x) {
        const int row = o / hs, c = o % hs;
        float s = 0.f;
        for (int e = 0; e < hdr.y; ++e) {
            const int2 ent = *reinterpret_cast<const int2*>(hl + 2 * (hdr.x + e));
            const typename TS::T* sb = sbase + (size_t)ent.x * (BS * BS);
            const typename TB::T* bp = Bm + ((size_t)n * rows_b + (size_t)ent.y * BS) * state + (size_t)h * hs + c;
#pragma unroll 4
            for (int x = 0; x < BS; ++x) {
                const float w = TS::to_f32(TRANS ? sb[x * BS + row] : sb[row * BS + x]);
                s = fmaf(w, TB::to_f32(bp[(size_t)x * state]), s);
            }
        }
        C[((size_t)n * rows_c + (size_t)oc * BS + row) * state + (size_t)h * hs + c] = TB::from_f32(s);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Softmax over the blocks of one query row-block.  grid (ctx_blks_q, heads, batch), BS * 8 threads: 8 adjacent lanes per
// query row, each covering VEC = BS / 8 consecutive keys of every block, so one block is read as BS*BS contiguous
// elements by the workgroup.  Three passes over the row's blocks (max, sum, write); the row's working set (<= a few
// tens of KiB) stays in L2 between them.
// ------------------------------------------------------------------------------------------------------------------
template <int BS> struct MaskT { typedef uint32_t T; };
template <> struct MaskT<64> { typedef unsigned long long T; };
template <> struct MaskT<16> { typedef uint16_t T; };
template <> struct MaskT<8> { typedef uint8_t T; };

__device__ __forceinline__ float row8_max(float v) {
    v = fmaxf(v, __shfl_xor(v, 1)); v = fmaxf(v, __shfl_xor(v, 2)); v = fmaxf(v, __shfl_xor(v, 4));
    return v;
}
__device__ __forceinline__ float row8_sum(float v) {
    v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4);
    return v;
}

// VEC consecutive 16-bit elements as packed words (one global access of 2 * VEC bytes)
template <int VEC> struct Raw16 { uint32_t w[(VEC + 1) / 2]; };
template <int VEC>
__device__ __forceinline__ Raw16<VEC> load_raw16(const uint16_t* p) {
    Raw16<VEC> r;
    if constexpr (VEC == 1) r.w[0] = *p;
    else if constexpr (VEC == 2) r.w[0] = *reinterpret_cast<const uint32_t*>(p);
    else if constexpr (VEC == 4) { const uint2 v = *reinterpret_cast<const uint2*>(p); r.w[0] = v.x; r.w[1] = v.y; }
    else { const uint4 v = *reinterpret_cast<const uint4*>(p); r.w[0] = v.x; r.w[1] = v.y; r.w[2] = v.z; r.w[3] = v.w; }
    return r;
}
template <int VEC>
__device__ __forceinline__ void store_raw16(uint16_t* p, const Raw16<VEC>& r) {
    if constexpr (VEC == 1) *p = (uint16_t)r.w[0];
    else if constexpr (VEC == 2) *reinterpret_cast<uint32_t*>(p) = r.w[0];
    else if constexpr (VEC == 4) *reinterpret_cast<uint2*>(p) = make_uint2(r.w[0], r.w[1]);
    else *reinterpret_cast<uint4*>(p) = make_uint4(r.w[0], r.w[1], r.w[2], r.w[3]);
}
template <class DT, int VEC>
__device__ __forceinline__ float raw_get(const Raw16<VEC>& r, int i) { return DT::to_f32((uint16_t)(r.w[i >> 1] >> (16 * (i & 1)))); }
template <class DT, int VEC>
__device__ __forceinline__ void raw_set(Raw16<VEC>& r, int i, float v) {
    const uint32_t x = DT::from_f32(v);
    if (i & 1) r.w[i >> 1] |= x << 16; else r.w[i >> 1] = x;
}

// Rows whose block list fits SM_CACHE entries (the usual case: 19 at BASELINE configs[4]) are read ONCE: all loads are
// issued up front and the (scaled, masked) values live in registers across the max / sum / write steps.  Longer rows
// fall back to three passes over global memory (the row then stays in L2).
template <int VEC> struct SmCache { static constexpr int N = VEC >= 8 ? 12 : 24; };

template <class TX, class TY, int BS>
__global__ void __launch_bounds__(BS * 8)
bst_softmax_kernel(const typename TX::T* __restrict__ X, typename TY::T* __restrict__ Y, const int32_t* __restrict__ lut, int lut_stride,
                   const typename MaskT<BS>::T* __restrict__ mask, int mask_stride, int blocks, int heads, float scale) {
    constexpr int VEC = BS / 8;
    constexpr int CACHE = SmCache<VEC>::N;
    typedef typename MaskT<BS>::T MT;
    const int Q = blockIdx.x, h = blockIdx.y, n = blockIdx.z;
    const int row = threadIdx.x >> 3, cg = threadIdx.x & 7;
    const int32_t* hl = lut + (size_t)h * lut_stride;
    const int2 hdr = *reinterpret_cast<const int2*>(hl + 2 * Q);
    if (hdr.y == 0) return;
    const int32_t* ent = hl + 2 * hdr.x;
    const size_t base = ((size_t)n * heads + h) * blocks * (BS * BS) + (size_t)row * BS + cg * VEC;
    const MT* mrow = mask ? mask + (size_t)h * mask_stride + (size_t)row * blocks : nullptr;
    const float NEG = -3.402823466e+38f;
    const float sc2 = scale * 1.4426950408889634f;          // exp(x) = exp2(x * log2 e)

    auto load = [&](int b, float (&v)[VEC]) {
        const Raw16<VEC> raw = load_raw16<VEC>(X + base + (size_t)b * (BS * BS));
        const MT m = mrow ? mrow[b] : (MT)~(MT)0;
#pragma unroll
        for (int i = 0; i < VEC; ++i) v[i] = ((m >> (cg * VEC + i)) & 1) ? raw_get<TX, VEC>(raw, i) * sc2 : NEG;
    };
    auto store = [&](int b, const float (&v)[VEC], float mx, float rcp) {
        Raw16<VEC> out;
#pragma unroll
        for (int i = 0; i < VEC; ++i) raw_set<TY, VEC>(out, i, exp2f(v[i] - mx) * rcp);
        store_raw16<VEC>(Y + base + (size_t)b * (BS * BS), out);
    };

    if (hdr.y <= CACHE) {
        // loads only in the first loop (each iteration is its own basic block because of the length test: a use of the
        // loaded value in there would put a full memory round trip into every iteration)
        Raw16<VEC> raw[CACHE];
        MT mk[CACHE];
#pragma unroll
        for (int e = 0; e < CACHE; ++e)
            if (e < hdr.y) {
                const int b = ent[2 * e];
                raw[e] = load_raw16<VEC>(X + base + (size_t)b * (BS * BS));
                mk[e] = mrow ? mrow[b] : (MT)~(MT)0;
            }
        float v[CACHE][VEC];
#pragma unroll
        for (int e = 0; e < CACHE; ++e)
#pragma unroll
            for (int i = 0; i < VEC; ++i)
                v[e][i] = (e < hdr.y && ((mk[e] >> (cg * VEC + i)) & 1)) ? raw_get<TX, VEC>(raw[e], i) * sc2 : NEG;
        float mx = NEG;
#pragma unroll
        for (int e = 0; e < CACHE; ++e)
#pragma unroll
            for (int i = 0; i < VEC; ++i) mx = fmaxf(mx, v[e][i]);
        mx = row8_max(mx);
        // one v_exp_f32 per element (arguments <= 0: no range fix-up needed)
        float sum = 0.f;
#pragma unroll
        for (int e = 0; e < CACHE; ++e)
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                v[e][i] = __builtin_amdgcn_exp2f(v[e][i] - mx);
                sum += (e < hdr.y) ? v[e][i] : 0.f;
            }
        sum = row8_sum(sum);
        const float rcp = 1.0f / sum;
#pragma unroll
        for (int e = 0; e < CACHE; ++e)
            if (e < hdr.y) {
                Raw16<VEC> out;
#pragma unroll
                for (int i = 0; i < VEC; ++i) raw_set<TY, VEC>(out, i, v[e][i] * rcp);
                store_raw16<VEC>(Y + base + (size_t)ent[2 * e] * (BS * BS), out);
            }
        return;
    }
    float mx = NEG;
    for (int e = 0; e < hdr.y; ++e) {
        float v[VEC];
        load(ent[2 * e], v);
#pragma unroll
        for (int i = 0; i < VEC; ++i) mx = fmaxf(mx, v[i]);
    }
    mx = row8_max(mx);
    float sum = 0.f;
    for (int e = 0; e < hdr.y; ++e) {
        float v[VEC];
        load(ent[2 * e], v);
#pragma unroll
        for (int i = 0; i < VEC; ++i) sum += exp2f(v[i] - mx);
    }
    sum = row8_sum(sum);
    const float rcp = 1.0f / sum;
    for (int e = 0; e < hdr.y; ++e) {
        float v[VEC];
        const int b = ent[2 * e];
        load(b, v);
        store(b, v, mx, rcp);
    }
}

template <class T16, int BS>
__global__ void __launch_bounds__(BS * 8)
bst_softmax_grad_kernel(const typename T16::T* __restrict__ DY, const typename T16::T* __restrict__ Y, typename T16::T* __restrict__ DX,
                        const int32_t* __restrict__ lut, int lut_stride, int blocks, int heads, float scale) {
    constexpr int VEC = BS / 8;
    constexpr int CACHE = SmCache<VEC>::N;
    const int Q = blockIdx.x, h = blockIdx.y, n = blockIdx.z;
    const int row = threadIdx.x >> 3, cg = threadIdx.x & 7;
    const int32_t* hl = lut + (size_t)h * lut_stride;
    const int2 hdr = *reinterpret_cast<const int2*>(hl + 2 * Q);
    if (hdr.y == 0) return;
    const int32_t* ent = hl + 2 * hdr.x;
    const size_t base = ((size_t)n * heads + h) * blocks * (BS * BS) + (size_t)row * BS + cg * VEC;
    auto emit = [&](size_t o, const Raw16<VEC>& d, const Raw16<VEC>& y, float s) {
        Raw16<VEC> out;
#pragma unroll
        for (int i = 0; i < VEC; ++i) raw_set<T16, VEC>(out, i, (raw_get<T16, VEC>(d, i) - s) * raw_get<T16, VEC>(y, i) * scale);
        store_raw16<VEC>(DX + o, out);
    };
    if (hdr.y <= CACHE) {
        Raw16<VEC> d[CACHE], y[CACHE];
#pragma unroll
        for (int e = 0; e < CACHE; ++e)
            if (e < hdr.y) {
                const size_t o = base + (size_t)ent[2 * e] * (BS * BS);
                d[e] = load_raw16<VEC>(DY + o);
                y[e] = load_raw16<VEC>(Y + o);
            }
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < CACHE; ++e)
            if (e < hdr.y)
#pragma unroll
                for (int i = 0; i < VEC; ++i) s = fmaf(raw_get<T16, VEC>(d[e], i), raw_get<T16, VEC>(y[e], i), s);
        s = row8_sum(s);
#pragma unroll
        for (int e = 0; e < CACHE; ++e)
            if (e < hdr.y) emit(base + (size_t)ent[2 * e] * (BS * BS), d[e], y[e], s);
        return;
    }
    float s = 0.f;
    for (int e = 0; e < hdr.y; ++e) {
        const size_t o = base + (size_t)ent[2 * e] * (BS * BS);
        const Raw16<VEC> d = load_raw16<VEC>(DY + o), y = load_raw16<VEC>(Y + o);
#pragma unroll
        for (int i = 0; i < VEC; ++i) s = fmaf(raw_get<T16, VEC>(d, i), raw_get<T16, VEC>(y, i), s);
    }
    s = row8_sum(s);
    for (int e = 0; e < hdr.y; ++e) {
        const size_t o = base + (size_t)ent[2 * e] * (BS * BS);
        emit(o, load_raw16<VEC>(DY + o), load_raw16<VEC>(Y + o), s);
    }
}

// mask [H][BS][blocks] -> same (src/bst_softmax_op_gpu.cu:461-503).  grid (ceil(blocks / 64), BS, H), 64 threads.
// ------------------------------------------------------------------------------------------------------------------
// Scores + softmax in ONE launch (round 6; bsize 32): Y[n][h][b] = softmax over the query row of (scale * bf16(Q . K^T) + mask).
//
// The two-launch form (bst_nt_mfma_kernel, then bst_softmax_kernel) writes 192 MB of raw scores at BASELINE configs[4], reads them back and
// writes 192 MB of probabilities: 0.150 + 0.090 ms.  Here one workgroup owns a query block ROW (<= NTS_MAXT * 4 = 20 blocks; longer rows: the
// caller takes the two launches): its four waves take every fourth block of the row's list (nn_lut: header (offset, count) per query block,
// entries (block, key block)), multiply exactly as bst_nt_mfma_kernel does (same tile images, same fragment reads, same piece products in the
// same order, the tile rounded to the score type -- the raw scores are what the two-launch form stores) and PARK the rounded tile in LDS instead
// of memory (2 KiB per tile, the image bst_nt_mfma_kernel stages its stores through).  The softmax then runs over those images in the store
// layout -- lane (row i, piece p) owns 8 consecutive keys of two rows per tile: max and sum meet across the four lanes of a row by two
// shuffles and across the four waves through 1 KiB of LDS -- and the probabilities leave as contiguous 1 KiB stores.  The scores never reach
// memory.  The Q tile is fetched once per workgroup (each wave a quarter of the DMA instructions).
// Same arithmetic as bst_softmax_kernel: exp2((s - max) * 1) with s = score * scale * log2 e, masked keys = -FLT_MAX, 1 / sum.
// ------------------------------------------------------------------------------------------------------------------
constexpr int NTS_MAXT = 5;          // tiles per wave: rows of up to 20 blocks

// GRAD: the same launch shape for the backward pair bst_nt (dP = E . V^T) + bst_softmax_grad: dX = (dP - sum_row(dP * P)) * P * scale with dP rounded
// to the score type as the two launches would store it; P (the forward's probabilities) is read in the store layout, requested before the tile loop.
template <class TA, class TS, int NCH, bool SPLIT, bool GRAD = false>
__global__ void __launch_bounds__(256)
bst_nt_softmax_kernel(const typename TA::T* __restrict__ A, const typename TA::T* __restrict__ B, typename TS::T* __restrict__ Y,
                      const int32_t* __restrict__ lut, int lut_stride, const uint32_t* __restrict__ mask, int mask_stride, int blocks, int heads,
                      int batch, int hs, int rows_q, int rows_k, int ctx_q, float scale, const typename TS::T* __restrict__ Pin = nullptr) {
    typedef typename TA::T T;
    typedef NtTile<T> TL;
    constexpr int BS = 32;
    static_assert(!SPLIT || !TA::is16, "the three-piece split is for fp32 activations");
    __shared__ __attribute__((aligned(16))) unsigned char lds_k[4][NCH][TL::BYTES];      // per wave: the K tile of its current block
    __shared__ __attribute__((aligned(16))) unsigned char lds_q[NCH][TL::BYTES];         // the row's Q tile (all waves)
    __shared__ __attribute__((aligned(16))) unsigned char olds[4][NTS_MAXT][32 * 64];    // per wave: its finished tiles (score type)
    // (row max / row sum of every wave: 256 B at the head of that wave's K buffer, idle by then -- 80 KiB in all: two workgroups per CU)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, hh = lane >> 5;
    int n, h, Q;
    if (!xcd_head_map(ctx_q, heads, batch, n, h, Q)) return;
    const int32_t* hl = lut + (size_t)h * lut_stride;
    const int2 hdr = *reinterpret_cast<const int2*>(hl + 2 * Q);
    const int cnt = __builtin_amdgcn_readfirstlane(hdr.y);
    if (cnt == 0) return;                                                                // (the whole workgroup: no barrier is passed by a part of it)
    const int32_t* ent = hl + 2 * __builtin_amdgcn_readfirstlane(hdr.x);
    const size_t state = (size_t)heads * hs;
    const int lrow = lane / TL::PPR, lp = lane % TL::PPR;
    const T* abase = A + (size_t)n * rows_q * state + (size_t)h * hs + (size_t)Q * BS * state;
    const T* bbase = B + (size_t)n * rows_k * state + (size_t)h * hs;
    const uint32_t a_k = lds_addr_of(&lds_k[wave][0][0]), a_q = lds_addr_of(&lds_q[0][0]);
    size_t soff[TL::NI];
#pragma unroll
    for (int i = 0; i < TL::NI; ++i) {
        const int row = TL::RPI * i + lrow;
        soff[i] = (size_t)row * state + (size_t)((lp ^ TL::sw(row)) * TL::EPP);
    }
    auto dma_tile = [&](const T* tile0, uint32_t dst) {
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int i = 0; i < TL::NI; ++i)
                glds16_asm(tile0 + soff[i] + 32 * c, __builtin_amdgcn_readfirstlane(dst + c * TL::BYTES + i * 1024));
    };
    auto frag = [&](const unsigned char* tile, float (&v)[16]) {
        constexpr int NP = 16 / TL::EPP;
#pragma unroll
        for (int g = 0; g < NP; ++g) {
            const uint4 x = *reinterpret_cast<const uint4*>(tile + r * TL::ROWB + (((NP * hh + g) ^ TL::sw(r)) << 4));
            const uint32_t w[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) v[4 * g + i] = __builtin_bit_cast(float, w[i]);
        }
    };
    // the Q tile: instruction (c, i) by wave (c * NI + i) % 4; my first K tile behind it
    {
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int i = 0; i < TL::NI; ++i)
                if (((c * TL::NI + i) & 3) == wave) glds16_asm(abase + soff[i] + 32 * c, __builtin_amdgcn_readfirstlane(a_q + c * TL::BYTES + i * 1024));
    }
    const int mine = wave < cnt ? (cnt - wave + 3) >> 2 : 0;                            // my tiles: entries wave, wave + 4, ...
    if (mine > 0) dma_tile(bbase + (size_t)ent[2 * wave + 1] * BS * state, a_k);
    // block ids and mask bytes of my tiles, for the softmax phase (requested here: their latency hides behind the tile loop; in the phase
    // itself 30 dependent loads per wave cost more than the launch the fusion saves) -- lane (row i0 / i0 + 16, piece p): keys 8p .. 8p + 7
    const int i0 = lane >> 2, p = lane & 3;
    const uint32_t* mrow = mask ? mask + (size_t)h * mask_stride : nullptr;
    int bt[NTS_MAXT];
    uint32_t mk[GRAD ? 1 : NTS_MAXT][2];
    uint4 pv[GRAD ? NTS_MAXT : 1][2];                                                    // GRAD: my 8 probabilities of (tile, row i0 / i0 + 16)
    typename TS::T* ybase = Y + ((size_t)n * heads + h) * blocks * (BS * BS);
#pragma unroll
    for (int t = 0; t < NTS_MAXT; ++t) {
        bt[t] = 0;
        if constexpr (!GRAD) mk[t][0] = mk[t][1] = 0xffffffffu;
        else pv[t][0] = pv[t][1] = make_uint4(0u, 0u, 0u, 0u);
        if (t < mine) {
            bt[t] = __builtin_amdgcn_readfirstlane(ent[2 * (wave + 4 * t)]);
            if constexpr (!GRAD) {
                if (mrow) { mk[t][0] = mrow[(size_t)i0 * blocks + bt[t]]; mk[t][1] = mrow[(size_t)(i0 + 16) * blocks + bt[t]]; }
            } else {
                const typename TS::T* pb = Pin + ((size_t)n * heads + h) * blocks * (BS * BS) + (size_t)bt[t] * (BS * BS) + p * 8;
                pv[t][0] = *reinterpret_cast<const uint4*>(pb + (size_t)i0 * BS);
                pv[t][1] = *reinterpret_cast<const uint4*>(pb + (size_t)(i0 + 16) * BS);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                                                     // the Q tile is everyone's
    // Q fragments / pieces once
    float fq[TA::is16 ? 1 : NCH][16];
    uint4 rq[TA::is16 ? NCH : 1][2];
    uint4 qp[SPLIT ? NCH : 1][3][2];
    auto split3 = [](const float (&v)[16], uint4 (&p)[3][2]) {                           // as bst_nt_mfma_kernel
        auto pack2 = [](float lo, float hi) { return bf16_pack2(lo, hi); };
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            uint32_t a[4], b[4], c3[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float v0 = v[8 * kk + 2 * i], v1 = v[8 * kk + 2 * i + 1];
                a[i] = pack2(v0, v1);
                const float r0 = v0 - __builtin_bit_cast(float, a[i] << 16), r1 = v1 - __builtin_bit_cast(float, a[i] & 0xffff0000u);
                b[i] = pack2(r0, r1);
                c3[i] = pack2(r0 - __builtin_bit_cast(float, b[i] << 16), r1 - __builtin_bit_cast(float, b[i] & 0xffff0000u));
            }
            p[0][kk] = make_uint4(a[0], a[1], a[2], a[3]);
            p[1][kk] = make_uint4(b[0], b[1], b[2], b[3]);
            p[2][kk] = make_uint4(c3[0], c3[1], c3[2], c3[3]);
        }
    };
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        if constexpr (TA::is16) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) rq[c][kk] = *reinterpret_cast<const uint4*>(&lds_q[c][0] + r * TL::ROWB + (((2 * kk + hh) ^ TL::sw(r)) << 4));
        } else {
            frag(&lds_q[c][0], fq[c]);
            if constexpr (SPLIT) split3(fq[c], qp[c]);
        }
    }
    for (int t = 0; t < mine; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                 // this tile's K has landed (the first time: the mask words too)
        float fk[TA::is16 ? 1 : NCH][16];
        uint4 rk[TA::is16 ? NCH : 1][2];
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            if constexpr (TA::is16) {
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) rk[c][kk] = *reinterpret_cast<const uint4*>(&lds_k[wave][c][0] + r * TL::ROWB + (((2 * kk + hh) ^ TL::sw(r)) << 4));
            } else {
                frag(&lds_k[wave][c][0], fk[c]);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                               // fragments in registers: the buffer is free
        if (t + 1 < mine) dma_tile(bbase + (size_t)ent[2 * (wave + 4 * (t + 1)) + 1] * BS * state, a_k);
        f32x16 acc, acc1;
#pragma unroll
        for (int i = 0; i < 16; ++i) { acc[i] = 0.f; acc1[i] = 0.f; }
        if constexpr (SPLIT) {
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                uint4 kp[3][2];
                split3(fk[c], kp);
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    acc1 = DTbf16::mfma32(kp[2][kk], qp[c][0][kk], acc1);
                    acc1 = DTbf16::mfma32(kp[0][kk], qp[c][2][kk], acc1);
                    acc1 = DTbf16::mfma32(kp[1][kk], qp[c][1][kk], acc1);
                    acc = DTbf16::mfma32(kp[1][kk], qp[c][0][kk], acc);
                    acc = DTbf16::mfma32(kp[0][kk], qp[c][1][kk], acc);
                    acc = DTbf16::mfma32(kp[0][kk], qp[c][0][kk], acc);
                }
            }
        } else if constexpr (TA::is16) {
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                acc = TA::mfma32(rk[c][0], rq[c][0], acc);
                acc1 = TA::mfma32(rk[c][1], rq[c][1], acc1);
            }
        } else {
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                if (c & 1) mma32_f32(fk[c], fq[c], acc1);
                else       mma32_f32(fk[c], fq[c], acc);
            }
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] += acc1[i];
        unsigned char* img = &olds[wave][0][0] + t * 2048;                               // park (bst_nt_mfma_kernel's image of a stored tile)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const uint32_t lo = (uint32_t)TS::from_f32(acc[4 * g + 0]) | ((uint32_t)TS::from_f32(acc[4 * g + 1]) << 16);
            const uint32_t hi = (uint32_t)TS::from_f32(acc[4 * g + 2]) | ((uint32_t)TS::from_f32(acc[4 * g + 3]) << 16);
            *reinterpret_cast<uint2*>(img + r * 64 + ((g ^ ((r >> 2) & 3)) << 4) + 8 * hh) = make_uint2(lo, hi);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#define XCH(w_, v_, i_) reinterpret_cast<float*>(&lds_k[v_][0][0])[32 * (w_) + (i_)]
    if constexpr (GRAD) {
        // ---- softmax gradient over the parked tiles of dP, in the store layout ----
        auto pair8 = [&](int t, int k, float (&d)[8], float (&y)[8]) {
            const int i = i0 + 16 * k;
            const uint4 x = *reinterpret_cast<const uint4*>(&olds[wave][0][0] + t * 2048 + i * 64 + ((p ^ ((i >> 2) & 3)) << 4));
            const uint32_t w[4] = {x.x, x.y, x.z, x.w}, q4[4] = {pv[t][k].x, pv[t][k].y, pv[t][k].z, pv[t][k].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                d[2 * j] = TS::to_f32((uint16_t)(w[j] & 0xffffu)); d[2 * j + 1] = TS::to_f32((uint16_t)(w[j] >> 16));
                y[2 * j] = TS::to_f32((uint16_t)(q4[j] & 0xffffu)); y[2 * j + 1] = TS::to_f32((uint16_t)(q4[j] >> 16));
            }
        };
        float sum[2] = {0.f, 0.f};
#pragma unroll
        for (int t = 0; t < NTS_MAXT; ++t)
            if (t < mine) {
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    float d[8], y[8];
                    pair8(t, k, d, y);
#pragma unroll
                    for (int j = 0; j < 8; ++j) sum[k] += d[j] * y[j];
                }
            }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            sum[k] += __shfl_xor(sum[k], 1); sum[k] += __shfl_xor(sum[k], 2);
            if (p == 0) XCH(1, wave, i0 + 16 * k) = sum[k];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 2; ++k) sum[k] = ((XCH(1, 0, i0 + 16 * k) + XCH(1, 1, i0 + 16 * k)) + XCH(1, 2, i0 + 16 * k)) + XCH(1, 3, i0 + 16 * k);
#pragma unroll
        for (int t = 0; t < NTS_MAXT; ++t)
            if (t < mine) {
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    float d[8], y[8];
                    pair8(t, k, d, y);
                    uint32_t o[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        o[j] = (uint32_t)TS::from_f32((d[2 * j] - sum[k]) * y[2 * j] * scale) | ((uint32_t)TS::from_f32((d[2 * j + 1] - sum[k]) * y[2 * j + 1] * scale) << 16);
                    *reinterpret_cast<uint4*>(ybase + (size_t)bt[t] * (BS * BS) + (size_t)(i0 + 16 * k) * BS + p * 8) = make_uint4(o[0], o[1], o[2], o[3]);
                }
            }
    } else {
    // ---- softmax over the parked tiles: lane -> (row i0 = lane >> 2 and i0 + 16, piece p = lane & 3: keys 8p .. 8p + 7) ----
    const float NEG = -3.402823466e+38f;
    const float sc2 = scale * 1.4426950408889634f;
    auto values = [&](int t, int k, uint32_t mbits, float (&v)[8]) {                    // scaled, masked scores of (tile t, row i0 + 16 k), my 8 keys
        const int i = i0 + 16 * k;
        const uint4 x = *reinterpret_cast<const uint4*>(&olds[wave][0][0] + t * 2048 + i * 64 + ((p ^ ((i >> 2) & 3)) << 4));
        const uint32_t w[4] = {x.x, x.y, x.z, x.w};
        const uint32_t m = mbits >> (8 * p);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v[2 * j] = ((m >> (2 * j)) & 1) ? TS::to_f32((uint16_t)(w[j] & 0xffffu)) * sc2 : NEG;
            v[2 * j + 1] = ((m >> (2 * j + 1)) & 1) ? TS::to_f32((uint16_t)(w[j] >> 16)) * sc2 : NEG;
        }
    };
    float mx[2] = {NEG, NEG};
#pragma unroll
    for (int t = 0; t < NTS_MAXT; ++t)
        if (t < mine) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                float v[8];
                values(t, k, mk[t][k], v);
#pragma unroll
                for (int j = 0; j < 8; ++j) mx[k] = fmaxf(mx[k], v[j]);
            }
        }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        mx[k] = fmaxf(mx[k], __shfl_xor(mx[k], 1)); mx[k] = fmaxf(mx[k], __shfl_xor(mx[k], 2));
        if (p == 0) XCH(0, wave, i0 + 16 * k) = mx[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 2; ++k) mx[k] = fmaxf(fmaxf(XCH(0, 0, i0 + 16 * k), XCH(0, 1, i0 + 16 * k)), fmaxf(XCH(0, 2, i0 + 16 * k), XCH(0, 3, i0 + 16 * k)));
    float sum[2] = {0.f, 0.f};
#pragma unroll
    for (int t = 0; t < NTS_MAXT; ++t)
        if (t < mine) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                float v[8];
                values(t, k, mk[t][k], v);
#pragma unroll
                for (int j = 0; j < 8; ++j) sum[k] += __builtin_amdgcn_exp2f(v[j] - mx[k]);
            }
        }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        sum[k] += __shfl_xor(sum[k], 1); sum[k] += __shfl_xor(sum[k], 2);
        if (p == 0) XCH(1, wave, i0 + 16 * k) = sum[k];
    }
    __syncthreads();
    float rcp[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) rcp[k] = 1.0f / (((XCH(1, 0, i0 + 16 * k) + XCH(1, 1, i0 + 16 * k)) + XCH(1, 2, i0 + 16 * k)) + XCH(1, 3, i0 + 16 * k));
#pragma unroll
    for (int t = 0; t < NTS_MAXT; ++t)
        if (t < mine) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                float v[8];
                values(t, k, mk[t][k], v);
                uint32_t o[4];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    o[j] = (uint32_t)TS::from_f32(__builtin_amdgcn_exp2f(v[2 * j] - mx[k]) * rcp[k]) | ((uint32_t)TS::from_f32(__builtin_amdgcn_exp2f(v[2 * j + 1] - mx[k]) * rcp[k]) << 16);
                *reinterpret_cast<uint4*>(ybase + (size_t)bt[t] * (BS * BS) + (size_t)(i0 + 16 * k) * BS + p * 8) = make_uint4(o[0], o[1], o[2], o[3]);
            }
        }
    }   // !GRAD
#undef XCH
}

template <int BS>
__global__ void bst_partial_ar_mask_kernel(const typename MaskT<BS>::T* __restrict__ in, typename MaskT<BS>::T* __restrict__ out,
                                           const int32_t* __restrict__ nt_lut, int blocks, int key) {
    typedef typename MaskT<BS>::T MT;
    const int b = blockIdx.x * 64 + threadIdx.x, qi = blockIdx.y, h = blockIdx.z;
    if (b >= blocks) return;
    const int2 qk = *reinterpret_cast<const int2*>(nt_lut + ((size_t)h * blocks + b) * 2);
    const size_t m = ((size_t)h * BS + qi) * blocks + b;
    const int K = qk.y * BS, q = qk.x * BS + qi;
    const int shift_a = BS - min(max(key - K, 0), BS);
    const int shift_b = min(max(BS - 1 + K - q, 0), BS);
    const int sh = min(shift_a, shift_b);
    const MT ones = (MT)~(MT)0;
    out[m] = in[m] & (sh >= BS ? (MT)0 : (MT)(ones >> sh));
}

}  // namespace bsmm
