// bsmm_xsmall0.h -- xprop for SMALL minibatches on feature_axis = 0 (round 6): bsize 32, 16-bit storage types, no plan needed.
//
// The reference's own benchmark runs exactly here (test/blocksparse_matmul_bench.py:37-78: feature axis 0, minibatch 64, ~13 MB of weights at
// every sparsity from 100 % to 1.4 %).  A pass is then ONE read of W (13 MB: ~2 us of HBM time) -- but the plan kernels have (N / 128) x groups
// = 5 .. 40 units for 256 CUs, and the per-segment kernel walks a whole output column (80 blocks when dense) in one serial chain per wave and,
// for fprop, pays a transposing pre-pass over W: 46 - 62 / 27 - 46 us (fprop / bprop) at those shapes.  Here, as in bsmm_xsmall.h on feature axis 1:
//   workgroup = one output block x 64 minibatch columns, XS0_NW = 16 waves; wave v multiplies the column's entries v, v + 16, ...; the partial
//   32 x 64 tiles meet in LDS (fp32, one barrier, summed in wave order) and are rounded ONCE.
// On feature axis 0 the contraction index of both operands is a ROW index (X is (C, N): 8 consecutive c of one minibatch column are 8 rows), so
// the activation tile of an entry -- [32 rows][64 columns], 128-byte row pieces -- goes through a wave-private 4 KiB of LDS as two [32][64 B]
// images and comes back with ds_read_b64_tr_b16; fprop's weight block (contraction over ITS rows too) likewise through 2 KiB: no transposed copy
// of W, no pre-pass, no workspace.  bprop reads its weight fragments (contraction over the block's columns) straight from global memory.  The
// next entry's six 16-byte loads per lane are in flight under the current entry's LDS round trip and MFMAs.
// Needs N % 8 == 0 (16-byte row pieces), no locks, no gate.
#pragma once
#include "bsmm_common.h"
#include "bsmm_updat_tr.h"   // ds_tr16

namespace bsmm {

constexpr int XS0_NW = 16;                                 // waves per workgroup = ways a column's entry list is cut
constexpr int XS0_C = 64;                                  // minibatch columns per workgroup
constexpr int XS0_PART = 32 * XS0_C * 4;                   // one wave's partial tile [2 column tiles][16 registers][64 lanes] fp32 = 8 KiB ...
constexpr int XS0_LDS = XS0_NW * XS0_PART;                 // ... whose first 6 KiB are the wave's staging (X tile 4 KiB | W block 2 KiB) during the loop
static_assert(XS0_PART >= 6144, "the staging of a wave lives inside its own partial tile");

template <class DT, bool TRANSW>
__global__ void __launch_bounds__(64 * XS0_NW)
xsmall32_a0_kernel(const typename DT::T* __restrict__ X, const typename DT::T* __restrict__ W, typename DT::T* __restrict__ Y,
                   const int32_t* __restrict__ lut, int N) {
    typedef typename DT::T T;
    static_assert(DT::is16, "small-minibatch kernel: 16-bit storage types");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int4 hdr = *reinterpret_cast<const int4*>(lut + 4 * blockIdx.x);
    const int cnt = __builtin_amdgcn_readfirstlane(hdr.y), ob = __builtin_amdgcn_readfirstlane(hdr.z);
    const int2* ent = reinterpret_cast<const int2*>(lut) + __builtin_amdgcn_readfirstlane(hdr.x);
    const int n0 = blockIdx.y * XS0_C;

    f32x16 acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

    unsigned char* xst = smem + wave * XS0_PART;           // my X staging: column tile t at t * 2048, [32 rows][64 B]
    unsigned char* wst = xst + 4096;                       // my W staging (fprop): [32 rows][64 B]
    // X tile loads: piece p = lane + 64 i of the tile's 256 16-byte pieces: row p >> 3, piece-in-row p & 7 (columns 8 (p & 7) ..); columns past N
    // are clamped re-reads of the row's last piece (their products land in output columns that are never stored)
    // (instruction i covers rows 8 i .. 8 i + 7: the same piece-in-row for all four)
    const int pc = lane & 7, row0 = lane >> 3;
    const int col = min(n0 + 8 * pc, N - 8);
    const int xoff0 = row0 * N + col, xoff1 = xoff0 + 8 * N, xoff2 = xoff0 + 16 * N, xoff3 = xoff0 + 24 * N;
    const int xdst0 = (pc >> 2) * 2048 + row0 * 64 + (pc & 3) * 16, xdst1 = xdst0 + 8 * 64, xdst2 = xdst0 + 16 * 64, xdst3 = xdst0 + 24 * 64;
    // fragment addresses (bsmm_xflow.h, TRANSW): lane (column = lane & 31 of the image, K half h) receives rows 16 kk + 8 h + {0..3 | 4..7}
    const int g16 = lane >> 4, t16 = lane & 15;
    const int frag = (8 * h + (t16 >> 2)) * 64 + (16 * (g16 & 1) + 4 * (t16 & 3)) * 2;

    // (plain scalars and a macro, not arrays behind a lambda: hipcc keeps arrays that a lambda captures by reference in scratch memory, with a
    //  vmcnt(0) around every access -- the first version of this kernel spent two thirds of its time there)
    uint4 xl0 = zero_u4(), xl1 = zero_u4(), xl2 = zero_u4(), xl3 = zero_u4(), wl0 = zero_u4(), wl1 = zero_u4();
#define XS0_LOAD(e_)                                                                                                        \
    do {                                                                                                                    \
        const int2 cw_ = ent[e_];                                                                                           \
        const int c_ = __builtin_amdgcn_readfirstlane(cw_.x), w_ = __builtin_amdgcn_readfirstlane(cw_.y);                   \
        const T* xb_ = X + (size_t)c_ * 32 * N;                                                                             \
        xl0 = *reinterpret_cast<const uint4*>(xb_ + xoff0);                                                                 \
        xl1 = *reinterpret_cast<const uint4*>(xb_ + xoff1);                                                                 \
        xl2 = *reinterpret_cast<const uint4*>(xb_ + xoff2);                                                                 \
        xl3 = *reinterpret_cast<const uint4*>(xb_ + xoff3);                                                                 \
        const T* wb_ = W + (size_t)w_ * 1024;                                                                               \
        if constexpr (TRANSW) {                                                                                             \
            wl0 = *reinterpret_cast<const uint4*>(wb_ + lane * 16);                                                         \
            wl1 = *reinterpret_cast<const uint4*>(wb_ + lane * 16 + 8);                                                     \
        } else {   /* bprop: A[m = ci][k = ko]: 8 consecutive outputs of row ci = 16 contiguous bytes */                    \
            wl0 = *reinterpret_cast<const uint4*>(wb_ + r * 32 + 8 * h);                                                    \
            wl1 = *reinterpret_cast<const uint4*>(wb_ + r * 32 + 16 + 8 * h);                                               \
        }                                                                                                                   \
    } while (0)
    if (wave < cnt) XS0_LOAD(wave);
    for (int e = wave; e < cnt; e += XS0_NW) {
        *reinterpret_cast<uint4*>(xst + xdst0) = xl0;
        *reinterpret_cast<uint4*>(xst + xdst1) = xl1;
        *reinterpret_cast<uint4*>(xst + xdst2) = xl2;
        *reinterpret_cast<uint4*>(xst + xdst3) = xl3;
        uint4 wq0, wq1;
        if constexpr (TRANSW) {
            *reinterpret_cast<uint4*>(wst + lane * 32) = wl0;
            *reinterpret_cast<uint4*>(wst + lane * 32 + 16) = wl1;
        } else {
            wq0 = wl0; wq1 = wl1;
        }
        if (e + XS0_NW < cnt) XS0_LOAD(e + XS0_NW);        // the next entry's loads, under this entry's LDS round trip and MFMAs
        uint4 xf[2][2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const unsigned char* p = xst + t * 2048 + 16 * kk * 64 + frag;
                const uint2 lo = ds_tr16(p), hi = ds_tr16(p + 4 * 64);
                xf[t][kk] = make_uint4(lo.x, lo.y, hi.x, hi.y);
            }
        if constexpr (TRANSW) {
            {
                const uint2 lo = ds_tr16(wst + frag), hi = ds_tr16(wst + frag + 4 * 64);
                wq0 = make_uint4(lo.x, lo.y, hi.x, hi.y);
            }
            {
                const uint2 lo = ds_tr16(wst + 16 * 64 + frag), hi = ds_tr16(wst + 16 * 64 + frag + 4 * 64);
                wq1 = make_uint4(lo.x, lo.y, hi.x, hi.y);
            }
        }
        acc[0] = DT::mfma32(wq0, xf[0][0], acc[0]);
        acc[1] = DT::mfma32(wq0, xf[1][0], acc[1]);
        acc[0] = DT::mfma32(wq1, xf[0][1], acc[0]);
        acc[1] = DT::mfma32(wq1, xf[1][1], acc[1]);
    }
#undef XS0_LOAD

    // the partial tiles meet in LDS: part[wave][t][reg][lane] (my staging is dead now: LDS operations of a wave execute in order)
    const int nparts = min(cnt, XS0_NW);
    if (wave < nparts) {
        float* part = reinterpret_cast<float*>(smem + wave * XS0_PART);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) part[(t * 16 + reg) * 64 + lane] = acc[t][reg];
    }
    __syncthreads();
    // D[o][n]: col n = lane & 31 (+ 32 t), rows o = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)  ->  Y[(ob * 32 + o) * N + n0 + n]
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int el = threadIdx.x + 1024 * i;             // (t * 16 + reg) * 64 + lane
        float s = 0.f;
        for (int v = 0; v < nparts; ++v) s += reinterpret_cast<const float*>(smem + v * XS0_PART)[el];
        const int ln = el & 63, treg = el >> 6, t = treg >> 4, reg = treg & 15;
        const int o = (reg & 3) + 8 * (reg >> 2) + 4 * (ln >> 5), n = n0 + 32 * t + (ln & 31);
        if (n < N) Y[(size_t)(ob * 32 + o) * N + n] = DT::from_f32(s);
    }
}

}  // namespace bsmm

// ---- bsize 16 (the reference benchmark's second block size on this axis) -------------------------------------------------------------------------
// Same decomposition: workgroup = one output block (16 features) x 64 minibatch columns, 16 waves over the column's entries, partial tiles meet in
// LDS.  One entry = one v_mfma_f32_16x16x16 per 16-column tile: A = W^T (fprop: the 512-byte block through LDS, one transposing read) or W (bprop:
// 8 contiguous bytes per lane straight from global memory), B = the entry's activation tile [16 rows][64 columns] (2 KiB, the plain image; lane
// (column, K group g) receives rows 4 g .. 4 g + 3 by one transposing read per tile).
namespace bsmm {

constexpr int XS16_PART = 16 * XS0_C * 4;                  // one wave's partial tile [4 column tiles][4 registers][64 lanes] fp32 = 4 KiB ...
constexpr int XS16_LDS = XS0_NW * XS16_PART;               // ... whose first 2.5 KiB are the wave's staging (X tile 2 KiB | W block 512 B) during the loop
static_assert(XS16_PART >= 2560, "the staging of a wave lives inside its own partial tile");

template <class DT, bool TRANSW>
__global__ void __launch_bounds__(64 * XS0_NW)
xsmall16_a0_kernel(const typename DT::T* __restrict__ X, const typename DT::T* __restrict__ W, typename DT::T* __restrict__ Y,
                   const int32_t* __restrict__ lut, int N) {
    typedef typename DT::T T;
    static_assert(DT::is16, "small-minibatch kernel: 16-bit storage types");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int4 hdr = *reinterpret_cast<const int4*>(lut + 4 * blockIdx.x);
    const int cnt = __builtin_amdgcn_readfirstlane(hdr.y), ob = __builtin_amdgcn_readfirstlane(hdr.z);
    const int2* ent = reinterpret_cast<const int2*>(lut) + __builtin_amdgcn_readfirstlane(hdr.x);
    const int n0 = blockIdx.y * XS0_C;

    f32x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

    unsigned char* xst = smem + wave * XS16_PART;          // my X staging: [16 rows][128 B]
    unsigned char* wst = xst + 2048;                       // my W staging (fprop): [16 rows][32 B]
    // X tile loads: instruction i covers rows 8 i .. 8 i + 7, lane -> (row lane >> 3, 16-byte piece lane & 7); columns past N are clamped re-reads
    const int pc = lane & 7, row0 = lane >> 3;
    const int col = min(n0 + 8 * pc, N - 8);
    const int xoff0 = row0 * N + col, xoff1 = xoff0 + 8 * N;
    const int xdst0 = row0 * 128 + pc * 16, xdst1 = xdst0 + 8 * 128;
    // transposing reads: the 16-lane group g = lane >> 4 points at rows 4 g .. 4 g + 3 of a [..][16 columns] patch, lane t16 at row t16 >> 2, columns 4 (t16 & 3) ..
    const int g = lane >> 4, t16 = lane & 15;
    const int xfrag = (4 * g + (t16 >> 2)) * 128 + 4 * (t16 & 3) * 2;
    const int wfrag = (4 * g + (t16 >> 2)) * 32 + 4 * (t16 & 3) * 2;

    uint4 xl0 = zero_u4(), xl1 = zero_u4();
    uint2 wl = make_uint2(0u, 0u);
#define XS16_LOAD(e_)                                                                                                       \
    do {                                                                                                                    \
        const int2 cw_ = ent[e_];                                                                                           \
        const int c_ = __builtin_amdgcn_readfirstlane(cw_.x), w_ = __builtin_amdgcn_readfirstlane(cw_.y);                   \
        const T* xb_ = X + (size_t)c_ * 16 * N;                                                                             \
        xl0 = *reinterpret_cast<const uint4*>(xb_ + xoff0);                                                                 \
        xl1 = *reinterpret_cast<const uint4*>(xb_ + xoff1);                                                                 \
        const T* wb_ = W + (size_t)w_ * 256;                                                                                \
        if constexpr (TRANSW) wl = *reinterpret_cast<const uint2*>(wb_ + lane * 4);                 /* the block as it lies */   \
        else                  wl = *reinterpret_cast<const uint2*>(wb_ + t16 * 16 + 4 * g);         /* A[m = ci][k = 4 g ..] */  \
    } while (0)
    if (wave < cnt) XS16_LOAD(wave);
    for (int e = wave; e < cnt; e += XS0_NW) {
        *reinterpret_cast<uint4*>(xst + xdst0) = xl0;
        *reinterpret_cast<uint4*>(xst + xdst1) = xl1;
        uint2 wq;
        if constexpr (TRANSW) *reinterpret_cast<uint2*>(wst + lane * 8) = wl;
        else wq = wl;
        if (e + XS0_NW < cnt) XS16_LOAD(e + XS0_NW);
        uint2 xf[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) xf[t] = ds_tr16(xst + xfrag + t * 32);
        if constexpr (TRANSW) wq = ds_tr16(wst + wfrag);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = DT::mfma16k16(wq, xf[t], acc[t]);
    }
#undef XS16_LOAD

    const int nparts = min(cnt, XS0_NW);
    if (wave < nparts) {
        float* part = reinterpret_cast<float*>(smem + wave * XS16_PART);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) part[(t * 4 + i) * 64 + lane] = acc[t][i];
    }
    __syncthreads();
    // D[o][n]: col n = lane & 15 (+ 16 t), rows o = 4 (lane >> 4) + i  ->  Y[(ob * 16 + o) * N + n0 + n]: one element per thread
    {
        const int el = threadIdx.x;                        // (t * 4 + i) * 64 + lane
        float s = 0.f;
        for (int v = 0; v < nparts; ++v) s += reinterpret_cast<const float*>(smem + v * XS16_PART)[el];
        const int ln = el & 63, ti = el >> 6, t = ti >> 2, i = ti & 3;
        const int o = 4 * (ln >> 4) + i, n = n0 + 16 * t + (ln & 15);
        if (n < N) Y[(size_t)(ob * 16 + o) * N + n] = DT::from_f32(s);
    }
}

}  // namespace bsmm

// ---- bsize 8 (the reference benchmark's third block size; before: the V_FMA kernel, 187 - 220 us per pass at its shapes) -----------------------------
// The bsize-16 kernel over PAIRS of entries: two 8-row activation tiles stack to the [16 rows][64 columns] image, the two 8 x 8 weight blocks to the
// K = 16 operand of v_mfma_f32_16x16x16 (fprop: [16 rows = (entry, ci)][8 outputs] through LDS, transposing read; bprop: lane (ci, K group g) takes
// W_{g >> 1}[ci][4 (g & 1) ..] straight from global memory); rows 8 .. 15 of the 16 x 16 result belong to nobody.  A column with an odd number of
// entries multiplies its last one against a zero block (and re-reads that entry's activations: finite values, times zero).
namespace bsmm {

template <class DT, bool TRANSW>
__global__ void __launch_bounds__(64 * XS0_NW)
xsmall8_a0_kernel(const typename DT::T* __restrict__ X, const typename DT::T* __restrict__ W, typename DT::T* __restrict__ Y,
                  const int32_t* __restrict__ lut, int N) {
    typedef typename DT::T T;
    static_assert(DT::is16, "small-minibatch kernel: 16-bit storage types");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int4 hdr = *reinterpret_cast<const int4*>(lut + 4 * blockIdx.x);
    const int cnt = __builtin_amdgcn_readfirstlane(hdr.y), ob = __builtin_amdgcn_readfirstlane(hdr.z);
    const int2* ent = reinterpret_cast<const int2*>(lut) + __builtin_amdgcn_readfirstlane(hdr.x);
    const int n0 = blockIdx.y * XS0_C;
    const int npairs = (cnt + 1) >> 1;

    f32x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

    unsigned char* xst = smem + wave * XS16_PART;          // [16 rows = (entry of the pair, its 8 rows)][128 B]
    unsigned char* wst = xst + 2048;                       // fprop: [16 rows = (entry, ci)][32 B], bytes 0 .. 15 of a row = the block's 8 outputs
    const int pc = lane & 7, row0 = lane >> 3;
    const int col = min(n0 + 8 * pc, N - 8);
    const int xoff = row0 * N + col;
    const int g = lane >> 4, t16 = lane & 15;
    const int xfrag = (4 * g + (t16 >> 2)) * 128 + 4 * (t16 & 3) * 2;
    const int wfrag = (4 * g + (t16 >> 2)) * 32 + 4 * (t16 & 3) * 2;

    uint4 xl0 = zero_u4(), xl1 = zero_u4(), wl4 = zero_u4();
    uint2 wl2 = make_uint2(0u, 0u);
#define XS8_LOAD(q_)                                                                                                        \
    do {                                                                                                                    \
        const int2 e0_ = ent[2 * (q_)];                                                                                     \
        const bool two_ = 2 * (q_) + 1 < cnt;                                                                               \
        const int2 e1_ = ent[two_ ? 2 * (q_) + 1 : 2 * (q_)];                                                               \
        const int c0_ = __builtin_amdgcn_readfirstlane(e0_.x), w0_ = __builtin_amdgcn_readfirstlane(e0_.y);                 \
        const int c1_ = __builtin_amdgcn_readfirstlane(e1_.x), w1_ = __builtin_amdgcn_readfirstlane(e1_.y);                 \
        xl0 = *reinterpret_cast<const uint4*>(X + (size_t)c0_ * 8 * N + xoff);                                              \
        xl1 = *reinterpret_cast<const uint4*>(X + (size_t)c1_ * 8 * N + xoff);                                              \
        if constexpr (TRANSW) {   /* lanes 0..7: row `lane` of block 0, lanes 8..15: row lane - 8 of block 1 (zeros if there is none) */ \
            wl4 = zero_u4();                                                                                                \
            if (lane < 8) wl4 = *reinterpret_cast<const uint4*>(W + (size_t)w0_ * 64 + lane * 8);                           \
            else if (lane < 16 && two_) wl4 = *reinterpret_cast<const uint4*>(W + (size_t)w1_ * 64 + (lane - 8) * 8);       \
        } else {                  /* A[m = ci][k = 8 (g >> 1) + 4 (g & 1) ..]: block g >> 1, row ci = t16, outputs 4 (g & 1) .. */       \
            wl2 = make_uint2(0u, 0u);                                                                                       \
            if (t16 < 8 && (g < 2 || two_))                                                                                 \
                wl2 = *reinterpret_cast<const uint2*>(W + (size_t)(g < 2 ? w0_ : w1_) * 64 + t16 * 8 + 4 * (g & 1));        \
        }                                                                                                                   \
    } while (0)
    if (wave < npairs) XS8_LOAD(wave);
    for (int q = wave; q < npairs; q += XS0_NW) {
        *reinterpret_cast<uint4*>(xst + lane * 16) = xl0;
        *reinterpret_cast<uint4*>(xst + 1024 + lane * 16) = xl1;
        uint2 wq;
        if constexpr (TRANSW) { if (lane < 16) *reinterpret_cast<uint4*>(wst + lane * 32) = wl4; }
        else wq = wl2;
        if (q + XS0_NW < npairs) XS8_LOAD(q + XS0_NW);
        uint2 xf[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) xf[t] = ds_tr16(xst + xfrag + t * 32);
        if constexpr (TRANSW) wq = ds_tr16(wst + wfrag);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = DT::mfma16k16(wq, xf[t], acc[t]);
    }
#undef XS8_LOAD

    const int nparts = min(npairs, XS0_NW);
    if (wave < nparts) {
        float* part = reinterpret_cast<float*>(smem + wave * XS16_PART);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) part[(t * 4 + i) * 64 + lane] = acc[t][i];
    }
    __syncthreads();
    // D[o][n]: col n = lane & 15 (+ 16 t), rows o = 4 (lane >> 4) + i: the block's 8 outputs are the rows of lanes 0 .. 31
    {
        const int el = threadIdx.x;
        const int ln = el & 63, ti = el >> 6, t = ti >> 2, i = ti & 3;
        if (ln < 32) {
            float s = 0.f;
            for (int v = 0; v < nparts; ++v) s += reinterpret_cast<const float*>(smem + v * XS16_PART)[el];
            const int o = 4 * (ln >> 4) + i, n = n0 + 16 * t + (ln & 15);
            if (n < N) Y[(size_t)(ob * 8 + o) * N + n] = DT::from_f32(s);
        }
    }
}

}  // namespace bsmm
