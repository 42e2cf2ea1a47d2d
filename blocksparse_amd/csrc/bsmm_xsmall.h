// bsmm_xsmall.h -- xprop for SMALL minibatches (round 4): feature_axis = 1, bsize 32, 16-bit storage types, no plan needed.
//
// Below a few hundred minibatch rows the plan kernels have too few (row tile, group) units for 256 CUs (N = 128: 8 units), and the
// per-segment kernel walks a whole output column (25 blocks at the bench layout) in ONE serial chain per wave: 14-16 us whatever
// N is.  The reference answers this regime by cutting long columns into segments that meet under locks (blocksparse/matmul.py:94-105,
// 203-231) so that segments x minibatch tiles cover the machine.  Here the cut is inside the workgroup, where meeting is cheap:
//   workgroup = one output block x 64 minibatch rows, XSM_NW = 8 waves; wave v multiplies the column's entries v, v + 8, ... for
//   all 64 rows (fragments straight from global memory: the activations are L2-resident at these sizes), the eight partial
//   32 x 64 tiles meet in LDS (fp32, one barrier) and are rounded ONCE -- same sums as the other kernels up to fp32 summation order.
// fprop (TRANSW): the weight block goes through a wave-private 2 KiB of LDS and comes back transposed (ds_read_b64_tr_b16): no
// transposed copy of W, no pre-pass, no workspace.
#pragma once
#include "bsmm_common.h"
#include "bsmm_updat_tr.h"   // ds_tr16

namespace bsmm {

constexpr int XSM_NW = 8;                                  // waves per workgroup = ways a column's entry list is cut
constexpr int XSM_R = 64;                                  // minibatch rows per workgroup
constexpr int XSM_PART = XSM_R * 128;                       // one wave's partial tile: [64 rows][32 outputs] fp32
constexpr int XSM_LDS = XSM_NW * XSM_PART + XSM_NW * 2048;    // partial tiles + the weight staging of fprop

template <class DT, bool TRANSW>
__global__ void __launch_bounds__(64 * XSM_NW)
xsmall32_kernel(const typename DT::T* __restrict__ X, const typename DT::T* __restrict__ W, typename DT::T* __restrict__ Y,
                const int32_t* __restrict__ lut, int N, int Cin, int Kout) {
    typedef typename DT::T T;
    static_assert(DT::is16, "small-minibatch kernel: 16-bit storage types");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int4 hdr = *reinterpret_cast<const int4*>(lut + 4 * blockIdx.x);
    const int cnt = __builtin_amdgcn_readfirstlane(hdr.y), ob = __builtin_amdgcn_readfirstlane(hdr.z);
    const int2* ent = reinterpret_cast<const int2*>(lut) + __builtin_amdgcn_readfirstlane(hdr.x);
    const int n0 = blockIdx.y * XSM_R;

    f32x16 acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    // this lane's rows of the activation tile (rows past N are clamped re-reads, never stored)
    const T* xrow[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) xrow[t] = X + (size_t)min(n0 + 32 * t + r, N - 1) * Cin + 8 * h;
    unsigned char* wst = smem + XSM_NW * XSM_PART + wave * 2048;          // fprop: my weight staging

    for (int e = wave; e < cnt; e += XSM_NW) {
        const int2 cw = ent[e];
        const int c = __builtin_amdgcn_readfirstlane(cw.x), w = __builtin_amdgcn_readfirstlane(cw.y);
        const T* wb = W + (size_t)w * 1024;
        uint4 wq[2], xf[2][2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) xf[t][kk] = *reinterpret_cast<const uint4*>(xrow[t] + c * 32 + 16 * kk);
        if constexpr (TRANSW) {
            // W[w][i][o] natural -> LDS as is (32 B per lane), read back with the transposing read: lane (o = r-ish, K half) gets 8 consecutive i
            const uint4 a = *reinterpret_cast<const uint4*>(wb + lane * 16), b = *reinterpret_cast<const uint4*>(wb + lane * 16 + 8);
            *reinterpret_cast<uint4*>(wst + lane * 32) = a;
            *reinterpret_cast<uint4*>(wst + lane * 32 + 16) = b;
            const int g16 = lane >> 4, t16 = lane & 15;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const unsigned char* p = wst + (16 * kk + 8 * h + (t16 >> 2)) * 64 + (16 * (g16 & 1) + 4 * (t16 & 3)) * 2;
                const uint2 lo = ds_tr16(p), hi = ds_tr16(p + 4 * 64);
                wq[kk] = make_uint4(lo.x, lo.y, hi.x, hi.y);
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) wq[kk] = *reinterpret_cast<const uint4*>(wb + r * 32 + 16 * kk + 8 * h);
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int t = 0; t < 2; ++t) acc[t] = DT::mfma32(wq[kk], xf[t][kk], acc[t]);
    }

    // the partial tiles meet in LDS: D[o][n] with col n = r, rows o = (reg & 3) + 8 (reg >> 2) + 4h -> part[wave][n][o], the eight float4
    // pieces of row n XOR-swizzled with n & 7
    const int nparts = min(cnt, XSM_NW);
    if (wave < nparts) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int n = 32 * t + r;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = make_float4(acc[t][4 * q + 0], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]);
                *reinterpret_cast<float4*>(smem + wave * XSM_PART + n * 128 + ((((2 * q + h)) ^ (n & 7)) << 4)) = v;
            }
        }
    }
    __syncthreads();
    {
        const int n = threadIdx.x >> 3, p = threadIdx.x & 7;             // 64 rows x 8 float4 pieces = 512 threads
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int v = 0; v < nparts; ++v) {
            const float4 a = *reinterpret_cast<const float4*>(smem + v * XSM_PART + n * 128 + ((p ^ (n & 7)) << 4));
            s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
        }
        if (n0 + n < N) {
            const uint32_t lo = (uint32_t)DT::from_f32(s.x) | ((uint32_t)DT::from_f32(s.y) << 16);
            const uint32_t hi = (uint32_t)DT::from_f32(s.z) | ((uint32_t)DT::from_f32(s.w) << 16);
            *reinterpret_cast<uint2*>(Y + (size_t)(n0 + n) * Kout + ob * 32 + 4 * p) = make_uint2(lo, hi);
        }
    }
}

}  // namespace bsmm

namespace bsmm {

// ---- bsize 16 / 8 on feature axis 1 (round 6) --------------------------------------------------------------------------------------------------------
// The reference runs 8 / 16-wide blocks on feature axis 0 only (blocksparse/matmul.py:84-89); here they run on both axes, and short minibatches on
// axis 1 used to fall to the per-segment kernel (bsize 16) or V_FMA (bsize 8).  Same decomposition as above: workgroup = one output block x 64
// minibatch rows, XSM_NW waves over the column's entries, partial tiles meet in LDS.  v_mfma_f32_16x16x16: A[m = row n][k] = 8 contiguous bytes of
// an activation row straight from global memory (bsize 8: K = 16 = two entries, lane group g takes entry g >> 1); B[k][col = output]: bprop 8
// contiguous bytes of a weight row from global memory, fprop the block(s) through 512 bytes of wave-private LDS and one transposing read.
template <class DT, int BS, bool TRANSW>
__global__ void __launch_bounds__(64 * XSM_NW)
xsmall_narrow_kernel(const typename DT::T* __restrict__ X, const typename DT::T* __restrict__ W, typename DT::T* __restrict__ Y,
                     const int32_t* __restrict__ lut, int N, int Cin, int Kout) {
    typedef typename DT::T T;
    static_assert(DT::is16 && (BS == 16 || BS == 8), "narrow small-minibatch kernel: 16-bit storage types, bsize 16 / 8");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, t16 = lane & 15;
    const int4 hdr = *reinterpret_cast<const int4*>(lut + 4 * blockIdx.x);
    const int cnt = __builtin_amdgcn_readfirstlane(hdr.y), ob = __builtin_amdgcn_readfirstlane(hdr.z);
    const int2* ent = reinterpret_cast<const int2*>(lut) + __builtin_amdgcn_readfirstlane(hdr.x);
    const int n0 = blockIdx.y * XSM_R;
    constexpr int EPS = BS == 16 ? 1 : 2;                  // entries per step (K = 16)
    const int nsteps = (cnt + EPS - 1) / EPS;

    f32x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    // my rows of the four 16-row tiles (rows past N are clamped re-reads, never stored) and my K offset inside an entry's features
    const T* xrow[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) xrow[t] = X + (size_t)min(n0 + 16 * t + t16, N - 1) * Cin + (BS == 16 ? 4 * g : 4 * (g & 1));
    unsigned char* wst = smem + wave * XSM_PART;           // fprop: my weight staging [16 rows (k)][32 B], inside my partial tile
    const int wfrag = (4 * g + (t16 >> 2)) * 32 + 4 * (t16 & 3) * 2;

    for (int s = wave; s < nsteps; s += XSM_NW) {
        int c, w, c1 = 0, w1 = 0;
        bool two = false;
        if constexpr (BS == 16) {
            const int2 cw = ent[s];
            c = __builtin_amdgcn_readfirstlane(cw.x); w = __builtin_amdgcn_readfirstlane(cw.y);
        } else {
            const int2 e0 = ent[2 * s];
            two = 2 * s + 1 < cnt;
            const int2 e1 = ent[two ? 2 * s + 1 : 2 * s];
            c = __builtin_amdgcn_readfirstlane(e0.x); w = __builtin_amdgcn_readfirstlane(e0.y);
            c1 = __builtin_amdgcn_readfirstlane(e1.x); w1 = __builtin_amdgcn_readfirstlane(e1.y);
        }
        const int cme = BS == 16 ? c : (g < 2 ? c : c1);    // the entry my K group belongs to (a missing second entry: the first one's features, times zero)
        uint2 xa[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) xa[t] = *reinterpret_cast<const uint2*>(xrow[t] + cme * BS);
        uint2 wq = make_uint2(0u, 0u);
        if constexpr (TRANSW) {
            // B[k = ci][col = ko] = W[ci][ko]: the block(s) as they lie -> LDS [16 rows][32 B] -> transposing read
            if constexpr (BS == 16) {
                *reinterpret_cast<uint2*>(wst + lane * 8) = *reinterpret_cast<const uint2*>(W + (size_t)w * 256 + lane * 4);
            } else {
                if (lane < 16) {
                    uint4 v = zero_u4();
                    if (lane < 8) v = *reinterpret_cast<const uint4*>(W + (size_t)w * 64 + lane * 8);
                    else if (two) v = *reinterpret_cast<const uint4*>(W + (size_t)w1 * 64 + (lane - 8) * 8);
                    *reinterpret_cast<uint4*>(wst + lane * 32) = v;
                }
            }
            wq = ds_tr16(wst + wfrag);
        } else {
            // B[k = ko][col = ci] = W[ci][ko]: lane (ci = t16, g) takes 4 consecutive outputs of row ci
            if constexpr (BS == 16) wq = *reinterpret_cast<const uint2*>(W + (size_t)w * 256 + t16 * 16 + 4 * g);
            else if (t16 < 8 && (g < 2 || two)) wq = *reinterpret_cast<const uint2*>(W + (size_t)(g < 2 ? w : w1) * 64 + t16 * 8 + 4 * (g & 1));
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = DT::mfma16k16(xa[t], wq, acc[t]);
    }

    // D[m = row][col]: col = t16, rows 4 g + i of tile t -> part[wave][t][i][lane]
    const int nparts = min(nsteps, XSM_NW);
    if (wave < nparts) {
        float* part = reinterpret_cast<float*>(smem + wave * XSM_PART);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) part[(t * 4 + i) * 64 + lane] = acc[t][i];
    }
    __syncthreads();
    for (int el = threadIdx.x; el < 1024; el += 64 * XSM_NW) {
        const int ln = el & 63, ti = el >> 6, t = ti >> 2, i = ti & 3;
        const int col = ln & 15, n = n0 + 16 * t + 4 * (ln >> 4) + i;
        if (col >= BS || n >= N) continue;
        float sum = 0.f;
        for (int v = 0; v < nparts; ++v) sum += reinterpret_cast<const float*>(smem + v * XSM_PART)[el];
        Y[(size_t)n * Kout + ob * BS + col] = DT::from_f32(sum);
    }
}

}  // namespace bsmm
