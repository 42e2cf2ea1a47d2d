// bsmm_updat.h -- weight-gradient kernels:  DW[w] = alpha * sum_p sum_n X_p[c,:,n] (x) DY_p[k,:,n] + beta * DW[w]
// with (c,k) = updat_lut[w].  Replaces gemm_blocksparse_{32,16,08}x64x*_updat
// (src/blocksparse_matmul_op_gpu.cu:960-1835,2424-2892) and hgemm_blocksparse_*_{nt_dds,tn_dds}.
//
// One workgroup (4 waves) per nonzero block; the minibatch reduction is split over the waves (wave v
// takes 32-wide n chunks v, v+4, ...), each wave accumulates a full bs x bs tile on the matrix cores
// (K dimension of the MFMA = n), then the 4 partial tiles are summed through LDS.
//   MFMA roles: A[ci][n] (M = c-in-block), B[n][ko] (N = k-in-block)  ->  D[ci][ko]
//   axis 0: both operands are K(=n)-contiguous in memory -> 16-byte loads
//   axis 1: both operands are K-strided (stride = feature count) -> element gathers (v1)
#pragma once
#include "bsmm_common.h"

namespace bsmm {

// block b -> weight block w so that each XCD (b % 8) walks a contiguous z-order range of blocks
// (neighbouring blocks share X rows / DY rows -> L2 hits).  Bijective for any `blocks`.
__device__ __forceinline__ int updat_block(int b, int blocks) {
    const int q8 = blocks >> 3, r8 = blocks & 7;
    const int xcd = b & 7, idx = b >> 3;
    const int len = q8 + (xcd < r8 ? 1 : 0);
    if (idx >= len) return -1;
    const int start = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    return start + idx;
}

template <class DT, int AXIS>
__global__ void __launch_bounds__(256)
updat32_kernel(PtrList8 Xs, PtrList8 Es, typename DT::T* __restrict__ DW, const int32_t* __restrict__ lut,
               int blocks, int N, int Cf, int Kf, int pcount, float alpha, float beta, const float* __restrict__ gate = nullptr,
               const int32_t* __restrict__ only_if = nullptr) {
    if (only_if && only_if[0] == 0) return;          // repair pass of the fp32 bf16-split paths (bsmm_api.hip): runs only when their flag is set
    typedef typename DT::T T;
    __shared__ float red[4 * 1024];
    const int w = updat_block(blockIdx.x, blocks);
    if (w < 0) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int c = lut[2 * w], k = lut[2 * w + 1];
    if (gate) alpha *= gate[w];                       // gated dw (updat_test(dw_gated=True), blocksparse/matmul.py:412-418)

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;

    for (int p = 0; p < pcount; ++p) {
        const T* X = static_cast<const T*>(Xs.p[p]);
        const T* E = static_cast<const T*>(Es.p[p]);
        for (int n0 = wave * 32; n0 < N; n0 += 128) {
            const int klim = N - n0;
            Frag32<DT> a, b;
            if constexpr (AXIS == 0) {
                a.load_contig_lim(X + (size_t)(c * 32 + r) * N + n0, h, klim);
                b.load_contig_lim(E + (size_t)(k * 32 + r) * N + n0, h, klim);
            } else {
                a.load_strided_lim(X + (size_t)n0 * Cf + c * 32 + r, (size_t)Cf, h, klim);
                b.load_strided_lim(E + (size_t)n0 * Kf + k * 32 + r, (size_t)Kf, h, klim);
            }
            mma32<DT>(a, b, acc);
        }
    }

#pragma unroll
    for (int reg = 0; reg < 16; ++reg) red[wave * 1024 + reg * 64 + lane] = acc[reg];
    __syncthreads();
    for (int slot = threadIdx.x; slot < 1024; slot += 256) {
        const float sum = red[slot] + red[1024 + slot] + red[2048 + slot] + red[3072 + slot];
        const int reg = slot >> 6, ln = slot & 63;
        const int ci = (reg & 3) + 8 * (reg >> 2) + 4 * (ln >> 5), ko = ln & 31;
        const size_t idx = (size_t)w * 1024 + ci * 32 + ko;
        float out = alpha * sum;
        if (beta != 0.f) out += beta * DT::to_f32(DW[idx]);
        DW[idx] = DT::from_f32(out);
    }
}

template <class DT, int AXIS>
__global__ void __launch_bounds__(256)
updat16_kernel(PtrList8 Xs, PtrList8 Es, typename DT::T* __restrict__ DW, const int32_t* __restrict__ lut,
               int blocks, int N, int Cf, int Kf, int pcount, float alpha, float beta, const float* __restrict__ gate = nullptr,
               const int32_t* __restrict__ only_if = nullptr) {
    if (only_if && only_if[0] == 0) return;          // repair pass of the fp32 bf16-split paths (bsmm_api.hip): runs only when their flag is set
    typedef typename DT::T T;
    constexpr int KS = Frag16<DT>::KS;   // n per MFMA slab: 32 (16-bit) / 16 (f32)
    constexpr int KL = Frag16<DT>::KL;   // n per lane per slab: 8 / 4
    __shared__ float red[4 * 256];
    const int w = updat_block(blockIdx.x, blocks);
    if (w < 0) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int c = lut[2 * w], k = lut[2 * w + 1];
    if (gate) alpha *= gate[w];                       // gated dw (updat_test(dw_gated=True), blocksparse/matmul.py:412-418)

    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int p = 0; p < pcount; ++p) {
        const T* X = static_cast<const T*>(Xs.p[p]);
        const T* E = static_cast<const T*>(Es.p[p]);
        for (int n0 = wave * KS; n0 < N; n0 += 4 * KS) {
            const int nl = n0 + KL * q;          // this lane's first n
            const int lim = N - nl;
            Frag16<DT> a, b;
            if constexpr (AXIS == 0) {
                a.load_contig_lim(X + (size_t)(c * 16 + r) * N + nl, lim);
                b.load_contig_lim(E + (size_t)(k * 16 + r) * N + nl, lim);
            } else {
                if (lim > 0) {
                    a.load_strided_lim(X + (size_t)nl * Cf + c * 16 + r, (size_t)Cf, lim);
                    b.load_strided_lim(E + (size_t)nl * Kf + k * 16 + r, (size_t)Kf, lim);
                } else {
                    a.zero();
                    b.zero();
                }
            }
            mma16<DT>(a, b, acc);
        }
    }

#pragma unroll
    for (int reg = 0; reg < 4; ++reg) red[wave * 256 + reg * 64 + lane] = acc[reg];
    __syncthreads();
    {
        const int slot = threadIdx.x;
        const float sum = red[slot] + red[256 + slot] + red[512 + slot] + red[768 + slot];
        const int reg = slot >> 6, ln = slot & 63;
        const int ci = 4 * (ln >> 4) + reg, ko = ln & 15;
        const size_t idx = (size_t)w * 256 + ci * 16 + ko;
        float out = alpha * sum;
        if (beta != 0.f) out += beta * DT::to_f32(DW[idx]);
        DW[idx] = DT::from_f32(out);
    }
}

// VALU kernel, any bsize.  A chunk of CH minibatch columns of the X rows of block-row c and the DY rows
// of block-row k is staged in LDS as fp32 ([feature][n], +1 pad); thread t owns output(s)
// (ci,ko) and, when bs*bs < 256, one of 256/(bs*bs) slices of the chunk; slices are summed via LDS.
template <class DT, int BS, int AXIS>
__global__ void __launch_bounds__(256)
updat_valu_kernel(PtrList8 Xs, PtrList8 Es, typename DT::T* __restrict__ DW, const int32_t* __restrict__ lut,
                  int blocks, int N, int Cf, int Kf, int pcount, float alpha, float beta, const float* __restrict__ gate = nullptr,
               const int32_t* __restrict__ only_if = nullptr) {
    if (only_if && only_if[0] == 0) return;          // repair pass of the fp32 bf16-split paths (bsmm_api.hip): runs only when their flag is set
    typedef typename DT::T T;
    constexpr int CH = 64;
    constexpr int OUTS = BS * BS;
    constexpr int SL = (OUTS >= 256) ? 1 : 256 / OUTS;     // n-slices per chunk
    constexpr int PER = (OUTS >= 256) ? OUTS / 256 : 1;    // outputs per thread
    constexpr int SLN = CH / SL;                            // n per slice
    __shared__ float xs[BS][CH + 1];
    __shared__ float es[BS][CH + 1];
    __shared__ float red[256];

    const int w = blockIdx.x;
    const int c = lut[2 * w], k = lut[2 * w + 1];
    if (gate) alpha *= gate[w];                       // gated dw (updat_test(dw_gated=True), blocksparse/matmul.py:412-418)
    const int tid = threadIdx.x;
    const int slice = (SL == 1) ? 0 : tid / OUTS;
    const int obase = (SL == 1) ? tid : tid % OUTS;

    float acc[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) acc[j] = 0.f;

    for (int p = 0; p < pcount; ++p) {
        const T* X = static_cast<const T*>(Xs.p[p]);
        const T* E = static_cast<const T*>(Es.p[p]);
        for (int n0 = 0; n0 < N; n0 += CH) {
            __syncthreads();
            for (int idx = tid; idx < BS * CH; idx += 256) {
                int f, nn;
                if (AXIS == 0) { f = idx / CH; nn = idx % CH; }     // n fastest (contiguous in memory)
                else           { nn = idx / BS; f = idx % BS; }     // feature fastest
                const int n = n0 + nn;
                float xv = 0.f, ev = 0.f;
                if (n < N) {
                    if (AXIS == 0) {
                        xv = DT::to_f32(X[(size_t)(c * BS + f) * N + n]);
                        ev = DT::to_f32(E[(size_t)(k * BS + f) * N + n]);
                    } else {
                        xv = DT::to_f32(X[(size_t)n * Cf + c * BS + f]);
                        ev = DT::to_f32(E[(size_t)n * Kf + k * BS + f]);
                    }
                }
                xs[f][nn] = xv;
                es[f][nn] = ev;
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                const int o = obase + j * 256;
                const int ci = o / BS, ko = o % BS;
                float a = acc[j];
#pragma unroll 8
                for (int nn = 0; nn < SLN; ++nn) a = fmaf(xs[ci][slice * SLN + nn], es[ko][slice * SLN + nn], a);
                acc[j] = a;
            }
        }
    }

    if (SL > 1) {
        __syncthreads();
        red[tid] = acc[0];
        __syncthreads();
        if (tid < OUTS) {
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < SL; ++j) s += red[j * OUTS + tid];
            const size_t idx = (size_t)w * OUTS + tid;
            float out = alpha * s;
            if (beta != 0.f) out += beta * DT::to_f32(DW[idx]);
            DW[idx] = DT::from_f32(out);
        }
    } else {
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const size_t idx = (size_t)w * OUTS + obase + j * 256;
            float out = alpha * acc[j];
            if (beta != 0.f) out += beta * DT::to_f32(DW[idx]);
            DW[idx] = DT::from_f32(out);
        }
    }
}

template <class DT>
__global__ void __launch_bounds__(256)
identity_init_kernel(typename DT::T* __restrict__ W, const int32_t* __restrict__ lut, int CB, int KB, int bsize, float scale) {
    const int w = blockIdx.x;
    const int c = lut[2 * w], k = lut[2 * w + 1];
    const bool diag = (c % KB) == (k % CB);
    const int n = bsize * bsize;
    for (int idx = threadIdx.x; idx < n; idx += 256) {
        const int i = idx / bsize, j = idx % bsize;
        W[(size_t)w * n + idx] = DT::from_f32((diag && i == j) ? scale : 0.f);
    }
}

// dw_out = dw * gate, dg[w] = sum(dw[w] * W[w])   (blocksparse_gate_grad, src/blocksparse_hgemm_cn_64_op_gpu.cu:1339-1392).
// One workgroup per block; dw_out may alias dw.
template <class DT>
__global__ void __launch_bounds__(256)
gate_grad_kernel(typename DT::T* __restrict__ dw_out, float* __restrict__ dg, const typename DT::T* __restrict__ dw,
                 const typename DT::T* __restrict__ W, const float* __restrict__ gate, int bsize) {
    __shared__ float red[4];
    const int w = blockIdx.x, n = bsize * bsize;
    const float g = gate[w];
    float s = 0.f;
    for (int idx = threadIdx.x; idx < n; idx += 256) {
        const float d = DT::to_f32(dw[(size_t)w * n + idx]);
        s = fmaf(d, DT::to_f32(W[(size_t)w * n + idx]), s);
        dw_out[(size_t)w * n + idx] = DT::from_f32(d * g);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) dg[w] = red[0] + red[1] + red[2] + red[3];
}

// Gated weight images for the fast ungated xprop kernels (round 6): out[0][w] = round(gate[w] * W[w]) and, with PIECES == 2,
// out[1][w] = round(gate[w] * W[w] - out[0][w]) -- hi + lo carry g * w to ~2^-17 (the same two pieces the staged GATED kernels form per
// fragment, bsmm_xcol_v2.h, here once per call instead of once per row tile).  gate 0 = the block contributes nothing whatever it holds
// (Inf / NaN included: the reference skips it, src/blocksparse_hgemm_cn_64_op_gpu.cu:96-100); gates 0 / 1 leave one exact image.
// One thread per 8 elements (16 bytes); 16-bit storage types.
template <class DT, int PIECES>
__global__ void __launch_bounds__(256)
gate_weights_kernel(const uint4* __restrict__ W, const float* __restrict__ gate, uint4* __restrict__ out, int per_block8, size_t total8) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total8) return;
    const float g = gate[i / (size_t)per_block8];
    const uint4 v = W[i];
    const uint32_t src[4] = {v.x, v.y, v.z, v.w};
    uint32_t hi[4], lo[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float p0 = g * DT::to_f32((uint16_t)(src[e] & 0xffffu)), p1 = g * DT::to_f32((uint16_t)(src[e] >> 16));
        const uint16_t h0 = g == 0.f ? (uint16_t)0 : DT::from_f32(p0), h1 = g == 0.f ? (uint16_t)0 : DT::from_f32(p1);
        hi[e] = (uint32_t)h0 | ((uint32_t)h1 << 16);
        if constexpr (PIECES == 2) {
            const uint16_t l0 = g == 0.f ? (uint16_t)0 : DT::from_f32(p0 - DT::to_f32(h0)), l1 = g == 0.f ? (uint16_t)0 : DT::from_f32(p1 - DT::to_f32(h1));
            lo[e] = (uint32_t)l0 | ((uint32_t)l1 << 16);
        }
    }
    out[i] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
    if constexpr (PIECES == 2) out[total8 + i] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
}

// bsize 8, feature axis 0, 16-bit types, short minibatches (round 6: the reference benchmark's (8, 0) shapes at N = 64 -- 41 - 225 us through the
// super-block path, 70 - 80 through the V_FMA kernel): one WAVE per PAIR of weight blocks on v_mfma_f32_16x16x16.  On this axis an operand fragment
// is contiguous in memory: lane (row t16, K group g) loads 16 bytes = 8 minibatch columns 32 j + 8 g .. of its row -- rows 0 .. 7 of the A side are
// the X rows of block 0, rows 8 .. 15 those of block 1, the B side the DY rows likewise -- and feeds the low / high 8 bytes to two MFMAs (A and B label
// the contraction index alike, which is all the hardware asks).  The diagonal 8 x 8 quadrants of the 16 x 16 result are the two blocks; the
// off-diagonal ones (block 0's X rows against block 1's DY rows) are discarded.  Needs N % 8 == 0.
template <class DT>
__global__ void __launch_bounds__(256)
updat8_a0_pairs_kernel(PtrList8 Xs, PtrList8 Es, typename DT::T* __restrict__ DW, const int32_t* __restrict__ lut,
                       int blocks, int N, int pcount, float alpha, float beta, const float* __restrict__ gate = nullptr) {
    typedef typename DT::T T;
    static_assert(DT::is16, "bsize-8 pair kernel: 16-bit storage types");
    const int lane = threadIdx.x & 63;
    const int pair = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int g = lane >> 4, t16 = lane & 15;
    const int w = 2 * pair + (t16 >> 3);                  // my block: rows 0 .. 7 -> block 0 of the pair, 8 .. 15 -> block 1
    const bool have = w < blocks;
    const int c = have ? lut[2 * w] : 0, k = have ? lut[2 * w + 1] : 0;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const size_t xrow = (size_t)(c * 8 + (t16 & 7)) * N, erow = (size_t)(k * 8 + (t16 & 7)) * N;
    for (int p = 0; p < pcount; ++p) {
        const T* X = static_cast<const T*>(Xs.p[p]) + xrow;
        const T* E = static_cast<const T*>(Es.p[p]) + erow;
        for (int nb = 0; nb < N; nb += 32) {                             // (uniform trip count: the matrix instruction wants every lane there)
            const int n = nb + 8 * g;
            const bool in = have && n < N;                               // a lane past the end of a ragged last chunk multiplies zeros
            const uint4 a = in ? *reinterpret_cast<const uint4*>(X + n) : zero_u4();
            const uint4 b = in ? *reinterpret_cast<const uint4*>(E + n) : zero_u4();
            acc = DT::mfma16k16(make_uint2(a.x, a.y), make_uint2(b.x, b.y), acc);
            acc = DT::mfma16k16(make_uint2(a.z, a.w), make_uint2(b.z, b.w), acc);
        }
    }
    // D[m][col]: col = t16, rows m = 4 g + i.  Block 0 = (m < 8, col < 8), block 1 = (m >= 8, col >= 8): lane (g, t16) holds a diagonal quadrant iff
    // (g >> 1) == (t16 >> 3); its block is the one it loaded for (w).
    if (have && (g >> 1) == (t16 >> 3)) {
        const float a = gate ? alpha * gate[w] : alpha;
        T* out = DW + (size_t)w * 64 + (t16 & 7);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ci = 4 * (g & 1) + i;
            float v = a * acc[i];
            if (beta != 0.f) v += beta * DT::to_f32(out[ci * 8]);
            out[ci * 8] = DT::from_f32(v);
        }
    }
}

// bsize 32 / 16, feature axis 0, 16-bit types, a few dozen minibatch columns (round 6: the reference benchmark's N = 64): ONE WAVE per weight block,
// operand fragments straight from global memory (on this axis lane (row, K group) needs 8 consecutive minibatch columns of its row: 16 contiguous
// bytes), no LDS, no reduction across waves -- the per-block kernels above cut the minibatch over four waves that meet in LDS, which at N = 64 is
// most of their 12.5 / 17 us.  Needs N % 8 == 0.
template <class DT, int BS>
__global__ void __launch_bounds__(256)
updat_a0_wave_kernel(PtrList8 Xs, PtrList8 Es, typename DT::T* __restrict__ DW, const int32_t* __restrict__ lut,
                     int blocks, int N, int pcount, float alpha, float beta, const float* __restrict__ gate = nullptr) {
    typedef typename DT::T T;
    static_assert(DT::is16 && (BS == 32 || BS == 16), "one-wave weight-gradient kernel: 16-bit storage types, bsize 32 / 16");
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= blocks) return;
    const int c = lut[2 * w], k = lut[2 * w + 1];
    const float a_eff = gate ? alpha * gate[w] : alpha;
    if constexpr (BS == 32) {
        const int r = lane & 31, h = lane >> 5;                  // A[m = ci][k = n]: lane (ci = r, K half h) holds columns 16 s + 8 h ..
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        for (int p = 0; p < pcount; ++p) {
            const T* X = static_cast<const T*>(Xs.p[p]) + (size_t)(c * 32 + r) * N;
            const T* E = static_cast<const T*>(Es.p[p]) + (size_t)(k * 32 + r) * N;
            for (int nb = 0; nb < N; nb += 16) {                 // (uniform trip count; a lane past the end multiplies zeros)
                const int n = nb + 8 * h;
                const uint4 av = n < N ? *reinterpret_cast<const uint4*>(X + n) : zero_u4();
                const uint4 bv = n < N ? *reinterpret_cast<const uint4*>(E + n) : zero_u4();
                acc = DT::mfma32(av, bv, acc);
            }
        }
        // D[ci][ko]: col = ko = lane & 31, row ci = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int ci = (reg & 3) + 8 * (reg >> 2) + 4 * h;
            const size_t idx = (size_t)w * 1024 + ci * 32 + r;
            float out = a_eff * acc[reg];
            if (beta != 0.f) out += beta * DT::to_f32(DW[idx]);
            DW[idx] = DT::from_f32(out);
        }
    } else {
        const int t16 = lane & 15, g = lane >> 4;                // v_mfma_f32_16x16x32: lane (row t16, K group g) holds columns 32 s + 8 g ..
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int p = 0; p < pcount; ++p) {
            const T* X = static_cast<const T*>(Xs.p[p]) + (size_t)(c * 16 + t16) * N;
            const T* E = static_cast<const T*>(Es.p[p]) + (size_t)(k * 16 + t16) * N;
            for (int nb = 0; nb < N; nb += 32) {
                const int n = nb + 8 * g;
                const uint4 av = n < N ? *reinterpret_cast<const uint4*>(X + n) : zero_u4();
                const uint4 bv = n < N ? *reinterpret_cast<const uint4*>(E + n) : zero_u4();
                acc = DT::mfma16(av, bv, acc);
            }
        }
        // D[ci][ko]: col = ko = t16, rows ci = 4 g + i
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const size_t idx = (size_t)w * 256 + (4 * g + i) * 16 + t16;
            float out = a_eff * acc[i];
            if (beta != 0.f) out += beta * DT::to_f32(DW[idx]);
            DW[idx] = DT::from_f32(out);
        }
    }
}

}  // namespace bsmm
