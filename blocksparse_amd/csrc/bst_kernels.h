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

__device__ __forceinline__ void mma32_f32(const float (&a)[16], const float (&b)[16], f32x16& acc) {
#pragma unroll
    for (int t = 0; t < 16; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b[t], acc, 0, 0, 0);
}

// ------------------------------------------------------------------------------------------------------------------
// nt: one wave per 32x32 tile of a block.  grid (ceil(blocks * SUB^2 / 4), heads, batch), 256 threads.
// ------------------------------------------------------------------------------------------------------------------
template <class TA, class TS, int BS>
__global__ void __launch_bounds__(256)
bst_nt_mfma_kernel(const typename TA::T* __restrict__ A, const typename TA::T* __restrict__ B, typename TS::T* __restrict__ S,
                   const int32_t* __restrict__ lut, int lut_stride, int blocks, int heads, int hs, int rows_q, int rows_k) {
    constexpr int SUB = BS / 32;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int tb = blockIdx.x * 4 + wave;
    if (tb >= blocks * SUB * SUB) return;
    const int b = tb / (SUB * SUB), ti = (tb / SUB) % SUB, tj = tb % SUB;
    const int h = blockIdx.y, n = blockIdx.z;
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
    // D[j][i]: i = r, j = (reg & 3) + 8 * (reg >> 2) + 4 * hh
    typename TS::T* out = S + (((size_t)n * heads + h) * blocks + b) * (BS * BS) + (size_t)(32 * ti + r) * BS + 32 * tj + 4 * hh;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const uint32_t lo = (uint32_t)TS::from_f32(acc[4 * g + 0]) | ((uint32_t)TS::from_f32(acc[4 * g + 1]) << 16);
        const uint32_t hi = (uint32_t)TS::from_f32(acc[4 * g + 2]) | ((uint32_t)TS::from_f32(acc[4 * g + 3]) << 16);
        *reinterpret_cast<uint2*>(out + 8 * g) = make_uint2(lo, hi);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// nn (TRANS = false) / tn (TRANS = true): one wave per (output block, 32-row sub tile, 32-feature tile).
// grid (ceil(ctx_c * SUB * NCT / 4), heads, batch), 256 threads.  lut = nn_lut / tn_lut (header + entries).
// ------------------------------------------------------------------------------------------------------------------
template <class TS, class TB, int BS, bool TRANS>
__global__ void __launch_bounds__(256)
bst_xn_mfma_kernel(const typename TS::T* __restrict__ S, const typename TB::T* __restrict__ Bm, typename TB::T* __restrict__ C,
                   const int32_t* __restrict__ lut, int lut_stride, int blocks, int heads, int hs, int ctx_c, int rows_b, int rows_c) {
    constexpr int SUB = BS / 32;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int nct = (hs + 31) / 32;
    const int wid = blockIdx.x * 4 + wave;
    if (wid >= ctx_c * SUB * nct) return;
    const int ct = wid % nct, ts = (wid / nct) % SUB, oc = wid / (nct * SUB);
    const int h = blockIdx.y, n = blockIdx.z;
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
    for (int e = 0; e < hdr.y; ++e) {
        const int2 ent = *reinterpret_cast<const int2*>(hl + 2 * (hdr.x + e));     // (block id, other-side block)
        const typename TS::T* sb = sbase + (size_t)ent.x * (BS * BS);
#pragma unroll
        for (int tk = 0; tk < SUB; ++tk) {
            float fa[16], fb[16];
            if constexpr (!TRANS) {       // A[i][j] = S_b[32ts + i][32tk + j], j = 16hh + t: contiguous
                load16_f32<TS>(sb + (size_t)(32 * ts + r) * BS + 32 * tk + 16 * hh, 0, 16, fa);
            } else {                      // A[j][i] = S_b[32tk + i][32ts + j], i = 16hh + t: stride BS (lanes walk j)
#pragma unroll
                for (int t = 0; t < 16; ++t) fa[t] = TS::to_f32(sb[(size_t)(32 * tk + 16 * hh + t) * BS + 32 * ts + r]);
            }
            const typename TB::T* bp = bcol + ((size_t)ent.y * BS + 32 * tk + 16 * hh) * state;
#pragma unroll
            for (int t = 0; t < 16; ++t) fb[t] = cvalid ? TB::to_f32(bp[(size_t)t * state]) : 0.f;
            mma32_f32(fa, fb, acc);
        }
    }
    if (!cvalid) return;
    typename TB::T* out = C + ((size_t)n * rows_c + (size_t)oc * BS + 32 * ts) * state + (size_t)h * hs + c;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int row = (reg & 3) + 8 * (reg >> 2) + 4 * hh;
        out[(size_t)row * state] = TB::from_f32(acc[reg]);
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
