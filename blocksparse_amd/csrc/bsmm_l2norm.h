// bsmm_l2norm.h -- block-sparse L2 weight normalisation over the output features of W.
//   forward   S[k] = sum over the blocks of column block K and rows i of W[w][i][j]^2       (k = K * bsize + j)
//             Y    = gain[k] * W / sqrt(max(S[k], eps))
//   backward  dW   = ( dY * gain + W * (S >= eps) * sum(-dY * gain * W / max(S, eps)) ) / sqrt(max(S, eps))
//             dG[k] = sum( dY * W / sqrt(max(S, eps)) )
// Replaces l2_normalize_CK_{32,16,8} / l2_normalize_grad_CK_* (src/blocksparse_l2_norm_op_gpu.cu:151-376,592-891; the
// formulas are the comment block at :705-708).  l2_lut: headers (offset, size, K, 0) per column block, then weight ids
// (blocksparse/matmul.py:254-268).  One workgroup per column block: thread t owns feature j = t % bsize for rows
// t / bsize, t / bsize + 256 / bsize, ...; partial sums meet in LDS.  W is small (13 MB at the headline size): this is a
// few-microsecond memory-bound pre-step, nothing to tile.
#pragma once
#include "bsmm_common.h"

namespace bsmm {

template <class TX, class TY, int BS>
__global__ void __launch_bounds__(256)
l2_normalize_kernel(typename TY::T* __restrict__ Y, float* __restrict__ S, const typename TX::T* __restrict__ X,
                    const float* __restrict__ G, const int32_t* __restrict__ lut, float eps) {
    constexpr int RG = 256 / BS;                       // row groups
    __shared__ float red[256];
    const int4 hdr = *reinterpret_cast<const int4*>(lut + 4 * blockIdx.x);
    const int32_t* ent = lut + hdr.x;
    const int j = threadIdx.x % BS, rg = threadIdx.x / BS, k = hdr.z * BS + j;
    float s = 0.f;
    for (int e = 0; e < hdr.y; ++e) {
        const typename TX::T* xb = X + (size_t)ent[e] * (BS * BS);
        for (int i = rg; i < BS; i += RG) {
            const float x = TX::to_f32(xb[i * BS + j]);
            s = fmaf(x, x, s);
        }
    }
    red[threadIdx.x] = s;
    __syncthreads();
    s = 0.f;
#pragma unroll
    for (int g = 0; g < RG; ++g) s += red[g * BS + j];
    if (rg == 0) S[k] = s;                             // also for empty column blocks (0)
    const float rn = rsqrtf(fmaxf(s, eps)) * (G ? G[k] : 1.f);
    for (int e = 0; e < hdr.y; ++e) {
        const size_t o = (size_t)ent[e] * (BS * BS);
        for (int i = rg; i < BS; i += RG) Y[o + i * BS + j] = TY::from_f32(TX::to_f32(X[o + i * BS + j]) * rn);
    }
}

template <class TX, class TY, int BS>
__global__ void __launch_bounds__(256)
l2_normalize_grad_kernel(typename TX::T* __restrict__ DX, float* __restrict__ DG, const typename TY::T* __restrict__ DY,
                         const typename TX::T* __restrict__ X, const float* __restrict__ G, const float* __restrict__ S,
                         const int32_t* __restrict__ lut, float eps) {
    constexpr int RG = 256 / BS;
    __shared__ float red1[256], red2[256];
    const int4 hdr = *reinterpret_cast<const int4*>(lut + 4 * blockIdx.x);
    const int32_t* ent = lut + hdr.x;
    const int j = threadIdx.x % BS, rg = threadIdx.x / BS, k = hdr.z * BS + j;
    const float gain = G ? G[k] : 1.f;
    const float ss = S[k], mx = fmaxf(ss, eps), ni = rsqrtf(mx), n2i = 1.0f / mx;
    float r = 0.f, dg = 0.f;
    for (int e = 0; e < hdr.y; ++e) {
        const size_t o = (size_t)ent[e] * (BS * BS);
        for (int i = rg; i < BS; i += RG) {
            const float x = TX::to_f32(X[o + i * BS + j]), dy = TY::to_f32(DY[o + i * BS + j]);
            r += (-dy * gain * x) * n2i;
            dg += dy * x * ni;
        }
    }
    red1[threadIdx.x] = r;
    red2[threadIdx.x] = dg;
    __syncthreads();
    r = 0.f;
    dg = 0.f;
#pragma unroll
    for (int g = 0; g < RG; ++g) { r += red1[g * BS + j]; dg += red2[g * BS + j]; }
    if (rg == 0 && DG) DG[k] = dg;
    r *= (ss >= eps) ? 1.f : 0.f;
    for (int e = 0; e < hdr.y; ++e) {
        const size_t o = (size_t)ent[e] * (BS * BS);
        for (int i = rg; i < BS; i += RG) {
            const float x = TX::to_f32(X[o + i * BS + j]), dy = TY::to_f32(DY[o + i * BS + j]);
            DX[o + i * BS + j] = TX::from_f32((dy * gain + x * r) * ni);
        }
    }
}

}  // namespace bsmm
