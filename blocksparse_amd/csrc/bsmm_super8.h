// bsmm_super8.h -- bsize 8 on the matrix cores (16-bit types): the two small data-movement kernels around the bsize-32
// kernels that do the work on the SUPER layout (plan 'BSS8', bsmm_plan.h).
//   expand8:  W (8x8 blocks)  ->  Wsel (32x32 super-blocks in the [out feature][in feature] order the xcol kernels multiply),
//             absent sub-blocks zero-filled.  ~2 KiB written per super-block, a few microseconds per pass.
//   gather8:  fp32 sums of the 32x32 super-blocks ([c][k]) -> DW (the present 8x8 blocks): alpha, beta, ONE rounding.
//   nonfinite16: does the activation tensor hold an Inf / NaN?  A super-block multiplies its zero-filled (absent) 8x8 parts with
//             live activations: 0 * x = 0 for every finite x, but 0 * Inf = NaN would reach outputs the reference leaves finite
//             (it walks only the lookup-table entries, blocksparse/matmul.py:353-392).  The flag this scan leaves makes the call
//             exact: when it is set, the per-entry V_FMA kernel recomputes the output after the matrix-core pass.
// Both: one workgroup of 256 threads per super-block, a thread moves 4 consecutive elements (8 bytes).
#pragma once
#include "bsmm_common.h"
#include "bsmm_plan.h"

namespace bsmm {

// FROM_IN_OUT: the source blocks are stored [in][out] (fprop: W[w][c][k], in = c) and are transposed on the way;
// otherwise they are stored [out][in] (bprop: out = c) and copied.
template <class DT, bool FROM_IN_OUT>
__global__ void __launch_bounds__(256)
expand8_kernel(const typename DT::T* __restrict__ W8, const int32_t* __restrict__ plan, typename DT::T* __restrict__ W32,
               int32_t* __restrict__ clear_flag = nullptr) {
    static_assert(DT::is16, "super8 path: 16-bit storage types");
    const int s = blockIdx.x;
    if (clear_flag && s == 0 && threadIdx.x == 0) clear_flag[0] = 0;      // the non-finite flag of this call (nonfinite16_kernel runs next)
    if (plan[0] != S8PLAN_MAGIC || plan[1] != S8PLAN_VERSION || s >= plan[2]) return;
    const int32_t* sub = plan + plan[3] + 16 * s;
    const int o32 = threadIdx.x >> 3, i0 = (threadIdx.x & 7) * 4;            // 4 consecutive in-features of one sub-block
    const int w = sub[4 * (i0 >> 3) + (o32 >> 3)];
    uint2 v = make_uint2(0u, 0u);
    if (w >= 0) {
        const uint16_t* src = reinterpret_cast<const uint16_t*>(W8) + (size_t)w * 64;
        if constexpr (FROM_IN_OUT) {
            const uint16_t* p = src + (i0 & 7) * 8 + (o32 & 7);
            v.x = (uint32_t)p[0] | ((uint32_t)p[8] << 16);
            v.y = (uint32_t)p[16] | ((uint32_t)p[24] << 16);
        } else {
            v = *reinterpret_cast<const uint2*>(src + (o32 & 7) * 8 + (i0 & 7));
        }
    }
    *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(W32) + (size_t)s * 1024 + o32 * 32 + i0) = v;
}

// flag[0] = 1 if any of the n8 * 8 16-bit elements at X has all exponent bits set (Inf / NaN).  EXPMASK: 0x7f80 (bf16) / 0x7c00 (f16).
// flag[0] must be 0 before (expand8_kernel of the same call clears it: `clear_flag`).
template <uint32_t EXPMASK>
__global__ void __launch_bounds__(256)
nonfinite16_kernel(const uint4* __restrict__ X, size_t n8, int32_t* __restrict__ flag) {
    constexpr uint32_t M = EXPMASK | (EXPMASK << 16);
    uint32_t hit = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
        const uint4 v = X[i];
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const uint32_t z = (w[e] & M) ^ M;                        // a 16-bit half of z is 0 <=> that element is Inf / NaN
            hit |= (z - 0x00010001u) & ~z & 0x80008000u;              // "has a zero half" (exact as an any-test)
        }
    }
    if (hit) flag[0] = 1;
}

template <class DT>
__global__ void __launch_bounds__(256)
gather8_kernel(const float* __restrict__ S32, const int32_t* __restrict__ plan, typename DT::T* __restrict__ DW8, float alpha, float beta) {
    static_assert(DT::is16, "super8 path: 16-bit storage types");
    const int s = blockIdx.x;
    if (plan[0] != S8PLAN_MAGIC || plan[1] != S8PLAN_VERSION || s >= plan[2]) return;
    const int32_t* sub = plan + plan[3] + 16 * s;
    const int c32 = threadIdx.x >> 3, k0 = (threadIdx.x & 7) * 4;
    const int w = sub[4 * (c32 >> 3) + (k0 >> 3)];
    if (w < 0) return;
    const float4 v = *reinterpret_cast<const float4*>(S32 + (size_t)s * 1024 + c32 * 32 + k0);
    float o[4] = {alpha * v.x, alpha * v.y, alpha * v.z, alpha * v.w};
    uint16_t* dst = reinterpret_cast<uint16_t*>(DW8) + (size_t)w * 64 + (c32 & 7) * 8 + (k0 & 7);
    if (beta != 0.f) {
        const uint2 old = *reinterpret_cast<const uint2*>(dst);
        o[0] += beta * DT::to_f32((typename DT::T)(old.x & 0xffffu));
        o[1] += beta * DT::to_f32((typename DT::T)(old.x >> 16));
        o[2] += beta * DT::to_f32((typename DT::T)(old.y & 0xffffu));
        o[3] += beta * DT::to_f32((typename DT::T)(old.y >> 16));
    }
    uint2 r;
    r.x = (uint32_t)DT::from_f32(o[0]) | ((uint32_t)DT::from_f32(o[1]) << 16);
    r.y = (uint32_t)DT::from_f32(o[2]) | ((uint32_t)DT::from_f32(o[3]) << 16);
    *reinterpret_cast<uint2*>(dst) = r;
}

// fp32 form (round 4: the fp32 weight gradient of bsize 8 through the bf16 streaming kernel, bsmm_api.hip::updat8_f32_split): the same gather,
// DW8 in fp32 (no rounding)
__global__ void __launch_bounds__(256)
gather8_f32_kernel(const float* __restrict__ S32, const int32_t* __restrict__ plan, float* __restrict__ DW8, float alpha, float beta,
                   const int32_t* __restrict__ skip_if = nullptr) {
    const int s = blockIdx.x;
    if (skip_if && skip_if[0] != 0) return;               // non-finite inputs: the repair pass (updat8_f32_split) writes DW8
    if (plan[0] != S8PLAN_MAGIC || plan[1] != S8PLAN_VERSION || s >= plan[2]) return;
    const int32_t* sub = plan + plan[3] + 16 * s;
    const int c32 = threadIdx.x >> 3, k0 = (threadIdx.x & 7) * 4;
    const int w = sub[4 * (c32 >> 3) + (k0 >> 3)];
    if (w < 0) return;
    const float4 v = *reinterpret_cast<const float4*>(S32 + (size_t)s * 1024 + c32 * 32 + k0);
    float4* dst = reinterpret_cast<float4*>(DW8 + (size_t)w * 64 + (c32 & 7) * 8 + (k0 & 7));
    float4 o = make_float4(alpha * v.x, alpha * v.y, alpha * v.z, alpha * v.w);
    if (beta != 0.f) {
        const float4 old = *dst;
        o.x += beta * old.x; o.y += beta * old.y; o.z += beta * old.z; o.w += beta * old.w;
    }
    *dst = o;
}

}  // namespace bsmm
