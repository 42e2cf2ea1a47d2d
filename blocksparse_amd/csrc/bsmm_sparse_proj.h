// bsmm_sparse_proj.h -- SparseProj row gather / scatter ops on (features, minibatch) tensors (blocksparse/matmul.py:835-921;
// kernels gather_scatter, scatter_add, scatter_mul, sparse_mul_grad: src/layer_norm_cn_op_gpu.cu:596-716).
//   OP_GAT  z[k][:] = x[lut[k]][:]                                   k over the K rows of z (lut = gather table)
//   OP_SCT  z[k][:] = lut[k] >= 0 ? x[lut[k]][:] : 0                 (lut = scatter table, -1 = unmapped)
//   OP_ADD  z = x;  z[lut[k]][:] += y[k][:]                          k over the K rows of y (lut = gather table)
//   OP_MUL  z[k][:] = lut[k] >= 0 ? x[k][:] * y[lut[k]][:] : x[k][:] k over the K rows of x (lut = scatter table)
//   mul grad  dx[lut[k]][:] = dz[lut[k]][:] * y[k][:] (unmapped rows: dx = dz),  dy[k][:] = dz[lut[k]][:] * x[lut[k]][:]
// Pure row copies: one thread per element, rows are contiguous (minibatch index fastest).
#pragma once
#include "bsmm_common.h"

namespace bsmm {

enum { SP_GAT = 0, SP_SCT = 1, SP_ADD = 2, SP_MUL = 3 };

template <class DT, int OP>
__global__ void __launch_bounds__(256)
sparse_proj_kernel(typename DT::T* __restrict__ Z, const typename DT::T* __restrict__ X, const typename DT::T* __restrict__ Y,
                   const int32_t* __restrict__ lut, int K, int N) {
    const int k = blockIdx.x;            // rows on grid.x (no 65535 limit), minibatch strips on grid.y
    const int src = lut[k];
    for (int n = blockIdx.y * 256 + threadIdx.x; n < N; n += gridDim.y * 256) {
        if constexpr (OP == SP_GAT) {
            Z[(size_t)k * N + n] = X[(size_t)src * N + n];
        } else if constexpr (OP == SP_SCT) {
            Z[(size_t)k * N + n] = src >= 0 ? X[(size_t)src * N + n] : DT::from_f32(0.f);
        } else if constexpr (OP == SP_ADD) {      // Z already holds X (the launcher copies unless they alias)
            Z[(size_t)src * N + n] = DT::from_f32(DT::to_f32(X[(size_t)src * N + n]) + DT::to_f32(Y[(size_t)k * N + n]));
        } else {
            const typename DT::T x = X[(size_t)k * N + n];
            Z[(size_t)k * N + n] = src >= 0 ? DT::from_f32(DT::to_f32(x) * DT::to_f32(Y[(size_t)src * N + n])) : x;
        }
    }
}

template <class DT>
__global__ void __launch_bounds__(256)
sparse_mul_grad_kernel(typename DT::T* __restrict__ DX, typename DT::T* __restrict__ DY, const typename DT::T* __restrict__ DZ,
                       const typename DT::T* __restrict__ X, const typename DT::T* __restrict__ Y, const int32_t* __restrict__ lut,
                       int K, int N) {
    const int k = blockIdx.x;            // rows on grid.x (no 65535 limit), minibatch strips on grid.y
    const int xk = lut[k];
    for (int n = blockIdx.y * 256 + threadIdx.x; n < N; n += gridDim.y * 256) {
        const float dz = DT::to_f32(DZ[(size_t)xk * N + n]);
        const float x = DT::to_f32(X[(size_t)xk * N + n]), y = DT::to_f32(Y[(size_t)k * N + n]);
        DX[(size_t)xk * N + n] = DT::from_f32(dz * y);
        DY[(size_t)k * N + n] = DT::from_f32(dz * x);
    }
}

}  // namespace bsmm
