// bsmm_b64.h -- bsize 64 on the bsize-32 kernels (feature axis 1; see the 'BS64' plan in bsmm_plan.h): the three small kernels around
// the nested bsize-32 call.  Quadrant (i, j) of weight block w = rows 32 i .. 32 i + 31, columns 32 j .. 32 j + 31 of W[w] (64 x 64,
// row-major) is weight block 4 w + 2 i + j of the quadrant view.
#pragma once
#include "bsmm_common.h"

namespace bsmm {

// W[blocks][64][64] -> Q[4 * blocks][32][32], 16 bytes per thread and step (ES = bytes per element).  An xprop lut names a block by
// (output block of the call, input block of the call): in fprop that is (column half j, row half i) of W's block, in bprop (i, j) -- the
// plan builder cannot know which lut it was handed, so it numbers the quadrant 2 * (input half) + (output half) either way and the bprop
// image stores quadrant (i, j) at 2 j + i (`swap`).
template <int ES>
__global__ void __launch_bounds__(256)
b64_split_kernel(const unsigned char* __restrict__ W, unsigned char* __restrict__ Q, int blocks, int swap) {
    constexpr int ROWB = 64 * ES, QROWB = 32 * ES, PIECES = 64 * ROWB / 16;      // 16-byte pieces of a 64 x 64 block
    const int w = blockIdx.x;
    if (w >= blocks) return;
    const unsigned char* src = W + (size_t)w * 64 * ROWB;
    unsigned char* dst = Q + (size_t)w * 64 * ROWB;
    for (int p = threadIdx.x; p < PIECES; p += 256) {
        const int row = (p * 16) / ROWB, colb = (p * 16) % ROWB;                 // position inside the 64 x 64 block
        const int i = row >> 5, j = colb / QROWB;
        const int q = swap ? 2 * j + i : 2 * i + j;
        *reinterpret_cast<uint4*>(dst + (size_t)q * 32 * QROWB + (row & 31) * QROWB + (colb % QROWB)) = *reinterpret_cast<const uint4*>(src + p * 16);
    }
}

// gate of the quadrant view: every quadrant carries its block's gate
__global__ void __launch_bounds__(256)
b64_gate_kernel(const float* __restrict__ gate, float* __restrict__ gate4, int blocks) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < 4 * blocks) gate4[i] = gate[i >> 2];
}

// (round 5: the pass that put the quadrant sums of the weight gradient together, b64_finalize_kernel, is gone -- the streaming kernel's summing
//  pass writes a quadrant straight into its 64 x 64 block: bsmm_updat_v2.h::updat2_reduce_kernel, q64)

}  // namespace bsmm
