// bsmm_xprop.h -- fprop / bprop kernels ("xprop": one kernel family, the weight operand decides which).
//
// Per lut segment (= output feature block `ob`, list of (input block, weight block) entries) and
// minibatch tile the kernels compute, in canonical form,
//        OUT[o][n] = sum_entries sum_i  Wop[w][o][i] * XT[ib][i][n]
// with Wop[w][o][i] = W[w][i][o] (fprop: o = k-in-block, i = c-in-block; the launcher passes a
// transposed copy of W so that i is contiguous) or W[w][o][i] (bprop), and
//        XT[ib][i][n] = X[(ib*bs+i)*N + n]   (axis 0)      X[n*Cin + ib*bs + i]   (axis 1).
// Replaces gemm_blocksparse_{32,16,08}x64x*_xprop (src/blocksparse_matmul_op_gpu.cu:8-958,1837-2422)
// and the tensor-core hgemm_blocksparse_*_{xn_sdd,nx_dsd} (src/blocksparse_hgemm_*_op_gpu.cu).
#pragma once
#include "bsmm_common.h"

namespace bsmm {

// ---- workgroup -> (segment, minibatch tile) mapping ------------------------------------------
// Blocks are dispatched round-robin over the 8 XCDs (block b -> XCD b % 8, observed; speed only).
// All segments of one minibatch tile re-read the same X panel, so we put them on one XCD's L2:
// "virtual tile" vt = tile * P + part runs on XCD vt % 8, where the segments are dealt over P parts
// when there are fewer than 8 tiles (P = ceil(8 / ntiles), else 1).
struct XMap {
    int ntiles, segments, P, SP;   // SP = ceil(segments / P)
    __host__ __device__ int grid() const { return 8 * SP * ((ntiles * P + 7) / 8); }
};
__device__ __forceinline__ bool xmap_decode(const XMap& m, int b, int& tile, int& seg) {
    const int xcd = b & 7, j = b >> 3;
    const int round = j / m.SP, sidx = j - round * m.SP;
    const int vt = round * 8 + xcd;
    tile = vt / m.P;
    seg = sidx * m.P + (vt - tile * m.P);
    return tile < m.ntiles && seg < m.segments;
}

// Locked segments (several lut segments write one output block): 16-bit storage types sum into the fp32 accumulator image
// Yacc (same indexing as Y, zeroed by the launcher) and lock_finalize_kernel rounds every locked block ONCE; fp32 adds
// straight into the zeroed Y.  (Yacc == nullptr with a 16-bit type keeps the storage-type CAS accumulation.)
template <class DT>
__device__ __forceinline__ void lock_accumulate(typename DT::T* Y, float* Yacc, size_t idx, float v) {
    if constexpr (DT::is16) {
        if (Yacc) { __hip_atomic_fetch_add(Yacc + idx, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }
    }
    atomic_accumulate<DT>(Y + idx, v);
}

// grid (segments, ceil(N / 256)): every segment with a lock id converts its output block (segments that share a block write
// identical values).
template <class DT, int BS, int AXIS>
__global__ void __launch_bounds__(256)
lock_finalize_kernel(const float* __restrict__ Yacc, typename DT::T* __restrict__ Y, const int32_t* __restrict__ lut, int N, int Kout) {
    const int4 hdr = *reinterpret_cast<const int4*>(lut + 4 * blockIdx.x);
    if (hdr.w == 0) return;
    const int ob = hdr.z;
    const int n = blockIdx.y * 256 + threadIdx.x;
    if (n >= N) return;
#pragma unroll
    for (int o = 0; o < BS; ++o) {
        const size_t idx = (AXIS == 1) ? ((size_t)n * Kout + ob * BS + o) : ((size_t)(ob * BS + o) * N + n);
        Y[idx] = DT::from_f32(Yacc[idx]);
    }
}

// =================================================================================================
// bsize 32, MFMA 32x32.  256 threads = 4 waves; wave w owns minibatch columns [tile*NT + w*32*NSUB, +32*NSUB).
// MFMA roles: A = Wop (M = o), B = XT (N = n)  ->  D[o][n]; every lane ends up with 4 consecutive o per
// register quad, which makes the axis-1 store a 8/16-byte vector store.
// =================================================================================================
// GATED: per-block fp32 gates (hgemm_blocksparse_*_sdd's `Gate`, src/blocksparse_hgemm_cn_64_op_gpu.cu:54-66,96-124): an
// entry with gate 0 is skipped, otherwise its 32x32 product is formed in a scratch accumulator and added scaled, i.e.
// the gate multiplies the fp32 block product exactly as in the reference (fprop_test, blocksparse/matmul.py:367-373).
template <class DT, int AXIS, int NSUB, bool GATED = false>
__global__ void __launch_bounds__(256)
xprop32_kernel(const typename DT::T* __restrict__ X, const typename DT::T* __restrict__ Wsel,
               typename DT::T* __restrict__ Y, const int32_t* __restrict__ lut, XMap map, int N, int Cin, int Kout,
               const float* __restrict__ gate = nullptr, float* __restrict__ Yacc = nullptr) {
    typedef typename DT::T T;
    constexpr int NT = 4 * 32 * NSUB;
    int tile, seg;
    if (!xmap_decode(map, blockIdx.x, tile, seg)) return;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int4 hdr = *reinterpret_cast<const int4*>(lut + 4 * seg);
    const int2* ent = reinterpret_cast<const int2*>(lut) + hdr.x;
    const int cnt = hdr.y, ob = hdr.z, lock = hdr.w;
    const int n_wave = tile * NT + wave * 32 * NSUB;
    if (n_wave >= N) return;   // whole wave out of range (no barriers in this kernel)

    f32x16 acc[NSUB];
#pragma unroll
    for (int s = 0; s < NSUB; ++s)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[s][i] = 0.f;

    for (int e = 0; e < cnt; ++e) {
        const int2 cw = ent[e];
        float g = 1.f;
        if constexpr (GATED) {
            g = gate[cw.y];
            if (g == 0.f) continue;
        }
        Frag32<DT> wf;
        wf.load_contig(Wsel + (size_t)cw.y * 1024 + r * 32, h);
#pragma unroll
        for (int s = 0; s < NSUB; ++s) {
            const int n = n_wave + s * 32 + r;
            Frag32<DT> xf;
            if (n < N) {
                if constexpr (AXIS == 1) xf.load_contig(X + (size_t)n * Cin + cw.x * 32, h);
                else                     xf.load_strided(X + (size_t)cw.x * 32 * N + n, (size_t)N, h);
            } else {
                xf.zero();
            }
            if constexpr (GATED) {
                f32x16 t;
#pragma unroll
                for (int i = 0; i < 16; ++i) t[i] = 0.f;
                mma32<DT>(wf, xf, t);
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[s][i] = fmaf(g, t[i], acc[s][i]);
            } else {
                mma32<DT>(wf, xf, acc[s]);
            }
        }
    }

#pragma unroll
    for (int s = 0; s < NSUB; ++s) {
        const int n = n_wave + s * 32 + r;
        if (n >= N) continue;
        if (lock == 0) {
            if constexpr (AXIS == 1) {
                T* yrow = Y + (size_t)n * Kout + ob * 32 + 4 * h;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    if constexpr (DT::is16) {
                        uint32_t lo = (uint32_t)DT::from_f32(acc[s][4 * g + 0]) | ((uint32_t)DT::from_f32(acc[s][4 * g + 1]) << 16);
                        uint32_t hi = (uint32_t)DT::from_f32(acc[s][4 * g + 2]) | ((uint32_t)DT::from_f32(acc[s][4 * g + 3]) << 16);
                        *reinterpret_cast<uint2*>(yrow + 8 * g) = make_uint2(lo, hi);
                    } else {
                        *reinterpret_cast<float4*>(yrow + 8 * g) =
                            make_float4(acc[s][4 * g + 0], acc[s][4 * g + 1], acc[s][4 * g + 2], acc[s][4 * g + 3]);
                    }
                }
            } else {
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int o = (reg & 3) + 8 * (reg >> 2) + 4 * h;
                    Y[(size_t)(ob * 32 + o) * N + n] = DT::from_f32(acc[s][reg]);
                }
            }
        } else {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int o = (reg & 3) + 8 * (reg >> 2) + 4 * h;
                const size_t idx = (AXIS == 1) ? ((size_t)n * Kout + ob * 32 + o) : ((size_t)(ob * 32 + o) * N + n);
                lock_accumulate<DT>(Y, Yacc, idx, acc[s][reg]);
            }
        }
    }
}

// =================================================================================================
// bsize 16, MFMA 16x16.  Wave owns 16*NSUB minibatch columns.
//   16-bit: v_mfma_f32_16x16x32 has K = 32 = two 16-wide blocks, so two lut entries are K-concatenated
//           per instruction: lanes with q = lane>>4 in {0,1} follow entry 2p, lanes with q in {2,3}
//           follow entry 2p+1 (zero operands when the segment has an odd tail).  The reference does the
//           same on Volta for bsize 8 (src/blocksparse_hgemm_cn_64_op_gpu.cu:541-624).
//   f32:    v_mfma_f32_16x16x4, 4 instructions per entry.
// =================================================================================================
// GATED (see xprop32_kernel): entries are taken one at a time (16-bit: the second K half of the instruction is zero).
template <class DT, int AXIS, int NSUB, bool GATED = false>
__global__ void __launch_bounds__(256)
xprop16_kernel(const typename DT::T* __restrict__ X, const typename DT::T* __restrict__ Wsel,
               typename DT::T* __restrict__ Y, const int32_t* __restrict__ lut, XMap map, int N, int Cin, int Kout,
               const float* __restrict__ gate = nullptr, float* __restrict__ Yacc = nullptr) {
    typedef typename DT::T T;
    constexpr int NT = 4 * 16 * NSUB;
    int tile, seg;
    if (!xmap_decode(map, blockIdx.x, tile, seg)) return;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int4 hdr = *reinterpret_cast<const int4*>(lut + 4 * seg);
    const int2* ent = reinterpret_cast<const int2*>(lut) + hdr.x;
    const int cnt = hdr.y, ob = hdr.z, lock = hdr.w;
    const int n_wave = tile * NT + wave * 16 * NSUB;
    if (n_wave >= N) return;

    f32x4 acc[NSUB];
#pragma unroll
    for (int s = 0; s < NSUB; ++s) acc[s] = f32x4{0.f, 0.f, 0.f, 0.f};

    if constexpr (DT::is16) {
        const int sub = q >> 1;          // which entry of the pair this lane follows
        const int i0 = 8 * (q & 1);      // first input feature (within the block) held by this lane
        for (int e0 = 0; e0 < cnt; e0 += (GATED ? 1 : 2)) {
            const int e = GATED ? e0 : e0 + sub;
            bool live = e < cnt;
            float g = 1.f;
            if constexpr (GATED) {
                g = gate[ent[e0].y];                       // wave-uniform
                if (g == 0.f) continue;
                live = sub == 0;
            }
            int2 cw = make_int2(0, 0);
            if (live) cw = ent[e];
            Frag16<DT> wf;
            if (live) wf.load_contig(Wsel + (size_t)cw.y * 256 + r * 16 + i0);
            else      wf.zero();
#pragma unroll
            for (int s = 0; s < NSUB; ++s) {
                const int n = n_wave + s * 16 + r;
                Frag16<DT> xf;
                if (live && n < N) {
                    if constexpr (AXIS == 1) xf.load_contig(X + (size_t)n * Cin + cw.x * 16 + i0);
                    else                     xf.load_strided(X + (size_t)(cw.x * 16 + i0) * N + n, (size_t)N);
                } else {
                    xf.zero();
                }
                if constexpr (GATED) {
                    f32x4 t = f32x4{0.f, 0.f, 0.f, 0.f};
                    mma16<DT>(wf, xf, t);
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[s][i] = fmaf(g, t[i], acc[s][i]);
                } else {
                    mma16<DT>(wf, xf, acc[s]);
                }
            }
        }
    } else {
        const int i0 = 4 * q;
        for (int e = 0; e < cnt; ++e) {
            const int2 cw = ent[e];
            float g = 1.f;
            if constexpr (GATED) {
                g = gate[cw.y];
                if (g == 0.f) continue;
            }
            Frag16<DT> wf;
            wf.load_contig(Wsel + (size_t)cw.y * 256 + r * 16 + i0);
#pragma unroll
            for (int s = 0; s < NSUB; ++s) {
                const int n = n_wave + s * 16 + r;
                Frag16<DT> xf;
                if (n < N) {
                    if constexpr (AXIS == 1) xf.load_contig(X + (size_t)n * Cin + cw.x * 16 + i0);
                    else                     xf.load_strided(X + (size_t)(cw.x * 16 + i0) * N + n, (size_t)N);
                } else {
                    xf.zero();
                }
                if constexpr (GATED) {
                    f32x4 t = f32x4{0.f, 0.f, 0.f, 0.f};
                    mma16<DT>(wf, xf, t);
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[s][i] = fmaf(g, t[i], acc[s][i]);
                } else {
                    mma16<DT>(wf, xf, acc[s]);
                }
            }
        }
    }

    // D[o][n]: n = r (col), o = 4q + reg (row)
#pragma unroll
    for (int s = 0; s < NSUB; ++s) {
        const int n = n_wave + s * 16 + r;
        if (n >= N) continue;
        if (lock == 0) {
            if constexpr (AXIS == 1) {
                T* yp = Y + (size_t)n * Kout + ob * 16 + 4 * q;
                if constexpr (DT::is16) {
                    uint32_t lo = (uint32_t)DT::from_f32(acc[s][0]) | ((uint32_t)DT::from_f32(acc[s][1]) << 16);
                    uint32_t hi = (uint32_t)DT::from_f32(acc[s][2]) | ((uint32_t)DT::from_f32(acc[s][3]) << 16);
                    *reinterpret_cast<uint2*>(yp) = make_uint2(lo, hi);
                } else {
                    *reinterpret_cast<float4*>(yp) = make_float4(acc[s][0], acc[s][1], acc[s][2], acc[s][3]);
                }
            } else {
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) Y[(size_t)(ob * 16 + 4 * q + reg) * N + n] = DT::from_f32(acc[s][reg]);
            }
        } else {
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int o = 4 * q + reg;
                const size_t idx = (AXIS == 1) ? ((size_t)n * Kout + ob * 16 + o) : ((size_t)(ob * 16 + o) * N + n);
                lock_accumulate<DT>(Y, Yacc, idx, acc[s][reg]);
            }
        }
    }
}

// =================================================================================================
// VALU kernel, any bsize (production path for bsize 8; independent cross-check for 16/32).
// One thread = one minibatch column n, all BS outputs of the segment; the weight block is staged in LDS
// as fp32 in Wop[o][i] order (read back as wave-uniform broadcasts).  Takes W in its natural layout
// (no transposed workspace), FPROP selects the operand orientation.
// =================================================================================================
template <class DT, int BS, int AXIS, bool FPROP>
__global__ void __launch_bounds__(256)
xprop_valu_kernel(const typename DT::T* __restrict__ X, const typename DT::T* __restrict__ W,
                  typename DT::T* __restrict__ Y, const int32_t* __restrict__ lut, int N, int Cin, int Kout,
                  const float* __restrict__ gate = nullptr, float* __restrict__ Yacc = nullptr, const int32_t* __restrict__ only_if = nullptr) {
    typedef typename DT::T T;
    __shared__ float Wl[BS * BS];
    if (only_if && only_if[0] == 0) return;          // repair pass of the bsize-8 super-block path: runs only when its flag is set
    const int seg = blockIdx.x;
    const int n = blockIdx.y * 256 + threadIdx.x;
    const int4 hdr = *reinterpret_cast<const int4*>(lut + 4 * seg);
    const int2* ent = reinterpret_cast<const int2*>(lut) + hdr.x;
    const int cnt = hdr.y, ob = hdr.z, lock = hdr.w;

    float acc[BS];
#pragma unroll
    for (int o = 0; o < BS; ++o) acc[o] = 0.f;

    for (int e = 0; e < cnt; ++e) {
        const int2 cw = ent[e];
        const float g = gate ? gate[cw.y] : 1.f;           // uniform over the workgroup
        if (g == 0.f) continue;
        __syncthreads();
        for (int idx = threadIdx.x; idx < BS * BS; idx += 256) {
            const int o = idx / BS, i = idx % BS;
            const T v = FPROP ? W[(size_t)cw.y * BS * BS + i * BS + o] : W[(size_t)cw.y * BS * BS + o * BS + i];
            Wl[idx] = DT::to_f32(v);
        }
        __syncthreads();
        if (n < N) {
            float x[BS];
#pragma unroll
            for (int i = 0; i < BS; ++i) {
                const T v = (AXIS == 1) ? X[(size_t)n * Cin + cw.x * BS + i] : X[(size_t)(cw.x * BS + i) * N + n];
                x[i] = DT::to_f32(v);
            }
#pragma unroll
            for (int o = 0; o < BS; ++o) {
                float a = 0.f;
#pragma unroll
                for (int i = 0; i < BS; ++i) a = fmaf(Wl[o * BS + i], x[i], a);
                acc[o] = fmaf(g, a, acc[o]);
            }
        }
    }
    if (n >= N) return;
#pragma unroll
    for (int o = 0; o < BS; ++o) {
        const size_t idx = (AXIS == 1) ? ((size_t)n * Kout + ob * BS + o) : ((size_t)(ob * BS + o) * N + n);
        if (lock == 0) Y[idx] = DT::from_f32(acc[o]);
        else lock_accumulate<DT>(Y, Yacc, idx, acc[o]);
    }
}

// Wt[w][o][i] = W[w][i][o]; one workgroup per block (fprop's weight operand wants i contiguous).
template <class DT, int BS>
__global__ void __launch_bounds__(256) transpose_blocks_kernel(const typename DT::T* __restrict__ W,
                                                               typename DT::T* __restrict__ Wt, int blocks) {
    typedef typename DT::T T;
    __shared__ T tile[BS][BS + 2];
    const size_t base = (size_t)blockIdx.x * BS * BS;
    for (int idx = threadIdx.x; idx < BS * BS; idx += 256) tile[idx / BS][idx % BS] = W[base + idx];
    __syncthreads();
    for (int idx = threadIdx.x; idx < BS * BS; idx += 256) Wt[base + idx] = tile[idx % BS][idx / BS];
}

}  // namespace bsmm
