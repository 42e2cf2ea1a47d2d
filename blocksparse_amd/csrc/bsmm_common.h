// bsmm_common.h -- device-side types shared by the gfx950 block-sparse matmul kernels.
//
// MFMA operand convention used everywhere in this library (wave64, CDNA4):
//   32x32 tiles  lane -> (r = lane & 31, h = lane >> 5).  A operand: row r of the M side; B operand:
//                column r of the N side; both hold the SAME K indices, chosen by us:
//                  16-bit (v_mfma_f32_32x32x16_{bf16,f16}, 2 instr per K=32): q[0] k = 8h+j, q[1] k = 16+8h+j
//                  f32    (v_mfma_f32_32x32x2_f32, 16 instr per K=32):        v[t] k = 16h+t
//                (the hardware pairs A's (h, j) with B's (h, j); any K labelling applied to both sides
//                 gives the same sum, so we pick the one that makes per-lane loads contiguous)
//                D: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
//   16x16 tiles  lane -> (r = lane & 15, q = lane >> 4).
//                  16-bit (v_mfma_f32_16x16x32_{bf16,f16}, K = 32 per instr): k = 8q+j
//                  f32    (v_mfma_f32_16x16x4_f32, 4 instr per K = 16):      v[t] k = 4q+t
//                D: col = lane & 15, row = 4 * (lane >> 4) + reg
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bsmm {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4v __attribute__((ext_vector_type(4)));

struct DTf32 {
    typedef float T;
    static constexpr bool is16 = false;
    static __device__ __forceinline__ float to_f32(float v) { return v; }
    static __device__ __forceinline__ float from_f32(float v) { return v; }
};

struct DTf16 {
    typedef uint16_t T;
    static constexpr bool is16 = true;
    static __device__ __forceinline__ float to_f32(uint16_t v) { return (float)__builtin_bit_cast(_Float16, v); }
    static __device__ __forceinline__ uint16_t from_f32(float v) {
        return __builtin_bit_cast(uint16_t, (_Float16)v);  // v_cvt_f16_f32: round-to-nearest-even
    }
    static __device__ __forceinline__ f32x16 mfma32(uint4 a, uint4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x4 mfma16(uint4 a, uint4 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
    // K = 16: lane (r, q) holds k = 4q .. 4q+3 (8 bytes); same time as the K = 32 instruction, half the products
    static __device__ __forceinline__ f32x4 mfma16k16(uint2 a, uint2 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(f16x4, a), __builtin_bit_cast(f16x4, b), c, 0, 0, 0);
    }
};

struct DTbf16 {
    typedef uint16_t T;
    static constexpr bool is16 = true;
    static __device__ __forceinline__ float to_f32(uint16_t v) { return __builtin_bit_cast(float, (uint32_t)v << 16); }
    static __device__ __forceinline__ uint16_t from_f32(float v) {
        return __builtin_bit_cast(uint16_t, (__bf16)v);    // v_cvt_pk_bf16_f32 on gfx950: round-to-nearest-even
    }
    static __device__ __forceinline__ f32x16 mfma32(uint4 a, uint4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x4 mfma16(uint4 a, uint4 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x4 mfma16k16(uint2 a, uint2 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4v, a), __builtin_bit_cast(s16x4v, b), c, 0, 0, 0);
    }
};

__device__ __forceinline__ uint4 zero_u4() { return make_uint4(0u, 0u, 0u, 0u); }

// gather 8 16-bit elements p[0], p[stride], ... into one 16-byte register group (element j in bits 16*(j&1) of word j/2)
__device__ __forceinline__ uint4 gather8_u16(const uint16_t* p, size_t stride) {
    uint32_t e[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) e[j] = p[j * stride];
    return make_uint4(e[0] | (e[1] << 16), e[2] | (e[3] << 16), e[4] | (e[5] << 16), e[6] | (e[7] << 16));
}

// same, but elements j >= lim read as zero
__device__ __forceinline__ uint4 gather8_u16_lim(const uint16_t* p, size_t stride, int lim) {
    uint32_t e[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) e[j] = (j < lim) ? (uint32_t)p[j * stride] : 0u;
    return make_uint4(e[0] | (e[1] << 16), e[2] | (e[3] << 16), e[4] | (e[5] << 16), e[6] | (e[7] << 16));
}

// 8 contiguous 16-bit elements, elements j >= lim read as zero (lim may be <= 0 or >= 8)
__device__ __forceinline__ uint4 load8_u16_lim(const uint16_t* p, int lim) {
    if (lim >= 8) return *reinterpret_cast<const uint4*>(p);
    if (lim <= 0) return zero_u4();
    return gather8_u16_lim(p, 1, lim);
}

// ------------------------------------------------------------------------------------------------
// Frag32: this lane's share of one [32 rows] x [K = 32] operand slab (see header comment).
// ------------------------------------------------------------------------------------------------
template <class DT, bool IS16 = DT::is16>
struct Frag32;

template <class DT>
struct Frag32<DT, true> {
    uint4 q[2];
    __device__ __forceinline__ void zero() { q[0] = zero_u4(); q[1] = zero_u4(); }
    // row -> element (r, k = 0), K contiguous, all 32 K valid
    __device__ __forceinline__ void load_contig(const uint16_t* row, int h) {
        q[0] = *reinterpret_cast<const uint4*>(row + 8 * h);
        q[1] = *reinterpret_cast<const uint4*>(row + 16 + 8 * h);
    }
    // only k < klim valid
    __device__ __forceinline__ void load_contig_lim(const uint16_t* row, int h, int klim) {
        q[0] = load8_u16_lim(row + 8 * h, klim - 8 * h);
        q[1] = load8_u16_lim(row + 16 + 8 * h, klim - 16 - 8 * h);
    }
    // col -> element (r, k = 0); element (r, k) at col[k * stride]
    __device__ __forceinline__ void load_strided(const uint16_t* col, size_t stride, int h) {
        q[0] = gather8_u16(col + (size_t)(8 * h) * stride, stride);
        q[1] = gather8_u16(col + (size_t)(16 + 8 * h) * stride, stride);
    }
    __device__ __forceinline__ void load_strided_lim(const uint16_t* col, size_t stride, int h, int klim) {
        q[0] = gather8_u16_lim(col + (size_t)(8 * h) * stride, stride, klim - 8 * h);
        q[1] = gather8_u16_lim(col + (size_t)(16 + 8 * h) * stride, stride, klim - 16 - 8 * h);
    }
};

template <class DT>
struct Frag32<DT, false> {
    float v[16];
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int t = 0; t < 16; ++t) v[t] = 0.f;
    }
    __device__ __forceinline__ void load_contig(const float* row, int h) {
        const float4* p = reinterpret_cast<const float4*>(row + 16 * h);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float4 x = p[g];
            v[4 * g + 0] = x.x; v[4 * g + 1] = x.y; v[4 * g + 2] = x.z; v[4 * g + 3] = x.w;
        }
    }
    __device__ __forceinline__ void load_contig_lim(const float* row, int h, int klim) {
        if (klim >= 32) { load_contig(row, h); return; }
#pragma unroll
        for (int t = 0; t < 16; ++t) v[t] = (16 * h + t < klim) ? row[16 * h + t] : 0.f;
    }
    __device__ __forceinline__ void load_strided(const float* col, size_t stride, int h) {
#pragma unroll
        for (int t = 0; t < 16; ++t) v[t] = col[(size_t)(16 * h + t) * stride];
    }
    __device__ __forceinline__ void load_strided_lim(const float* col, size_t stride, int h, int klim) {
#pragma unroll
        for (int t = 0; t < 16; ++t) v[t] = (16 * h + t < klim) ? col[(size_t)(16 * h + t) * stride] : 0.f;
    }
};

// acc(32x32) += A(32 x K32) * B(K32 x 32)
template <class DT>
__device__ __forceinline__ void mma32(const Frag32<DT, true>& a, const Frag32<DT, true>& b, f32x16& acc) {
    acc = DT::mfma32(a.q[0], b.q[0], acc);
    acc = DT::mfma32(a.q[1], b.q[1], acc);
}
template <class DT>
__device__ __forceinline__ void mma32(const Frag32<DT, false>& a, const Frag32<DT, false>& b, f32x16& acc) {
#pragma unroll
    for (int t = 0; t < 16; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[t], b.v[t], acc, 0, 0, 0);
}

// ------------------------------------------------------------------------------------------------
// Frag16: this lane's share of a [16 rows] x [K] operand slab for the 16x16 MFMAs.
//   16-bit: 8 elements, K = 32 per slab (k = 8q + j);   f32: 4 elements, K = 16 per slab (k = 4q + t).
// Loads take a pointer to the LANE's first element (the caller applies q).
// ------------------------------------------------------------------------------------------------
template <class DT, bool IS16 = DT::is16>
struct Frag16;

template <class DT>
struct Frag16<DT, true> {
    static constexpr int KL = 8;    // K elements per lane
    static constexpr int KS = 32;   // K per slab
    uint4 q;
    __device__ __forceinline__ void zero() { q = zero_u4(); }
    __device__ __forceinline__ void load_contig(const uint16_t* p) { q = *reinterpret_cast<const uint4*>(p); }
    __device__ __forceinline__ void load_contig_lim(const uint16_t* p, int lim) { q = load8_u16_lim(p, lim); }
    __device__ __forceinline__ void load_strided(const uint16_t* p, size_t stride) { q = gather8_u16(p, stride); }
    __device__ __forceinline__ void load_strided_lim(const uint16_t* p, size_t stride, int lim) { q = gather8_u16_lim(p, stride, lim); }
};

template <class DT>
struct Frag16<DT, false> {
    static constexpr int KL = 4;
    static constexpr int KS = 16;
    float v[4];
    __device__ __forceinline__ void zero() { v[0] = v[1] = v[2] = v[3] = 0.f; }
    __device__ __forceinline__ void load_contig(const float* p) {
        float4 x = *reinterpret_cast<const float4*>(p);
        v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w;
    }
    __device__ __forceinline__ void load_contig_lim(const float* p, int lim) {
        if (lim >= 4) { load_contig(p); return; }
#pragma unroll
        for (int t = 0; t < 4; ++t) v[t] = (t < lim) ? p[t] : 0.f;
    }
    __device__ __forceinline__ void load_strided(const float* p, size_t stride) {
#pragma unroll
        for (int t = 0; t < 4; ++t) v[t] = p[(size_t)t * stride];
    }
    __device__ __forceinline__ void load_strided_lim(const float* p, size_t stride, int lim) {
#pragma unroll
        for (int t = 0; t < 4; ++t) v[t] = (t < lim) ? p[(size_t)t * stride] : 0.f;
    }
};

template <class DT>
__device__ __forceinline__ void mma16(const Frag16<DT, true>& a, const Frag16<DT, true>& b, f32x4& acc) {
    acc = DT::mfma16(a.q, b.q, acc);
}
template <class DT>
__device__ __forceinline__ void mma16(const Frag16<DT, false>& a, const Frag16<DT, false>& b, f32x4& acc) {
#pragma unroll
    for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[t], b.v[t], acc, 0, 0, 0);
}

// ------------------------------------------------------------------------------------------------
// accumulate-into-memory for lut segments that share an output block (lock id != 0).  The launcher
// zero-fills the output first.  fp32: hardware float atomic; 16-bit: CAS on the containing dword
// (one rounding per contribution -- the reference's locked path also accumulates in the storage type,
// src/gpu_hmma.h:120-129).
// ------------------------------------------------------------------------------------------------
template <class DT>
__device__ __forceinline__ void atomic_accumulate(typename DT::T* p, float v) {
    if constexpr (!DT::is16) {
        __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        uintptr_t a = reinterpret_cast<uintptr_t>(p);
        uint32_t* wp = reinterpret_cast<uint32_t*>(a & ~(uintptr_t)3);
        const bool hi = (a & 2) != 0;
        uint32_t old = __hip_atomic_load(wp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint32_t assumed;
        do {
            assumed = old;
            uint16_t cur = hi ? (uint16_t)(assumed >> 16) : (uint16_t)(assumed & 0xffffu);
            uint32_t nv = DT::from_f32(DT::to_f32(cur) + v);
            uint32_t nw = hi ? ((assumed & 0x0000ffffu) | (nv << 16)) : ((assumed & 0xffff0000u) | nv);
            old = atomicCAS(wp, assumed, nw);
        } while (old != assumed);
    }
}

// two floats -> two bf16 (round to nearest even) in one dword, lo in bits 0..15: ONE v_cvt_pk_bf16_f32.  (Written as two scalar conversions
// and a shift / or, hipcc emits two single-operand conversions plus an SDWA or: the three-piece splits of the fp32 paths spent most of their
// vector instructions there -- 488 per 32 x 32 x 64 tile in bst_nt_mfma_kernel, for 24 matrix instructions; round 6.)
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t bf16_pack2(float lo, float hi) {
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}

struct PtrList8 {
    const void* p[8];
};

// LDS-DMA of 16 bytes per lane (1 KiB per wave) to LDS byte address `lds_byte_addr` + lane * 16, issued from inline asm: with the
// builtin form hipcc drains vmcnt(0) in front of every later LDS read, and DMAs issued from asm are neither counted nor
// drained by the compiler's own waits (callers place s_waitcnt vmcnt themselves).  M0 is saved and restored.
__device__ __forceinline__ void glds16_asm(const void* gsrc, uint32_t lds_byte_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_byte_addr)
                 : "memory");
}
// Same with only the lanes of `mask` active (the LDS slot of a lane is still lane * 16; inactive lanes move nothing).
// The caller must be in wave-uniform control flow with ALL lanes active: EXEC is set to -1 afterwards.  (An `if` around
// glds16_asm costs a saveexec / branch / restore sequence per instruction, measurably more than this.)
__device__ __forceinline__ void glds16_asm_masked(const void* gsrc, uint32_t lds_byte_addr, uint64_t mask) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_mov_b64 exec, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
                 "s_mov_b64 exec, -1\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_byte_addr), "s"(mask)
                 : "memory");
}
__device__ __forceinline__ uint32_t lds_addr_of(const void* p) {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}

}  // namespace bsmm
