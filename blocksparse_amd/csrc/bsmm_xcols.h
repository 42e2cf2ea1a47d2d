// bsmm_xcols.h -- fp32 xprop (bsize 32, both feature axes; BASELINE configs[1]) on the 16-bit matrix cores, EXACTLY.
//
// v_mfma_f32_32x32x2_f32 needs 1024 cycles per (block, 32-row tile); the bf16 instruction v_mfma_f32_32x32x16_bf16 does the
// same K = 32 in 64.  Every fp32 value is the exact sum of three bf16 pieces (x = b1 + b2 + b3: 8 + 8 + 8 significand bits,
// the two subtractions are exact in fp32), a bf16 x bf16 product is exact in fp32, and the accumulation stays fp32, so
//     x * w  =  b1c1 + (b1c2 + b2c1) + (b1c3 + b2c2 + b3c1)  +  O(2^-26 |x w|)
// with six MFMAs (384 cycles) -- the three dropped terms are below a quarter of an fp32 ulp of the product.  The result
// differs from the fp32-MFMA kernel (xcol32f) only in the order of fp32 additions.  (Same device as the fp32 attention
// kernels, bst_kernels.h.)  Inf / NaN inputs give NaN (inf - inf in the split), like any 0 * inf.
//
// Two pre-passes write the pieces as bf16 arrays into the caller's workspace: split3_x_kernel (activations, [3] x the input layout)
// and split3_w_kernel (weights, [3][blocks][32][32], transposed per block for fprop).  The main kernel is the WIDE xcol
// kernel (bsmm_xcol.h: 16 waves, wave v owns output block v of the group for all 128 rows, 'BSXC' plan with G = 16) with
// three slabs per pair step: ring of 2 steps x 3 x 16 KiB = 96 KiB, one barrier per step (a step now carries 6x the MFMA
// work of the bf16 kernel's).  Weight pieces (24 VGPRs) are requested right after the last use of the previous ones --
// from inline asm, with the waits placed by hand, because hipcc's own vmcnt for an ordinary load would also drain the
// LDS-DMA of the next slab that was issued in between.
#pragma once
#include "bsmm_common.h"
#include "bsmm_plan.h"
#include "bsmm_xcol.h"

namespace bsmm {

// The first piece never exceeds the input: a first piece that ROUNDED UP to Inf (finite 3.39e38 < |x| <= FLT_MAX; it would leave
// Inf - Inf = NaN behind) is stepped back to the largest finite bf16, 0x7f7f, and the split stays exact (x - 0x7f7f0000 has at most 16
// significant bits).  Branch-free on the 16-bit piece (and / compare / subtract-with-carry, no constant in a vector register): round 5's
// two-level `if` in here was inlined into the LDS -> LDS conversion loop of xcol32sf_kernel and pushed it from 127 registers to 128 + 8
// spilled -- BASELINE configs[1] lost a fifth of its throughput; a v_med3_f32 clamp of x still spilled 4 (its two
// constants live in vector registers across the loop).  tests/test_codeobj.py now fails on any spill of a hot kernel.
// A non-finite x leaves NaN pieces (Inf: 0x7f7f, Inf, NaN; NaN: NaN all the way), so every output that x enters is NaN -- non-finite
// exactly where the unsplit IEEE product is non-finite (Inf or NaN there).  The weight-gradient paths do better: split3_x_kernel raises a
// flag for such inputs and the call is re-run on the fp32 kernels (bsmm_api.hip::f32_split_repair), which reproduces IEEE's Inf / NaN sets.
__device__ __forceinline__ void split3(float x, uint32_t& b1, uint32_t& b2, uint32_t& b3) {
    uint16_t p1 = DTbf16::from_f32(x);
    p1 -= (uint16_t)((p1 & 0x7fffu) == 0x7f80u);      // rounded up to Inf (or was Inf): the largest finite bf16 instead
    const float r1 = x - DTbf16::to_f32(p1);
    const uint16_t p2 = DTbf16::from_f32(r1);
    const float r2 = r1 - DTbf16::to_f32(p2);
    b1 = p1; b2 = p2; b3 = DTbf16::from_f32(r2);
}

// P[q][i] = piece q of X[i], i < n (n % 8 == 0); 8 elements per thread
__global__ void __launch_bounds__(256)
split3_x_kernel(const float* __restrict__ X, uint16_t* __restrict__ P, size_t n, int32_t* __restrict__ nonfinite = nullptr) {
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 8;
    if (i >= n) return;
    const float4 a = *reinterpret_cast<const float4*>(X + i), b = *reinterpret_cast<const float4*>(X + i + 4);
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    uint32_t p[3][8];
#pragma unroll
    for (int j = 0; j < 8; ++j) split3(v[j], p[0][j], p[1][j], p[2][j]);
    if (nonfinite) {
        bool bad = false;
#pragma unroll
        for (int j = 0; j < 8; ++j) bad |= (__builtin_bit_cast(uint32_t, v[j]) & 0x7f800000u) == 0x7f800000u;
        if (bad) nonfinite[0] = 1;
    }
#pragma unroll
    for (int q = 0; q < 3; ++q)
        *reinterpret_cast<uint4*>(P + q * n + i) = make_uint4(p[q][0] | (p[q][1] << 16), p[q][2] | (p[q][3] << 16),
                                                              p[q][4] | (p[q][5] << 16), p[q][6] | (p[q][7] << 16));
}

// P[q][w][o][i] = piece q of (TRANS ? W[w][i][o] : W[w][o][i]); one workgroup per 32x32 block, 4 elements per thread
template <bool TRANS>
__global__ void __launch_bounds__(256)
split3_w_kernel(const float* __restrict__ W, uint16_t* __restrict__ P, int blocks) {
    const int w = blockIdx.x;
    const int o = threadIdx.x >> 3, i0 = (threadIdx.x & 7) * 4;
    const float* src = W + (size_t)w * 1024;
    float v[4];
    if constexpr (TRANS) {
        // through LDS: the block is read as full rows (the strided gather of round 3 made this launch 35 us at the bench shape, against
        // 8 us for the other orientation), rows padded to 33 words so that the column reads below hit 32 different banks
        __shared__ float tile[32 * 33];
        const float4 a = *reinterpret_cast<const float4*>(src + o * 32 + i0);
        tile[o * 33 + i0 + 0] = a.x; tile[o * 33 + i0 + 1] = a.y; tile[o * 33 + i0 + 2] = a.z; tile[o * 33 + i0 + 3] = a.w;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = tile[(i0 + j) * 33 + o];
    } else {
        const float4 a = *reinterpret_cast<const float4*>(src + o * 32 + i0);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
    }
    uint32_t p[3][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) split3(v[j], p[0][j], p[1][j], p[2][j]);
    const size_t stride = (size_t)blocks * 1024;
#pragma unroll
    for (int q = 0; q < 3; ++q)
        *reinterpret_cast<uint2*>(P + q * stride + (size_t)w * 1024 + o * 32 + i0) = make_uint2(p[q][0] | (p[q][1] << 16), p[q][2] | (p[q][3] << 16));
}

constexpr int XS_G = 16;                    // waves = output blocks per workgroup (the wide 'BSXC' plan)
constexpr int XS_SLOT = 3 * XC_SLAB;        // the three piece slabs of one pair step: 48 KiB
constexpr int XS_LDS = 2 * XS_SLOT;         // ring of two steps; also holds the 64 KiB epilogue staging tile
constexpr int XS_ROWB = XS_G * 128;         // bytes per staged output row (16 blocks x 32 fp32)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// 16-byte global load from inline asm (not tracked by hipcc's waitcnt insertion; see the file comment)
__device__ __forceinline__ void gload16_asm(u32x4& dst, const void* src) {
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(src) : "memory");
}
__device__ __forceinline__ f32x16 mfma32_bf16(u32x4 a, uint4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// AXIS = 1: activations (N, C), slab rows are minibatch rows (xcol32_a1_kernel).  AXIS = 0: activations (C, N), slab rows are
// features, B-operand fragments by transposing reads, direct stores (xcol32_a0_kernel; needs N % 8 == 0).
template <int AXIS>
__global__ void __launch_bounds__(64 * XS_G, 4)
xcol32s_kernel(const uint16_t* __restrict__ Xp, const uint16_t* __restrict__ Wp, float* __restrict__ Y,
                  const int32_t* __restrict__ plan, XMap map, int N, int Cin, int Kout, int blocks) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int tile, grp;
    if (!xmap_decode(map, blockIdx.x, tile, grp)) return;
    if (plan[0] != XCPLAN_MAGIC || plan[1] != XCPLAN_VERSION || plan[2] != XS_G) return;
    const int4 gh = *reinterpret_cast<const int4*>(plan + plan[5] + 4 * grp);
    const int step_off = gh.x, nsteps = gh.y, ob0 = gh.z, nob = gh.w;
    const int32_t* pairs = plan + plan[6] + step_off;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int32_t* wt0 = plan + plan[7] + 2 * XS_G * step_off + (2 * wave) * nsteps;   // my column, even half of the pair
    const int32_t* wt1 = wt0 + nsteps;                                                  // odd half
    const int r = lane & 31, h = lane >> 5;
    const int n_tile = tile * XC_R;
    const uint32_t base_addr = lds_addr_of(smem);
    const int npairs_full = Cin / 64;
    const size_t pstride = (size_t)N * Cin, wstride = (size_t)blocks * 1024;

    // X DMA: a slab is 16 instructions of 1 KiB, wave v issues instruction v of each of the three slabs.
    //   axis 1: 8 minibatch rows x 128 B per instruction; axis 0: 4 feature rows x 256 B (XC0_* geometry of bsmm_xcol.h)
    static_assert(XC0_SLAB == XC_SLAB, "both slab shapes are 16 KiB");
    const int drow = AXIS == 1 ? 8 * wave + (lane >> 3) : XC0_RPI * wave + lane / XC0_PPR;
    const int dpiece = AXIS == 1 ? (lane & 7) ^ ((drow >> 1) & 7) : (lane % XC0_PPR) ^ (4 * (drow & 3));
    // axis 1: rows past N are clamped (never stored); axis 0: columns past N are clamped re-reads (never stored)
    const uint16_t* xsrc = AXIS == 1 ? Xp + (size_t)min(n_tile + drow, N - 1) * Cin + dpiece * 8 : Xp + min(n_tile + dpiece * 8, N - 8);
    const int oddsub = (dpiece & 4) ? 32 : 0;                                              // axis 1: trailing pair without its odd block
    auto issue_x = [&](int p, int pos) {
        const uint16_t* src;
        if constexpr (AXIS == 1) src = xsrc + (p * 64 - (p < npairs_full ? 0 : oddsub));
        else                     src = xsrc + (size_t)min(p * 64 + drow, Cin - 1) * N;     // a missing odd half re-reads the last row
#pragma unroll
        for (int q = 0; q < 3; ++q)
            glds16_asm(src + q * pstride, __builtin_amdgcn_readfirstlane(base_addr + pos * XS_SLOT + q * XC_SLAB + wave * 1024));
    };
    const int xsw = (r >> 1) & 7;
    int xrd[2][2];
#pragma unroll
    for (int half = 0; half < 2; ++half)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) xrd[half][kk] = r * 128 + (((4 * half + 2 * kk + h) ^ xsw) << 4);
    // axis 0, transposing reads: 16-lane group g16 -> minibatch columns 16*(g16&1) .. +15 of a 32-column tile, K half g16 >> 1;
    // lane t16 points at row (t16 >> 2) of a 4-row band, 8 bytes at column 4*(t16 & 3)
    const int g16 = lane >> 4, t16 = lane & 15, trow = t16 >> 2;
    const int tcolb = (16 * (g16 & 1) + 4 * (t16 & 3)) * 2;

    f32x16 acc[XC_RT];
#pragma unroll
    for (int t = 0; t < XC_RT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

    u32x4 wq[3][2];      // weight pieces of the entry this wave handles next: [piece][K half]
#pragma unroll
    for (int q = 0; q < 3; ++q) wq[q][0] = wq[q][1] = u32x4{0u, 0u, 0u, 0u};
    auto request_w = [&](int w) {
        const uint16_t* row = Wp + (size_t)w * 1024 + r * 32 + 8 * h;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            gload16_asm(wq[q][0], row + q * wstride);
            gload16_asm(wq[q][1], row + q * wstride + 16);
        }
    };
    auto wait_all = [&]() {      // everything this wave requested has landed; ties the weight registers to the wait
        asm volatile("s_waitcnt vmcnt(0)"
                     : "+v"(wq[0][0]), "+v"(wq[0][1]), "+v"(wq[1][0]), "+v"(wq[1][1]), "+v"(wq[2][0]), "+v"(wq[2][1])
                     :
                     : "memory");
    };
    auto block = [&](const unsigned char* slot, int half) {
#pragma unroll
        for (int t = 0; t < XC_RT; ++t)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                uint4 x0, x1, x2;
                if constexpr (AXIS == 1) {
                    const unsigned char* p = slot + t * 4096 + xrd[half][kk];
                    x0 = *reinterpret_cast<const uint4*>(p);
                    x1 = *reinterpret_cast<const uint4*>(p + XC_SLAB);
                    x2 = *reinterpret_cast<const uint4*>(p + 2 * XC_SLAB);
                } else {
                    // rows (features) 32*half + 16*kk + 8*(g16>>1) + {0..3 | 4..7}; row & 3 == trow for both bands
                    const int row0 = 32 * half + 16 * kk + 8 * (g16 >> 1) + trow;
                    const int byte = 64 * t + tcolb;
                    const unsigned char* p = slot + row0 * XC0_ROWB + ((((byte >> 4) ^ (4 * trow)) << 4) | (byte & 15));
                    auto tr = [&](const unsigned char* q) {
                        const uint2 lo = ds_tr16(q), hi = ds_tr16(q + 4 * XC0_ROWB);
                        return make_uint4(lo.x, lo.y, hi.x, hi.y);
                    };
                    x0 = tr(p); x1 = tr(p + XC_SLAB); x2 = tr(p + 2 * XC_SLAB);
                }
                // smallest terms first
                acc[t] = mfma32_bf16(wq[2][kk], x0, acc[t]);
                acc[t] = mfma32_bf16(wq[1][kk], x1, acc[t]);
                acc[t] = mfma32_bf16(wq[0][kk], x2, acc[t]);
                acc[t] = mfma32_bf16(wq[1][kk], x0, acc[t]);
                acc[t] = mfma32_bf16(wq[0][kk], x1, acc[t]);
                acc[t] = mfma32_bf16(wq[0][kk], x0, acc[t]);
            }
    };

    const bool owner = wave < nob;
    if (nsteps > 0) {
        for (int tb = 0; tb < nsteps; tb += 64) {     // lane-indexed tables for steps [tb, tb+64)
            const int idx = min(tb + lane, nsteps - 1);
            const int pv = pairs[idx];
            const int tend = min(64, nsteps - tb);
            const int w0v = (owner && lane < tend) ? wt0[idx] : -1, w1v = (owner && lane < tend) ? wt1[idx] : -1;
            const uint64_t m0 = __ballot(w0v >= 0), m1 = __ballot(w1v >= 0);     // bit s: this wave has a block in half 0 / 1 of step s
            // entries of this wave in walk order: e = 2 * step + half
            auto next_entry = [&](int e) -> int {      // first entry >= e, or 128
                const int s = e >> 1;
                if (s >= 64) return 128;
                uint64_t a = m0 >> s;
                const uint64_t b = m1 >> s;
                if (e & 1) a &= ~1ull;
                const int ea = a ? 2 * (s + __builtin_ctzll(a)) : 128;
                const int eb = b ? 2 * (s + __builtin_ctzll(b)) + 1 : 128;
                return min(ea, eb);
            };
            auto entry_block = [&](int e) -> int {
                const int s = e >> 1;
                return (e & 1) ? __builtin_amdgcn_readlane(w1v, s) : __builtin_amdgcn_readlane(w0v, s);
            };
            int ne = __builtin_amdgcn_readfirstlane(next_entry(0));
            if (ne < 128) request_w(entry_block(ne));
            issue_x(__builtin_amdgcn_readlane(pv, 0), 0);
            for (int s = 0; s < tend; ++s) {
                wait_all();            // my share of this step's slabs and my next weight pieces have landed
                __syncthreads();       // everyone's did; everyone has left step s - 1, whose slot is refilled now
                if (s + 1 < tend) issue_x(__builtin_amdgcn_readlane(pv, s + 1), (s + 1) & 1);
                const unsigned char* slot = smem + (s & 1) * XS_SLOT;
                while ((ne >> 1) == s) {
                    block(slot, ne & 1);
                    ne = __builtin_amdgcn_readfirstlane(next_entry(ne + 1));
                    if (ne < 128) request_w(entry_block(ne));      // into the registers just used (the MFMAs have read them)
                    if ((ne >> 1) == s) wait_all();                // second block of the same step: needed right away
                }
            }
            wait_all();
            __syncthreads();   // the next batch re-primes slot 0
        }
    }

    if constexpr (AXIS == 0) {
        // D[o][n]: col = n = r, rows o = (reg & 3) + 8 * (reg >> 2) + 4h  ->  Y[(ob*32 + o) * N + n]: 128 contiguous bytes per half wave
        if (!owner) return;
#pragma unroll
        for (int t = 0; t < XC_RT; ++t) {
            const int n = n_tile + t * 32 + r;
            if (n >= N) continue;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int o = (reg & 3) + 8 * (reg >> 2) + 4 * h;
                Y[(size_t)((ob0 + wave) * 32 + o) * N + n] = acc[t][reg];
            }
        }
    } else {
    // Epilogue.  D[o][n]: col = n = r, rows o = (reg & 3) + 8 * (reg >> 2) + 4h: 4 consecutive o = one 16-byte piece.
    // 32 rows at a time are staged as [32][2 KiB] (pieces XOR-swizzled with n) and stored as full rows.
    const int rowbytes = nob * 128;
    float* ybase = Y + (size_t)ob0 * 32;
#pragma unroll
    for (int t = 0; t < XC_RT; ++t) {
        __syncthreads();
        if (owner) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int piece = wave * 8 + 2 * g + h;
                *reinterpret_cast<float4*>(smem + r * XS_ROWB + ((piece ^ r) << 4)) =
                    make_float4(acc[t][4 * g + 0], acc[t][4 * g + 1], acc[t][4 * g + 2], acc[t][4 * g + 3]);
            }
        }
        __syncthreads();
        constexpr int PPR = XS_ROWB / 16;      // 128 pieces per row
        for (int i = threadIdx.x; i < 32 * PPR; i += 64 * XS_G) {
            const int nn = i / PPR, piece = i % PPR, row = n_tile + 32 * t + nn;
            if (row < N && piece * 16 < rowbytes) {
                const float4 v = *reinterpret_cast<const float4*>(smem + nn * XS_ROWB + ((piece ^ nn) << 4));
                *reinterpret_cast<float4*>(reinterpret_cast<unsigned char*>(ybase + (size_t)row * Kout) + piece * 16) = v;
            }
        }
    }
    }
}

// ---- feature axis 1, round 4: the activation split FUSED into the kernel ---------------------------------------
// The pre-pass above reads the fp32 activations and writes 6 bytes per element of pieces (335 MB, 52 us at the bench shape) that the main
// kernel then streams as three 16 KiB slabs per pair step.  Here the fp32 slab itself (128 rows x 64 features = 32 KiB, two thirds of the
// bytes) is staged by LDS-DMA, and the pieces are made on the way from LDS to LDS: wave v requests rows 8v .. 8v + 7 of a step's slab
// (two 1 KiB instructions: 4 rows x 256 B) and converts exactly THOSE rows -- so a counted wait on its own requests is all it needs
// before it reads them -- into the piece slabs of the layout the fragment reads expect (the image the bf16 DMA of xcol32s_kernel
// produces: row r, 16-byte piece j at position j ^ ((r >> 1) & 7)).  Thread (row R = tid >> 3, piece j = tid & 7) splits 8 values: two
// 16-byte reads, ~60 vector operations, three 16-byte writes -- per step and wave, next to 48 MFMAs per block.  Ring: two stage slots of
// 32 KiB (requested TWO steps ahead) + two piece slots of 48 KiB = 160 KiB, one barrier per step as before: at the top of iteration s
// the barrier says that every wave has converted step s (done in iteration s - 1) and finished the blocks of step s - 1, whose piece
// slot the conversion of step s + 1 now overwrites.  Same pieces, same MFMA order: bit-identical to the pre-pass form.
constexpr int XSF_STAGE = XC_R * 256;                 // fp32 slab of a pair step: 32 KiB
constexpr int XSF_LDS = 2 * XS_SLOT + 2 * XSF_STAGE;  // 160 KiB
static_assert(XC_R == 128 && XSF_LDS == 163840, "the fused-split kernel fills the LDS of a CU");

__global__ void __launch_bounds__(64 * XS_G, 4)
xcol32sf_kernel(const float* __restrict__ Xf, const uint16_t* __restrict__ Wp, float* __restrict__ Y,
                const int32_t* __restrict__ plan, XMap map, int N, int Cin, int Kout, int blocks) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int tile, grp;
    if (map.P == 0) {
        // group-per-XCD mapping (the launcher sets map.P = 0 when the groups are a multiple of 8): workgroup b runs on XCD b % 8 and takes
        // group (b % 8) + 8 * ((b / 8) / ntiles), row tile (b / 8) % ntiles.  An XCD then multiplies with the weight pieces of ONE group at a
        // time (2.5 MB at the bench shape: they stay in its 4 MB L2) and streams all of X -- which is requested two steps ahead and does not
        // mind coming from the Infinity Cache -- instead of sharing X and taking every weight piece from beyond the L2 right when it is needed.
        const int j = blockIdx.x >> 3;
        grp = (blockIdx.x & 7) + 8 * (j / map.ntiles);
        tile = j - (j / map.ntiles) * map.ntiles;
        if (grp >= map.segments) return;
    } else if (!xmap_decode(map, blockIdx.x, tile, grp)) return;
    if (plan[0] != XCPLAN_MAGIC || plan[1] != XCPLAN_VERSION || plan[2] != XS_G) return;
    const int4 gh = *reinterpret_cast<const int4*>(plan + plan[5] + 4 * grp);
    const int step_off = gh.x, nsteps = gh.y, ob0 = gh.z, nob = gh.w;
    const int32_t* pairs = plan + plan[6] + step_off;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int32_t* wt0 = plan + plan[7] + 2 * XS_G * step_off + (2 * wave) * nsteps;   // my column, even half of the pair
    const int32_t* wt1 = wt0 + nsteps;                                                  // odd half
    const int r = lane & 31, h = lane >> 5;
    const int n_tile = tile * XC_R;
    const uint32_t base_addr = lds_addr_of(smem);
    unsigned char* stage = smem + 2 * XS_SLOT;
    const int npairs_full = Cin / 64;
    const size_t wstride = (size_t)blocks * 1024;

    // stage DMA: instruction i of a slab = rows 4 i .. 4 i + 3 (256 B each), this lane's 16 bytes = floats 4 (lane & 15) .. of row
    // 4 i + (lane >> 4); wave v issues instructions 2 v and 2 v + 1.  Rows past N re-read row N - 1 (never stored); a trailing pair
    // without its odd block re-reads the even one (never multiplied: the plan has no entry there).
    const int d_row0 = 8 * wave + (lane >> 4), d_q = lane & 15;
    const float* xsrc0 = Xf + (size_t)min(n_tile + d_row0, N - 1) * Cin + 4 * d_q;
    const float* xsrc1 = Xf + (size_t)min(n_tile + d_row0 + 4, N - 1) * Cin + 4 * d_q;
    const int oddsub = d_q >= 8 ? 32 : 0;
    auto issue_x = [&](int p, int pos) {
        const int off = p * 64 - (p < npairs_full ? 0 : oddsub);
        const uint32_t dst = __builtin_amdgcn_readfirstlane(base_addr + 2 * XS_SLOT + pos * XSF_STAGE + wave * 2048);
        glds16_asm(xsrc0 + off, dst);
        glds16_asm(xsrc1 + off, dst + 1024);
    };
    // conversion of my rows of a stage slot into a piece slot
    const int cR = 8 * wave + (lane >> 3), cj = lane & 7;
    const int c_rd = cR * 256 + cj * 32, c_wr = cR * 128 + ((cj ^ ((cR >> 1) & 7)) << 4);
    auto convert = [&](int pos) {
        // two values at a time, packed as they are made: the live set is 2 floats + 3 packed words per pair, not 8 + 24
        const unsigned char* src = stage + pos * XSF_STAGE + c_rd;
        unsigned char* dst = smem + pos * XS_SLOT + c_wr;
        uint32_t pk[3][4];
#pragma unroll
        for (int hlf = 0; hlf < 2; ++hlf) {
            const float4 a = *reinterpret_cast<const float4*>(src + 16 * hlf);
            const float v[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                uint32_t lo[3], hi[3];
                split3(v[2 * j], lo[0], lo[1], lo[2]);
                split3(v[2 * j + 1], hi[0], hi[1], hi[2]);
#pragma unroll
                for (int q = 0; q < 3; ++q) pk[q][2 * hlf + j] = lo[q] | (hi[q] << 16);
            }
        }
#pragma unroll
        for (int q = 0; q < 3; ++q) *reinterpret_cast<uint4*>(dst + q * XC_SLAB) = make_uint4(pk[q][0], pk[q][1], pk[q][2], pk[q][3]);
    };
    const int xsw = (r >> 1) & 7;
    int xrd[2][2];
#pragma unroll
    for (int half = 0; half < 2; ++half)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) xrd[half][kk] = r * 128 + (((4 * half + 2 * kk + h) ^ xsw) << 4);

    f32x16 acc[XC_RT];
#pragma unroll
    for (int t = 0; t < XC_RT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

    u32x4 wq[3][2];      // weight pieces of the entry this wave handles next: [piece][K half]
#pragma unroll
    for (int q = 0; q < 3; ++q) wq[q][0] = wq[q][1] = u32x4{0u, 0u, 0u, 0u};
    auto request_w_half = [&](int w, int kk) {      // K half kk of the pieces of block w (w < 0: nothing)
        if (w < 0) return;
        const uint16_t* row = Wp + (size_t)w * 1024 + r * 32 + 8 * h + 16 * kk;
#pragma unroll
        for (int q = 0; q < 3; ++q) gload16_asm(wq[q][kk], row + q * wstride);
    };
    auto request_w = [&](int w) { request_w_half(w, 0); request_w_half(w, 1); };
    auto wait_all = [&]() {      // everything this wave requested has landed; ties the weight registers to the wait
        asm volatile("s_waitcnt vmcnt(0)"
                     : "+v"(wq[0][0]), "+v"(wq[0][1]), "+v"(wq[1][0]), "+v"(wq[1][1]), "+v"(wq[2][0]), "+v"(wq[2][1])
                     :
                     : "memory");
    };
    // K half kk of the block for all four row tiles, then the next block's pieces of that half are requested into the registers just
    // used -- half a block (768 cycles of MFMAs) earlier than after the whole block: the weight pieces come from beyond the L2 (20 MB of
    // them at the bench shape) and the wave that has blocks in consecutive steps waits for them at every barrier.  Per accumulator the
    // order of the products is unchanged (K half 0, then 1).
    auto block = [&](const unsigned char* slot, int half, int wnext) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
            for (int t = 0; t < XC_RT; ++t) {
                const unsigned char* p = slot + t * 4096 + xrd[half][kk];
                const uint4 x0 = *reinterpret_cast<const uint4*>(p);
                const uint4 x1 = *reinterpret_cast<const uint4*>(p + XC_SLAB);
                const uint4 x2 = *reinterpret_cast<const uint4*>(p + 2 * XC_SLAB);
                // smallest terms first (the order of xcol32s_kernel)
                acc[t] = mfma32_bf16(wq[2][kk], x0, acc[t]);
                acc[t] = mfma32_bf16(wq[1][kk], x1, acc[t]);
                acc[t] = mfma32_bf16(wq[0][kk], x2, acc[t]);
                acc[t] = mfma32_bf16(wq[1][kk], x0, acc[t]);
                acc[t] = mfma32_bf16(wq[0][kk], x1, acc[t]);
                acc[t] = mfma32_bf16(wq[0][kk], x0, acc[t]);
            }
            request_w_half(wnext, kk);
        }
    };

    const bool owner = wave < nob;
    if (nsteps > 0) {
        for (int tb = 0; tb < nsteps; tb += 64) {     // lane-indexed tables for steps [tb, tb+64)
            const int idx = min(tb + lane, nsteps - 1);
            const int pv = pairs[idx];
            const int tend = min(64, nsteps - tb);
            const int w0v = (owner && lane < tend) ? wt0[idx] : -1, w1v = (owner && lane < tend) ? wt1[idx] : -1;
            const uint64_t m0 = __ballot(w0v >= 0), m1 = __ballot(w1v >= 0);     // bit s: this wave has a block in half 0 / 1 of step s
            auto next_entry = [&](int e) -> int {      // first entry >= e (e = 2 * step + half), or 128
                const int s = e >> 1;
                if (s >= 64) return 128;
                uint64_t a = m0 >> s;
                const uint64_t b = m1 >> s;
                if (e & 1) a &= ~1ull;
                const int ea = a ? 2 * (s + __builtin_ctzll(a)) : 128;
                const int eb = b ? 2 * (s + __builtin_ctzll(b)) + 1 : 128;
                return min(ea, eb);
            };
            auto entry_block = [&](int e) -> int {
                const int s = e >> 1;
                return (e & 1) ? __builtin_amdgcn_readlane(w1v, s) : __builtin_amdgcn_readlane(w0v, s);
            };
            int ne = __builtin_amdgcn_readfirstlane(next_entry(0));
            // prologue: stages 0 and 1 requested, step 0 converted (my rows; the barrier of iteration 0 makes it everyone's)
            issue_x(__builtin_amdgcn_readlane(pv, 0), 0);
            if (tend > 1) issue_x(__builtin_amdgcn_readlane(pv, 1), 1);
            if (tend > 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            convert(0);
            if (ne < 128) request_w(entry_block(ne));
            for (int s = 0; s < tend; ++s) {
                wait_all();            // my rows of stage s + 1 and my next weight pieces have landed
                __syncthreads();       // everyone converted step s and has left step s - 1
                if (s + 2 < tend) issue_x(__builtin_amdgcn_readlane(pv, s + 2), s & 1);      // (stage s: converted by me an iteration ago, reads returned)
                if (s + 1 < tend) convert((s + 1) & 1);
                const unsigned char* slot = smem + (s & 1) * XS_SLOT;
                while ((ne >> 1) == s) {
                    const int half = ne & 1;
                    ne = __builtin_amdgcn_readfirstlane(next_entry(ne + 1));
                    block(slot, half, ne < 128 ? entry_block(ne) : -1);      // (requests the next block's pieces as its own are used up)
                    if ((ne >> 1) == s) wait_all();                          // second block of the same step: needed right away
                }
            }
            wait_all();
            __syncthreads();   // the next batch re-primes the ring
        }
    }

    // Epilogue (as xcol32s_kernel, axis 1): 32 rows at a time staged as [32][2 KiB] (pieces XOR-swizzled with n), stored as full rows.
    const int rowbytes = nob * 128;
    float* ybase = Y + (size_t)ob0 * 32;
#pragma unroll
    for (int t = 0; t < XC_RT; ++t) {
        __syncthreads();
        if (owner) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int piece = wave * 8 + 2 * g + h;
                *reinterpret_cast<float4*>(smem + r * XS_ROWB + ((piece ^ r) << 4)) =
                    make_float4(acc[t][4 * g + 0], acc[t][4 * g + 1], acc[t][4 * g + 2], acc[t][4 * g + 3]);
            }
        }
        __syncthreads();
        constexpr int PPR = XS_ROWB / 16;      // 128 pieces per row
        for (int i = threadIdx.x; i < 32 * PPR; i += 64 * XS_G) {
            const int nn = i / PPR, piece = i % PPR, row = n_tile + 32 * t + nn;
            if (row < N && piece * 16 < rowbytes) {
                const float4 v = *reinterpret_cast<const float4*>(smem + nn * XS_ROWB + ((piece ^ nn) << 4));
                *reinterpret_cast<float4*>(reinterpret_cast<unsigned char*>(ybase + (size_t)row * Kout) + piece * 16) = v;
            }
        }
    }
}

}  // namespace bsmm
