// bsmm_updat16_rows.h -- row-owner weight-gradient kernel for bsize 16 on feature axis 0 (round 5; 'BSU6' section of the 'BSUP' plan,
// bsmm_plan.h::build_updat16_rows_section).  BASELINE configs[2] (4096^2, bsize 16, 10 %, feature axis 0, bf16).
//
// Why.  The windowed kernel (bsmm_updat_win.h) stages the X rows AND the DY rows of a 256 x 256-feature window in LDS: 64 KiB per 64-wide
// minibatch chunk, two slots -- the window cannot grow, and every window streams 2 x 256 rows x N: 2.1 GB through the L2 -> LDS path at
// 4096^2, N = 8192, which is what its 117-122 us are (both weight-gradient kernels run at the rate their slab traffic allows).
// On feature axis 0 the minibatch index is contiguous in both operands, so an MFMA operand fragment (16 rows x 8 consecutive minibatch
// entries per lane quarter) IS 16 contiguous bytes of a row: only rows that several waves need have to go through LDS.  Here a wave OWNS
// block rows of the window -- it loads their X fragments straight into registers, once per chunk, one chunk ahead -- and multiplies them
// with every block of those rows; only the DY rows are shared, by LDS-DMA.  LDS then holds 512 DY rows x 128 B = 64 KiB per chunk and the
// window is 512 x 512 features: half the bytes per block (1.07 GB at the shape above).
//
//   workgroup = 16 waves, one window x one part of the minibatch (grid = items x split, linear id L: part = L % split, item = L / split:
//               the parts of a window run on neighbouring XCDs, consecutive items = one row of windows = the same X rows);
//               a wave owns at most 2 block rows and 12 blocks (the plan deals the rows, longest first onto the lightest wave);
//   per chunk:  wait for everything the wave asked for (its X pieces, its part of this chunk's slab), barrier (so has every other
//               wave's part), X pieces into operand order (they are loaded 4 rows x 64 contiguous bytes per quarter wave -- a quarter of the
//               requests of loading in operand order -- and permuted with ds_bpermute), request the NEXT chunk (X pieces into the freed
//               registers, then the wave's DMA duty of the next slab: 1 KiB instructions, rows XOR-swizzled by 16-byte piece as in the
//               windowed kernel), then per owned block two v_mfma_f32_16x16x32 (K = the chunk's 64 entries): A = the wave's X fragment of
//               the block's row, B = ds_read_b128 of the DY rows of the block's column, four blocks' reads in flight; barrier (the slot
//               is free);
//   epilogue:   the 16 x 16 sums of a block go to DW (split = 1: alpha / beta here, one rounding) or into the part's own fp32 image in the
//               workspace, and updat16_rows_finalize_kernel adds the images and rounds once (deterministic: no atomics -- fp32 atomics into
//               one image cost 6 us per part at BASELINE configs[2]).
// N % 8 == 0 and 16-byte aligned operands (row pieces of 16 bytes); a ragged last chunk re-reads the last 8 entries and zeroes the X pieces.
//
// Measured (profiles/r05_updat16_rows_*.txt): BASELINE configs[2] 88-92 us against 115-125 for the windowed kernel; 4096^2 20 % 138 against 241;
// 8192^2 5 % 292 against 481.  Cycle stamps: the data of a chunk (128 KiB per CU) takes 1.8 us = 30 B/clk/CU, what every L2 -> CU stream of
// this chip gets, and a step takes 2.2 us: a wave STALLS AT ISSUE while the memory pipeline is full (the wait itself is 40-80 cycles), so
// the part of a step in which the waves request cannot overlap the part in which the same waves multiply; with two slab slots (2 x 64 KiB
// of the 160) there is no second chunk to work on meanwhile.  Who requests the slab (every wave 4 instructions, or the waves with few
// blocks all of them: U6_DMA_WEIGHT) makes no difference: 97.5-99.1 us.
#pragma once
#include "bsmm_common.h"
#include "bsmm_plan.h"

// ablation switches of experiment builds (scripts/build_variants.py): 0 in the product
#ifndef U6_NO_MATH
#define U6_NO_MATH 0
#endif
#ifndef U6_NO_X
#define U6_NO_X 0
#endif
#ifndef U6_NO_DMA
#define U6_NO_DMA 0
#endif
#ifndef U6_NO_BREAD
#define U6_NO_BREAD 0
#endif
#ifndef U6_NO_BPERM
#define U6_NO_BPERM 0
#endif

namespace bsmm {

#ifdef U6_STAMPS
// experiment builds: per wave of the first 64 workgroups, cycles spent in [0] requesting the next chunk, [1] the counted wait, [2] first
// barrier, [3] the matrix work, [4] second barrier, [5] epilogue, [6] whole kernel, [7] chunks; bsmm_debug_u6_trace_copy()
__device__ unsigned long long g_u6_trace[64 * 16 * 8];
#define U6_STAMP(k) { const unsigned long long t_ = __builtin_readcyclecounter(); tacc[k] += t_ - tlast; tlast = t_; }
#else
#define U6_STAMP(k)
#endif

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int WK> struct U6Geom {
    static constexpr int SLAB = WK * 16 * 128;            // DY rows of the window x 128 bytes (64 minibatch entries)
    static constexpr int LDS = 2 * SLAB;
};

template <class DT, int WK>
__global__ void __launch_bounds__(64 * U6_WAVES, 1)
updat16_rows_kernel(PtrList8 Xs, PtrList8 Es, typename DT::T* __restrict__ DW, float* __restrict__ scratch, const int32_t* __restrict__ sec,
                    int N, int Cf, int Kf, int pcount, float alpha, float beta, int split, size_t nel) {
    typedef typename DT::T T;
    static_assert(DT::is16, "16-bit storage types");
    typedef U6Geom<WK> G;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if (sec[0] != U6PLAN_MAGIC || sec[1] != U6PLAN_VERSION || sec[2] != U6_WC || sec[3] != WK ||
        sec[5] != (U6_WAVES | (U6_ROWS << 8) | (U6_MAXB << 16)) || sec[6] != U6_ITEM) return;
    // workgroup -> (part of the minibatch, item).  Workgroup L runs on XCD L % 8 (observed; speed only).  split = 1 / 2 / 4 / 8: XCD x takes
    // part x / (8 / split) and nothing else, and of that part every (8 / split)-th run of items: an XCD's L2 sees one part of X and DY, and the
    // 8 / split XCDs of a part divide the windows instead of all reading all of them (the launcher rounds the grid up to whole rounds of 8).
    // Other splits: parts interleaved.
    int part, it;
    if (split == 1 || split == 2 || split == 4 || split == 8) {
        const int per = 8 / split, x = blockIdx.x & 7;
        part = x / per;
        it = (blockIdx.x >> 3) * per + x % per;
    } else {
        part = blockIdx.x % split; it = blockIdx.x / split;
    }
    if (it >= sec[4]) return;
    const int32_t* item = sec + U6_HDR + (size_t)it * U6_ITEM;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c0 = item[0], k0 = item[1];
    const int32_t* wv = item + 4 + wave * U6_WAVE;
    const uint32_t rowsw = (uint32_t)wv[0];
    const int nb = wv[1] & 255, ii0 = (wv[1] >> 8) & 255, ni = (wv[1] >> 16) & 255;       // blocks; DMA duty: slab instructions ii0 .. ii0 + ni - 1
    uint32_t mw[U6_MAXB / 2];                     // (column | row slot << 8) of the wave's blocks, 16 bits each: stays in scalar registers
#pragma unroll
    for (int i = 0; i < U6_MAXB / 2; ++i) mw[i] = (uint32_t)wv[2 + i];
    // blocks [0, e0) are of row slot 0, [e0, nb) of slot 1 (sorted by slot; e0 = nb when the wave has one row); the LDS byte offset of
    // every block's DY rows in scalar registers
    static_assert(U6_ROWS == 2, "one switch of the A fragment per K-step");
    int e0 = nb;
    uint32_t koff[U6_MAXB];
#pragma unroll
    for (int jj = U6_MAXB - 1; jj >= 0; --jj) {
        const uint32_t m = (mw[jj >> 1] >> (16 * (jj & 1))) & 0xffffu;
        koff[jj] = (m & 255u) * 2048u;
        if (jj < nb && (m >> 8) >= 1) e0 = jj;
    }
    const int nchunks = (N + 63) >> 6;
    const int per = (nchunks + split - 1) / split;
    const int q_beg = part * per, q_end = min(nchunks, q_beg + per);
    const uint32_t base_addr = lds_addr_of(smem);
    const int f = lane & 15, q = lane >> 4;

    // DY slab: DMA instruction ii (of 2 WK) covers slab rows 8 ii .. 8 ii + 7 (lane: row 8 ii + lane / 8; position lane % 8 holds piece
    // pos ^ ((row >> 1) & 7) = pos ^ (lane >> 4) ^ 4 (ii & 1)).  Offsets in elements, 32 bits (the launcher checks N * max(C, K) < 2^31).
    const int erow0 = k0 * 16 + (lane >> 3);
    const int ecol0 = ((lane & 7) ^ (lane >> 4)) * 8;
    // X fragments of the wave's own block rows.  The MFMA operand wants lane (f, q) to hold entries 32 ks + 8 q .. + 7 of feature row f --
    // loaded that way a quarter wave touches 16 different rows with 16 bytes each.  Loaded with lane l on row l / 4, piece l % 4 instead (a
    // quarter wave = 4 rows x 64 contiguous bytes: a quarter of the requests), and put in operand order by one ds_bpermute per dword
    // (lane (f, q) pulls from lane 4 f + q) when the chunk is used.  The loads come from inline asm: hipcc's own wait counts for a tracked
    // load would also drain the LDS-DMA issued next to it; every wait here is the loop's vmcnt(0).
    uint32_t xrow[U6_ROWS];
#pragma unroll
    for (int r = 0; r < U6_ROWS; ++r) {
        const int rb = (rowsw >> (8 * r)) & 255;
        xrow[r] = (uint32_t)min((c0 + (rb != 255 ? rb : 0)) * 16 + (lane >> 2), Cf - 1) * (uint32_t)N;      // (a slot without a row re-reads the window's first)
    }
    const int xpiece = 8 * (lane & 3);
    const int pull = (4 * f + q) << 2;            // ds_bpermute address: byte offset of the source lane
    const int fsw = (f >> 1) & 7;                 // piece swizzle of slab row 16 kidx + f
    const int boff = f * 128;

    f32x4 acc[U6_MAXB];
#pragma unroll
    for (int j = 0; j < U6_MAXB; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#ifdef U6_STAMPS
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_readcyclecounter();
    const unsigned long long tstart = tlast;
#endif

    for (int p = 0; p < pcount; ++p) {
        const T* X = static_cast<const T*>(Xs.p[p]);
        const T* E = static_cast<const T*>(Es.p[p]);
        if (q_beg >= q_end) break;
        auto issue = [&](int qq, int pos) {
            const int n0 = qq * 64;
            const uint32_t slot = base_addr + pos * G::SLAB;
            if (U6_NO_DMA) return;
            for (int ii = ii0; ii < ii0 + ni; ++ii) {
                const uint32_t off = (uint32_t)min(erow0 + 8 * ii, Kf - 1) * (uint32_t)N + (uint32_t)min(n0 + (ecol0 ^ (32 * (ii & 1))), N - 8);
                glds16_asm(E + off, __builtin_amdgcn_readfirstlane(slot + ii * 1024));
            }
        };
        // X pieces as loaded, [K-step][row slot]: requested for the NEXT chunk as soon as this chunk's have been put in operand order
        u32x4 raw[2][U6_ROWS];
        auto load_x = [&](int qq) {
            const int n0 = qq * 64;
            if (U6_NO_X) return;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int r = 0; r < U6_ROWS; ++r)
                    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(raw[ks][r]) : "v"(X + (xrow[r] + (uint32_t)min(n0 + 32 * ks + xpiece, N - 8))) : "memory");
        };
        // operand order (lane (f, q) pulls from lane 4 f + q), all 16 dwords of the chunk in one batch; zero past N in a ragged last chunk
        // (N % 8 == 0: whole pieces)
        uint4 x[2][U6_ROWS];
        auto shuffle = [&](int qq) {
            const int n0 = qq * 64;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const bool live = n0 + 32 * ks + 8 * q < N;
#pragma unroll
                for (int r = 0; r < U6_ROWS; ++r) {
                    u32x4 v = raw[ks][r];
                    if (U6_NO_X) v = u32x4(0x3c003c00u);
                    uint4 o;
                    if (U6_NO_BPERM) o = make_uint4(v[0], v[1], v[2], v[3]);
                    else {
                        o.x = (uint32_t)__builtin_amdgcn_ds_bpermute(pull, (int)v[0]);
                        o.y = (uint32_t)__builtin_amdgcn_ds_bpermute(pull, (int)v[1]);
                        o.z = (uint32_t)__builtin_amdgcn_ds_bpermute(pull, (int)v[2]);
                        o.w = (uint32_t)__builtin_amdgcn_ds_bpermute(pull, (int)v[3]);
                    }
                    x[ks][r] = live ? o : zero_u4();
                }
            }
            // (the next X loads overwrite raw: they must not pass the permutes that read it)
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x[0][0].x), "+v"(x[0][1].x), "+v"(x[1][0].x), "+v"(x[1][1].x) :: "memory");
        };
        // a chunk: per K-step the blocks in groups of four -- four B fragments requested together, then four MFMAs.  The A fragment is the
        // wave's row slot of the block: blocks are sorted by slot, so `a` changes once per K-step (at e0: a real branch -- the empty asm
        // keeps hipcc from turning it into selects per block).  Slots past nb multiply leftovers into accumulators nobody stores.
        auto compute = [&](int pos) {
            const unsigned char* slot = smem + pos * G::SLAB;
            if (U6_NO_MATH) {
                if (U6_NO_MATH == 2) { __builtin_amdgcn_s_sleep(24); __builtin_amdgcn_s_sleep(24); }      // ~3000 cycles asleep instead of the matrix work
                if (x[0][0].x == 0x12345u && x[1][1].y == 77u) acc[0][0] += 1.f;
                return;
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const unsigned char* sl = slot + boff + (((4 * ks + q) ^ fsw) << 4);
                uint4 a = x[ks][0];
#pragma unroll
                for (int g = 0; g < U6_MAXB / 4; ++g) {
                    if (4 * g < nb) {
                        uint4 b[4];
#pragma unroll
                        for (int t = 0; t < 4; ++t) b[t] = U6_NO_BREAD ? make_uint4(koff[4 * g + t], 1u, 2u, 3u) : *reinterpret_cast<const uint4*>(sl + koff[4 * g + t]);
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            const int jj = 4 * g + t;
                            if (__builtin_expect(jj == e0, 0)) { a = x[ks][1]; asm volatile("" : "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(a.w)); }
                            acc[jj] = DT::mfma16(a, b[t], acc[jj]);
                        }
                    }
                }
            }
        };
        // One chunk of the ring.  Everything the wave asked for has landed (its X pieces, its part of this chunk's slab); barrier: so has
        // every other wave's part.  The X pieces go into operand order (one batch of permutes), the registers they came in are handed to
        // the next chunk's loads, the wave's DMA duty for the next slab follows (the other slot is free since the barrier that ended the
        // previous step), then the blocks.  A wave STALLS AT ISSUE while the memory pipeline is full
        // (profiles/r05_updat16_rows_stamps.txt: a chunk's requests are accepted about as fast as the previous chunk's data arrives),
        // which is why the plan gives the DMA duty to the waves with few blocks: they stall while the others multiply.
        auto step = [&](int qq, int pos) {
#ifdef U6_STAMPS
            tlast = __builtin_readcyclecounter(); tacc[7] += 1;
#endif
            // (the registers ride through the asm: nothing that reads them may be scheduled above the wait)
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(raw[0][0]), "+v"(raw[0][1]), "+v"(raw[1][0]), "+v"(raw[1][1]) :: "memory");
            U6_STAMP(1)
            __syncthreads();
            U6_STAMP(2)
            shuffle(qq);
            if (qq + 1 < q_end) { load_x(qq + 1); issue(qq + 1, pos ^ 1); }
            U6_STAMP(0)
            compute(pos);
            U6_STAMP(3)
            __syncthreads();
            U6_STAMP(4)
        };
        load_x(q_beg);
        issue(q_beg, 0);
        int pos = 0;
        for (int qq = q_beg; qq < q_end; ++qq) {
            step(qq, pos);
            pos ^= 1;
        }
    }

    // D[ci][ko]: col = ko = lane & 15, row ci = 4 * (lane >> 4) + reg.  split = 1: DW here.  Otherwise this part's sums of the block into
    // ITS image of the workspace ([part][block][256] floats, plain stores -- every (part, block) has exactly one writer, a part without
    // chunks writes zeros); updat16_rows_finalize_kernel adds the images in part order.  (fp32 atomics into one image, as the windowed kernel
    // does: 6 us per part at BASELINE configs[2], 31 us at split 4.)
#pragma unroll
    for (int j = 0; j < U6_MAXB; ++j) {
        if (j >= nb) continue;
        const int wid = wv[2 + U6_MAXB / 2 + j];
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const size_t idx = (size_t)wid * 256 + (4 * q + reg) * 16 + f;
            if (scratch == nullptr) {
                float out = alpha * acc[j][reg];
                if (beta != 0.f) out += beta * DT::to_f32(DW[idx]);
                DW[idx] = DT::from_f32(out);
            } else {
                scratch[(size_t)part * nel + idx] = acc[j][reg];
            }
        }
    }
#ifdef U6_STAMPS
    U6_STAMP(5)
    tacc[6] = __builtin_readcyclecounter() - tstart;
    if (blockIdx.x < 64 && lane == 0)
        for (int k = 0; k < 8; ++k) g_u6_trace[(blockIdx.x * 16 + wave) * 8 + k] = tacc[k];
#endif
}

// DW = alpha * [gate *] (sum over the parts' images) + beta * DW, rounded once (fp32 DW: the fp32 call through six bf16 piece pairs,
// bsmm_api.hip::updat16_f32_rows -- skip_if is its non-finite flag: the repair pass writes DW then)
template <class DT>
__global__ void __launch_bounds__(256)
updat16_rows_finalize_kernel(const float* __restrict__ scratch, typename DT::T* __restrict__ DW, size_t nel, int split, float alpha, float beta,
                             const float* __restrict__ gate = nullptr, const int32_t* __restrict__ skip_if = nullptr) {
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= nel || (skip_if && skip_if[0] != 0)) return;
    float4 s = *reinterpret_cast<const float4*>(scratch + i);
    for (int part = 1; part < split; ++part) {
        const float4 t = *reinterpret_cast<const float4*>(scratch + (size_t)part * nel + i);
        s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
    }
    const float a = gate ? alpha * gate[i >> 8] : alpha;
    float v[4] = {a * s.x, a * s.y, a * s.z, a * s.w};
    if (beta != 0.f) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += beta * DT::to_f32(DW[i + e]);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) DW[i + e] = DT::from_f32(v[e]);
}

}  // namespace bsmm
