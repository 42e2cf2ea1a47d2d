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
//               with split = 8 an XCD reads ONE eighth of the minibatch of X and DY -- every re-read of a row by the windows of that
//               XCD's workgroups stays in its L2, consecutive items = one row of windows = the same X rows);
//   per chunk:  request the NEXT chunk (the wave's X pieces + 4 x 1 KiB DMA instructions of the slab, rows XOR-swizzled by 16-byte piece as
//               in the windowed kernel), wait for THIS one (counted vmcnt: the queue never drains), barrier, then per owned block two
//               v_mfma_f32_16x16x32 (K = the chunk's 64 entries): A = the wave's X fragment of the block's row, B = ds_read_b128 of the DY
//               rows of the block's column; barrier (the slot is free);
//   epilogue:   the 16 x 16 sums of a block go to DW (split = 1: alpha / beta here, one rounding) or into the part's own fp32 image in the
//               workspace, and updat16_rows_finalize_kernel adds the images and rounds once (deterministic: no atomics).
// N % 8 == 0 and 16-byte aligned operands (row pieces of 16 bytes); a ragged last chunk re-reads the last 8 entries and zeroes both fragments.
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
    static constexpr int NI = SLAB / 1024 / U6_WAVES;     // DMA instructions per wave and slab
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
    const int part = blockIdx.x % split, it = blockIdx.x / split;
    if (it >= sec[4]) return;
    const int32_t* item = sec + U6_HDR + (size_t)it * U6_ITEM;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c0 = item[0], k0 = item[1];
    const int32_t* wv = item + 4 + wave * U6_WAVE;
    const uint32_t rowsw = (uint32_t)wv[0];
    const int nb = wv[1];
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

    // DY slab: DMA instruction ii = NI * wave + i covers slab rows 8 ii .. 8 ii + 7 (lane: row 8 ii + lane / 8; position lane % 8 holds piece
    // pos ^ ((row >> 1) & 7) = pos ^ (lane >> 4) ^ 4 (i & 1): NI is even).  Offsets in elements, 32 bits (the launcher checks N * max(C, K) < 2^31).
    static_assert(G::NI % 2 == 0, "piece swizzle by the parity of i");
    uint32_t erow[G::NI];
#pragma unroll
    for (int i = 0; i < G::NI; ++i) erow[i] = (uint32_t)min(k0 * 16 + 8 * (G::NI * wave + i) + (lane >> 3), Kf - 1) * (uint32_t)N;
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
#pragma unroll
            for (int i = 0; i < G::NI; ++i) {
                const uint32_t dst = __builtin_amdgcn_readfirstlane(slot + (G::NI * wave + i) * 1024);
                glds16_asm(E + (erow[i] + (uint32_t)min(n0 + (ecol0 ^ (32 * (i & 1))), N - 8)), dst);
            }
        };
        // X pieces as loaded, [K-step][row slot], two sets: one being used, one landing
        u32x4 rawA[2][U6_ROWS], rawB[2][U6_ROWS];
        auto load_x = [&](int qq, u32x4 (&raw)[2][U6_ROWS]) {
            const int n0 = qq * 64;
            if (U6_NO_X) return;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int r = 0; r < U6_ROWS; ++r)
                    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(raw[ks][r]) : "v"(X + (xrow[r] + (uint32_t)min(n0 + 32 * ks + xpiece, N - 8))) : "memory");
        };
        // operand order (lane (f, q) pulls from lane 4 f + q); zero past N in a ragged last chunk (N % 8 == 0: whole pieces)
        auto operand = [&](u32x4 v, bool live) {
            if (U6_NO_X) v = u32x4(0x3c003c00u);
            if (U6_NO_BPERM) return live ? make_uint4(v[0], v[1], v[2], v[3]) : zero_u4();
            uint4 o;
            o.x = (uint32_t)__builtin_amdgcn_ds_bpermute(pull, (int)v[0]);
            o.y = (uint32_t)__builtin_amdgcn_ds_bpermute(pull, (int)v[1]);
            o.z = (uint32_t)__builtin_amdgcn_ds_bpermute(pull, (int)v[2]);
            o.w = (uint32_t)__builtin_amdgcn_ds_bpermute(pull, (int)v[3]);
            return live ? o : zero_u4();
        };
        // a chunk: per K-step the blocks in groups of four -- four B fragments requested together, then four MFMAs.  The A fragment is the
        // wave's row slot of the block: blocks are sorted by slot, so `a` changes once per K-step (at e0: a real branch -- the empty asm
        // keeps hipcc from turning it into selects per block).  Slots past nb multiply leftovers into accumulators nobody stores.
        auto compute = [&](int qq, int pos, u32x4 (&raw)[2][U6_ROWS]) {
            const unsigned char* slot = smem + pos * G::SLAB;
            const int n0 = qq * 64;
            if (U6_NO_MATH) {
                if (U6_NO_MATH == 2) { __builtin_amdgcn_s_sleep(24); __builtin_amdgcn_s_sleep(24); }      // ~3000 cycles asleep instead of the matrix work
                if (raw[0][0][0] == 0x12345u && raw[1][1][1] == 77u) acc[0][0] += 1.f;
                return;
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const unsigned char* sl = slot + boff + (((4 * ks + q) ^ fsw) << 4);
                const bool live = n0 + 32 * ks + 8 * q < N;
                uint4 a = operand(raw[ks][0], live);
#pragma unroll
                for (int g = 0; g < U6_MAXB / 4; ++g) {
                    if (4 * g < nb) {
                        uint4 b[4];
#pragma unroll
                        for (int t = 0; t < 4; ++t) b[t] = U6_NO_BREAD ? make_uint4(koff[4 * g + t], 1u, 2u, 3u) : *reinterpret_cast<const uint4*>(sl + koff[4 * g + t]);
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            const int jj = 4 * g + t;
                            if (__builtin_expect(jj == e0, 0)) { a = operand(raw[ks][1], live); asm volatile("" : "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(a.w)); }
                            acc[jj] = DT::mfma16(a, b[t], acc[jj]);
                        }
                    }
                }
            }
        };
        // One chunk of the ring.  On entry every wave is past the previous chunk's MFMAs (barrier at the end of step): the other slot is free,
        // so the NEXT chunk is requested first -- X pieces into the other register set, then the wave's part of the slab -- and only then
        // does the wave wait for THIS chunk: everything but the 2 U6_ROWS + NI requests just made (the memory queue never drains).  Second
        // barrier: every wave's part of this chunk's slab is in LDS.
        auto step = [&](int qq, int pos, u32x4 (&cur)[2][U6_ROWS], u32x4 (&nxt)[2][U6_ROWS]) {
#ifdef U6_STAMPS
            tlast = __builtin_readcyclecounter(); tacc[7] += 1;
#endif
            if (qq + 1 < q_end) {
                load_x(qq + 1, nxt);
                issue(qq + 1, pos ^ 1);
                U6_STAMP(0)
                constexpr int INFLIGHT = (U6_NO_X ? 0 : 2 * U6_ROWS) + (U6_NO_DMA ? 0 : G::NI);
                asm volatile("s_waitcnt vmcnt(%0)" :: "n"(INFLIGHT) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            // (the registers ride through an asm: nothing that reads them may be scheduled above the wait)
            asm volatile("" : "+v"(cur[0][0]), "+v"(cur[0][1]), "+v"(cur[1][0]), "+v"(cur[1][1]) :: "memory");
            U6_STAMP(1)
            __syncthreads();
            U6_STAMP(2)
            compute(qq, pos, cur);
            U6_STAMP(3)
            __syncthreads();
            U6_STAMP(4)
        };
        load_x(q_beg, rawA);
        issue(q_beg, 0);
        for (int qq = q_beg; qq < q_end; qq += 2) {
            step(qq, 0, rawA, rawB);
            if (qq + 1 < q_end) step(qq + 1, 1, rawB, rawA);
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

// DW = alpha * (sum over the parts' images) + beta * DW, rounded once
template <class DT>
__global__ void __launch_bounds__(256)
updat16_rows_finalize_kernel(const float* __restrict__ scratch, typename DT::T* __restrict__ DW, size_t nel, int split, float alpha, float beta) {
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= nel) return;
    float4 s = *reinterpret_cast<const float4*>(scratch + i);
    for (int part = 1; part < split; ++part) {
        const float4 t = *reinterpret_cast<const float4*>(scratch + (size_t)part * nel + i);
        s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
    }
    float v[4] = {alpha * s.x, alpha * s.y, alpha * s.z, alpha * s.w};
    if (beta != 0.f) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += beta * DT::to_f32(DW[i + e]);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) DW[i + e] = DT::from_f32(v[e]);
}

}  // namespace bsmm
