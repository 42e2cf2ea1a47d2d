// bsmm_xcol_v2.h -- xprop kernel with activations AND weights staged through LDS by a static software pipeline ('BSX2'
// plans), feature_axis = 1, bsize 32, 16-bit storage types.
//
// What bounded bsmm_xcol.h (99 us per pass at the bench shape against ~22 us of matrix work): a wave fetched its weight
// fragments into registers with ordinary loads one step ahead, and the vector-memory counter is in order -- so the
// `s_waitcnt vmcnt(0)` in front of every step also waited for the activation slabs of the NEXT phase that had just been
// requested; a phase was one memory round trip long whatever the matrix work.  The first staged version (weights by
// LDS-DMA too, two ring halves, ONE vmcnt(0) per phase of two steps) reached 81 us and its ablation builds showed why
// (profiles/r02_xcol_v2_ablation.md): the DMAs alone took 68 us -- a request burst, then a full drain, 0.5 us of miss latency
// per phase with nothing in flight behind it -- and the matrix work alone 63 us, a phase lasting as long as its busiest SIMD.
// Hence this shape:
//   * workgroup = 16 output blocks x 128 minibatch rows, 16 waves; wave 4c + t owns row tile t of output blocks 4c .. 4c+3
//     (4 x 16 accumulators).  The four waves of a class c do identical work on the four SIMDs: balanced by construction.
//   * the kernel runs ROWS of the plan (bsmm_plan.h).  In row r every wave requests one 1 KiB piece of the activation slab
//     that row r+3 multiplies (ring of four 16 KiB slabs) and NW half weight blocks for later rows (circular pool of 47
//     slots of 2 KiB; which block, when and where is decided on the host) -- always 1 + NW DMA instructions, so
//     `s_waitcnt vmcnt(2 * (1 + NW))` + barrier at the top of a row means "everything requested three rows ago is in LDS":
//     three rows of requests are always in flight and nothing ever drains.
//   * per row and wave: per half of the pair with a block in the class 2 ds_read_b128 of the activation fragment and, per
//     block, 2 ds_read_b128 + 2 MFMAs.
//   activation slab: rows of 128 B, the eight 16-byte pieces of row r XOR-swizzled with (r >> 1) & 7;
//   weight block:   rows of 64 B, the four pieces of row r XOR-swizzled with (r >> 2) & 3 (both conflict-free for b128).
// Pool safety (why the host schedule cannot deadlock): slots are handed out in request order and a class's run of <= 8
// blocks never wraps, so the previous occupant of a slot is >= 40 blocks older; a step has <= 32 blocks, hence that occupant
// belongs to an EARLIER step, all of whose requests precede in the queue: it gets scheduled, runs, and frees the slot.
#pragma once
#include "bsmm_common.h"
#include "bsmm_plan.h"
#include "bsmm_updat_v2.h"   // uniform_ptr
#include "bsmm_xprop.h"      // XMap

namespace bsmm {

// measurement switches (ablation builds, scripts/build_variants.py): wrong results by construction
#ifndef X2_NO_XDMA
#define X2_NO_XDMA 0
#endif
#ifndef X2_NO_WDMA
#define X2_NO_WDMA 0
#endif
#ifndef X2_NO_READS
#define X2_NO_READS 0
#endif
#ifndef X2_NO_MFMA
#define X2_NO_MFMA 0
#endif
#ifndef X2_LATE
#define X2_LATE 0            // waves 8..15 request AFTER their blocks: while one half of the waves waits for the memory pipe to
#endif                       // accept its DMAs the other half multiplies
#ifndef X2_BATCH_W
#define X2_BATCH_W 1         // all weight fragments of a half are read before its first MFMA (one LDS round trip per half)
#endif
#ifndef X2_CHEAP_DUMMY
#define X2_CHEAP_DUMMY 1     // requests with nothing to fetch read one 16-byte piece with all lanes (64 B from L2, not 1 KiB)
#endif
#ifndef X2_NO_EPILOGUE
#define X2_NO_EPILOGUE 0
#endif
#ifdef BSMM_XC_TRACE
// cycle stamps of the first 8 workgroups: [wg][wave][row][5] = before the wait, after it, after the barrier, after the requests,
// after the blocks; read back with bsmm_debug_trace_copy2().  Debug builds only.
__device__ unsigned long long g_x2_trace[8 * 16 * 48 * 5];
#define X2_STAMP(k) do { if (blockIdx.x < 8 && row < 48 && lane == 0) g_x2_trace[((blockIdx.x * 16 + wave) * 48 + row) * 5 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define X2_STAMP(k) do { } while (0)
#endif
constexpr int X2_R = 128;                          // minibatch rows per workgroup
constexpr int X2_SLAB = X2_R * 128;                // 16 KiB: one pair step of activations
constexpr int X2_WBASE = 4 * X2_SLAB;              // weight pool behind the ring of four slabs
constexpr int X2_LDS = X2_WBASE + (X2_POOL + 1) * 2048;
static_assert(X2_LDS <= 163840 && X2_R * X2_G * 64 <= X2_LDS && X2_AHEAD == 3, "ring, pool and epilogue tile must fit the LDS");

// one LDS-DMA of 1 KiB: scalar base + 32-bit scalar byte offset (added here), per-lane byte offset, LDS destination.
// M0 is left holding the destination: the kernel below uses M0 for nothing else (v_readlane takes SGPR selectors).
__device__ __forceinline__ void x2_dma(const void* sbase, uint32_t soff, uint32_t voff, uint32_t lds_byte_addr) {
    const unsigned char* p = static_cast<const unsigned char*>(sbase) + soff;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(p), "s"(lds_byte_addr)
                 : "memory");
}

template <class DT, int NW>
__global__ void __launch_bounds__(64 * X2_G, 4)
xcol32_a1_v2_kernel(const typename DT::T* __restrict__ X, const typename DT::T* __restrict__ Wsel,
                    typename DT::T* __restrict__ Y, const int32_t* __restrict__ plan, XMap map, int N, int Cin, int Kout) {
    typedef typename DT::T T;
    static_assert(DT::is16 && NW >= 1 && NW <= 4, "xcol v2 kernel: 16-bit storage types");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int tile, grp;
    if (!xmap_decode(map, blockIdx.x, tile, grp)) return;
    const int4 gh = *reinterpret_cast<const int4*>(plan + plan[5] + 4 * grp);
    const int row_off = __builtin_amdgcn_readfirstlane(gh.x), nrows = __builtin_amdgcn_readfirstlane(gh.y);
    const int ob0 = __builtin_amdgcn_readfirstlane(gh.z), nob = __builtin_amdgcn_readfirstlane(gh.w);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tw = wave & 3, cls = wave >> 2;      // my row tile, my class of four output blocks
    const int32_t* pxt = plan + plan[6] + row_off;
    const int32_t* cwt = plan + plan[7] + (size_t)row_off * 4 + cls;
    const int32_t* dut = plan + plan[10] + ((size_t)row_off * X2_G + wave) * (2 * NW);
    const int r = lane & 31, h = lane >> 5;
    const int n_tile = tile * X2_R;
    const uint32_t base_addr = lds_addr_of(smem);
    const int npairs_full = Cin / 64;

    // activation DMA: a slab is 16 instructions of 1 KiB (8 rows of 128 B); wave v issues instruction v of each slab
    uint32_t xvoff, xvoff_tail;
    {
        const int row = 8 * wave + (lane >> 3);
        const int xr = min(n_tile + row, N - 1) - n_tile;            // rows past N are clamped (never stored)
        const int piece = (lane & 7) ^ ((row >> 1) & 7);
        xvoff = (uint32_t)xr * (uint32_t)Cin * 2u + piece * 16;
        xvoff_tail = xvoff - ((piece & 4) ? 64 : 0);                  // last pair of an odd block count: re-read its even half
    }
    const unsigned char* xtile = static_cast<const unsigned char*>(uniform_ptr(reinterpret_cast<const unsigned char*>(X) + (size_t)n_tile * Cin * 2));
    const unsigned char* wsel = static_cast<const unsigned char*>(uniform_ptr(Wsel));
    // weight DMA: lane i of an instruction writes piece i of a 1 KiB half block (rows 16*hb + (i >> 2)); it fetches the
    // piece that the swizzle puts there: (i & 3) ^ ((row >> 2) & 3) = (i & 3) ^ ((i >> 4) & 3)
    const uint32_t wvoff = (uint32_t)((lane >> 2) * 64 + (((lane & 3) ^ ((lane >> 4) & 3)) << 4));
    const uint32_t xdst0 = base_addr + wave * 1024, wdst0 = base_addr + X2_WBASE;

    // fragment read offsets
    const int xsw = (r >> 1) & 7;
    uint32_t xrd[2][2];      // [half][kk], my 32-row band of slab 0
#pragma unroll
    for (int half = 0; half < 2; ++half)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) xrd[half][kk] = tw * 4096 + r * 128 + (((4 * half + 2 * kk + h) ^ xsw) << 4);
    uint32_t wrd[2];         // [kk], slot 0 of the pool
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) wrd[kk] = X2_WBASE + r * 64 + (((2 * kk + h) ^ ((r >> 2) & 3)) << 4);

    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;

    // requests of one row: my piece of the slab of pair p_ into ring slab (row_ & 3), my NW half weight blocks
#define X2_ISSUE_X(p_, row_)                                                                                                \
    if (!X2_NO_XDMA) x2_dma(xtile, (uint32_t)(p_) * 128u, (p_) < npairs_full ? xvoff : xvoff_tail, xdst0 + (((row_) & 3) << 14))
#define X2_ISSUE_W(src_, dst_)                                                                                              \
    if (!X2_NO_WDMA) x2_dma(wsel, (uint32_t)(src_), (X2_CHEAP_DUMMY && (dst_) == X2_POOL * 2048) ? 0u : wvoff, wdst0 + (uint32_t)(dst_))

    // the blocks of my class at one half of the pair: mask m over its four columns, `wslot` = byte offset of the next weight
    // slot.  One activation fragment serves all of them.
#define X2_COL(j_)                                                                                                          \
    if (m & (1u << (j_))) {                                                                                                 \
        uint4 wq[2];                                                                                                        \
        _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) wq[kk] = *reinterpret_cast<const uint4*>(smem + wrd[kk] + wslot); \
        wslot += 2048;                                                                                                      \
        _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) {                                                                  \
            if (X2_NO_MFMA) asm volatile("" ::"v"(wq[kk].x), "v"(wq[kk].w), "v"(xf[kk].x), "v"(xf[kk].w));                  \
            else acc[j_] = DT::mfma32(wq[kk], xf[kk], acc[j_]);                                                             \
        }                                                                                                                   \
    }
#define X2_RD(j_)                                                                                                           \
    if (m & (1u << (j_))) {                                                                                                 \
        _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) wb[(j_) & 1][kk] = *reinterpret_cast<const uint4*>(smem + wrd[kk] + wslot); \
        wslot += 2048;                                                                                                      \
    }
#define X2_MM(j_)                                                                                                           \
    if (m & (1u << (j_))) {                                                                                                 \
        _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) {                                                                  \
            if (X2_NO_MFMA) asm volatile("" ::"v"(wb[(j_) & 1][kk].x), "v"(wb[(j_) & 1][kk].w), "v"(xf[kk].x), "v"(xf[kk].w)); \
            else acc[j_] = DT::mfma32(wb[(j_) & 1][kk], xf[kk], acc[j_]);                                                   \
        }                                                                                                                   \
    }
    // X2_BATCH_W: the weight fragment of column j+1 is requested before the MFMAs of column j (two register sets, chosen by
    // the column's parity): one LDS round trip per half instead of one per block on the wave's critical path
#define X2_HALF(half_)                                                                                                      \
    do {                                                                                                                    \
        const uint32_t m = (cw >> (8 + 4 * (half_))) & 15u;                                                                 \
        if (m != 0 && !X2_NO_READS) {                                                                                       \
            uint4 xf[2];                                                                                                    \
            _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                                                \
                xf[kk] = *reinterpret_cast<const uint4*>(smem + xrd[half_][kk] + xo);                                       \
            if (X2_BATCH_W) {                                                                                               \
                uint4 wb[2][2];                                                                                             \
                X2_RD(0) X2_RD(1) X2_MM(0) X2_RD(2) X2_MM(1) X2_RD(3) X2_MM(2) X2_MM(3)                                     \
            } else {                                                                                                        \
                X2_COL(0) X2_COL(1) X2_COL(2) X2_COL(3)                                                                     \
            }                                                                                                               \
        }                                                                                                                   \
    } while (0)

    if (nrows > X2_AHEAD) {
        // lane-indexed tables of rows [tb, tb + 64): loaded (and waited for) BEFORE the requests of the prologue, so that no
        // wait of the compiler's ends up draining the DMA queue inside the loop
        for (int tb = X2_AHEAD; tb < nrows; tb += 64) {
            const int idx = min(tb + lane, nrows - 1);
            int pxv = pxt[idx], cwv = cwt[(size_t)idx * 4];
            int dv[2 * NW];
#pragma unroll
            for (int k = 0; k < 2 * NW; ++k) dv[k] = dut[(size_t)idx * X2_G * 2 * NW + k];
            asm volatile("" : "+v"(pxv), "+v"(cwv));
#pragma unroll
            for (int k = 0; k < 2 * NW; ++k) asm volatile("" : "+v"(dv[k]));
            if (tb == X2_AHEAD) {    // prologue: rows 0 .. 2 only request
#pragma unroll
                for (int row = 0; row < X2_AHEAD; ++row) {
                    const int p = __builtin_amdgcn_readfirstlane(pxt[row]);
                    X2_ISSUE_X(p, row);
#pragma unroll
                    for (int k = 0; k < NW; ++k) {
                        const int src = __builtin_amdgcn_readfirstlane(dut[(size_t)row * X2_G * 2 * NW + 2 * k]);
                        const int dst = __builtin_amdgcn_readfirstlane(dut[(size_t)row * X2_G * 2 * NW + 2 * k + 1]);
                        X2_ISSUE_W(src, dst);
                    }
                }
            }
            const int tend = min(64, nrows - tb);
            for (int qi = 0; qi < tend; ++qi) {
                const int row = tb + qi;
                X2_STAMP(0);
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (1 + NW)) : "memory");   // my requests of row - 3 have landed
                X2_STAMP(1);
                __builtin_amdgcn_s_barrier();                                          // everyone's have; everyone left row - 1
                X2_STAMP(2);
#define X2_REQUESTS()                                                                                                        \
                {                                                                                                           \
                    const int p = __builtin_amdgcn_readlane(pxv, qi);                                                       \
                    X2_ISSUE_X(p, row);                                                                                     \
                    _Pragma("unroll") for (int k = 0; k < NW; ++k) {                                                        \
                        const int src = __builtin_amdgcn_readlane(dv[2 * k], qi), dst = __builtin_amdgcn_readlane(dv[2 * k + 1], qi); \
                        X2_ISSUE_W(src, dst);                                                                               \
                    }                                                                                                       \
                }
                const bool late = X2_LATE && wave >= 8;
                if (!late) X2_REQUESTS()
                X2_STAMP(3);
                const uint32_t cw = (uint32_t)__builtin_amdgcn_readlane(cwv, qi);
                const uint32_t xo = (uint32_t)((row + 1) & 3) << 14;                   // slab of row - 3
                uint32_t wslot = (cw & 0xff) << 11;
                X2_HALF(0); X2_HALF(1);
                X2_STAMP(4);
                if (late) X2_REQUESTS()
#undef X2_REQUESTS
            }
        }
    }
#undef X2_ISSUE_X
#undef X2_ISSUE_W
#undef X2_HALF
#undef X2_RD
#undef X2_MM
#undef X2_COL

    // Epilogue: D[o][n]: col = n = r (lane) of my row tile, rows o = (reg & 3) + 8 * (reg >> 2) + 4h of output block 4c + j.
    // The 16 output blocks are 1024 contiguous bytes per minibatch row: staged through the idle ring as [128 rows][1024 B]
    // (16-byte pieces of row n XOR-swizzled with n & 31) and stored as full rows.
    constexpr int ROWB = X2_G * 64;
    if (X2_NO_EPILOGUE) { if (acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] == 12345.f) Y[0] = DT::from_f32(1.f); return; }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the trailing (dummy) requests must not land in the staging tile
    __syncthreads();
    {
        const int n = tw * 32 + r;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t lo = (uint32_t)DT::from_f32(acc[j][4 * q + 0]) | ((uint32_t)DT::from_f32(acc[j][4 * q + 1]) << 16);
                const uint32_t hi = (uint32_t)DT::from_f32(acc[j][4 * q + 2]) | ((uint32_t)DT::from_f32(acc[j][4 * q + 3]) << 16);
                const int piece = (4 * cls + j) * 4 + q;
                *reinterpret_cast<uint2*>(smem + n * ROWB + ((piece ^ (n & 31)) << 4) + 8 * h) = make_uint2(lo, hi);
            }
        }
    }
    __syncthreads();
    {
        const int rowbytes = nob * 64;
        T* ybase = Y + (size_t)ob0 * 32;
        constexpr int PPR = ROWB / 16;
        for (int i = threadIdx.x; i < X2_R * PPR; i += 64 * X2_G) {
            const int n = i / PPR, piece = i % PPR;
            if (n_tile + n < N && piece * 16 < rowbytes) {
                const uint4 v = *reinterpret_cast<const uint4*>(smem + n * ROWB + ((piece ^ (n & 31)) << 4));
                *reinterpret_cast<uint4*>(reinterpret_cast<unsigned char*>(ybase + (size_t)(n_tile + n) * Kout) + piece * 16) = v;
            }
        }
    }
}

}  // namespace bsmm
