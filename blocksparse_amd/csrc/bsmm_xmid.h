// bsmm_xmid.h -- xprop for MEDIUM minibatches (round 4): feature_axis = 1, bsize 32, 16-bit storage types, no plan needed.
//
// Between the small-minibatch kernel (bsmm_xsmall.h: a column's entry list cut over 8 waves that meet in LDS -- right below ~256 rows,
// where nothing else fills the machine, but 80 KiB of LDS and a reduction per 64 rows: 22-26 us at N = 512) and the flow kernel
// (bsmm_xflow.h: needs >= 256 units of 64 x 512 outputs) there are a few hundred to a few thousand minibatch rows.  There
// (output block, 64 rows) tasks alone cover the SIMDs, so nothing has to meet anywhere:
//   one WAVE = one output block x 64 (or 128) minibatch rows; it walks the column's whole entry list (reference: the lut segment of the
//   block, blocksparse/matmul.py:353-392) with its operands several entries ahead in wave-PRIVATE LDS rings filled by LDS-DMA
//   (activation tile [64 or 128 rows][64 B] + weight block 2 KiB per entry; `s_waitcnt vmcnt` counted, no barrier, no other wave involved);
//   a workgroup is four such waves, either on ONE output block and four consecutive row chunks (they share the weight blocks in the L1:
//   right while the activations fit an XCD's L2) or on four consecutive output blocks and the SAME rows with all workgroups of a row
//   chunk on one XCD (XMap, bsmm_xprop.h): an XCD's L2 then holds 1/8 of the activations instead of all of them (8192 features x 512
//   rows = 8.4 MB against 4 MB of L2: with blocks dealt round-robin over the XCDs every tile came from the Infinity Cache, 22 us instead
//   of 17 at that shape; at 4096 features x 512 rows the first mapping is the faster one, 13.4 against 15.7 us).
// fprop (TRANSW) reads the weight block with the transposing read (no transposed copy of W, no pre-pass, no workspace).
// Sums are formed in lut order, one fp32 accumulator per output: bit-identical to the staged / flow kernels.
#pragma once
#include "bsmm_common.h"
#include "bsmm_updat_tr.h"   // ds_tr16
#include "bsmm_updat_v2.h"   // glds16_saddr_x2 / _x4, uniform_ptr
#include "bsmm_xprop.h"      // XMap

namespace bsmm {

constexpr int XMD_NW = 4;                                   // waves per workgroup (independent: no barrier in the kernel)
constexpr int XMD_DX = 3;                                   // activation tiles in a wave's ring: entries idx .. idx + 2
// RT = 32-row tiles per wave (2: 64 rows, 4: 128 rows -- half the weight traffic per output, half the waves); DWS = weight blocks in a
// wave's ring.  The weight block of an entry is used by ONE wave per XCD, so its fetch always misses that XCD's L2 (~2 us from the
// Infinity Cache) while the activation tile, shared by every column of the row chunk, hits it (< 1 us): the weights run DWS - 1 entries
// ahead, the tiles two.  LDS-DMA results arrive in request order, so one counted `vmcnt` covers both.
constexpr int xmd_rows(int rt) { return 32 * rt; }
constexpr int xmd_wave_bytes(int rt, int dws) { return XMD_DX * rt * 2048 + dws * 2048; }
constexpr int xmd_lds(int rt, int dws) { return XMD_NW * xmd_wave_bytes(rt, dws); }            // <2, 4>: 80 KiB (two workgroups per CU); <4, 6>: 144 KiB

template <class DT, bool TRANSW, int RT, int DWS>
__global__ void __launch_bounds__(64 * XMD_NW)
xmid32_kernel(const typename DT::T* __restrict__ X, const typename DT::T* __restrict__ W, typename DT::T* __restrict__ Y,
              const int32_t* __restrict__ lut, XMap map, int segments, int N, int Cin, int Kout, int by_column) {
    typedef typename DT::T T;
    static_assert(DT::is16, "medium-minibatch kernel: 16-bit storage types");
    static_assert((RT == 2 || RT == 4) && DWS >= 4 && DWS <= 8, "medium-minibatch kernel: tile / ring geometry");
    constexpr int R = 32 * RT, XT = RT * 2048, XI = RT * 2;          // rows per wave, bytes and DMA instructions of an activation tile
    constexpr int WRING = XMD_DX * XT;                                // byte offset of the weight ring inside the wave's LDS
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // by_column = 0: the workgroup's waves take four consecutive output blocks and the same rows (map: row chunks x block quads);
    // by_column = 1: one output block and four consecutive row chunks (map: chunk quads x blocks) -- the waves share the weight blocks in
    // the L1; right while the whole activation matrix fits an XCD's L2 next to the weights
    int mt, ms;
    if (!xmap_decode(map, blockIdx.x, mt, ms)) return;
    const int seg = by_column ? ms : ms * XMD_NW + wave;
    const int n0 = (by_column ? mt * XMD_NW + wave : mt) * R;
    if (seg >= segments || n0 >= N) return;                 // (no barrier below: a wave may leave alone)
    const int r = lane & 31, h = lane >> 5;
    const int4 hdr = *reinterpret_cast<const int4*>(lut + 4 * seg);
    const int cnt = __builtin_amdgcn_readfirstlane(hdr.y), ob = __builtin_amdgcn_readfirstlane(hdr.z);
    const int2* ent = reinterpret_cast<const int2*>(lut) + __builtin_amdgcn_readfirstlane(hdr.x);
    unsigned char* ring = smem + wave * xmd_wave_bytes(RT, DWS);
    const uint32_t ring_addr = lds_addr_of(ring);

    // ---- DMA geometry.  Activation tile: instruction i (1 KiB) = rows 16 i + (lane >> 2), this lane's 16-byte LDS piece lane & 3 holds
    //      source piece (lane & 3) ^ ((row >> 2) & 3) (the swizzle of the fragment reads below); rows past N re-read row N - 1 (never
    //      stored).  Weight block: as in bsmm_xflow.h (natural for the transposing reads of fprop, swizzled otherwise). ----
    const unsigned char* xt = static_cast<const unsigned char*>(uniform_ptr(X));
    const unsigned char* wt = static_cast<const unsigned char*>(uniform_ptr(W));
    uint32_t xoff[XI];
#pragma unroll
    for (int i = 0; i < XI; ++i) {
        const int row = 16 * i + (lane >> 2);
        const int piece = (lane & 3) ^ ((row >> 2) & 3);
        xoff[i] = (uint32_t)min(n0 + row, N - 1) * (uint32_t)Cin * 2u + (uint32_t)piece * 16u;
    }
    const uint32_t wvoff = TRANSW ? (uint32_t)lane * 16u : (uint32_t)((lane >> 2) * 64 + (((lane & 3) ^ ((lane >> 4) & 3)) << 4));
    // ---- fragment geometry (stage 0 of either ring) ----
    uint32_t xrd[2], wrd[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        xrd[kk] = r * 64 + (((2 * kk + h) ^ ((r >> 2) & 3)) << 4);
        if constexpr (TRANSW) {
            const int g16 = lane >> 4, t16 = lane & 15;
            wrd[kk] = WRING + (16 * kk + 8 * h + (t16 >> 2)) * 64 + (16 * (g16 & 1) + 4 * (t16 & 3)) * 2;
        } else {
            wrd[kk] = WRING + r * 64 + (((2 * kk + h) ^ ((r >> 2) & 3)) << 4);
        }
    }

    f32x16 acc[RT];
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

    // the entry list, lane-resident in chunks of 64 (one vector load instead of a scalar load chain per entry)
    for (int eb = 0; eb < cnt; eb += 64) {
        const int2 cw = ent[min(eb + lane, cnt - 1)];
        uint32_t v_c = (uint32_t)cw.x * 64u, v_w = (uint32_t)cw.y << 11;          // byte offsets: inside an activation row, of the weight block
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("" : "+v"(v_c), "+v"(v_w));
        const int nchunk = min(64, cnt - eb);
        // (entries past the end of the chunk re-fetch its last entry: every iteration issues the same number of requests, so the wait below
        //  is one constant)
        auto issue_x = [&](int idx, int stage) {
            const uint32_t co = (uint32_t)__builtin_amdgcn_readlane(v_c, min(idx, nchunk - 1));
            const uint32_t dst = ring_addr + stage * XT;
#pragma unroll
            for (int i = 0; i < XI; i += 4) glds16_saddr_x4(xt, xoff[i] + co, xoff[i + 1] + co, xoff[i + 2] + co, xoff[i + 3] + co, dst + i * 1024);
        };
        auto issue_w = [&](int idx, int stage) {
            const uint32_t wo = (uint32_t)__builtin_amdgcn_readlane(v_w, min(idx, nchunk - 1));
            glds16_saddr_x2(wt, wvoff + wo, wvoff + wo + 1024u, ring_addr + WRING + stage * 2048);
        };
        // prologue in the order of the loop (tile, then weight), so that the count below holds from entry 0 on: weights 0 .. DWS - 4 first
#pragma unroll
        for (int k = 0; k < DWS - 3; ++k) issue_w(k, k);
        issue_x(0, 0);
        issue_w(DWS - 3, DWS - 3);
        issue_x(1, 1);
        issue_w(DWS - 2, DWS - 2);
        int xs = 0, ws = 0;                                                        // ring stages of entry idx
        for (int idx = 0; idx < nchunk; ++idx) {
            // tile idx + 2 -> the stage tile idx - 1 was multiplied from; weight idx + DWS - 1 -> the stage of weight idx - 1 (their fragment
            // reads were issued an iteration ago, in program order before these requests)
            issue_x(idx + 2, xs == 0 ? XMD_DX - 1 : xs - 1);
            issue_w(idx + DWS - 1, ws == 0 ? DWS - 1 : ws - 1);
            // requests younger than tile idx: weight, tile idx + 1, weight, tile idx + 2, weight; weight idx is older than tile idx (DWS >= 4)
#ifdef XMD_WAIT_ALL
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * XI + 6) : "memory");      // tile idx and weight idx have landed
#endif
            const unsigned char* sx = ring + xs * XT;
            const unsigned char* sw = ring + ws * 2048;
            uint4 wq[2], xf[RT][2];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                if constexpr (TRANSW) {
                    const uint2 lo = ds_tr16(sw + wrd[kk]), hi = ds_tr16(sw + wrd[kk] + 4 * 64);
                    wq[kk] = make_uint4(lo.x, lo.y, hi.x, hi.y);
                } else {
                    wq[kk] = *reinterpret_cast<const uint4*>(sw + wrd[kk]);
                }
#pragma unroll
                for (int t = 0; t < RT; ++t) xf[t][kk] = *reinterpret_cast<const uint4*>(sx + xrd[kk] + t * 2048);
            }
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int t = 0; t < RT; ++t) acc[t] = DT::mfma32(wq[kk], xf[t][kk], acc[t]);
            xs = xs == XMD_DX - 1 ? 0 : xs + 1;
            ws = ws == DWS - 1 ? 0 : ws + 1;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                           // the re-fetches past the end of the chunk
    }

    // Epilogue through the wave's ring (idle now), 64 rows at a time: D[o][n] with col n = r, rows o = (reg & 3) + 8 (reg >> 2) + 4h ->
    // [64 rows][64 B], the four 16-byte pieces of row n XOR-swizzled with (n >> 2) & 3; read back as full 64-byte rows and stored
    // (16 rows per instruction).
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    unsigned char* ybase = reinterpret_cast<unsigned char*>(Y + (size_t)ob * 32);
#pragma unroll
    for (int pass = 0; pass < RT / 2; ++pass) {
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const int t = 2 * pass + tt, n = tt * 32 + r;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t lo = (uint32_t)DT::from_f32(acc[t][4 * q + 0]) | ((uint32_t)DT::from_f32(acc[t][4 * q + 1]) << 16);
                const uint32_t hi = (uint32_t)DT::from_f32(acc[t][4 * q + 2]) | ((uint32_t)DT::from_f32(acc[t][4 * q + 3]) << 16);
                *reinterpret_cast<uint2*>(ring + pass * 4096 + n * 64 + ((q ^ ((n >> 2) & 3)) << 4) + 8 * h) = make_uint2(lo, hi);
            }
        }
    }
    asm volatile("" ::: "memory");
#pragma unroll
    for (int i = 0; i < 2 * RT; ++i) {
        const int n = 16 * i + (lane >> 2), pc = lane & 3;
        const uint4 v = *reinterpret_cast<const uint4*>(ring + n * 64 + ((pc ^ ((n >> 2) & 3)) << 4));
        const int gn = n0 + n;
        if (gn < N) *reinterpret_cast<uint4*>(ybase + (size_t)gn * Kout * 2 + pc * 16) = v;
    }
}

}  // namespace bsmm
