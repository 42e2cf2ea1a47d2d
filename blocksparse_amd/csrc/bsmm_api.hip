// bsmm_api.hip -- C-ABI entry points (include/bsmm.h) and kernel dispatch for gfx950.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

#include "bsmm.h"
#include "bsmm_plan.h"
#include "bsmm_l2norm.h"
#include "bsmm_sparse_proj.h"
#include "bsmm_updat.h"
#include "bsmm_updat_tr.h"
#include "bsmm_updat_win.h"
#include "bsmm_updat_v2.h"
#include "bsmm_super8.h"
#include "bsmm_xcols.h"
#include "bsmm_xcol.h"
#include "bsmm_xcol_v2.h"
#include "bsmm_xflow.h"
#include "bsmm_updat16_rows.h"
#include "bsmm_xsmall.h"
#include "bsmm_xsmall0.h"
#include "bsmm_xmid.h"
#include "bsmm_xcol16_v2.h"
#include "bsmm_b64.h"
#include "bsmm_xprop.h"

using namespace bsmm;

namespace {

// kernel-choice overrides come with the call (bsmm_args.flags), never from process state:
//   0 production dispatch, 1 = BSMM_FLAG_FORCE_VALU, 2 = BSMM_FLAG_NO_PLAN, 3 = BSMM_FLAG_FORCE_PLAN
inline int call_variant(const bsmm_args* a) {
    if (a->flags & BSMM_FLAG_FORCE_VALU) return 1;
    if (a->flags & BSMM_FLAG_NO_PLAN) return 2;
    if (a->flags & BSMM_FLAG_FORCE_PLAN) return 3;
    return 0;
}
inline void trace(const bsmm_args* a, int k) { if (a->trace) *a->trace = k; }
#ifndef UTS_NMAX
#define UTS_NMAX 768      // minibatch rows (x pairs) up to which the one-wave-per-block updat kernel takes the per-block calls (experiment switch: 0 = never)
#endif

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel, device); the result is checked.  The kernel is a template
// ARGUMENT, so every kernel instantiation has its own latch (a latch per function TYPE would be shared by all kernels of one
// signature, e.g. every xcol32_v3_kernel<...>: only the first of them would get the attribute).  The latch is a cache of an
// idempotent driver call, not a switch: it never changes what a later call computes.
template <auto Kernel>
inline int ensure_lds(int bytes) {
    static std::atomic<uint64_t> done{0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return (int)hipGetLastError();
    const uint64_t bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return 0;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(Kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return (int)e;
    done.fetch_or(bit, std::memory_order_release);
    return 0;
}

inline size_t elem_size(int dtype) { return dtype == BSMM_F32 ? 4 : 2; }

// compute units of the current device (the kernel-choice cost models and the streaming updat grid scale with it); the driver is
// asked once per device -- a cache of a constant, not a switch
inline int device_cus() {
    static std::atomic<int> cached[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    const int slot = dev & 63;
    int cus = cached[slot].load(std::memory_order_relaxed);
    if (cus > 0) return cus;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    cached[slot].store(cus, std::memory_order_relaxed);
    return cus;
}

int check_common(const bsmm_args* a) {
    if (!a || !a->lut) return BSMM_ERR_ARG;
    if (a->blocks <= 0 || a->N <= 0 || a->C <= 0 || a->K <= 0) return BSMM_ERR_ARG;
    if (a->bsize != 8 && a->bsize != 16 && a->bsize != 32 && !(a->bsize == 64 && a->axis == 1)) return BSMM_ERR_UNSUPPORTED;   // 64: axis 1 only, as the reference
    if (a->axis != 0 && a->axis != 1) return BSMM_ERR_UNSUPPORTED;
    if (a->dtype != BSMM_F32 && a->dtype != BSMM_F16 && a->dtype != BSMM_BF16) return BSMM_ERR_UNSUPPORTED;
    if (a->C % a->bsize || a->K % a->bsize) return BSMM_ERR_ARG;
    if (reinterpret_cast<uintptr_t>(a->lut) & 15) return BSMM_ERR_ARG;
    return BSMM_OK;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// A plan must be one this library built for THIS kind of call (the descriptor comes from bsmm_plan_attach): anything else
// is refused here, on the host, instead of reaching a kernel that would not recognise it.
int check_plan(bool updat, const bsmm_args* a) {
    if (!a->plan) return BSMM_OK;
    if (reinterpret_cast<uintptr_t>(a->plan) & 15) return BSMM_ERR_ARG;
    const int32_t m = a->plan_magic;
    if (a->bsize == 64) return (m == B64PLAN_MAGIC && a->plan_inner == (updat ? 1 : 0)) ? BSMM_OK : BSMM_ERR_ARG;
    if (a->bsize == 8) return (m == S8PLAN_MAGIC && a->plan_width > 0 && (a->plan_items > 0) == updat && (a->dtype != BSMM_F32 || updat)) ? BSMM_OK : BSMM_ERR_ARG;   // (fp32: updat only, updat8_f32_split)
    if (updat && a->dtype == BSMM_F32)      // fp32: the 16-bit plans of bsize 32 / 16 (the bf16-split paths updat32_f32_split / updat16_f32_split use them)
        return (((m == U2PLAN_MAGIC && a->bsize == 32) || (m == UPLAN_MAGIC && a->bsize == 16)) && a->plan_items > 0) ? BSMM_OK : BSMM_ERR_ARG;
    if (updat) return (((m == UPLAN_MAGIC && a->bsize == 16) || (m == U2PLAN_MAGIC && a->bsize == 32)) && a->plan_items > 0) ? BSMM_OK : BSMM_ERR_ARG;   // (bsize-32 'BSUP' plans: retired in round 4, refused as bsmm.h says)
    if (a->bsize == 16) return (m == X7PLAN_MAGIC && a->plan_width == X7_G && a->dtype != BSMM_F32) ? BSMM_OK : BSMM_ERR_ARG;
    if (a->dtype == BSMM_F32) return (m == XCPLAN_MAGIC && a->plan_width == XS_G) ? BSMM_OK : BSMM_ERR_ARG;
    return (m == XCPLAN_MAGIC || (m == X2PLAN_MAGIC && a->plan_width == X2_G) || (m == X4PLAN_MAGIC && a->plan_width == X4_G && a->axis == 1)) ? BSMM_OK : BSMM_ERR_ARG;
}

// ---------------------------------------------------------------------------------------------
// xprop
// ---------------------------------------------------------------------------------------------
template <class DT, int BS, int AXIS, bool FPROP>
int launch_xprop_valu(const void* X, const void* W, void* Y, const bsmm_args* a, hipStream_t st, float* yacc) {
    typedef typename DT::T T;
    dim3 grid(a->segments, (a->N + 255) / 256);    // segments on x: no 65535 limit
    trace(a, BSMM_K_XPROP_VALU);
    xprop_valu_kernel<DT, BS, AXIS, FPROP><<<grid, 256, 0, st>>>(static_cast<const T*>(X), static_cast<const T*>(W),
                                                                 static_cast<T*>(Y), a->lut, a->N, a->C, a->K, a->gate, yacc);
    return (int)hipGetLastError();
}

template <class DT, int BS, int AXIS>
int launch_xprop_mfma(const void* X, const void* Wsel, void* Y, const bsmm_args* a, hipStream_t st, float* yacc) {
    typedef typename DT::T T;
    const int N = a->N;
    trace(a, BSMM_K_XPROP_SEGMENT);
    auto go = [&](auto nsub_tag) {
        constexpr int NSUB = decltype(nsub_tag)::value;
        constexpr int NT = 4 * BS * NSUB;
        XMap m;
        m.ntiles = (N + NT - 1) / NT;
        m.segments = a->segments;
        m.P = m.ntiles >= 8 ? 1 : (8 + m.ntiles - 1) / m.ntiles;
        if (m.P > m.segments) m.P = m.segments;
        m.SP = (m.segments + m.P - 1) / m.P;
        if (a->gate) {
            if constexpr (BS == 32)
                xprop32_kernel<DT, AXIS, NSUB, true><<<m.grid(), 256, 0, st>>>(static_cast<const T*>(X), static_cast<const T*>(Wsel),
                                                                               static_cast<T*>(Y), a->lut, m, N, a->C, a->K, a->gate, yacc);
            else
                xprop16_kernel<DT, AXIS, NSUB, true><<<m.grid(), 256, 0, st>>>(static_cast<const T*>(X), static_cast<const T*>(Wsel),
                                                                               static_cast<T*>(Y), a->lut, m, N, a->C, a->K, a->gate, yacc);
            return;
        }
        if constexpr (BS == 32)
            xprop32_kernel<DT, AXIS, NSUB><<<m.grid(), 256, 0, st>>>(static_cast<const T*>(X), static_cast<const T*>(Wsel),
                                                                     static_cast<T*>(Y), a->lut, m, N, a->C, a->K, nullptr, yacc);
        else
            xprop16_kernel<DT, AXIS, NSUB><<<m.grid(), 256, 0, st>>>(static_cast<const T*>(X), static_cast<const T*>(Wsel),
                                                                     static_cast<T*>(Y), a->lut, m, N, a->C, a->K, nullptr, yacc);
    };
    // per-wave minibatch extent: BS*NSUB columns; use the wide tile only when it still fills the chip
    if constexpr (BS == 32) {
        if (N >= 2048) go(std::integral_constant<int, 2>{});
        else           go(std::integral_constant<int, 1>{});
    } else {
        if (N >= 2048) go(std::integral_constant<int, 4>{});
        else if (N >= 512) go(std::integral_constant<int, 2>{});
        else           go(std::integral_constant<int, 1>{});
    }
    return (int)hipGetLastError();
}

// grouped (xcol) kernels need args->plan built by bsmm_xprop_plan_build for args->lut
#ifndef BSMM_XC_WIDE_PH
#define BSMM_XC_WIDE_PH 4
#endif
inline bool use_xcol() { return true; }

// bsize 16, 'BSX7' plan: which inner loop.  The list-driven kernel multiplies every block on its own with the K = 16 instruction; where most
// blocks have their pair partner (dense layouts) the round-2 kernel's K = 32 instruction per PAIR wins on feature axis 1 -- measured at 4096^2,
// N = 8192 (profiles/r03_x7_density.txt): 20 % 152 / 145 against 166 / 149 us, 30 % 201 / 193 against 210 / 184, 50 % 315 / 319 against 284 / 263;
// on feature axis 0 the list kernel wins at every density (50 %: 251 / 237 against 321 / 301).
inline bool x7_use_list(const bsmm_args* a) {
#ifdef X7_POSITIONAL
    return a->axis == 0 && !a->gate;      // (measurement build: the pair kernel wherever it exists)
#else
    if (a->gate) return false;
    const double dens = (double)a->blocks / std::max(1.0, (a->C / 16.0) * (a->K / 16.0));
    return a->axis == 0 || dens < 0.4;
#endif
}

template <class DT, int AXIS>
int launch_xcol16_v2(const void* X, const void* Wsel, void* Y, const bsmm_args* a, hipStream_t st, bool transw) {
    typedef typename DT::T T;
    const int n_out = a->K / 16;
    XMap m;
    m.ntiles = (a->N + XC_R - 1) / XC_R;
    m.segments = (n_out + X7_G - 1) / X7_G;
    m.P = m.ntiles >= 8 ? 1 : (8 + m.ntiles - 1) / m.ntiles;
    if (m.P > m.segments) m.P = m.segments;
    m.SP = (m.segments + m.P - 1) / m.P;
    trace(a, BSMM_K_XCOL16_STAGED);
    if (a->gate) {
        return BSMM_ERR_ARG;                     // (xprop_path never sends a gated call here)
    } else if (!x7_use_list(a)) {
        // (feature axis 1, dense layouts only -- x7_use_list: on feature axis 0 the list kernel wins at every density, and that instantiation
        //  of the pair kernel spilled 4 registers: not built)
        if constexpr (AXIS == 1) {
            if (int rc = ensure_lds<&xcol16_v2_kernel<DT, 1, false>>(X7_LDS)) return rc;
            xcol16_v2_kernel<DT, 1, false><<<m.grid(), 1024, X7_LDS, st>>>(static_cast<const T*>(X), static_cast<const T*>(Wsel), static_cast<T*>(Y), a->plan, m,
                                                                           a->N, a->C, a->K, nullptr);
        } else {
            return BSMM_ERR_ARG;
        }
    } else if (transw) {
        if (int rc = ensure_lds<&xcol16_list_kernel<DT, AXIS, true>>(X7_LDS)) return rc;
        xcol16_list_kernel<DT, AXIS, true><<<m.grid(), 1024, X7_LDS, st>>>(static_cast<const T*>(X), static_cast<const T*>(Wsel), static_cast<T*>(Y), a->plan, m,
                                                                          a->N, a->C, a->K);
    } else {
        if (int rc = ensure_lds<&xcol16_list_kernel<DT, AXIS, false>>(X7_LDS)) return rc;
        xcol16_list_kernel<DT, AXIS, false><<<m.grid(), 1024, X7_LDS, st>>>(static_cast<const T*>(X), static_cast<const T*>(Wsel), static_cast<T*>(Y), a->plan, m,
                                                                           a->N, a->C, a->K);
    }
    return (int)hipGetLastError();
}

template <class DT, int AXIS>
int launch_xcol16(const void* X, const void* Wsel, void* Y, const bsmm_args* a, hipStream_t st, bool transw) {
    // ('BSX7' plans only: the round-1 kernel and its 'BSX6' plans were retired in round 4)
    return (a->plan_magic == X7PLAN_MAGIC && a->plan_width == X7_G) ? launch_xcol16_v2<DT, AXIS>(X, Wsel, Y, a, st, transw) : BSMM_ERR_ARG;
}

// fp32 on the bf16 matrix cores: split pre-passes into the workspace ([3][N*C] activation pieces, then [3][blocks*1024]
// weight pieces), then the wide xcol kernel with three slabs per step.
inline size_t xcols_w_bytes(const bsmm_args* a) { return 6 * (size_t)a->blocks * 1024; }
#ifndef XS_NO_FUSE      // (experiment switch: -DXS_NO_FUSE=1 keeps the activation-split pre-pass on feature axis 1 too, for A/B)
#define XS_NO_FUSE 0
#endif
// feature axis 1 (round 4): the activations are split inside the kernel (xcol32sf_kernel) -- no pieces of X in the workspace
inline bool xcols_fused(const bsmm_args* a) { return a->axis == 1 && !XS_NO_FUSE; }
inline size_t xcols_workspace_bytes(const bsmm_args* a) {
    return (xcols_fused(a) ? 0 : 6 * (size_t)a->N * a->C) + (a->prepared_w ? 0 : xcols_w_bytes(a));
}
template <int AXIS>
int launch_xcol32s(bool fprop, const void* X, const void* W, void* Y, const bsmm_args* a, hipStream_t st) {
    const size_t nx = (size_t)a->N * a->C;
    if (a->plan_magic != XCPLAN_MAGIC || a->plan_width != XS_G) return BSMM_ERR_ARG;
    const bool fused = AXIS == 1 && xcols_fused(a);
    const size_t need = xcols_workspace_bytes(a);
    if (need && (!a->workspace || a->workspace_bytes < need || !aligned16(a->workspace))) return BSMM_ERR_WORKSPACE;
    if (a->prepared_w && !aligned16(a->prepared_w)) return BSMM_ERR_ARG;
    if (fused && !aligned16(X)) return BSMM_ERR_ARG;                       // (16-byte row pieces of the fp32 activations; C % 32 == 0 is checked by the dispatch)
    uint16_t* xp = static_cast<uint16_t*>(a->workspace);
    const uint16_t* wp = static_cast<const uint16_t*>(a->prepared_w);
    if (!fused) split3_x_kernel<<<(unsigned)((nx / 8 + 255) / 256), 256, 0, st>>>(static_cast<const float*>(X), xp, nx);
    if (!wp) {      // W's pieces are a constant of the pass: a caller that holds them (bsmm_prepare_weights) skips this launch
        uint16_t* wq = fused ? xp : xp + 3 * nx;
        if (fprop) split3_w_kernel<true><<<a->blocks, 256, 0, st>>>(static_cast<const float*>(W), wq, a->blocks);
        else       split3_w_kernel<false><<<a->blocks, 256, 0, st>>>(static_cast<const float*>(W), wq, a->blocks);
        wp = wq;
    }
    const int n_out = a->K / 32;
    XMap m;
    m.ntiles = (a->N + XC_R - 1) / XC_R;
    m.segments = (n_out + XS_G - 1) / XS_G;
    m.P = m.ntiles >= 8 ? 1 : (8 + m.ntiles - 1) / m.ntiles;
    if (m.P > m.segments) m.P = m.segments;
    m.SP = (m.segments + m.P - 1) / m.P;
    trace(a, BSMM_K_XCOL32_F32SPLIT);
    if constexpr (AXIS == 1) {
        if (fused) {
            if (int rc = ensure_lds<&xcol32sf_kernel>(XSF_LDS)) return rc;
#ifndef XSF_GMAP
#define XSF_GMAP 1      // (experiment switch: 0 = the row-tile-per-XCD mapping of the other grouped kernels)
#endif
            unsigned grid = (unsigned)m.grid();
            if (XSF_GMAP && m.segments % 8 == 0 && m.ntiles >= 4) { m.P = 0; grid = (unsigned)(m.segments * m.ntiles); }   // one group per XCD at a time
            xcol32sf_kernel<<<grid, 64 * XS_G, XSF_LDS, st>>>(static_cast<const float*>(X), wp, static_cast<float*>(Y), a->plan, m, a->N, a->C, a->K, a->blocks);
            return (int)hipGetLastError();
        }
    }
    if (int rc = ensure_lds<&xcol32s_kernel<AXIS>>(XS_LDS)) return rc;
    xcol32s_kernel<AXIS><<<m.grid(), 64 * XS_G, XS_LDS, st>>>(xp, wp, static_cast<float*>(Y), a->plan, m, a->N, a->C, a->K, a->blocks);
    return (int)hipGetLastError();
}

template <class DT, bool TRANSW, int G, int PH>
int launch_xcol0_g(const void* X, const void* Wsel, void* Y, const bsmm_args* a, hipStream_t st) {
    typedef typename DT::T T;
    const int n_out = a->K / 32;
    XMap m;
    m.ntiles = (a->N + XC_R - 1) / XC_R;
    m.segments = (n_out + G - 1) / G;
    m.P = m.ntiles >= 8 ? 1 : (8 + m.ntiles - 1) / m.ntiles;
    if (m.P > m.segments) m.P = m.segments;
    m.SP = (m.segments + m.P - 1) / m.P;
    constexpr int LDS = 2 * PH * XC0_SLAB;
    if (int rc = ensure_lds<&xcol32_a0_kernel<DT, TRANSW, G, PH>>(LDS)) return rc;
    trace(a, BSMM_K_XCOL32);
    xcol32_a0_kernel<DT, TRANSW, G, PH><<<m.grid(), 64 * G, LDS, st>>>(static_cast<const T*>(X), static_cast<const T*>(Wsel), static_cast<T*>(Y), a->plan, m,
                                                                     a->N, a->C, a->K);
    return (int)hipGetLastError();
}

template <class DT, bool TRANSW>
int launch_xcol0(const void* X, const void* Wsel, void* Y, const bsmm_args* a, hipStream_t st) {
    if (a->plan_width == 16)   return launch_xcol0_g<DT, TRANSW, 16, BSMM_XC_WIDE_PH>(X, Wsel, Y, a, st);
    if (a->plan_width == XC_G) return launch_xcol0_g<DT, TRANSW, XC_G, XC_PH>(X, Wsel, Y, a, st);
    return BSMM_ERR_ARG;
}

template <class DT, bool TRANSW, int G, int PH>
int launch_xcol_g(const void* X, const void* Wsel, void* Y, const bsmm_args* a, hipStream_t st) {
    typedef typename DT::T T;
    const int n_out = a->K / 32;
    XMap m;
    m.ntiles = (a->N + XC_R - 1) / XC_R;
    m.segments = (n_out + G - 1) / G;
    m.P = m.ntiles >= 8 ? 1 : (8 + m.ntiles - 1) / m.ntiles;
    if (m.P > m.segments) m.P = m.segments;
    m.SP = (m.segments + m.P - 1) / m.P;
    constexpr int LDS = xc_lds_bytes(G, PH);
    if (int rc = ensure_lds<&xcol32_a1_kernel<DT, TRANSW, G, PH>>(LDS)) return rc;
    trace(a, BSMM_K_XCOL32);
    xcol32_a1_kernel<DT, TRANSW, G, PH><<<m.grid(), 64 * G, LDS, st>>>(static_cast<const T*>(X), static_cast<const T*>(Wsel), static_cast<T*>(Y), a->plan, m,
                                                                     a->N, a->C, a->K);
    return (int)hipGetLastError();
}

template <class DT, bool TRANSW>
int launch_xcol(const void* X, const void* Wsel, void* Y, const bsmm_args* a, hipStream_t st) {
    if (a->plan_width == 16)   return launch_xcol_g<DT, TRANSW, 16, BSMM_XC_WIDE_PH>(X, Wsel, Y, a, st);
    if (a->plan_width == XC_G) return launch_xcol_g<DT, TRANSW, XC_G, XC_PH>(X, Wsel, Y, a, st);
    return BSMM_ERR_ARG;
}

template <class DT, bool TRANSW, int AXIS, bool GATED, int PH>
int launch_xcol_v2_ph(const void* X, const void* Wsel, void* Y, const bsmm_args* a, hipStream_t st) {
    typedef typename DT::T T;
    const int n_out = a->K / 32;
    XMap m;
    m.ntiles = (a->N + X2_R - 1) / X2_R;
    m.segments = (n_out + X2_G - 1) / X2_G;
    m.P = m.ntiles >= 8 ? 1 : (8 + m.ntiles - 1) / m.ntiles;
    if (m.P > m.segments) m.P = m.segments;
    m.SP = (m.segments + m.P - 1) / m.P;
    if (int rc = ensure_lds<&xcol32_v2_kernel<DT, TRANSW, AXIS, GATED, PH>>(X2_LDS)) return rc;
    trace(a, BSMM_K_XCOL32_STAGED);
    xcol32_v2_kernel<DT, TRANSW, AXIS, GATED, PH><<<m.grid(), 64 * X2_G, X2_LDS, st>>>(static_cast<const T*>(X), static_cast<const T*>(Wsel), static_cast<T*>(Y), a->plan, m,
                                                                                 a->N, a->C, a->K, GATED ? a->gate : nullptr);
    return (int)hipGetLastError();
}

template <class DT, bool TRANSW, int AXIS, bool GATED = false>
int launch_xcol_v2(const void* X, const void* Wsel, void* Y, const bsmm_args* a, hipStream_t st) {
    switch (a->plan_inner) {     // steps per phase the plan was cut for
        case 2: return launch_xcol_v2_ph<DT, TRANSW, AXIS, GATED, 2>(X, Wsel, Y, a, st);
        case 3: return launch_xcol_v2_ph<DT, TRANSW, AXIS, GATED, 3>(X, Wsel, Y, a, st);
        case 4: return launch_xcol_v2_ph<DT, TRANSW, AXIS, GATED, 4>(X, Wsel, Y, a, st);
    }
    return BSMM_ERR_ARG;
}

#ifndef BSMM_FLOW_P
#define BSMM_FLOW_P 0             // measurement builds: groups of a row tile spread over P XCDs (measured at the bench shape, profiles/r05_flow_xcd_map.txt: no effect)
#endif
// barrier-free persistent kernel ('BSX4' plans, bsmm_xflow.h): one workgroup per CU walks its (row tile, group) units
template <class DT, bool TRANSW, int RT>
int launch_xflow_rt(const void* X, const void* Wsel, void* Y, const bsmm_args* a, hipStream_t st) {
    typedef typename DT::T T;
    const int n_out = a->K / 32;
    constexpr int R = 32 * RT;
    XMap m;
    m.ntiles = (a->N + R - 1) / R;
    m.segments = (n_out + X4_G - 1) / X4_G;
    m.P = m.ntiles >= 8 ? 1 : (8 + m.ntiles - 1) / m.ntiles;
#if BSMM_FLOW_P
    if (m.ntiles >= 8) m.P = BSMM_FLOW_P;      // measurement builds: groups of a row tile spread over P XCDs (an XCD then runs segments / P groups at a time)
#endif
    if (m.P > m.segments) m.P = m.segments;
    m.SP = (m.segments + m.P - 1) / m.P;
    if (int rc = ensure_lds<&xflow32_kernel<DT, TRANSW, RT>>(X4_LDS)) return rc;
    trace(a, BSMM_K_XCOL32_FLOW | (RT == 2 ? (BSMM_KV_FLOW_HALF_UNITS << 8) : 0));
    const int cus = device_cus();
    const int grid = std::min(m.grid(), std::max(8, cus / 8 * 8));
    xflow32_kernel<DT, TRANSW, RT><<<grid, 64 * X4_G, X4_LDS, st>>>(static_cast<const T*>(X), static_cast<const T*>(Wsel), static_cast<T*>(Y), a->plan,
                                                                    m, a->N, a->C, a->K);
    return (int)hipGetLastError();
}

// Units of 128 rows unless they leave CUs idle: 64-row units (twice as many, each with half the multiplies per weight block) from there down.
#ifndef BSMM_FLOW_RT
#define BSMM_FLOW_RT 0            // 0: by the unit count; 2 / 4: forced (measurement builds)
#endif
template <class DT, bool TRANSW>
int launch_xflow(const void* X, const void* Wsel, void* Y, const bsmm_args* a, hipStream_t st) {
    const long units128 = (long)((a->N + X4_R - 1) / X4_R) * ((a->K / 32 + X4_G - 1) / X4_G);
    const bool half = BSMM_FLOW_RT == 2 || (BSMM_FLOW_RT == 0 && units128 < device_cus());
    return half ? launch_xflow_rt<DT, TRANSW, 2>(X, Wsel, Y, a, st) : launch_xflow_rt<DT, TRANSW, 4>(X, Wsel, Y, a, st);
}

template <class DT, int AXIS>
int launch_xgroup32(const void* X, const void* Wsel, void* Y, const bsmm_args* a, hipStream_t st, bool transw) {
    if (a->plan_magic == X4PLAN_MAGIC) {
        if (a->plan_width != X4_G || a->gate) return BSMM_ERR_ARG;
        if constexpr (AXIS == 1) return transw ? launch_xflow<DT, true>(X, Wsel, Y, a, st) : launch_xflow<DT, false>(X, Wsel, Y, a, st);
        else return BSMM_ERR_ARG;
    }
    if (a->plan_magic == X2PLAN_MAGIC) {
        if (a->plan_width != X2_G) return BSMM_ERR_ARG;
        if (a->gate) return transw ? launch_xcol_v2<DT, true, AXIS, true>(X, Wsel, Y, a, st) : launch_xcol_v2<DT, false, AXIS, true>(X, Wsel, Y, a, st);
        return transw ? launch_xcol_v2<DT, true, AXIS>(X, Wsel, Y, a, st) : launch_xcol_v2<DT, false, AXIS>(X, Wsel, Y, a, st);
    }
    // round-1 grouped kernels ('BSXC' plans: the bsize-8 super-block path): fprop always comes with the transposed copy of W (`transw` is
    // only ever set for the staged / flow plans above), so the TRANSW = true instantiations -- the axis-0 one spilled 3 registers -- are not built
    if (a->plan_magic != XCPLAN_MAGIC || transw) return BSMM_ERR_ARG;
    if constexpr (AXIS == 1) return launch_xcol<DT, false>(X, Wsel, Y, a, st);
    else                     return launch_xcol0<DT, false>(X, Wsel, Y, a, st);
}

template <class DT, int BS>
int launch_transpose(const void* W, void* Wt, int blocks, hipStream_t st) {
    typedef typename DT::T T;
    transpose_blocks_kernel<DT, BS><<<blocks, 256, 0, st>>>(static_cast<const T*>(W), static_cast<T*>(Wt), blocks);
    return (int)hipGetLastError();
}

// Workspace layout of an xprop call (both parts 16-byte aligned): [0, wt) the transposed / expanded / split weights of
// the kernel that runs, then [wt, wt + N*K*4) the fp32 accumulators of LOCKED output blocks when the reference-policy
// table is walked by the per-segment kernels with a 16-bit storage type (lut segments that share an output block are
// summed in fp32 and rounded ONCE by lock_finalize_kernel -- the reference rounds every partial sum to 16 bit with
// red.add.f16x2, src/blocksparse_hgemm_cn_64_op_gpu.cu:211-240).
inline size_t round16(size_t b) { return (b + 15) & ~(size_t)15; }
inline size_t wt_bytes(const bsmm_args* a) { return round16((size_t)a->blocks * a->bsize * a->bsize * elem_size(a->dtype)); }
inline size_t lock_acc_bytes(const bsmm_args* a) {
    return (a->locks > 0 && a->dtype != BSMM_F32) ? (size_t)a->N * a->K * sizeof(float) : 0;
}

enum XPath { XP_VALU, XP_SEGMENT, XP_XCOL32, XP_XCOL16, XP_F32SPLIT, XP_SUPER8, XP_SMALL, XP_MID };
#ifndef BSMM_SMALL_N_MAX
#define BSMM_SMALL_N_MAX 4096     // the small-minibatch kernel (bsmm_xsmall.h) is considered up to this many minibatch rows (the cost model decides)
#endif
#ifndef XS0_NMAX
#define XS0_NMAX 512              // feature axis 0: the small-minibatch kernel (bsmm_xsmall0.h) up to this many minibatch columns
#endif
#ifndef XSN_NMAX
#define XSN_NMAX 512              // bsize 16 / 8, feature axis 1: the narrow small-minibatch kernel (bsmm_xsmall.h) up to this many minibatch rows
#endif
#ifndef UAW_NMAX
#define UAW_NMAX 128              // bsize 32 / 16, feature axis 0, weight gradient: the one-wave-per-block kernel up to this many minibatch columns x pairs
#endif
#ifndef U8P_ON
#define U8P_ON 1                  // bsize 8, feature axis 0, weight gradient: the pair kernel (bsmm_updat.h) where the cost model picks it (0: never; measurement builds)
#endif
#ifndef BSMM_MID_MODE
#define BSMM_MID_MODE 0           // medium-minibatch kernel (bsmm_xmid.h): 0 = by the cost model, 1 = whenever it can run, -1 = never (measurement builds)
#endif
#ifndef BSMM_MID_MAP
#define BSMM_MID_MAP 0            // its workgroup shape: 0 = by the size of the activations, 1 = one block x 4 row chunks, 2 = 4 blocks x one row chunk
#endif
#ifndef BSMM_MID_RT
#define BSMM_MID_RT 0             // its rows per wave: 0 = by the wave count, 2 = 64, 4 = 128 (measurement builds)
#endif

// ONE decision, used for the workspace layout, the zero-fill of locked outputs and the launch (the three used to be
// derived separately and could disagree).
template <class DT, int BS, int AXIS>
XPath xprop_path(const void* X, const void* W, const void* Y, const bsmm_args* a, bool fprop = true) {
    const int variant = call_variant(a);
    const bool vec_ok = aligned16(X) && aligned16(W) && aligned16(Y);
    // gated calls: the staged bsize-32 kernel applies gates (exactly: bsmm_xcol_v2.h); everything else runs the per-segment kernels.
    // (Round 6: the GATED instantiation of the bsize-16 pair kernel is no longer built -- 22 spilled registers, 280-308 us at BASELINE
    //  configs[2]'s shape against 85 us ungated; a gated call on the fast kernels is bsmm_gate_weights + the UNGATED call, include/bsmm.h.)
    const bool gate_ok = a->gate == nullptr || (DT::is16 && BS == 32 && a->plan_magic == X2PLAN_MAGIC);
    const bool plan_ok = a->plan != nullptr && gate_ok && vec_ok && (variant == 0 || variant == 3);
    const bool force = variant == 3;
    if constexpr (BS == 8) {
        if constexpr (DT::is16 && AXIS == 0) {      // short minibatches on feature axis 0 (bsmm_xsmall0.h, round 6: the reference benchmark's (8, 0) shapes)
            // (BS=8 in scripts/gpu_a0_xprop_sweep.py: at N = 1024 114 against 365, 51 against 108, 162 against 226, 249 against 451 us)
            const bool n_ok8 = a->N <= 2 * XS0_NMAX;
            if (variant == 0 && vec_ok && !a->gate && a->locks == 0 && a->N % 8 == 0 && n_ok8 && a->segments > 0) return XP_SMALL;
        }
        if constexpr (DT::is16 && AXIS == 1) {      // short minibatches on feature axis 1 (bsmm_xsmall.h::xsmall_narrow_kernel, round 6)
            // (the same sweep: against the V_FMA kernel 11.6 against 32 us at 4096^2 10 % N = 64, 43 against 216 at 20480 / 1.5 %; up to 512 rows it also
            //  beats the super-block path where that applies: 291 against 426 us at 20480 / 1.5 % N = 512)
            if (variant == 0 && vec_ok && !a->gate && a->locks == 0 && a->N <= XSN_NMAX && a->C % 8 == 0 && a->K % 8 == 0 && a->segments > 0) return XP_SMALL;
        }
        if constexpr (DT::is16) {
            // bsize 8 on the matrix cores: expand W into the 32x32 super-blocks of the 'BSS8' plan and run the bsize-32 kernel
            const bool shape_ok = a->C % 32 == 0 && a->K % 32 == 0 && !(AXIS == 0 && (a->N % 8 != 0));
            const bool fill = (long)((a->N + XC_R - 1) / XC_R) * ((a->K / 32 + XC_G - 1) / XC_G) >= device_cus() * 7 / 8;
            // (locked reference-policy tables stay on the exact kernel: the repair pass of the super-block path writes Y directly)
            if (plan_ok && a->plan_magic == S8PLAN_MAGIC && a->plan_width > 0 && shape_ok && a->locks == 0 && (fill || force)) return XP_SUPER8;
        }
        if constexpr (DT::is16 && AXIS == 0) {
            // no super-block path for this call (no plan, or too few units for the chip -- hidden 2560 at N = 2048: 643 us on the V_FMA kernel): the
            // pair kernel at any minibatch (its time grows with blocks x N like the V_FMA kernel's, at a third of it)
            if (variant == 0 && vec_ok && !a->gate && a->locks == 0 && a->N % 8 == 0 && a->segments > 0) return XP_SMALL;
        }
        return XP_VALU;
    }
    if (variant == 1 || !vec_ok) return XP_VALU;
    // small minibatches on feature axis 1: every output block's entry list cut over the 8 waves of a workgroup (bsmm_xsmall.h); needs no plan.
    // Measured as hipGraph replays (scripts/gpu_smalln_sweep.py, profiles/r04_smalln.txt): 3 + 1.35e-5 us per (block, minibatch row)
    bool small_ok = false;
    double t_small = 0.0;
    if constexpr (BS == 32 && DT::is16 && AXIS == 1) {
        small_ok = variant == 0 && !a->gate && a->locks == 0 && a->N <= BSMM_SMALL_N_MAX && a->C % 32 == 0 && a->K % 32 == 0;
        t_small = 3.0 + 1.35e-5 * (double)a->blocks * a->N;
    }
    // medium minibatches: one wave per (output block, 64 rows) walks the block's whole entry list from a private LDS ring (bsmm_xmid.h)
    bool mid_ok = false;
    double t_mid = 0.0;
    if constexpr (BS == 32 && DT::is16 && AXIS == 1) {
        mid_ok = BSMM_MID_MODE >= 0 && (variant == 0 || (a->flags & BSMM_FLAG_FORCE_MID)) && !a->gate && a->locks == 0 && a->C % 32 == 0 && a->K % 32 == 0 && a->segments > 0 &&
                 (long)a->N * a->C * 2 < (1L << 32) && a->blocks < (1 << 21);
        // measured as hipGraph replays (profiles/r04_smalln.txt): ~0.9 us per entry of a column and round of 8 waves per CU; a half-empty
        // machine is not proportionally faster (occupancy o: o >= 1 -> o, below -> 0.25 + 0.5 o)
        const double o = (double)a->segments * ((a->N + 63) / 64) / (8.0 * device_cus());
        t_mid = 4.5 + 0.9 * ((double)a->blocks / a->segments) * std::max(1.12 * o, 0.25 + 0.5 * o);   // (full rounds: 12 % more per round, measured)
        if ((BSMM_MID_MODE > 0 || (a->flags & BSMM_FLAG_FORCE_MID)) && mid_ok) return XP_MID;
    }
    // small minibatches on feature axis 0 (round 6, bsmm_xsmall0.h): the regime of the reference's own benchmark (N = 64).  Measured against the
    // per-segment and the plan kernels in scripts/gpu_a0_xprop_sweep.py
    if constexpr ((BS == 32 || BS == 16) && DT::is16 && AXIS == 0) {
        // (hipGraph replays, fprop / bprop us, this kernel against what ran before: hidden 2560 dense N = 64 8.3 / 8.1 against 45.8 / 27.1, N = 512 22.7
        //  against 64.6 / 54.3, N = 1024 38.5 against 95; 20480 at 1.7 % N = 64 10.5 against 63.8 / 44.3, N = 512 60 against 82 / 73, N = 1024 116
        //  against 91 / 81: up to 512 columns always, up to 1024 where the columns are long -- >= 16 blocks on average)
        // (bsize 16, the same sweep with BS=16: at N = 1024 the kernel still wins at every shape -- 51 against 118, 25 against 41, 72 against 83, 109 against 210 us)
        const bool n_ok = a->N <= XS0_NMAX || (a->N <= 2 * XS0_NMAX && (BS == 16 || (long)a->blocks >= 16L * a->segments));
        if (variant == 0 && !a->gate && a->locks == 0 && a->N % 8 == 0 && n_ok && a->C % BS == 0 && a->K % BS == 0 && a->segments > 0) return XP_SMALL;
    }
    if constexpr (BS == 16 && DT::is16 && AXIS == 1) {
        // (scripts/gpu_a1_narrow_sweep.py, hipGraph replays: against the per-segment kernel the narrow kernel wins fprop up to 256 rows -- 7.2 against
        //  14.8 us at 4096^2 10 % N = 64, 58 against 73 at hidden 2560 dense N = 256: that kernel transposes W in a pre-pass -- and bprop up to 64)
        if (variant == 0 && !a->gate && a->locks == 0 && a->N <= (fprop ? XSN_NMAX / 2 : XSN_NMAX / 8) && a->C % 16 == 0 && a->K % 16 == 0 && a->segments > 0) return XP_SMALL;
    }
    if (!plan_ok) {
        if (mid_ok && t_mid < (small_ok ? t_small : 1e30) && t_mid < 6.0 + 1.2e-5 * (double)a->blocks * a->N + 8.0) return XP_MID;
        if (small_ok && t_small < 6.0 + 1.2e-5 * (double)a->blocks * a->N + 8.0) return XP_SMALL;     // (the per-segment fprop also pays a transpose pre-pass)
        return XP_SEGMENT;
    }
    // grouped kernels need enough (row tile x group) workgroups to fill 256 CUs; below that the per-segment kernel,
    // which has segments x tiles workgroups, is faster
    if constexpr (BS == 16) {
        if constexpr (!DT::is16) return XP_SEGMENT;
        if (AXIS == 0 && (a->N % 8 != 0)) return XP_SEGMENT;            // 16-byte aligned row pieces
        if (a->plan_magic == X7PLAN_MAGIC && (AXIS == 1 ? (long)a->C : (long)a->N) * 256 >= (1L << 32)) return XP_SEGMENT;   // 32-bit lane offsets
        const bool enough = (long)((a->N + XC_R - 1) / XC_R) * ((a->K / 16 + 15) / 16) >= device_cus() * 7 / 8;
        return (enough || force) ? XP_XCOL16 : XP_SEGMENT;
    }
    if constexpr (BS == 32 && !DT::is16) {
        if (a->plan_magic != XCPLAN_MAGIC) return XP_SEGMENT;           // 'BSXC' (G = 16): the exact bf16 split
        if (AXIS == 0 && (a->N % 8 != 0)) return XP_SEGMENT;
        if (a->C % 32 != 0) return XP_SEGMENT;
        const bool enough = (long)((a->N + 127) / 128) * ((a->K / 32 + XC_G - 1) / XC_G) >= device_cus() * 7 / 8;
        if (!(enough || force)) return XP_SEGMENT;
        return XP_F32SPLIT;
    }
    if constexpr (BS == 32 && DT::is16) {
        if (AXIS == 0 && (a->N % 8 != 0)) return XP_SEGMENT;            // axis-0 xcol needs 16-byte aligned row pieces
        // staged kernel: 32-bit per-lane byte offsets inside a slab's source (128 rows of C elements / 64 rows of N elements)
        if ((a->plan_magic == X2PLAN_MAGIC || a->plan_magic == X4PLAN_MAGIC) && (AXIS == 1 ? (long)a->C : (long)a->N) * 256 >= (1L << 32)) return XP_SEGMENT;
        if (a->plan_magic == X4PLAN_MAGIC && AXIS != 1) return XP_SEGMENT;
        if (force) return XP_XCOL32;
        // Cost model fitted to the measurements in profiles/r01_sweeps.md (4096^2 / 20 % and 8192^2 / 5 %, N = 512 .. 8192):
        // the grouped kernel pays ~0.48 us per pair step of a group plus ~0.045 us per block, once per round of 256
        // workgroups, whatever the density; the per-segment kernel pays ~1.04e-5 us per (block, minibatch row).
        const int G = a->plan_width > 0 ? a->plan_width : 16;
        const double CB = a->C / 32.0, KB = a->K / 32.0;
        const double ngroups = (double)((a->K / 32 + G - 1) / G), ntiles = (double)((a->N + XC_R - 1) / XC_R);
        const double cus = (double)device_cus();
        const double rounds = std::max(1.0, std::ceil(ntiles * ngroups / cus));
        const double dens = std::min(1.0, a->blocks / std::max(1.0, CB * KB));
        const double steps = std::ceil(CB / 2.0) * (1.0 - std::pow(1.0 - dens, 2.0 * G));
        double t_group = rounds * (0.48 * steps + 0.045 * a->blocks / ngroups) + 8.0;
        double t_segment = 17.0 + 1.04e-5 * (double)a->blocks * a->N;
        if (a->plan_magic == X2PLAN_MAGIC || a->plan_magic == X4PLAN_MAGIC) {
            // staged kernel, refit (scripts/gpu_xprop_sweep.py, 4096^2 20 % / 5 %, 8192^2 5 %, 2048^2 20 %, N = 128 .. 8192): a round
            // costs 0.28 us per pair step + 0.040 us per block of the group, up to 20 % more when the round fills all CUs
            const double fill = std::min(1.0, ntiles * ngroups / rounds / cus);
            t_group = rounds * (0.28 * steps + 0.040 * a->blocks / ngroups) * (1.0 + 0.2 * fill) + 4.0;
            t_segment = 6.0 + 1.2e-5 * (double)a->blocks * a->N;        // (round 4: refit on graph replays -- the 14.4 us floor of round 2 was the host's)
        }
        if (mid_ok && t_mid <= t_group && t_mid <= t_segment && (!small_ok || t_mid < t_small)) return XP_MID;
        if (small_ok && t_small <= t_group && t_small <= t_segment) return XP_SMALL;
        return t_group < t_segment ? XP_XCOL32 : XP_SEGMENT;
    }
    return XP_SEGMENT;
}

// the bsize-32 call nested in a 'BSS8' plan: plan_inner = nested width | nested format << 8 (0: round-1 'BSXC' / 'BSUP', 1: staged 'BSX2',
// 2: streaming 'BSU2') | the nested plan's own descriptor word << 11 (bsmm_plan_attach)
inline bsmm_args s8_inner(const bsmm_args* a, bool updat) {
    const int ns = a->plan_width, code = (a->plan_inner >> 8) & 7;
    bsmm_args b = *a;
    b.bsize = 32; b.blocks = ns; b.plan = a->plan + s8_off_nested(ns);
    b.plan_width = a->plan_inner & 0xff;
    b.plan_inner = (int32_t)((uint32_t)a->plan_inner >> 11);
    b.plan_magic = updat ? (code == 2 ? U2PLAN_MAGIC : UPLAN_MAGIC) : (code == 1 ? X2PLAN_MAGIC : XCPLAN_MAGIC);
    return b;
}

template <class DT, int BS, int AXIS>
int xprop_typed(bool fprop, const void* X, const void* W, void* Y, const bsmm_args* a) {
    typedef typename DT::T T;
    hipStream_t st = static_cast<hipStream_t>(a->stream);
    const XPath path = xprop_path<DT, BS, AXIS>(X, W, Y, a, fprop);
    if (path == XP_SUPER8) {
        if constexpr (BS == 8 && DT::is16) {
            // Exactness with non-finite activations: a super-block multiplies its zero-filled (absent) 8x8 parts with live
            // activations, and 0 * Inf = NaN would reach outputs the reference leaves finite (it walks only the lookup-table entries,
            // blocksparse/matmul.py:353-392).  One scan of X leaves a flag behind the expanded weights; when it is set -- never, for a
            // healthy network -- the per-entry V_FMA kernel recomputes Y after the matrix-core pass (same stream, Y fully rewritten).
            const int ns = a->plan_width;
            const size_t wbytes = round16((size_t)ns * 1024 * elem_size(a->dtype)), need = wbytes + 16;
            if (!a->workspace || a->workspace_bytes < need || !aligned16(a->workspace)) return BSMM_ERR_WORKSPACE;
            int32_t* flag = reinterpret_cast<int32_t*>(static_cast<char*>(a->workspace) + wbytes);
            if (fprop) expand8_kernel<DT, true><<<ns, 256, 0, st>>>(static_cast<const T*>(W), a->plan, static_cast<T*>(a->workspace), flag);
            else       expand8_kernel<DT, false><<<ns, 256, 0, st>>>(static_cast<const T*>(W), a->plan, static_cast<T*>(a->workspace), flag);
            const size_t n8 = (size_t)a->N * a->C / 8;                 // (C % 32 == 0 and X 16-byte aligned on this path)
            const int sgrid = (int)std::min<size_t>((n8 + 255) / 256, (size_t)device_cus() * 8);
            if (a->dtype == BSMM_BF16) nonfinite16_kernel<0x7f80u><<<sgrid, 256, 0, st>>>(static_cast<const uint4*>(X), n8, flag);
            else                       nonfinite16_kernel<0x7c00u><<<sgrid, 256, 0, st>>>(static_cast<const uint4*>(X), n8, flag);
            bsmm_args b = s8_inner(a, false);
            int rc = launch_xgroup32<DT, AXIS>(X, a->workspace, Y, &b, st, false);
            if (rc) return rc;
            dim3 grid(a->segments, (a->N + 255) / 256);
            if (fprop) xprop_valu_kernel<DT, BS, AXIS, true><<<grid, 256, 0, st>>>(static_cast<const T*>(X), static_cast<const T*>(W), static_cast<T*>(Y), a->lut,
                                                                                  a->N, a->C, a->K, nullptr, nullptr, flag);
            else       xprop_valu_kernel<DT, BS, AXIS, false><<<grid, 256, 0, st>>>(static_cast<const T*>(X), static_cast<const T*>(W), static_cast<T*>(Y), a->lut,
                                                                                   a->N, a->C, a->K, nullptr, nullptr, flag);
            rc = (int)hipGetLastError();
            trace(a, BSMM_K_XPROP_SUPER8);
            return rc;
        }
    }
    if (path == XP_SMALL) {
        if constexpr ((BS == 8 || BS == 16) && DT::is16 && AXIS == 1) {
            constexpr int LDSN = XSM_NW * XSM_PART;
            if (fprop) { if (int rc = ensure_lds<&xsmall_narrow_kernel<DT, BS, true>>(LDSN)) return rc; }
            else       { if (int rc = ensure_lds<&xsmall_narrow_kernel<DT, BS, false>>(LDSN)) return rc; }
            trace(a, BSMM_K_XPROP_SMALL);
            dim3 grid(a->segments, (a->N + XSM_R - 1) / XSM_R);
            if (fprop) xsmall_narrow_kernel<DT, BS, true><<<grid, 64 * XSM_NW, LDSN, st>>>(static_cast<const T*>(X), static_cast<const T*>(W), static_cast<T*>(Y), a->lut, a->N, a->C, a->K);
            else       xsmall_narrow_kernel<DT, BS, false><<<grid, 64 * XSM_NW, LDSN, st>>>(static_cast<const T*>(X), static_cast<const T*>(W), static_cast<T*>(Y), a->lut, a->N, a->C, a->K);
            return (int)hipGetLastError();
        }
        if constexpr (BS == 8 && DT::is16 && AXIS == 0) {
            if (fprop) { if (int rc = ensure_lds<&xsmall8_a0_kernel<DT, true>>(XS16_LDS)) return rc; }
            else       { if (int rc = ensure_lds<&xsmall8_a0_kernel<DT, false>>(XS16_LDS)) return rc; }
            trace(a, BSMM_K_XPROP_SMALL);
            dim3 grid(a->segments, (a->N + XS0_C - 1) / XS0_C);
            if (fprop) xsmall8_a0_kernel<DT, true><<<grid, 64 * XS0_NW, XS16_LDS, st>>>(static_cast<const T*>(X), static_cast<const T*>(W), static_cast<T*>(Y), a->lut, a->N);
            else       xsmall8_a0_kernel<DT, false><<<grid, 64 * XS0_NW, XS16_LDS, st>>>(static_cast<const T*>(X), static_cast<const T*>(W), static_cast<T*>(Y), a->lut, a->N);
            return (int)hipGetLastError();
        }
        if constexpr (BS == 16 && DT::is16 && AXIS == 0) {
            if (fprop) { if (int rc = ensure_lds<&xsmall16_a0_kernel<DT, true>>(XS16_LDS)) return rc; }
            else       { if (int rc = ensure_lds<&xsmall16_a0_kernel<DT, false>>(XS16_LDS)) return rc; }
            trace(a, BSMM_K_XPROP_SMALL);
            dim3 grid(a->segments, (a->N + XS0_C - 1) / XS0_C);
            if (fprop) xsmall16_a0_kernel<DT, true><<<grid, 64 * XS0_NW, XS16_LDS, st>>>(static_cast<const T*>(X), static_cast<const T*>(W), static_cast<T*>(Y), a->lut, a->N);
            else       xsmall16_a0_kernel<DT, false><<<grid, 64 * XS0_NW, XS16_LDS, st>>>(static_cast<const T*>(X), static_cast<const T*>(W), static_cast<T*>(Y), a->lut, a->N);
            return (int)hipGetLastError();
        }
        if constexpr (BS == 32 && DT::is16 && AXIS == 0) {
            if (fprop) { if (int rc = ensure_lds<&xsmall32_a0_kernel<DT, true>>(XS0_LDS)) return rc; }
            else       { if (int rc = ensure_lds<&xsmall32_a0_kernel<DT, false>>(XS0_LDS)) return rc; }
            trace(a, BSMM_K_XPROP_SMALL);
            dim3 grid(a->segments, (a->N + XS0_C - 1) / XS0_C);
            if (fprop) xsmall32_a0_kernel<DT, true><<<grid, 64 * XS0_NW, XS0_LDS, st>>>(static_cast<const T*>(X), static_cast<const T*>(W), static_cast<T*>(Y), a->lut, a->N);
            else       xsmall32_a0_kernel<DT, false><<<grid, 64 * XS0_NW, XS0_LDS, st>>>(static_cast<const T*>(X), static_cast<const T*>(W), static_cast<T*>(Y), a->lut, a->N);
            return (int)hipGetLastError();
        }
        if constexpr (BS == 32 && DT::is16 && AXIS == 1) {
            if (fprop) { if (int rc = ensure_lds<&xsmall32_kernel<DT, true>>(XSM_LDS)) return rc; }
            else       { if (int rc = ensure_lds<&xsmall32_kernel<DT, false>>(XSM_LDS)) return rc; }
            trace(a, BSMM_K_XPROP_SMALL);
            dim3 grid(a->segments, (a->N + XSM_R - 1) / XSM_R);
            if (fprop) xsmall32_kernel<DT, true><<<grid, 64 * XSM_NW, XSM_LDS, st>>>(static_cast<const T*>(X), static_cast<const T*>(W), static_cast<T*>(Y), a->lut, a->N, a->C, a->K);
            else       xsmall32_kernel<DT, false><<<grid, 64 * XSM_NW, XSM_LDS, st>>>(static_cast<const T*>(X), static_cast<const T*>(W), static_cast<T*>(Y), a->lut, a->N, a->C, a->K);
            return (int)hipGetLastError();
        }
    }
    if (path == XP_MID) {
        if constexpr (BS == 32 && DT::is16 && AXIS == 1) {
            trace(a, BSMM_K_XPROP_MID);
            // (128-row waves -- <4, 6>, half the weight traffic per output -- measured no faster anywhere the flow kernel is not faster
            //  still: BSMM_MID_RT=4 builds them for measurements)
            const bool big = BSMM_MID_RT == 4;
            auto go = [&](auto rt_tag, auto dws_tag) -> int {
                constexpr int RT = decltype(rt_tag)::value, DWS = decltype(dws_tag)::value;
                // which four tasks make a workgroup: see the kernel
                const int by_column = BSMM_MID_MAP == 1 || (BSMM_MID_MAP == 0 && (long)a->N * a->C * 2 <= (5L << 20)) ? 1 : 0;
                const int nchunks = (a->N + xmd_rows(RT) - 1) / xmd_rows(RT);
                XMap m;
                m.ntiles = by_column ? (nchunks + XMD_NW - 1) / XMD_NW : nchunks;
                m.segments = by_column ? a->segments : (a->segments + XMD_NW - 1) / XMD_NW;
                m.P = m.ntiles >= 8 ? 1 : (8 + m.ntiles - 1) / m.ntiles;
                if (m.P > m.segments) m.P = m.segments;
                m.SP = (m.segments + m.P - 1) / m.P;
                constexpr int LDS = xmd_lds(RT, DWS);
                if (fprop) {
                    if (int rc = ensure_lds<&xmid32_kernel<DT, true, RT, DWS>>(LDS)) return rc;
                    xmid32_kernel<DT, true, RT, DWS><<<m.grid(), 64 * XMD_NW, LDS, st>>>(static_cast<const T*>(X), static_cast<const T*>(W), static_cast<T*>(Y), a->lut, m,
                                                                                        a->segments, a->N, a->C, a->K, by_column);
                } else {
                    if (int rc = ensure_lds<&xmid32_kernel<DT, false, RT, DWS>>(LDS)) return rc;
                    xmid32_kernel<DT, false, RT, DWS><<<m.grid(), 64 * XMD_NW, LDS, st>>>(static_cast<const T*>(X), static_cast<const T*>(W), static_cast<T*>(Y), a->lut, m,
                                                                                         a->segments, a->N, a->C, a->K, by_column);
                }
                return (int)hipGetLastError();
            };
            return big ? go(std::integral_constant<int, 4>{}, std::integral_constant<int, 6>{}) : go(std::integral_constant<int, 2>{}, std::integral_constant<int, 4>{});
        }
    }
    if (path == XP_F32SPLIT) {
        if constexpr (BS == 32 && !DT::is16) return launch_xcol32s<AXIS>(fprop, X, W, Y, a, st);
    }
    // the remaining kernels read the weights with the contraction index contiguous: fprop needs the transposed copy
    const bool generic = path == XP_VALU || path == XP_SEGMENT;
    float* yacc = nullptr;
    size_t off = 0;
    const void* Wsel = W;
    const bool staged = path == XP_XCOL32 && (a->plan_magic == X2PLAN_MAGIC || a->plan_magic == X4PLAN_MAGIC);   // transposes the staged blocks itself
    const bool staged16 = path == XP_XCOL16 && a->plan_magic == X7PLAN_MAGIC && x7_use_list(a);   // the list kernel reads them transposed
    if (fprop && path != XP_VALU && !staged && !staged16) {
        if constexpr (BS != 8) {
            if (!a->workspace || a->workspace_bytes < wt_bytes(a) || !aligned16(a->workspace)) return BSMM_ERR_WORKSPACE;
            const int rc = launch_transpose<DT, BS>(W, a->workspace, a->blocks, st);
            if (rc) return rc;
            Wsel = a->workspace;
            off = wt_bytes(a);
        }
    }
    if (generic && a->locks > 0) {   // several segments accumulate into the same output block: start from zero
        if (DT::is16) {
            const size_t need = off + lock_acc_bytes(a);
            if (!a->workspace || a->workspace_bytes < need || !aligned16(a->workspace)) return BSMM_ERR_WORKSPACE;
            yacc = reinterpret_cast<float*>(static_cast<char*>(a->workspace) + off);
            hipError_t e = hipMemsetAsync(yacc, 0, lock_acc_bytes(a), st);
            if (e != hipSuccess) return (int)e;
        } else {
            hipError_t e = hipMemsetAsync(Y, 0, (size_t)a->N * a->K * elem_size(a->dtype), st);
            if (e != hipSuccess) return (int)e;
        }
    }
    int rc = BSMM_ERR_UNSUPPORTED;
    switch (path) {
        case XP_VALU:
            rc = fprop ? launch_xprop_valu<DT, BS, AXIS, true>(X, W, Y, a, st, yacc) : launch_xprop_valu<DT, BS, AXIS, false>(X, W, Y, a, st, yacc);
            break;
        case XP_SEGMENT:
            if constexpr (BS != 8) rc = launch_xprop_mfma<DT, BS, AXIS>(X, Wsel, Y, a, st, yacc);
            break;
        case XP_XCOL32:
            if constexpr (BS == 32 && DT::is16) rc = launch_xgroup32<DT, AXIS>(X, Wsel, Y, a, st, staged && fprop);
            break;
        case XP_XCOL16:
            if constexpr (BS == 16 && DT::is16) rc = launch_xcol16<DT, AXIS>(X, Wsel, Y, a, st, staged16 && fprop);
            break;
        default: break;
    }
    if (rc == 0 && yacc) {
        dim3 grid(a->segments, (a->N + 255) / 256);
        lock_finalize_kernel<DT, BS, AXIS><<<grid, 256, 0, st>>>(yacc, static_cast<T*>(Y), a->lut, a->N, a->K);
        rc = (int)hipGetLastError();
    }
    return rc;
}

template <class DT>
int xprop_dt(bool fprop, const void* X, const void* W, void* Y, const bsmm_args* a) {
#define BSMM_CASE(BS, AX) \
    if (a->bsize == BS && a->axis == AX) return xprop_typed<DT, BS, AX>(fprop, X, W, Y, a);
    BSMM_CASE(32, 0) BSMM_CASE(32, 1) BSMM_CASE(16, 0) BSMM_CASE(16, 1) BSMM_CASE(8, 0) BSMM_CASE(8, 1)
#undef BSMM_CASE
    return BSMM_ERR_UNSUPPORTED;
}

// ---- bsize 64 (feature axis 1): the quadrant view on the bsize-32 path ('BS64' plans, bsmm_plan.h / bsmm_b64.h) ----
// descriptor of a 'BS64' plan in bsmm_args (bsmm_plan_attach): plan_width / plan_items = the nested plan's, plan_waves = the nested plan's
// waves (5 bits) | code of its format << 5 | its `inner` word << 8, plan_inner = 0 (xprop) / 1 (updat)
const int32_t kNestedMagic[8] = {0, XCPLAN_MAGIC, X2PLAN_MAGIC, 0 /* (3: 'BSXF', retired) */, UPLAN_MAGIC, U2PLAN_MAGIC, X4PLAN_MAGIC, 0};
inline int nested_code(int32_t magic) { for (int i = 1; i < 8; ++i) if (kNestedMagic[i] == magic) return i; return 0; }
inline size_t b64_w_bytes(const bsmm_args* a) { return round16((size_t)a->blocks * 4096 * elem_size(a->dtype)); }
inline size_t b64_gate_bytes(const bsmm_args* a) { return a->gate ? round16((size_t)a->blocks * 4 * sizeof(float)) : 0; }
// the bsize-32 call behind a bsize-64 one (workspace fields left to the caller)
inline bsmm_args b64_inner(const bsmm_args* a, bool updat) {
    bsmm_args b = *a;
    b.bsize = 32; b.blocks = 4 * a->blocks; b.prepared_w = nullptr;
    b.lut = a->plan + B64_HDR;
    if (!updat) { b.segments = 2 * a->segments; b.locks = 2 * a->locks; }
    b.plan = a->plan + b64_off_nested(updat ? 1 : 0, updat ? 0 : a->segments, a->blocks);
    b.plan_magic = kNestedMagic[(a->plan_waves >> 5) & 7];
    b.plan_waves = a->plan_waves & 31;
    b.plan_inner = (int32_t)((uint32_t)a->plan_waves >> 8);
    if (b.plan_magic == 0) { b.plan = nullptr; b.plan_width = b.plan_waves = b.plan_items = b.plan_inner = 0; }
    return b;
}

int xprop(bool fprop, const void* X, const void* W, void* Y, const bsmm_args* a);

int xprop64(bool fprop, const void* X, const void* W, void* Y, const bsmm_args* a) {
    hipStream_t st = static_cast<hipStream_t>(a->stream);
    if (!a->plan || !aligned16(W)) return BSMM_ERR_ARG;       // (check_plan: a 'BS64' xprop plan)
    bsmm_args b = b64_inner(a, false);
    // workspace: [quadrant copy of W, unless the caller prepared one][gates of the quadrants][the inner call's]
    const size_t wq = a->prepared_w ? 0 : b64_w_bytes(a), gq = b64_gate_bytes(a);
    const size_t inner = bsmm_workspace_bytes(fprop ? BSMM_OP_FPROP : BSMM_OP_BPROP, &b);
    if (wq + gq + inner > 0 && (!a->workspace || a->workspace_bytes < wq + gq + inner || !aligned16(a->workspace))) return BSMM_ERR_WORKSPACE;
    char* ws = static_cast<char*>(a->workspace);
    const void* Wq = a->prepared_w;
    if (!Wq) {
        if (elem_size(a->dtype) == 4) b64_split_kernel<4><<<a->blocks, 256, 0, st>>>(static_cast<const unsigned char*>(W), reinterpret_cast<unsigned char*>(ws), a->blocks, fprop ? 0 : 1);
        else                          b64_split_kernel<2><<<a->blocks, 256, 0, st>>>(static_cast<const unsigned char*>(W), reinterpret_cast<unsigned char*>(ws), a->blocks, fprop ? 0 : 1);
        Wq = ws;
    } else if (!aligned16(Wq)) return BSMM_ERR_ARG;
    if (a->gate) {
        b64_gate_kernel<<<(4 * a->blocks + 255) / 256, 256, 0, st>>>(a->gate, reinterpret_cast<float*>(ws + wq), a->blocks);
        b.gate = reinterpret_cast<const float*>(ws + wq);
    }
    b.workspace = inner ? ws + wq + gq : nullptr;
    b.workspace_bytes = inner ? a->workspace_bytes - wq - gq : 0;
    return xprop(fprop, X, Wq, Y, &b);
}

int xprop(bool fprop, const void* X, const void* W, void* Y, const bsmm_args* a) {
    int rc = check_common(a);
    if (rc) return rc;
    if (!X || !W || !Y || a->segments <= 0) return BSMM_ERR_ARG;
    if (a->bsize == 64 && !a->plan) return BSMM_ERR_UNSUPPORTED;      // bsize 64 runs through its plan only
    if ((rc = check_plan(false, a))) return rc;
    if (a->bsize == 64) return xprop64(fprop, X, W, Y, a);
    switch (a->dtype) {
        case BSMM_F32:  return xprop_dt<DTf32>(fprop, X, W, Y, a);
        case BSMM_F16:  return xprop_dt<DTf16>(fprop, X, W, Y, a);
        case BSMM_BF16: return xprop_dt<DTbf16>(fprop, X, W, Y, a);
    }
    return BSMM_ERR_UNSUPPORTED;
}

// ---------------------------------------------------------------------------------------------
// updat
// ---------------------------------------------------------------------------------------------
// workgroups per work item of the windowed kernels (each takes a slice of the minibatch): the caller's choice
// (bsmm_args.split), else enough to give every CU a workgroup while each keeps >= 8 chunks
// internal to the library (never set by callers: bsmm_updat clears it on entry of a public call): the blocks of this bsize-32 call are the
// quadrants of 64 x 64 blocks and DW is the 64-layout (updat64)
constexpr int32_t FLAG_INTERNAL_Q64 = 1 << 30;
static thread_local bool tl_inside_updat64 = false;      // (a public call that carries the bit is refused: bsmm_updat)

inline int updat_split(const bsmm_args* a, int nitems, int nchunks) {
    if (a->split > 0) return std::min(a->split, std::max(1, nchunks));
    int split = 1;
    const int cus = device_cus();
    while (nitems * split < cus && split * 2 <= nchunks / 8 && split < 8) split *= 2;
    return split;
}

// Streaming kernel (bsmm_updat_v2.h, 'BSU2' plans).  Default grid: 8 x U workgroups (U = CUs / 8) in the XCD-aware schedule of
// the plan (plan_inner = NSETS | common set size << 8).  Every item is ONE workgroup's -- and is stored directly -- when the plan
// has 8 sets of equal size that fill whole rounds of U; otherwise partial sums meet in the fp32 workspace.  The caller can ask
// for `split` workgroups per item instead (1: one per item, direct).
// partial-sum path of the streaming kernel: the workspace holds the fp32 sums [blocks][1024] (only written for
// BSMM_FLAG_DW_SUMS) and behind them one region of 64 accumulator slots x 4 KiB per (round, workgroup)
struct U2Launch { int grid; bool scratch; int flat; int rounds; int direct; };      // grid: the schedule's workgroups; direct: workgroups of the direct blocks behind them
// descriptor of a 'BSU2' plan (describe_flat): item sets (4 bits) | all sets equally long (bit 4) | longest set << 8 (13 bits) | direct blocks << 21 (10 bits)
inline int u2_direct_blocks(const bsmm_args* a) { return (a->plan_inner >> 21) & 0x3ff; }
inline int u2_longest_set(const bsmm_args* a) { return (a->plan_inner >> 8) & 0x1fff; }
// (+ 2 KiB: the fused data-parallel reduction reads / writes the sums in `world` 32-byte aligned shards, include/bsmm_dist.h)
inline size_t u2_sums_bytes(const bsmm_args* a) { return round16((size_t)a->blocks * 1024 * sizeof(float)) + 2048; }
inline size_t u2_region_bytes() { return (size_t)U2_WAVES * U2_SLOTS * 4096; }
// (a direct block's workgroup leaves ONE 4 KiB partial sum: packed behind the regions of the schedule's workgroups, 64 to a region's worth)
inline size_t u2_direct_bytes(int direct_wgs) { return (size_t)((direct_wgs + 63) / 64) * u2_region_bytes(); }
inline int u2_chunk(const bsmm_args* a) { return a->axis == 1 ? U2_CH : U2_CH0; }   // minibatch entries per chunk of the streaming kernel
inline U2Launch updat2_shape(const bsmm_args* a, bool gated) {
    const int cus = device_cus();
    const int nsets = a->plan_inner & 15, longest = u2_longest_set(a), common = (a->plan_inner & 16) ? longest : 0;
    U2Launch L;
    L.direct = u2_direct_blocks(a) * U2_DIRECT_PARTS;           // (their partial sums meet in the summing pass: such a plan always takes the scratch path)
    if (a->split >= 1) {
        const long nchunks = (long)a->pcount * ((a->N + u2_chunk(a) - 1) / u2_chunk(a));
        const long sp = std::max<long>(1, std::min<long>(a->split, nchunks));
        L.grid = (int)(a->plan_items * sp); L.scratch = sp > 1 || gated || L.direct > 0; L.flat = 1;
        L.rounds = (a->plan_items + L.grid - 1) / L.grid;
        return L;
    }
    const int U = std::max(1, cus / 8);
    L.grid = 8 * U; L.flat = 0;
    L.scratch = gated || L.direct > 0 || !(nsets == 8 && common > 0 && common % U == 0);
    L.rounds = (longest + U - 1) / U;
    return L;
}

template <class DT, int AXIS>
int launch_updat2(const PtrList8& xs, const PtrList8& es, void* DW, const bsmm_args* a, const float* gate) {
    typedef typename DT::T T;
    hipStream_t st = static_cast<hipStream_t>(a->stream);
    if (a->plan_magic != U2PLAN_MAGIC || a->plan_waves != U2_WAVES || a->plan_items <= 0 || (a->plan_width != 16 && a->plan_width != 8 && !(a->plan_width == 32 && AXIS == 1))) return BSMM_ERR_ARG;
    U2Launch L = updat2_shape(a, gate != nullptr);
    if (L.direct > 0 && AXIS != 1) return BSMM_ERR_ARG;         // (the builder makes direct blocks for feature axis 1 only)
    const bool sums_only = (a->flags & BSMM_FLAG_DW_SUMS) != 0;
    const bool q64 = (a->flags & FLAG_INTERNAL_Q64) != 0;       // (updat64: the summing pass writes the quadrants into their 64 x 64 blocks)
    if (sums_only || q64) L.scratch = true;
    float* scratch = nullptr;      // the partial-sum regions
    float* sums = nullptr;
    if (L.scratch) {
        const size_t need = u2_sums_bytes(a) + (size_t)L.rounds * L.grid * u2_region_bytes() + u2_direct_bytes(L.direct);
        if (!a->workspace || a->workspace_bytes < need || !aligned16(a->workspace)) return BSMM_ERR_WORKSPACE;
        sums = static_cast<float*>(a->workspace);
        scratch = reinterpret_cast<float*>(static_cast<char*>(a->workspace) + u2_sums_bytes(a));
    }
    trace(a, BSMM_K_UPDAT_STREAM);
    constexpr int LDS16 = AXIS == 1 ? u2_lds_bytes(16) : u2_lds_bytes0(16), LDS8 = AXIS == 1 ? u2_lds_bytes(8) : u2_lds_bytes0(8);
    if (a->plan_width == 32) {
        if constexpr (AXIS == 1) {
            if (int rc = ensure_lds<&updat32_a1_v2_kernel<DT, 32, 1>>(u2_lds_bytes(32))) return rc;
            updat32_a1_v2_kernel<DT, 32, 1><<<L.grid + L.direct, 64 * U2_WAVES, u2_lds_bytes(32), st>>>(xs, es, static_cast<T*>(DW), scratch, a->plan, a->N, a->C, a->K,
                                                                                            a->pcount, a->alpha, a->beta, L.flat, L.grid, L.rounds);
        } else {
            return BSMM_ERR_ARG;
        }
    } else if (a->plan_width == 16) {
        if (int rc = ensure_lds<&updat32_a1_v2_kernel<DT, 16, AXIS>>(LDS16)) return rc;
        updat32_a1_v2_kernel<DT, 16, AXIS><<<L.grid + L.direct, 64 * U2_WAVES, LDS16, st>>>(xs, es, static_cast<T*>(DW), scratch, a->plan, a->N, a->C, a->K,
                                                                                a->pcount, a->alpha, a->beta, L.flat, L.grid, L.rounds);
    } else {
        if (int rc = ensure_lds<&updat32_a1_v2_kernel<DT, 8, AXIS>>(LDS8)) return rc;
        updat32_a1_v2_kernel<DT, 8, AXIS><<<L.grid + L.direct, 64 * U2_WAVES, LDS8, st>>>(xs, es, static_cast<T*>(DW), scratch, a->plan, a->N, a->C, a->K,
                                                                              a->pcount, a->alpha, a->beta, L.flat, L.grid, L.rounds);
    }
    if (scratch) {
        const int CPI = a->pcount * ((a->N + u2_chunk(a) - 1) / u2_chunk(a));
        const int32_t* bmap = a->plan + U2_HDR + (size_t)a->plan_items * U2_ITEM;     // behind the items (bsmm_plan.h)
        if (sums_only) updat2_reduce_kernel<DT, true><<<a->blocks, 128, 0, st>>>(scratch, nullptr, sums, a->plan, bmap, nullptr, L.rounds * L.grid, L.grid, L.flat, CPI, 1.f, 0.f);
        else           updat2_reduce_kernel<DT, false><<<a->blocks, 128, 0, st>>>(scratch, static_cast<T*>(DW), nullptr, a->plan, bmap, gate, L.rounds * L.grid, L.grid, L.flat, CPI, a->alpha, a->beta, q64 ? 1 : 0);
    }
    return (int)hipGetLastError();
}

// Row-owner kernel for bsize 16 on feature axis 0 (bsmm_updat16_rows.h).  grid = items x split, part of the minibatch = linear id % split.
// split: the caller's (bsmm_args.split, at most U6_MAX_SPLIT), else the smallest power of two that gives every CU a workgroup.  Measured
// against the windowed kernel (scripts/gpu_updat16_rows_sweep.py, profiles/r05_updat16_rows_sweep.txt): the row-owner kernel wins once
// three quarters of the CUs get a workgroup and a workgroup keeps 8 chunks of 64 minibatch entries (16 when four or more parts leave
// their sums for the finalize pass) -- 17 to 43 % at 4096^2 / 8192^2, N >= 4096; otherwise BSMM_ERR_UNSUPPORTED: the caller takes the windowed kernel
// (2048^2 has 16 windows of 512 x 512: never).
constexpr int U6_MAX_SPLIT = 8;
// f32: the fp32 call through six bf16 piece pairs (updat16_f32_rows below): `pairs` pairs in xs / es, the images at `images` (always, also for
// one part), DW in fp32 with alpha / beta / gate by the finalize pass unless *skip_if (the non-finite flag of the split).
template <class DT>
int launch_updat16_rows(const PtrList8& xs, const PtrList8& es, void* DW, const bsmm_args* a, int pairs = 0, float* images = nullptr,
                        const float* gate = nullptr, const int32_t* skip_if = nullptr) {
    typedef typename DT::T T;
    hipStream_t st = static_cast<hipStream_t>(a->stream);
    const bool f32 = images != nullptr;
    const int pcount = f32 ? pairs : a->pcount;
    const int32_t* sec = a->plan + a->plan_inner;
    const int nchunks = (a->N + 63) / 64;
    const int nitems = a->plan_waves >> 8, wk = a->plan_width >> 8;      // (bsmm_plan_attach packs the section's item count / window width there)
    if (nitems <= 0 || (wk != 32 && wk != 16)) return BSMM_ERR_ARG;
    int split;
    if (a->split > 0) {
        split = std::min(std::min(a->split, nchunks), U6_MAX_SPLIT);
    } else {
        const int cus = device_cus();
        split = 1;
        while (split < U6_MAX_SPLIT && nitems * split < cus) split *= 2;
        const bool pays = 4L * nitems * split >= 3L * cus && (long)pcount * nchunks / split >= (split >= 4 ? 16 : 8);
        if (!pays && call_variant(a) != 3) return BSMM_ERR_UNSUPPORTED;
        split = std::max(1, std::min(split, nchunks));
    }
    float* scratch = images;
    const size_t nel = (size_t)a->blocks * 256;
    if (!f32 && (split > 1 || gate)) {      // one fp32 image of the sums per part (a gated call: always -- the finalize pass applies the gate)
        if (!a->workspace || a->workspace_bytes < (size_t)split * nel * sizeof(float) || !aligned16(a->workspace)) return BSMM_ERR_WORKSPACE;
        scratch = static_cast<float*>(a->workspace);
    }
    trace(a, BSMM_K_UPDAT16_ROWS);
    // (split 1 / 2 / 4 / 8: rounds of 8 workgroups = 8 / split items x split parts, see the kernel's workgroup map)
    const bool pow2 = split == 1 || split == 2 || split == 4 || split == 8;
    const unsigned grid = pow2 ? 8u * (unsigned)((nitems + 8 / split - 1) / (8 / split)) : (unsigned)nitems * split;
    T* dw16 = f32 ? nullptr : static_cast<T*>(DW);
    if (wk == 32) {
        if (int rc = ensure_lds<&updat16_rows_kernel<DT, 32>>(U6Geom<32>::LDS)) return rc;
        updat16_rows_kernel<DT, 32><<<grid, 64 * U6_WAVES, U6Geom<32>::LDS, st>>>(xs, es, dw16, scratch, sec, a->N, a->C, a->K, pcount, a->alpha, a->beta, split, nel);
    } else {
        if (int rc = ensure_lds<&updat16_rows_kernel<DT, 16>>(U6Geom<16>::LDS)) return rc;
        updat16_rows_kernel<DT, 16><<<grid, 64 * U6_WAVES, U6Geom<16>::LDS, st>>>(xs, es, dw16, scratch, sec, a->N, a->C, a->K, pcount, a->alpha, a->beta, split, nel);
    }
    const unsigned fgrid = (unsigned)((nel / 4 + 255) / 256);
    if (f32)          updat16_rows_finalize_kernel<DTf32><<<fgrid, 256, 0, st>>>(scratch, static_cast<float*>(DW), nel, split, a->alpha, a->beta, gate, skip_if);
    else if (scratch) updat16_rows_finalize_kernel<DT><<<fgrid, 256, 0, st>>>(scratch, dw16, nel, split, a->alpha, a->beta, gate);
    return (int)hipGetLastError();
}

template <class DT, int BS, int AXIS>
int updat_typed(const PtrList8& xs, const PtrList8& es, void* DW, const bsmm_args* a) {
    typedef typename DT::T T;
    hipStream_t st = static_cast<hipStream_t>(a->stream);
    const int N = a->N;
    bool vec_ok = aligned16(DW);
    if (AXIS == 0) {
        // 16-byte row loads need every row start aligned
        vec_ok = vec_ok && (N % (DT::is16 ? 8 : 4) == 0);
        for (int p = 0; p < a->pcount; ++p) vec_ok = vec_ok && aligned16(xs.p[p]) && aligned16(es.p[p]);
    }
    const int variant = call_variant(a);
    const bool use_valu = (BS == 8) || variant == 1 || !vec_ok;
    const bool sums_only = (a->flags & BSMM_FLAG_DW_SUMS) != 0;     // only the streaming kernel can leave raw sums
    const float* ug = (a->flags & BSMM_FLAG_GATED_DW) ? a->gate : nullptr;   // gated dw: per-block kernels only
    const bool gated = ug != nullptr;
    if constexpr (BS == 8 && DT::is16 && AXIS == 0) {
        // short minibatches (round 6, scripts/gpu_a0_updat8_sweep.py): one wave per pair of blocks, operand fragments straight from global memory
        bool al8 = aligned16(DW) && N % 8 == 0;
        for (int p = 0; p < a->pcount; ++p) al8 = al8 && aligned16(xs.p[p]) && aligned16(es.p[p]);
        // (hipGraph replays, us: the pair kernel costs 6 + 3.4e-6 per (block, minibatch column); the super-block path about 30 + (0.0022 + 2e-6 N) per
        //  super-block -- hidden 2560 dense N = 64: 29 against 41; 20480 at 1.4 %: 26.5 against 228 at N = 64, 333 against 724 at N = 1024; 4096 at 10 %:
        //  9.4 against 45.5 at N = 64, 94 against 66 at N = 1024.  Without a plan the alternative is the V_FMA kernel: always the pair kernel)
        const bool s8 = a->plan != nullptr && a->plan_magic == S8PLAN_MAGIC && a->plan_width > 0;
        const double t_pair = 6.0 + 3.4e-6 * (double)a->blocks * N * a->pcount;
        const double t_s8 = 30.0 + (double)a->plan_width * (0.0022 + 2e-6 * (double)N * a->pcount);
        if (al8 && variant == 0 && a->split == 0 && !sums_only && (!s8 || t_pair < t_s8) && U8P_ON) {
            trace(a, BSMM_K_UPDAT_BLOCK);
            const int pairs = (a->blocks + 1) / 2;
            updat8_a0_pairs_kernel<DT><<<(pairs + 3) / 4, 256, 0, st>>>(xs, es, static_cast<T*>(DW), a->lut, a->blocks, N, a->pcount, a->alpha, a->beta, ug);
            return (int)hipGetLastError();
        }
    }
    if constexpr (BS == 8 && DT::is16) {
        // bsize 8 on the matrix cores: fp32 sums of whole 32x32 super-blocks ('BSS8' plan) into the workspace, then the
        // present 8x8 parts get alpha / beta and are rounded once
        bool al = aligned16(DW) && a->C % 32 == 0 && a->K % 32 == 0 && !(AXIS == 0 && (N % 8 != 0));
        for (int p = 0; p < a->pcount; ++p) al = al && aligned16(xs.p[p]) && aligned16(es.p[p]);
        // (operands the streaming kernel's 32-bit byte offsets cannot address fall through to the generic kernels below: ADVICE r3)
        const bool s8_big = (long)N * std::max(a->C, a->K) >= (1L << 30);
        if (a->plan != nullptr && a->plan_magic == S8PLAN_MAGIC && a->plan_width > 0 && a->plan_items > 0 && !gated && al && (variant == 0 || variant == 3) &&
            !(s8_inner(a, true).plan_magic == U2PLAN_MAGIC && s8_big)) {
            const int ns = a->plan_width;
            bsmm_args b = s8_inner(a, true);
            b.flags = 0; b.gate = nullptr; b.trace = nullptr;
            b.lut = a->plan + s8_off_lut32(ns);
            if (b.plan_magic != U2PLAN_MAGIC) return BSMM_ERR_ARG;      // (the nested plan is the streaming kernel's since round 3)
            b.flags = BSMM_FLAG_DW_SUMS; b.split = 0;      // raw fp32 sums of the super-blocks at the start of the workspace
            if (int rc = launch_updat2<DT, AXIS>(xs, es, nullptr, &b, nullptr)) return rc;
            trace(a, BSMM_K_UPDAT_SUPER8);
            gather8_kernel<DT><<<ns, 256, 0, st>>>(static_cast<const float*>(a->workspace), a->plan, static_cast<T*>(DW), a->alpha, a->beta);
            return (int)hipGetLastError();
        }
    }
    if constexpr (BS == 32 && DT::is16) {
        bool al = vec_ok;   // axis 0: 16-byte row pieces (checked above); axis 1: operand base pointers
        if (AXIS == 1) {
            al = aligned16(DW);
            for (int p = 0; p < a->pcount; ++p) al = al && aligned16(xs.p[p]) && aligned16(es.p[p]);
        }
        {
            // streaming kernel ('BSU2' plan): 32-bit byte offsets inside an operand
            if (!use_valu && al && (variant == 0 || variant == 3) && a->plan != nullptr && a->plan_magic == U2PLAN_MAGIC &&
                (long)N * std::max(a->C, a->K) < (1L << 30)) {
                bool stream = true;
                if (variant == 0 && AXIS == 0 && a->split == 0 && !sums_only && !(a->flags & FLAG_INTERNAL_Q64)) {
                    // feature axis 0 (round 6; scripts/gpu_a0_updat_sweep.py, hipGraph replays): the per-block kernel costs 7 + 1.37e-5 us per (block,
                    // minibatch column) -- 12.8 against the streaming kernel's 33.9 us at the reference benchmark's hidden 2560 / N = 64, 12.8 against 46
                    // at 20480 / 1.7 %.  It is taken while that stays under 45 us, and whenever the windows are nearly empty (< 8 blocks per item: the
                    // streaming kernel then moves a 32 KiB window per chunk for a handful of blocks, 537 against 227 us at 20480 / 1.7 % / N = 2048)
                    const double t_blk0 = 7.0 + 1.37e-5 * (double)a->blocks * N * a->pcount;
                    if (a->blocks < 8L * a->plan_items || t_blk0 < 45.0) stream = false;
                }
                if (variant == 0 && AXIS == 1) {
                    // Fitted to scripts/gpu_updat_sweep.py (4096^2 20 % / 5 %, 8192^2 5 %, 2048^2 20 %, N = 128 .. 8192, us):
                    // streaming kernel: 6 + 0.40 per 16-row chunk of a workgroup's share + 0.45 per MiB of partial sums (written by the
                    // kernel, read back by the reduce pass; a block has nparts partial sums, times the slices of the last round);
                    // per-block transposing-read kernel: 8 + rounds of 512 blocks * N * r, r = 0.004 while X and DY are small
                    // (16 MiB), rising to 0.0105 at 128 MiB.
                    const U2Launch L = updat2_shape(a, gated);
                    const double chunks = (double)a->plan_items * a->pcount * ((N + 15) / 16);      // (the fit is per 16 minibatch entries)
                    double t_stream = 6.0 + chunks / L.grid * 0.40;
                    if (L.scratch && !L.flat) {
                        const int nsets = a->plan_inner & 15, longest = u2_longest_set(a), U = std::max(1, L.grid / 8);
                        const int m_last = longest % U;
                        const double sliced = longest > 0 ? (double)m_last / longest : 0.0;
                        const double mult = (8.0 / std::max(1, nsets)) * ((1.0 - sliced) + sliced * (m_last > 0 ? std::min(U / m_last, U2_MAX_SLICES) : 1));
                        t_stream += 0.45 * mult * a->blocks * 4096.0 / 1048576.0;
                    } else if (L.scratch) {
                        t_stream += 8.0;
                    }
                    const double rounds_b = std::max(1.0, std::ceil(a->blocks / (2.0 * device_cus())));
                    const double foot = (double)N * a->pcount * (a->C + a->K) * 2.0 / 1048576.0;                 // MiB of X and DY
                    const double rate = std::min(0.0105, std::max(0.004, 0.004 + 0.0065 * (foot - 16.0) / 112.0));
                    double t_blk = 8.0 + rounds_b * (double)N * a->pcount * rate;
                    // small minibatches: the one-wave-per-block kernel (updat32_a1_small_kernel; hipGraph replays at 4096^2 20 %: 5.3 / 6.6 / 9.3 /
                    // 12.1 / 18.1 / 26.8 us at N = 64 / 128 / 256 / 384 / 512 / 768 against 15.9 / 17.1 / 18.2 / 19.7 / 22.9 / 28.5 for the kernel above)
                    const long rows = (long)N * a->pcount;
                    if (rows <= UTS_NMAX) {
                        const double rounds_s = std::max(1.0, std::ceil(a->blocks / (20.0 * device_cus())));
                        t_blk = std::min(t_blk, 4.0 + rounds_s * (double)rows * (rows <= 384 ? 0.021 : 0.033));
                    }
                    stream = t_stream <= t_blk || sums_only;
                    if (a->flags & FLAG_INTERNAL_Q64) {      // quadrants of 64 x 64 blocks: the streaming kernel, or (round 6) the one-wave-per-block kernel
                        const double rounds_s = std::max(1.0, std::ceil(a->blocks / (20.0 * device_cus())));
                        const double t_small = rows <= UTS_NMAX ? 4.0 + rounds_s * (double)rows * (rows <= 384 ? 0.021 : 0.033) : 1e30;
                        stream = t_stream <= t_small;
                    }
                }
                if (stream) return launch_updat2<DT, AXIS>(xs, es, DW, a, ug);
            }
        }
        // (only the streaming kernel's summing pass leaves raw sums; the quadrant layout: that pass or the small-minibatch kernel below)
        if (sums_only || ((a->flags & FLAG_INTERNAL_Q64) && !(AXIS == 1 && variant == 0 && (long)N * a->pcount <= UTS_NMAX))) return BSMM_ERR_UNSUPPORTED;
    }
    if constexpr (BS == 32 && AXIS == 1 && DT::is16) {
        bool al = aligned16(DW);
        for (int p = 0; p < a->pcount; ++p) al = al && aligned16(xs.p[p]) && aligned16(es.p[p]);
        if (!use_valu && al && variant != 1 && (long)N * a->pcount <= UTS_NMAX) {   // small minibatches: one wave per block, no reduction (gate in its epilogue)
            if (int rc = ensure_lds<&updat32_a1_small_kernel<DT>>(UTS_LDS)) return rc;
            trace(a, BSMM_K_UPDAT_BLOCK_TR | (BSMM_KV_ONE_WAVE << 8));
            const int grid = 8 * (((a->blocks + 3) / 4 + 7) / 8);
            updat32_a1_small_kernel<DT><<<grid, 256, UTS_LDS, st>>>(xs, es, static_cast<T*>(DW), a->lut, a->blocks, N, a->C, a->K, a->pcount,
                                                                   a->alpha, a->beta, ug, (a->flags & FLAG_INTERNAL_Q64) ? 1 : 0);
            return (int)hipGetLastError();
        }
        if (a->flags & FLAG_INTERNAL_Q64) return BSMM_ERR_UNSUPPORTED;
        if (!use_valu && !gated && al && variant != 1) {   // LDS-DMA + transposing-read kernel
            if (int rc = ensure_lds<&updat32_a1_tr_kernel<DT>>(UT_LDS)) return rc;
            trace(a, BSMM_K_UPDAT_BLOCK_TR);
            const int grid = 8 * ((a->blocks + 7) / 8);
            updat32_a1_tr_kernel<DT><<<grid, 256, UT_LDS, st>>>(xs, es, static_cast<T*>(DW), a->lut, a->blocks, N, a->C, a->K, a->pcount,
                                                              a->alpha, a->beta);
            return (int)hipGetLastError();
        }
    }
    if constexpr (BS == 16 && DT::is16) {
        bool al16 = aligned16(DW) && (AXIS == 1 || N % 8 == 0);
        for (int p = 0; p < a->pcount; ++p) al16 = al16 && aligned16(xs.p[p]) && aligned16(es.p[p]);
        if constexpr (AXIS == 0) {
            // a GATED call can take the row-owner kernel too (round 5: its finalize pass applies the gate); the windowed kernel has no gate
            if (!use_valu && gated && al16 && (variant == 0 || variant == 3) && a->plan != nullptr && a->plan_magic == UPLAN_MAGIC && a->plan_inner > 0 &&
                (long)N * std::max(a->C, a->K) < (1L << 31)) {
                const int rc = launch_updat16_rows<DT>(xs, es, DW, a, 0, nullptr, ug);
                if (rc != BSMM_ERR_UNSUPPORTED) return rc;
            }
        }
        bool blk16 = false;
        if constexpr (AXIS == 0) {
            // (round 6, the same sweep: the per-block kernel costs 7 + 6.6e-6 us per (block, minibatch column); the windowed kernel about 5 + 0.03 per
            //  window + 1.1e-4 per (window, column).  Nearly empty windows -- the reference benchmark's sparse shapes: 73 against 18.6 us at 20480 /
            //  1.5 % / N = 64 -- always take the per-block kernel)
            if (variant == 0 && a->split == 0 && a->plan != nullptr && a->plan_items > 0) {      // (a caller who names a split wants the plan kernels)
                const double t_blk0 = 7.0 + 6.6e-6 * (double)a->blocks * N * a->pcount;
                const double t_win0 = 5.0 + 0.03 * a->plan_items + 1.1e-4 * (double)a->plan_items * N * a->pcount;
                blk16 = a->blocks < 8L * a->plan_items || t_blk0 < 0.6 * t_win0;
            }
        }
        if (!use_valu && !gated && !blk16 && al16 && (variant == 0 || variant == 3) && a->plan != nullptr && a->plan_items > 0) {   // windowed, 16x16 blocks
            if (a->plan_magic != UPLAN_MAGIC || (a->plan_width & 255) != UW16 || (a->plan_waves & 255) != UP_WAVES) return BSMM_ERR_ARG;   // (bits 8..: the 'BSU6' section's window width / items)
            if constexpr (AXIS == 0) {
                // row-owner kernel ('BSU6' section, bsmm_updat16_rows.h): half the bytes per block of the windowed kernel, 64 (or fewer) work items --
                // taken when every workgroup gets at least U6_MIN_CHUNKS chunks of 64 minibatch entries
                if (a->plan_inner > 0 && (long)N * std::max(a->C, a->K) < (1L << 31)) {
                    const int rc = launch_updat16_rows<DT>(xs, es, DW, a);
                    if (rc != BSMM_ERR_UNSUPPORTED) return rc;
                }
            }
            if (int rc = ensure_lds<&updat16_win_kernel<DT, AXIS>>(2 * UWN_SLOT)) return rc;
            trace(a, BSMM_K_UPDAT16_WIN);
            const int nitems = a->plan_items;
            const int nchunks = (N + 63) / 64;
            const int split = updat_split(a, nitems, nchunks);
            float* scratch = nullptr;
            const size_t nel = (size_t)a->blocks * 256;
            if (split > 1) {
                if (!a->workspace || a->workspace_bytes < nel * sizeof(float) || !aligned16(a->workspace)) return BSMM_ERR_WORKSPACE;
                scratch = static_cast<float*>(a->workspace);
                hipError_t e = hipMemsetAsync(scratch, 0, nel * sizeof(float), st);
                if (e != hipSuccess) return (int)e;
            }
            updat16_win_kernel<DT, AXIS><<<dim3(nitems, split), 512, 2 * UWN_SLOT, st>>>(xs, es, static_cast<T*>(DW), scratch, a->plan, N, a->C,
                                                                                      a->K, a->pcount, a->alpha, a->beta);
            if (split > 1)
                updat_finalize_kernel<DT><<<(unsigned)((nel / 4 + 255) / 256), 256, 0, st>>>(scratch, static_cast<T*>(DW), nel, a->alpha, a->beta);
            return (int)hipGetLastError();
        }
    }
    if constexpr (BS == 16 && AXIS == 1 && DT::is16) {
        bool al = aligned16(DW);
        for (int p = 0; p < a->pcount; ++p) al = al && aligned16(xs.p[p]) && aligned16(es.p[p]);
        if (!use_valu && !gated && al && variant != 1) {   // LDS-DMA + transposing-read kernel, 16x16 blocks
            if (int rc = ensure_lds<&updat16_a1_tr_kernel<DT>>(UT16_LDS)) return rc;
            trace(a, BSMM_K_UPDAT_BLOCK_TR);
            const int grid = 8 * ((a->blocks + 7) / 8);
            updat16_a1_tr_kernel<DT><<<grid, 256, UT16_LDS, st>>>(xs, es, static_cast<T*>(DW), a->lut, a->blocks, N, a->C, a->K, a->pcount,
                                                                a->alpha, a->beta);
            return (int)hipGetLastError();
        }
    }
    if constexpr ((BS == 32 || BS == 16) && AXIS == 0 && DT::is16) {
        // a few dozen minibatch columns (round 6): one wave per block, fragments straight from global memory (bsmm_updat.h::updat_a0_wave_kernel)
        bool alw = aligned16(DW) && N % 8 == 0;
        for (int p = 0; p < a->pcount; ++p) alw = alw && aligned16(xs.p[p]) && aligned16(es.p[p]);
        // (hipGraph replays at the reference benchmark's shapes, N = 64: bsize 32 10.6 against 12.6 us, bsize 16 15 - 16 against 17 - 19; from 128 columns
        //  on the four-wave kernels below are as fast or faster at bsize 32)
        if (!use_valu && alw && variant == 0 && a->split == 0 && (long)N * a->pcount <= (BS == 32 ? UAW_NMAX / 2 : UAW_NMAX)) {
            trace(a, BSMM_K_UPDAT_BLOCK);
            updat_a0_wave_kernel<DT, BS><<<(a->blocks + 3) / 4, 256, 0, st>>>(xs, es, static_cast<T*>(DW), a->lut, a->blocks, N, a->pcount, a->alpha, a->beta, ug);
            return (int)hipGetLastError();
        }
    }
    trace(a, use_valu ? BSMM_K_UPDAT_VALU : BSMM_K_UPDAT_BLOCK);
    if (use_valu) {
        updat_valu_kernel<DT, BS, AXIS><<<a->blocks, 256, 0, st>>>(xs, es, static_cast<T*>(DW), a->lut, a->blocks, N, a->C,
                                                                   a->K, a->pcount, a->alpha, a->beta, ug);
    } else if constexpr (BS == 32) {
        const int grid = 8 * ((a->blocks + 7) / 8);
        updat32_kernel<DT, AXIS><<<grid, 256, 0, st>>>(xs, es, static_cast<T*>(DW), a->lut, a->blocks, N, a->C, a->K,
                                                       a->pcount, a->alpha, a->beta, ug);
    } else if constexpr (BS == 16) {
        const int grid = 8 * ((a->blocks + 7) / 8);
        updat16_kernel<DT, AXIS><<<grid, 256, 0, st>>>(xs, es, static_cast<T*>(DW), a->lut, a->blocks, N, a->C, a->K,
                                                       a->pcount, a->alpha, a->beta, ug);
    }
    return (int)hipGetLastError();
}

template <class DT>
int updat_dt(const PtrList8& xs, const PtrList8& es, void* DW, const bsmm_args* a) {
#define BSMM_CASE(BS, AX) \
    if (a->bsize == BS && a->axis == AX) return updat_typed<DT, BS, AX>(xs, es, DW, a);
    BSMM_CASE(32, 0) BSMM_CASE(32, 1) BSMM_CASE(16, 0) BSMM_CASE(16, 1) BSMM_CASE(8, 0) BSMM_CASE(8, 1)
#undef BSMM_CASE
    return BSMM_ERR_UNSUPPORTED;
}

// Non-finite activations in the fp32 split paths below.  A finite x beyond the bf16 range still splits exactly
// (bsmm_xcols.h::split3), an Inf or NaN cannot: its products with the partner's three pieces come out as Inf or NaN as their signs fall,
// where the fp32 kernels -- and the reference's fp32 kernels -- give what IEEE gives for the unsplit product.  So the split kernels raise a
// flag (4 bytes behind the pieces in the workspace), the finalize pass leaves DW alone when it is set, and the per-block fp32 kernel of
// the shape -- always launched, the flag is its condition: ~2 us of empty workgroups on finite inputs -- then computes the call the way
// fp32 ran before round 4.  (A call that asks for the raw sums, BSMM_FLAG_DW_SUMS, gets the split path's sums as they are.)
constexpr size_t F32_SPLIT_FLAG_BYTES = 16;
template <int BS, int AXIS>
int f32_split_repair(const void* const* X, const void* const* DY, void* DW, const bsmm_args* a, const int32_t* flag) {
    hipStream_t st = static_cast<hipStream_t>(a->stream);
    PtrList8 xs, es;
    for (int p = 0; p < 8; ++p) { xs.p[p] = p == 0 ? X[0] : nullptr; es.p[p] = p == 0 ? DY[0] : nullptr; }
    const float* ug = (a->flags & BSMM_FLAG_GATED_DW) ? a->gate : nullptr;
    const int grid = 8 * ((a->blocks + 7) / 8);
    if constexpr (BS == 32)
        updat32_kernel<DTf32, AXIS><<<grid, 256, 0, st>>>(xs, es, static_cast<float*>(DW), a->lut, a->blocks, a->N, a->C, a->K, 1, a->alpha, a->beta, ug, flag);
    else if constexpr (BS == 16)
        updat16_kernel<DTf32, AXIS><<<grid, 256, 0, st>>>(xs, es, static_cast<float*>(DW), a->lut, a->blocks, a->N, a->C, a->K, 1, a->alpha, a->beta, ug, flag);
    else
        updat_valu_kernel<DTf32, BS, AXIS><<<a->blocks, 256, 0, st>>>(xs, es, static_cast<float*>(DW), a->lut, a->blocks, a->N, a->C, a->K, 1, a->alpha, a->beta, ug, flag);
    return (int)hipGetLastError();
}

// fp32 weight gradient on feature axis 1, bsize 32, with a streaming plan (round 4).  The per-block fp32 kernel gathers its fragments at a
// stride of C elements there (25 TF: 2.2 ms at the bench shape).  Instead: X and DY are split into three bf16 pieces each (exact:
// bsmm_xcols.h) and the SIX significant piece products  x3 y1, x2 y2, x1 y3, x2 y1, x1 y2, x1 y1  (smallest first; the other three are below
// 2^-26 of the product) go through the bf16 streaming kernel as six (x, dy) PAIRS of ONE launch -- its pair list is what the reference's
// Plist<T, 8> is (src/gpu_types.h:167-170) -- with fp32 sums, and the finalize pass writes the fp32 DW with alpha / beta / gate.  bf16 x bf16
// products are exact in fp32 and the accumulation is fp32: the result has the accuracy of an fp32 matrix-core product.
// Workspace: [what the bf16 call needs][pieces of X: 3 N C bf16][pieces of DY: 3 N K bf16][the non-finite flag, 16 bytes].
inline size_t updat_f32_split_inner_bytes(const bsmm_args* a) {
    bsmm_args b = *a;
    b.dtype = BSMM_BF16; b.pcount = 6; b.flags = BSMM_FLAG_DW_SUMS; b.split = 0; b.gate = nullptr;
    return round16(bsmm_workspace_bytes(BSMM_OP_UPDAT, &b));
}
inline bool updat_f32_split_applies(const bsmm_args* a) {
    return a->dtype == BSMM_F32 && a->bsize == 32 && a->axis == 1 && a->plan && a->plan_magic == U2PLAN_MAGIC && a->pcount == 1 && a->split == 0 &&
           a->C % 32 == 0 && a->K % 32 == 0 && (long)a->N * std::max(a->C, a->K) < (1L << 30) && call_variant(a) != 1 && call_variant(a) != 2;
}
int updat32_f32_split(const void* const* X, const void* const* DY, void* DW, const bsmm_args* a) {
    hipStream_t st = static_cast<hipStream_t>(a->stream);
    const bool sums_only = (a->flags & BSMM_FLAG_DW_SUMS) != 0;
    const size_t inner = updat_f32_split_inner_bytes(a), nx = (size_t)a->N * a->C, ne = (size_t)a->N * a->K;
    if (!a->workspace || !aligned16(a->workspace) || a->workspace_bytes < inner + 6 * (nx + ne) + F32_SPLIT_FLAG_BYTES) return BSMM_ERR_WORKSPACE;
    if (!aligned16(X[0]) || !aligned16(DY[0]) || (DW && !aligned16(DW))) return BSMM_ERR_ARG;
    uint16_t* xp = reinterpret_cast<uint16_t*>(static_cast<char*>(a->workspace) + inner);
    uint16_t* ep = xp + 3 * nx;
    int32_t* flag = reinterpret_cast<int32_t*>(ep + 3 * ne);
    if (hipError_t e = hipMemsetAsync(flag, 0, F32_SPLIT_FLAG_BYTES, st); e != hipSuccess) return (int)e;
    split3_x_kernel<<<(unsigned)((nx / 8 + 255) / 256), 256, 0, st>>>(static_cast<const float*>(X[0]), xp, nx, flag);
    split3_x_kernel<<<(unsigned)((ne / 8 + 255) / 256), 256, 0, st>>>(static_cast<const float*>(DY[0]), ep, ne, flag);
    bsmm_args b = *a;
    b.dtype = BSMM_BF16; b.pcount = 6; b.flags = BSMM_FLAG_DW_SUMS; b.split = 0; b.gate = nullptr; b.trace = nullptr;
    b.workspace_bytes = inner;
    static const int xi[6] = {2, 1, 0, 1, 0, 0}, ei[6] = {0, 1, 2, 0, 1, 0};      // piece indices of the six products, smallest first
    PtrList8 xs, es;
    for (int p = 0; p < 8; ++p) {
        xs.p[p] = p < 6 ? xp + xi[p] * nx : nullptr;
        es.p[p] = p < 6 ? ep + ei[p] * ne : nullptr;
    }
    if (int rc = launch_updat2<DTbf16, 1>(xs, es, nullptr, &b, nullptr)) return rc;
    trace(a, BSMM_K_UPDAT_STREAM);
    if (sums_only) return (int)hipGetLastError();      // the raw fp32 sums stay at the start of the workspace, as for the 16-bit types
    const float* ug = (a->flags & BSMM_FLAG_GATED_DW) ? a->gate : nullptr;
    const size_t nel = (size_t)a->blocks * 1024;
    updat_finalize_gated_kernel<DTf32><<<(unsigned)((nel / 4 + 255) / 256), 256, 0, st>>>(static_cast<const float*>(a->workspace), static_cast<float*>(DW), nel, 1024,
                                                                                         a->alpha, a->beta, ug, flag);
    return f32_split_repair<32, 1>(X, DY, DW, a, flag);
}

// The same for bsize 8 (either feature axis): the six piece products run as six pairs of the streaming launch over the 32x32 SUPER-blocks of
// the 'BSS8' plan (bsmm_super8.h), gather8_f32_kernel writes the present 8x8 parts in fp32.  (The V_FMA kernel it replaces: 5.9 / 2.0 ms on
// feature axis 1 / 0 at 4096^2, 10 %, N = 8192.)
inline bsmm_args updat8_f32_inner(const bsmm_args* a) {
    bsmm_args b = s8_inner(a, true);
    b.dtype = BSMM_BF16; b.pcount = 6; b.flags = BSMM_FLAG_DW_SUMS; b.split = 0; b.gate = nullptr; b.trace = nullptr;
    b.lut = a->plan ? a->plan + s8_off_lut32(a->plan_width) : nullptr;
    return b;
}
inline bool updat8_f32_split_applies(const bsmm_args* a) {
    if (!(a->dtype == BSMM_F32 && a->bsize == 8 && a->plan && a->plan_magic == S8PLAN_MAGIC && a->plan_width > 0 && a->plan_items > 0)) return false;
    if (((a->plan_inner >> 8) & 7) != 2) return false;                                  // the nested plan must be the streaming kernel's
    return a->pcount == 1 && a->split == 0 && !(a->flags & (BSMM_FLAG_GATED_DW | BSMM_FLAG_DW_SUMS)) && a->C % 32 == 0 && a->K % 32 == 0 &&
           !(a->axis == 0 && a->N % 8 != 0) && (long)a->N * std::max(a->C, a->K) < (1L << 30) && call_variant(a) != 1 && call_variant(a) != 2;
}
inline size_t updat8_f32_inner_bytes(const bsmm_args* a) {
    bsmm_args b = updat8_f32_inner(a);
    return round16(bsmm_workspace_bytes(BSMM_OP_UPDAT, &b));
}
int updat8_f32_split(const void* const* X, const void* const* DY, void* DW, const bsmm_args* a) {
    hipStream_t st = static_cast<hipStream_t>(a->stream);
    const size_t inner = updat8_f32_inner_bytes(a), nx = (size_t)a->N * a->C, ne = (size_t)a->N * a->K;
    if (!a->workspace || !aligned16(a->workspace) || a->workspace_bytes < inner + 6 * (nx + ne) + F32_SPLIT_FLAG_BYTES) return BSMM_ERR_WORKSPACE;
    if (!aligned16(X[0]) || !aligned16(DY[0]) || !aligned16(DW)) return BSMM_ERR_ARG;
    uint16_t* xp = reinterpret_cast<uint16_t*>(static_cast<char*>(a->workspace) + inner);
    uint16_t* ep = xp + 3 * nx;
    int32_t* flag = reinterpret_cast<int32_t*>(ep + 3 * ne);
    if (hipError_t e = hipMemsetAsync(flag, 0, F32_SPLIT_FLAG_BYTES, st); e != hipSuccess) return (int)e;
    split3_x_kernel<<<(unsigned)((nx / 8 + 255) / 256), 256, 0, st>>>(static_cast<const float*>(X[0]), xp, nx, flag);
    split3_x_kernel<<<(unsigned)((ne / 8 + 255) / 256), 256, 0, st>>>(static_cast<const float*>(DY[0]), ep, ne, flag);
    bsmm_args b = updat8_f32_inner(a);
    b.workspace_bytes = inner;
    static const int xi[6] = {2, 1, 0, 1, 0, 0}, ei[6] = {0, 1, 2, 0, 1, 0};      // the six products, smallest first (see updat32_f32_split)
    PtrList8 xs, es;
    for (int p = 0; p < 8; ++p) {
        xs.p[p] = p < 6 ? xp + xi[p] * nx : nullptr;
        es.p[p] = p < 6 ? ep + ei[p] * ne : nullptr;
    }
    const int rc = a->axis == 1 ? launch_updat2<DTbf16, 1>(xs, es, nullptr, &b, nullptr) : launch_updat2<DTbf16, 0>(xs, es, nullptr, &b, nullptr);
    if (rc) return rc;
    trace(a, BSMM_K_UPDAT_SUPER8);
    gather8_f32_kernel<<<a->plan_width, 256, 0, st>>>(static_cast<const float*>(a->workspace), a->plan, static_cast<float*>(DW), a->alpha, a->beta, flag);
    return a->axis == 1 ? f32_split_repair<8, 1>(X, DY, DW, a, flag) : f32_split_repair<8, 0>(X, DY, DW, a, flag);
}

// ... and for bsize 16 on feature axis 1 on the windowed kernel: six pairs of one launch, raw fp32 sums by its scratch path, fp32 finalize:
// 1.44 -> 1.12 ms at 4096^2, 10 %, N = 8192.  (Feature axis 0 measured too: 1.05 ms against 1.01 for the per-block fp32 kernel, whose
// fragments are contiguous there -- not taken.)
inline bool updat16_f32_split_applies(const bsmm_args* a) {
    return a->dtype == BSMM_F32 && a->bsize == 16 && a->axis == 1 && a->plan && a->plan_magic == UPLAN_MAGIC && (a->plan_width & 255) == UW16 && (a->plan_waves & 255) == UP_WAVES &&
           a->plan_items > 0 && a->pcount == 1 && a->split == 0 && !(a->flags & BSMM_FLAG_DW_SUMS) && a->C % 16 == 0 && a->K % 16 == 0 &&
           !(a->axis == 0 && a->N % 8 != 0) && call_variant(a) != 1 && call_variant(a) != 2;
}
inline size_t updat16_f32_sums_bytes(const bsmm_args* a) { return round16((size_t)a->blocks * 256 * sizeof(float)); }
template <int AXIS>
int updat16_f32_split(const void* const* X, const void* const* DY, void* DW, const bsmm_args* a) {
    hipStream_t st = static_cast<hipStream_t>(a->stream);
    const size_t sums_b = updat16_f32_sums_bytes(a), nx = (size_t)a->N * a->C, ne = (size_t)a->N * a->K, nel = (size_t)a->blocks * 256;
    if (!a->workspace || !aligned16(a->workspace) || a->workspace_bytes < sums_b + 6 * (nx + ne) + F32_SPLIT_FLAG_BYTES) return BSMM_ERR_WORKSPACE;
    if (!aligned16(X[0]) || !aligned16(DY[0]) || !aligned16(DW)) return BSMM_ERR_ARG;
    float* sums = static_cast<float*>(a->workspace);
    uint16_t* xp = reinterpret_cast<uint16_t*>(static_cast<char*>(a->workspace) + sums_b);
    uint16_t* ep = xp + 3 * nx;
    int32_t* flag = reinterpret_cast<int32_t*>(ep + 3 * ne);
    if (hipError_t e = hipMemsetAsync(flag, 0, F32_SPLIT_FLAG_BYTES, st); e != hipSuccess) return (int)e;
    split3_x_kernel<<<(unsigned)((nx / 8 + 255) / 256), 256, 0, st>>>(static_cast<const float*>(X[0]), xp, nx, flag);
    split3_x_kernel<<<(unsigned)((ne / 8 + 255) / 256), 256, 0, st>>>(static_cast<const float*>(DY[0]), ep, ne, flag);
    if (hipError_t e = hipMemsetAsync(sums, 0, nel * sizeof(float), st); e != hipSuccess) return (int)e;
    static const int xi[6] = {2, 1, 0, 1, 0, 0}, ei[6] = {0, 1, 2, 0, 1, 0};      // the six products, smallest first (see updat32_f32_split)
    PtrList8 xs, es;
    for (int p = 0; p < 8; ++p) {
        xs.p[p] = p < 6 ? xp + xi[p] * nx : nullptr;
        es.p[p] = p < 6 ? ep + ei[p] * ne : nullptr;
    }
    if (int rc = ensure_lds<&updat16_win_kernel<DTbf16, AXIS>>(2 * UWN_SLOT)) return rc;
    trace(a, BSMM_K_UPDAT16_WIN);
    const int split = updat_split(a, a->plan_items, (a->N + 63) / 64);
    updat16_win_kernel<DTbf16, AXIS><<<dim3(a->plan_items, split), 512, 2 * UWN_SLOT, st>>>(xs, es, nullptr, sums, a->plan, a->N, a->C, a->K, 6, 1.f, 0.f);
    const float* ug = (a->flags & BSMM_FLAG_GATED_DW) ? a->gate : nullptr;
    updat_finalize_gated_kernel<DTf32><<<(unsigned)((nel / 4 + 255) / 256), 256, 0, st>>>(sums, static_cast<float*>(DW), nel, 256, a->alpha, a->beta, ug, flag);
    return f32_split_repair<16, AXIS>(X, DY, DW, a, flag);
}

// ... and on feature axis 0 (round 5) the six pairs go through the row-owner kernel (bsmm_updat16_rows.h) where it pays -- its images take the
// place of the sums: [U6_MAX_SPLIT images][pieces of X][pieces of DY][flag].  BSMM_ERR_UNSUPPORTED (short minibatches, few windows): the
// caller takes the per-block fp32 kernel, as before.
inline bool updat16_f32_rows_applies(const bsmm_args* a) {
    return a->dtype == BSMM_F32 && a->bsize == 16 && a->axis == 0 && a->plan && a->plan_magic == UPLAN_MAGIC && (a->plan_width & 255) == UW16 && (a->plan_waves & 255) == UP_WAVES &&
           a->plan_inner > 0 && a->pcount == 1 && !(a->flags & BSMM_FLAG_DW_SUMS) && a->N % 8 == 0 && (long)a->N * std::max(a->C, a->K) < (1L << 31) &&
           call_variant(a) != 1 && call_variant(a) != 2;
}
inline size_t updat16_f32_rows_images_bytes(const bsmm_args* a) { return round16((size_t)U6_MAX_SPLIT * a->blocks * 256 * sizeof(float)); }
// will the call run on the row-owner kernel?  ONE rule for the dispatch and for bsmm_workspace_bytes (the sizing used to reserve the
// images and the pieces -- several hundred MB -- for shapes the dispatch then sent to the per-block kernel, which needs no workspace)
inline bool updat16_f32_rows_taken(const bsmm_args* a) {
    if (!updat16_f32_rows_applies(a)) return false;
    if (call_variant(a) == 3) return true;
    if (a->N < 256) return false;
    const int nitems = a->plan_waves >> 8, nchunks = (a->N + 63) / 64, cus = device_cus();
    int split = 1;
    while (split < U6_MAX_SPLIT && nitems * split < cus) split *= 2;
    return a->split > 0 || (4L * nitems * split >= 3L * cus && 6L * nchunks / split >= (split >= 4 ? 16 : 8));
}
int updat16_f32_rows(const void* const* X, const void* const* DY, void* DW, const bsmm_args* a) {
    hipStream_t st = static_cast<hipStream_t>(a->stream);
    const size_t img_b = updat16_f32_rows_images_bytes(a), nx = (size_t)a->N * a->C, ne = (size_t)a->N * a->K;
    if (!a->workspace || !aligned16(a->workspace) || a->workspace_bytes < img_b + 6 * (nx + ne) + F32_SPLIT_FLAG_BYTES) return BSMM_ERR_WORKSPACE;
    if (!aligned16(X[0]) || !aligned16(DY[0]) || !aligned16(DW)) return BSMM_ERR_ARG;
    if (!updat16_f32_rows_taken(a)) return BSMM_ERR_UNSUPPORTED;      // (asked before the pieces are made)
    float* images = static_cast<float*>(a->workspace);
    uint16_t* xp = reinterpret_cast<uint16_t*>(static_cast<char*>(a->workspace) + img_b);
    uint16_t* ep = xp + 3 * nx;
    int32_t* flag = reinterpret_cast<int32_t*>(ep + 3 * ne);
    if (hipError_t e = hipMemsetAsync(flag, 0, F32_SPLIT_FLAG_BYTES, st); e != hipSuccess) return (int)e;
    split3_x_kernel<<<(unsigned)((nx / 8 + 255) / 256), 256, 0, st>>>(static_cast<const float*>(X[0]), xp, nx, flag);
    split3_x_kernel<<<(unsigned)((ne / 8 + 255) / 256), 256, 0, st>>>(static_cast<const float*>(DY[0]), ep, ne, flag);
    static const int xi[6] = {2, 1, 0, 1, 0, 0}, ei[6] = {0, 1, 2, 0, 1, 0};      // the six products, smallest first (see updat32_f32_split)
    PtrList8 xs, es;
    for (int p = 0; p < 8; ++p) {
        xs.p[p] = p < 6 ? xp + xi[p] * nx : nullptr;
        es.p[p] = p < 6 ? ep + ei[p] * ne : nullptr;
    }
    const float* ug = (a->flags & BSMM_FLAG_GATED_DW) ? a->gate : nullptr;
    if (int rc = launch_updat16_rows<DTbf16>(xs, es, DW, a, 6, images, ug, flag)) return rc;
    return f32_split_repair<16, 0>(X, DY, DW, a, flag);
}

int updat64(const void* const* X, const void* const* DY, void* DW, const bsmm_args* a) {
    // the streaming bsize-32 kernel on the quadrants.  16-bit types with the streaming plan only (what the reference runs bsize 64 in: fp16
    // tensor cores).
    if (!a->plan) return BSMM_ERR_UNSUPPORTED;
    bsmm_args b = b64_inner(a, true);
    if (a->dtype == BSMM_F32 || b.plan_magic != U2PLAN_MAGIC) return BSMM_ERR_UNSUPPORTED;
    const bool sums_only = (a->flags & BSMM_FLAG_DW_SUMS) != 0;
    if (sums_only) return BSMM_ERR_UNSUPPORTED;                  // (the raw sums of a bsize-64 call would be in quadrant order)
    // (round 5: the streaming kernel's summing pass writes every quadrant into its place in the 64 x 64 block with alpha / beta / the block's
    //  gate and ONE rounding -- the separate pass over the quadrant sums, b64_finalize_kernel: 20 us at the bench shape, is gone)
    b.flags = a->flags | FLAG_INTERNAL_Q64;
    b.gate = a->gate; b.alpha = a->alpha; b.beta = a->beta;
    b.workspace = a->workspace; b.workspace_bytes = a->workspace_bytes;
    if (!a->workspace || a->workspace_bytes < bsmm_workspace_bytes(BSMM_OP_UPDAT, &b)) return BSMM_ERR_WORKSPACE;
    tl_inside_updat64 = true;
    const int rc = bsmm_updat(X, DY, DW, &b);
    tl_inside_updat64 = false;
    return rc;
}

}  // namespace

namespace {
template <class F>
int l2_by_types(int xd, int yd, int bsize, F&& f) {
    auto with_bs = [&](auto tx, auto ty) {
        switch (bsize) {
            case 8: return f(tx, ty, std::integral_constant<int, 8>{});
            case 16: return f(tx, ty, std::integral_constant<int, 16>{});
            case 32: return f(tx, ty, std::integral_constant<int, 32>{});
            default: return (int)BSMM_ERR_UNSUPPORTED;
        }
    };
    auto with_y = [&](auto tx) {
        if (yd == BSMM_F32) return with_bs(tx, DTf32{});
        if (yd == BSMM_F16) return with_bs(tx, DTf16{});
        if (yd == BSMM_BF16) return with_bs(tx, DTbf16{});
        return (int)BSMM_ERR_UNSUPPORTED;
    };
    if (xd == BSMM_F32) return with_y(DTf32{});
    if (xd == BSMM_F16) return with_y(DTf16{});
    if (xd == BSMM_BF16) return with_y(DTbf16{});
    return (int)BSMM_ERR_UNSUPPORTED;
}
}  // namespace

extern "C" {

int bsmm_fprop(const void* X, const void* W, void* Y, const bsmm_args* args) { return xprop(true, X, W, Y, args); }

int bsmm_bprop(const void* DY, const void* W, void* DX, const bsmm_args* args) { return xprop(false, DY, W, DX, args); }

int bsmm_updat(const void* const* X, const void* const* DY, void* DW, const bsmm_args* a) {
    int rc = check_common(a);
    if (rc) return rc;
    if (!X || !DY || (!DW && !(a->flags & BSMM_FLAG_DW_SUMS))) return BSMM_ERR_ARG;      // (DW is not written in sums mode)
    if ((a->flags & FLAG_INTERNAL_Q64) && !tl_inside_updat64) return BSMM_ERR_ARG;       // (the library's own bit)
    if (a->pcount < 1 || a->pcount > 8) return BSMM_ERR_ARG;
    if (a->bsize == 64 && !a->plan) return BSMM_ERR_UNSUPPORTED;
    if ((rc = check_plan(true, a))) return rc;
    if (a->bsize == 64) return DW ? updat64(X, DY, DW, a) : (int)BSMM_ERR_ARG;
    if (a->dtype == BSMM_F32 && a->plan) {      // fp32 with a (streaming) plan: the bf16-split path where it applies, else the kernels without a plan
        // (minibatches of a few rows: two split launches + a six-pair stream cost more than the per-block kernel)
        if (updat_f32_split_applies(a) && (a->N >= 256 || call_variant(a) == 3 || (a->flags & BSMM_FLAG_DW_SUMS))) return updat32_f32_split(X, DY, DW, a);
        if (updat8_f32_split_applies(a) && (a->N >= 256 || call_variant(a) == 3)) return updat8_f32_split(X, DY, DW, a);
        if (updat16_f32_split_applies(a) && (a->N >= 256 || call_variant(a) == 3)) return updat16_f32_split<1>(X, DY, DW, a);
        if (updat16_f32_rows_taken(a)) {
            const int rc16 = updat16_f32_rows(X, DY, DW, a);
            if (rc16 != BSMM_ERR_UNSUPPORTED) return rc16;
        }
        if (a->flags & BSMM_FLAG_DW_SUMS) return BSMM_ERR_UNSUPPORTED;
        bsmm_args b = *a;
        b.plan = nullptr; b.plan_magic = b.plan_width = b.plan_waves = b.plan_items = b.plan_inner = 0;
        return bsmm_updat(X, DY, DW, &b);
    }
    if ((a->flags & BSMM_FLAG_DW_SUMS) && !(a->bsize == 32 && a->dtype != BSMM_F32 && a->plan && a->plan_magic == U2PLAN_MAGIC))
        return BSMM_ERR_UNSUPPORTED;
    PtrList8 xs, es;
    for (int p = 0; p < 8; ++p) {
        xs.p[p] = p < a->pcount ? X[p] : nullptr;
        es.p[p] = p < a->pcount ? DY[p] : nullptr;
        if (p < a->pcount && (!xs.p[p] || !es.p[p])) return BSMM_ERR_ARG;
    }
    switch (a->dtype) {
        case BSMM_F32:  return updat_dt<DTf32>(xs, es, DW, a);
        case BSMM_F16:  return updat_dt<DTf16>(xs, es, DW, a);
        case BSMM_BF16: return updat_dt<DTbf16>(xs, es, DW, a);
    }
    return BSMM_ERR_UNSUPPORTED;
}

size_t bsmm_prepared_bytes(int op, const bsmm_args* a) {
    if (!a || (op != BSMM_OP_FPROP && op != BSMM_OP_BPROP)) return 0;
    if (a->bsize == 64) return (a->axis == 1 && a->blocks > 0 && a->plan && a->plan_magic == B64PLAN_MAGIC) ? b64_w_bytes(a) : 0;   // the quadrant copy of W
    if (a->dtype == BSMM_F32 && a->bsize == 32 && a->plan && a->plan_magic == XCPLAN_MAGIC && a->plan_width == XS_G && a->blocks > 0) return xcols_w_bytes(a);
    return 0;
}

int bsmm_prepare_weights(int op, const void* W, void* prepared, const bsmm_args* a) {
    if (!a || !W || !prepared || (op != BSMM_OP_FPROP && op != BSMM_OP_BPROP)) return BSMM_ERR_ARG;
    if (bsmm_prepared_bytes(op, a) == 0) return BSMM_ERR_UNSUPPORTED;
    if (!aligned16(W) || !aligned16(prepared)) return BSMM_ERR_ARG;
    hipStream_t st = static_cast<hipStream_t>(a->stream);
    if (a->bsize == 64) {      // (one image per op: bprop numbers the quadrants the other way round, see bsmm_b64.h)
        const int swap = op == BSMM_OP_BPROP;
        if (elem_size(a->dtype) == 4) b64_split_kernel<4><<<a->blocks, 256, 0, st>>>(static_cast<const unsigned char*>(W), static_cast<unsigned char*>(prepared), a->blocks, swap);
        else                          b64_split_kernel<2><<<a->blocks, 256, 0, st>>>(static_cast<const unsigned char*>(W), static_cast<unsigned char*>(prepared), a->blocks, swap);
        return (int)hipGetLastError();
    }
    if (op == BSMM_OP_FPROP) split3_w_kernel<true><<<a->blocks, 256, 0, st>>>(static_cast<const float*>(W), static_cast<uint16_t*>(prepared), a->blocks);
    else                     split3_w_kernel<false><<<a->blocks, 256, 0, st>>>(static_cast<const float*>(W), static_cast<uint16_t*>(prepared), a->blocks);
    return (int)hipGetLastError();
}

int bsmm_updat_finalize(const float* sums, void* DW, const float* gate, int32_t blocks, int32_t bsize, int32_t dtype, float alpha, float beta,
                        void* stream) {
    if (!sums || !DW || blocks <= 0) return BSMM_ERR_ARG;
    if (bsize != 8 && bsize != 16 && bsize != 32) return BSMM_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(sums) & 15) || (reinterpret_cast<uintptr_t>(DW) & 7)) return BSMM_ERR_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t nel = (size_t)blocks * bsize * bsize;
    const unsigned grid = (unsigned)((nel / 4 + 255) / 256);
    switch (dtype) {
        case BSMM_F16:  updat_finalize_gated_kernel<DTf16><<<grid, 256, 0, st>>>(sums, static_cast<uint16_t*>(DW), nel, bsize * bsize, alpha, beta, gate); break;
        case BSMM_BF16: updat_finalize_gated_kernel<DTbf16><<<grid, 256, 0, st>>>(sums, static_cast<uint16_t*>(DW), nel, bsize * bsize, alpha, beta, gate); break;
        case BSMM_F32:  updat_finalize_gated_kernel<DTf32><<<grid, 256, 0, st>>>(sums, static_cast<float*>(DW), nel, bsize * bsize, alpha, beta, gate); break;
        default: return BSMM_ERR_UNSUPPORTED;
    }
    return (int)hipGetLastError();
}

int bsmm_l2_normalize(void* y, float* sum_sqr, const void* x, const float* gain, const int32_t* l2_lut, int32_t cols, int32_t bsize,
                      int32_t x_dtype, int32_t y_dtype, float epsilon, void* stream) {
    if (!y || !sum_sqr || !x || !l2_lut || cols <= 0) return BSMM_ERR_ARG;
    if (reinterpret_cast<uintptr_t>(l2_lut) & 15) return BSMM_ERR_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    return l2_by_types(x_dtype, y_dtype, bsize, [&](auto tx, auto ty, auto bs) {
        typedef decltype(tx) TX;
        typedef decltype(ty) TY;
        l2_normalize_kernel<TX, TY, decltype(bs)::value><<<cols, 256, 0, st>>>(static_cast<typename TY::T*>(y), sum_sqr, static_cast<const typename TX::T*>(x),
                                                                               gain, l2_lut, epsilon);
        return (int)hipGetLastError();
    });
}

int bsmm_l2_normalize_grad(void* dx, float* dgain, const void* dy, const void* x, const float* gain, const float* sum_sqr, const int32_t* l2_lut,
                           int32_t cols, int32_t bsize, int32_t x_dtype, int32_t y_dtype, float epsilon, void* stream) {
    if (!dx || !dy || !x || !sum_sqr || !l2_lut || cols <= 0) return BSMM_ERR_ARG;
    if (gain && !dgain) return BSMM_ERR_ARG;
    if (reinterpret_cast<uintptr_t>(l2_lut) & 15) return BSMM_ERR_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    return l2_by_types(x_dtype, y_dtype, bsize, [&](auto tx, auto ty, auto bs) {
        typedef decltype(tx) TX;
        typedef decltype(ty) TY;
        l2_normalize_grad_kernel<TX, TY, decltype(bs)::value><<<cols, 256, 0, st>>>(static_cast<typename TX::T*>(dx), dgain, static_cast<const typename TY::T*>(dy),
                                                                                    static_cast<const typename TX::T*>(x), gain, sum_sqr, l2_lut, epsilon);
        return (int)hipGetLastError();
    });
}

int bsmm_sparse_op(void* z, const void* x, const void* y, const int32_t* lut, int32_t op, int32_t K, int32_t rows_z, int32_t N, int32_t dtype,
                   void* stream) {
    if (!z || !x || !lut || K <= 0 || N <= 0 || rows_z <= 0) return BSMM_ERR_ARG;
    if ((op == SP_ADD || op == SP_MUL) && !y) return BSMM_ERR_ARG;
    if (op < SP_GAT || op > SP_MUL) return BSMM_ERR_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype != BSMM_F32 && dtype != BSMM_F16 && dtype != BSMM_BF16) return BSMM_ERR_UNSUPPORTED;
    const size_t esz = elem_size(dtype);
    if (op == SP_ADD && z != x) {                        // the unmapped rows pass through
        hipError_t e = hipMemcpyAsync(z, x, (size_t)rows_z * N * esz, hipMemcpyDeviceToDevice, st);
        if (e != hipSuccess) return (int)e;
    }
    dim3 grid(K, std::min((N + 255) / 256, 64));
    auto go = [&](auto t) {
        typedef decltype(t) DT;
        typedef typename DT::T T;
        switch (op) {
            case SP_GAT: sparse_proj_kernel<DT, SP_GAT><<<grid, 256, 0, st>>>(static_cast<T*>(z), static_cast<const T*>(x), nullptr, lut, K, N); break;
            case SP_SCT: sparse_proj_kernel<DT, SP_SCT><<<grid, 256, 0, st>>>(static_cast<T*>(z), static_cast<const T*>(x), nullptr, lut, K, N); break;
            case SP_ADD: sparse_proj_kernel<DT, SP_ADD><<<grid, 256, 0, st>>>(static_cast<T*>(z), static_cast<const T*>(z), static_cast<const T*>(y), lut, K, N); break;
            default:     sparse_proj_kernel<DT, SP_MUL><<<grid, 256, 0, st>>>(static_cast<T*>(z), static_cast<const T*>(x), static_cast<const T*>(y), lut, K, N); break;
        }
        return (int)hipGetLastError();
    };
    if (dtype == BSMM_F32) return go(DTf32{});
    if (dtype == BSMM_F16) return go(DTf16{});
    return go(DTbf16{});
}

int bsmm_sparse_mul_grad(void* dx, void* dy, const void* dz, const void* x, const void* y, const int32_t* lut, int32_t K, int32_t rows_x, int32_t N,
                         int32_t dtype, void* stream) {
    if (!dx || !dy || !dz || !x || !y || !lut || K <= 0 || N <= 0 || rows_x <= 0) return BSMM_ERR_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype != BSMM_F32 && dtype != BSMM_F16 && dtype != BSMM_BF16) return BSMM_ERR_UNSUPPORTED;
    const size_t esz = elem_size(dtype);
    if (dx != dz) {
        hipError_t e = hipMemcpyAsync(dx, dz, (size_t)rows_x * N * esz, hipMemcpyDeviceToDevice, st);
        if (e != hipSuccess) return (int)e;
    }
    dim3 grid(K, std::min((N + 255) / 256, 64));
    auto go = [&](auto t) {
        typedef decltype(t) DT;
        typedef typename DT::T T;
        // reads dz through dx (identical after the copy / when aliased): every mapped element is read before it is overwritten by the same thread
        sparse_mul_grad_kernel<DT><<<grid, 256, 0, st>>>(static_cast<T*>(dx), static_cast<T*>(dy), static_cast<const T*>(dx), static_cast<const T*>(x),
                                                         static_cast<const T*>(y), lut, K, N);
        return (int)hipGetLastError();
    };
    if (dtype == BSMM_F32) return go(DTf32{});
    if (dtype == BSMM_F16) return go(DTf16{});
    return go(DTbf16{});
}

int bsmm_gate_grad(void* dw_out, float* dg, const void* dw, const void* W, const float* gate, int32_t blocks, int32_t bsize,
                   int32_t dtype, void* stream) {
    if (!dw_out || !dg || !dw || !W || !gate || blocks <= 0) return BSMM_ERR_ARG;
    if (bsize != 8 && bsize != 16 && bsize != 32) return BSMM_ERR_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (dtype) {
        case BSMM_F32: gate_grad_kernel<DTf32><<<blocks, 256, 0, st>>>(static_cast<float*>(dw_out), dg, static_cast<const float*>(dw), static_cast<const float*>(W), gate, bsize); break;
        case BSMM_F16: gate_grad_kernel<DTf16><<<blocks, 256, 0, st>>>(static_cast<uint16_t*>(dw_out), dg, static_cast<const uint16_t*>(dw), static_cast<const uint16_t*>(W), gate, bsize); break;
        case BSMM_BF16: gate_grad_kernel<DTbf16><<<blocks, 256, 0, st>>>(static_cast<uint16_t*>(dw_out), dg, static_cast<const uint16_t*>(dw), static_cast<const uint16_t*>(W), gate, bsize); break;
        default: return BSMM_ERR_UNSUPPORTED;
    }
    return (int)hipGetLastError();
}

int bsmm_gate_weights(const void* W, const float* gate, void* out, int32_t blocks, int32_t bsize, int32_t dtype, int32_t pieces, void* stream) {
    if (!W || !gate || !out || blocks <= 0) return BSMM_ERR_ARG;
    if ((bsize != 8 && bsize != 16 && bsize != 32 && bsize != 64) || (dtype != BSMM_F16 && dtype != BSMM_BF16) || (pieces != 1 && pieces != 2)) return BSMM_ERR_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int per8 = bsize * bsize / 8;
    const size_t total8 = (size_t)blocks * per8;
    const unsigned grid = (unsigned)((total8 + 255) / 256);
    const uint4* w = static_cast<const uint4*>(W);
    uint4* o = static_cast<uint4*>(out);
    if (dtype == BSMM_BF16) {
        if (pieces == 1) gate_weights_kernel<DTbf16, 1><<<grid, 256, 0, st>>>(w, gate, o, per8, total8);
        else             gate_weights_kernel<DTbf16, 2><<<grid, 256, 0, st>>>(w, gate, o, per8, total8);
    } else {
        if (pieces == 1) gate_weights_kernel<DTf16, 1><<<grid, 256, 0, st>>>(w, gate, o, per8, total8);
        else             gate_weights_kernel<DTf16, 2><<<grid, 256, 0, st>>>(w, gate, o, per8, total8);
    }
    return (int)hipGetLastError();
}

#ifdef X4_TIMELINE
extern "C" int bsmm_debug_x4_timeline_copy(void* host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(bsmm::g_x4_tl), sizeof(bsmm::g_x4_tl)); }
#endif
#ifdef X4_ENDSTAMPS
extern "C" int bsmm_debug_x4_ends_copy(void* host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(bsmm::g_x4_ends), sizeof(bsmm::g_x4_ends)); }
#endif
#ifdef U2_STAMPS
extern "C" int bsmm_debug_u2_trace_copy(void* host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(bsmm::g_u2_trace), sizeof(bsmm::g_u2_trace)); }
#endif
#ifdef U6_STAMPS
extern "C" int bsmm_debug_u6_trace_copy(void* host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(bsmm::g_u6_trace), sizeof(bsmm::g_u6_trace)); }
#endif
#ifdef X4_STAMPS
extern "C" int bsmm_debug_x4_trace_copy(void* host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(bsmm::g_x4_trace), sizeof(bsmm::g_x4_trace)); }
#endif
#ifdef BSMM_XC_TRACE
int bsmm_debug_trace_copy(void* host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(bsmm::g_xc_trace), sizeof(bsmm::g_xc_trace)); }
#endif
#ifdef X7L_TRACE
int bsmm_debug_x7_trace_copy(void* host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(bsmm::g_x7_trace), sizeof(bsmm::g_x7_trace)); }
#endif

int bsmm_identity_init(void* W, const int32_t* updat_lut, int32_t CB, int32_t KB, int32_t blocks, int32_t bsize,
                       float scale, int32_t dtype, void* stream) {
    if (!W || !updat_lut || CB <= 0 || KB <= 0 || blocks <= 0 || bsize <= 0) return BSMM_ERR_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (dtype) {
        case BSMM_F32:
            identity_init_kernel<DTf32><<<blocks, 256, 0, st>>>(static_cast<float*>(W), updat_lut, CB, KB, bsize, scale);
            break;
        case BSMM_F16:
            identity_init_kernel<DTf16><<<blocks, 256, 0, st>>>(static_cast<uint16_t*>(W), updat_lut, CB, KB, bsize, scale);
            break;
        case BSMM_BF16:
            identity_init_kernel<DTbf16><<<blocks, 256, 0, st>>>(static_cast<uint16_t*>(W), updat_lut, CB, KB, bsize, scale);
            break;
        default:
            return BSMM_ERR_UNSUPPORTED;
    }
    return (int)hipGetLastError();
}

// widths the plan options select (defaults: the wide shapes)
static inline int opt_xc_group(int32_t options) { return (options & BSMM_PLAN_XCOL_NARROW) ? XC_G : 16; }

static long xprop_plan(const int32_t* lut, int32_t segments, int32_t blocks, int32_t n_out, int32_t bsize, int32_t dtype, int32_t axis,
                       int32_t options, int32_t* out) {
    if (axis != 0 && axis != 1) return 0;
    if (bsize == 64) {      // 'BS64': the quadrant view's lut + the bsize-32 plan built from it (feature axis 1 only)
        if (axis != 1 || !lut || segments <= 0 || blocks <= 0 || blocks >= (1 << 28)) return axis != 1 ? 0 : -1;
        std::vector<int32_t> lut32, nested;
        if (!b64_expand_xprop_lut(lut, segments, blocks, lut32)) return -1;
        const int32_t nopt = options;
        const long nw = xprop_plan(lut32.data(), 2 * segments, 4 * blocks, 2 * n_out, 32, dtype, axis, nopt, nullptr);
        if (nw < 0) return -1;
        nested.resize((size_t)nw);
        if (nw > 0 && out) xprop_plan(lut32.data(), 2 * segments, 4 * blocks, 2 * n_out, 32, dtype, axis, nopt, nested.data());
        return b64_emit(0, blocks, segments, lut32, nested, out);
    }
    if (bsize == 8) return dtype == BSMM_F32 ? 0 : build_super8_xprop_plan(lut, segments, blocks, n_out, out, opt_xc_group(options));   // 'BSS8'
    if (bsize != 32 && bsize != 16) return 0;   // plan kernels: bsize 32 (any dtype) / 16 and 8 (16-bit)
    if (dtype == BSMM_F32) {
        if (bsize != 32) return 0;
        return build_xcol_plan(lut, segments, blocks, n_out, out, XS_G);      // (BSMM_PLAN_F32_MFMA named the retired fp32 matrix-core kernel: ignored)
    }
    if (bsize == 16)         // 'BSX7' (staged / list kernels); BSMM_PLAN_XCOL_UNSTAGED / _NARROW named the round-1 kernel, retired in round 4: ignored
        return build_xcol16s_plan(lut, segments, blocks, n_out, out, !(options & BSMM_PLAN_FLOW_CONSECUTIVE));      // (0: the layout does not fit the table fields -> no plan, per-segment kernels)
    if ((options & BSMM_PLAN_XCOL_FLOW) && axis == 1 && !(options & (BSMM_PLAN_XCOL_UNSTAGED | BSMM_PLAN_XCOL_NARROW))) {   // barrier-free persistent kernel
        const long n = build_xflow_plan(lut, segments, blocks, n_out, out, (options & BSMM_PLAN_FLOW_SCHEDULED) != 0, (options & BSMM_PLAN_FLOW_CONSECUTIVE) != 0);
        if (n != 0) return n;
    }
    if (!(options & (BSMM_PLAN_XCOL_UNSTAGED | BSMM_PLAN_XCOL_NARROW))) {   // default: the staged kernel (either feature axis)
        const long n = build_xcol2_plan(lut, segments, blocks, n_out, out, (options >> BSMM_PLAN_XPROP_PH_SHIFT) & 7, axis == 0 && !(options & BSMM_PLAN_FLOW_CONSECUTIVE));
        if (n != 0) return n;                                            // 0: the layout does not fit the table fields
    }
    return build_xcol_plan(lut, segments, blocks, n_out, out, opt_xc_group(options));
}

long bsmm_xprop_plan_words(const int32_t* host_lut, int32_t segments, int32_t blocks, int32_t n_out_blocks, int32_t bsize,
                           int32_t dtype, int32_t axis, int32_t options) {
    return xprop_plan(host_lut, segments, blocks, n_out_blocks, bsize, dtype, axis, options, nullptr);
}

int bsmm_xprop_plan_build(const int32_t* host_lut, int32_t segments, int32_t blocks, int32_t n_out_blocks, int32_t bsize,
                          int32_t dtype, int32_t axis, int32_t options, int32_t* host_plan_out) {
    if (!host_plan_out) return BSMM_ERR_ARG;
    const long n = xprop_plan(host_lut, segments, blocks, n_out_blocks, bsize, dtype, axis, options, host_plan_out);
    return n > 0 ? BSMM_OK : (n == 0 ? BSMM_ERR_UNSUPPORTED : BSMM_ERR_ARG);
}

static long updat_plan(const int32_t* lut, int32_t blocks, int32_t CB, int32_t KB, int32_t bsize, int32_t dtype, int32_t axis, int32_t options,
                       int32_t* out) {
    if (dtype == BSMM_F32 || (axis != 0 && axis != 1)) return 0;   // windowed kernels: 16-bit types
    if (bsize == 64) {      // 'BS64': the quadrant view's updat lut + the bsize-32 (streaming) plan built from it
        if (axis != 1 || !lut || blocks <= 0 || blocks >= (1 << 28)) return axis != 1 ? 0 : -1;
        std::vector<int32_t> lut32((size_t)8 * blocks), nested;
        for (int w = 0; w < blocks; ++w)
            for (int q = 0; q < 4; ++q) {
                lut32[(size_t)2 * (4 * w + q)] = 2 * lut[2 * w] + (q >> 1);
                lut32[(size_t)2 * (4 * w + q) + 1] = 2 * lut[2 * w + 1] + (q & 1);
            }
        const int32_t nopt = options | BSMM_PLAN_UPDAT_NO_DIRECT;      // (the composite call derives the nested plan's descriptor itself: no direct blocks there)
        const long nw = updat_plan(lut32.data(), 4 * blocks, 2 * CB, 2 * KB, 32, dtype, axis, nopt, nullptr);
        if (nw < 0) return -1;
        nested.resize((size_t)nw);
        if (nw > 0 && out) updat_plan(lut32.data(), 4 * blocks, 2 * CB, 2 * KB, 32, dtype, axis, nopt, nested.data());
        return b64_emit(1, blocks, 0, lut32, nested, out);
    }
    if (bsize == 8) return build_super8_updat_plan(lut, blocks, CB, KB, out);   // 'BSS8'
    if (bsize == 16) {
        // 'BSUP' items of the windowed kernel; on feature axis 0 the 'BSU6' section of the row-owner kernel behind them (header word [8]):
        // the call picks per minibatch size (updat_typed)
        const long base = build_updat_plan(lut, blocks, CB, KB, UW16, UP16_MAXB, out);
        if (base <= 0 || axis != 0 || (options & BSMM_PLAN_UPDAT16_WINDOWED)) return base;
        const long off = (base + 3) & ~3L;
        const long sec = build_updat16_rows_section(lut, blocks, CB, KB, out ? out + off : nullptr);
        if (sec <= 0) return base;
        if (out) { std::fill(out + base, out + off, 0); out[8] = (int32_t)off; }
        return off + sec;
    }
    if (bsize != 32) return 0;
    int force = options & BSMM_PLAN_WINDOW_MASK;
    // BSMM_PLAN_WINDOW_8 / _16 / _16W named the windowed bsize-32 kernels of round 1 (retired in round 4): the same window side on the streaming kernel
    if (force == BSMM_PLAN_WINDOW_8) force = BSMM_PLAN_STREAM_8;
    if (force == BSMM_PLAN_WINDOW_16 || force == BSMM_PLAN_WINDOW_16W) force = BSMM_PLAN_STREAM_16;
    {     // either feature axis
        // streaming kernel: 16x16 windows while a window's blocks fit the 64 accumulator slots of a workgroup (with some
        // slack for the rows that do not pack: <= 56 on average), 8x8 windows for denser layouts
        const double windows = (double)((CB + 15) / 16) * ((KB + 15) / 16);
        int ws = force == BSMM_PLAN_STREAM_16 ? 16 : (force == BSMM_PLAN_STREAM_8 ? 8 : (blocks <= 56.0 * windows ? 16 : 8));
        // very sparse layouts on feature axis 1: 32 x 32 windows (a 16 x 16 window then holds < 10 blocks for its 32 KiB per chunk).  Measured at
        // 8192^2, N = 4096 (profiles/r03_updat_ws32.txt): 3 % 71 against 95 us; 5 % (BASELINE configs[3]) 102 against 96 -- so only below ~3.7 %
        // (Round 6: with DIRECT blocks the bigger window also wins at 4 - 5 % on a grid of >= 64 such windows -- 8192^2 N = 4096, BASELINE configs[3]:
        //  81 / 85 / 91 us at 4 / 4.5 / 5 % against 99 / 98 / 100 -- but only for LONG minibatches: at N = 1024 / 2048 it loses, 42 / 57 against 29 / 50 us,
        //  because 64 items need the partial sums and their summing pass where 256 items store directly.  A plan does not know the minibatch, so the
        //  rule stays; the host class builds the BSMM_PLAN_STREAM_32 plan as well and picks per call: profiles/r06_updat_ws32.txt)
        const double windows32 = (double)((CB + 31) / 32) * ((KB + 31) / 32);
        if (force == BSMM_PLAN_STREAM_32 || (force == 0 && axis == 1 && windows32 >= 16 && blocks <= 38.0 * windows32)) ws = 32;
        // direct blocks (round 6): what a window's 16 waves cannot hold gets its own workgroups instead of a sliced last round (feature axis 1)
        const int direct_max = (axis == 1 && !(options & BSMM_PLAN_UPDAT_NO_DIRECT)) ? U2_DIRECT_MAX : 0;
        return build_updat2_plan(lut, blocks, CB, KB, ws, out, (options >> BSMM_PLAN_UPDAT_SETS_SHIFT) & 15, direct_max);
    }
}

long bsmm_updat_plan_words(const int32_t* host_updat_lut, int32_t blocks, int32_t CB, int32_t KB, int32_t bsize, int32_t dtype,
                           int32_t axis, int32_t options) {
    return updat_plan(host_updat_lut, blocks, CB, KB, bsize, dtype, axis, options, nullptr);
}

int bsmm_updat_plan_build(const int32_t* host_updat_lut, int32_t blocks, int32_t CB, int32_t KB, int32_t bsize, int32_t dtype,
                          int32_t axis, int32_t options, int32_t* host_plan_out) {
    if (!host_plan_out) return BSMM_ERR_ARG;
    const long n = updat_plan(host_updat_lut, blocks, CB, KB, bsize, dtype, axis, options, host_plan_out);
    return n > 0 ? BSMM_OK : (n == 0 ? BSMM_ERR_UNSUPPORTED : BSMM_ERR_ARG);
}

// descriptor of a flat (non-composite) plan: (magic, width, waves, items)
static bool describe_flat(const int32_t* p, long words, int32_t d[5]) {
    if (words < 8) return false;
    d[4] = 0;
    switch (p[0]) {
        case XCPLAN_MAGIC:   if (p[1] != XCPLAN_VERSION || words < XC_HDR) return false;   d[1] = p[2]; d[2] = p[2]; d[3] = 0; break;
        case X2PLAN_MAGIC:   if (p[1] != X2PLAN_VERSION || words < X2_HDR || p[11] < 2 || p[11] > 4 || p[12] < X2_HDR || words < (long)p[12] + (long)p[3] * X2_G) return false;   d[1] = p[2]; d[2] = p[2]; d[3] = 0; d[4] = p[11]; break;
        case X4PLAN_MAGIC:   if (p[1] != X4PLAN_VERSION || words < X4_HDR || p[2] != X4_G) return false;   d[1] = p[2]; d[2] = p[2]; d[3] = 0; break;
        case X7PLAN_MAGIC:   if (p[1] != X7PLAN_VERSION || words < X7_HDR || p[12] < X7_HDR || words < (long)p[12] + (long)p[3] * X7_G) return false;   d[1] = p[2]; d[2] = 16; d[3] = 0; break;
        case UPLAN_MAGIC:    if (p[1] != UPLAN_VERSION || words < UP_HDR || p[8] < 0 || (p[8] > 0 && (p[8] + U6_HDR > words || p[p[8]] != U6PLAN_MAGIC ||
                                 p[8] + U6_HDR + (long)p[p[8] + 4] * U6_ITEM > words))) return false;
                             d[1] = p[2]; d[2] = p[7]; d[3] = p[4]; d[4] = p[8];               // (plan_inner: word offset of the 'BSU6' section, 0 = none;
                             if (p[8] > 0) {                                                   //  its window width / item count in bits 8.. of width / waves)
                                 if (p[2] > 255 || p[7] > 255 || p[p[8] + 4] <= 0 || p[p[8] + 4] >= (1 << 23)) return false;
                                 d[1] |= p[p[8] + 3] << 8; d[2] |= p[p[8] + 4] << 8;
                             }
                             break;
        case U2PLAN_MAGIC:   if (p[1] != U2PLAN_VERSION || words < U2_HDR || p[26] != U2_HDR + p[4] * U2_ITEM || words < (long)p[26] + p[5]) return false;   // (the launcher addresses the block map behind the items)
                               if (p[27] < 0 || p[27] > 0x1fff || p[28] < 0 || p[28] > U2_DIRECT_MAX) return false;
                               if (p[28] > 0 && (p[30] != U2_DIRECT_PARTS || p[29] < p[26] + p[5] || words < (long)p[29] + 4L * p[28])) return false;      // direct blocks
                               d[1] = p[2]; d[2] = p[7]; d[3] = p[4]; d[4] = p[8] | (p[25] > 0 ? 16 : 0) | (p[27] << 8) | (p[28] << 21); break;   // item sets | all equally long | longest set | direct blocks
        default: return false;
    }
    d[0] = p[0];
    return true;
}

int bsmm_plan_attach(bsmm_args* a, const int32_t* host_plan, long words, const int32_t* device_plan) {
    if (!a) return BSMM_ERR_ARG;
    a->plan = nullptr;
    a->plan_magic = a->plan_width = a->plan_waves = a->plan_items = a->plan_inner = 0;
    if (!device_plan) return BSMM_OK;
    if (!host_plan || words < 8 || (reinterpret_cast<uintptr_t>(device_plan) & 15)) return BSMM_ERR_ARG;
    int32_t d[5];
    if (host_plan[0] == S8PLAN_MAGIC) {
        if (host_plan[1] != S8PLAN_VERSION || host_plan[2] <= 0 || host_plan[6] != words) return BSMM_ERR_ARG;
        const int32_t off = host_plan[5];
        if (off < S8_HDR || off >= words || !describe_flat(host_plan + off, words - off, d)) return BSMM_ERR_ARG;
        int code = -1;                                                                    // word [7]: 0 = xprop, 1 = updat
        if (!host_plan[7] && d[0] == XCPLAN_MAGIC) code = 0;
        if (!host_plan[7] && d[0] == X2PLAN_MAGIC) code = 1;
        if (host_plan[7] && d[0] == UPLAN_MAGIC) code = 0;
        if (host_plan[7] && d[0] == U2PLAN_MAGIC) code = 2;
        if (code < 0 || d[1] > 255 || (uint32_t)d[4] >= (1u << 21)) return BSMM_ERR_ARG;
        a->plan_magic = S8PLAN_MAGIC; a->plan_width = host_plan[2]; a->plan_waves = d[2]; a->plan_items = d[3];
        a->plan_inner = d[1] | (code << 8) | (int32_t)((uint32_t)(code ? d[4] : 0) << 11);
    } else if (host_plan[0] == B64PLAN_MAGIC) {
        if (host_plan[1] != B64PLAN_VERSION || host_plan[2] <= 0 || host_plan[6] != words || (host_plan[3] != 0 && host_plan[3] != 1)) return BSMM_ERR_ARG;
        const int32_t off = host_plan[5];
        if (off != b64_off_nested(host_plan[3], host_plan[7], host_plan[2]) || off > words) return BSMM_ERR_ARG;
        a->plan_magic = B64PLAN_MAGIC; a->plan_inner = host_plan[3];
        if (off < words) {      // a nested bsize-32 plan (none: the nested call runs the kernels without a plan)
            if (!describe_flat(host_plan + off, words - off, d) || nested_code(d[0]) == 0 || d[2] > 31 || (uint32_t)d[4] > 0xffffffu) return BSMM_ERR_ARG;
            a->plan_width = d[1]; a->plan_items = d[3];
            a->plan_waves = d[2] | (nested_code(d[0]) << 5) | (int32_t)((uint32_t)d[4] << 8);
        }
    } else {
        if (!describe_flat(host_plan, words, d)) return BSMM_ERR_ARG;
        a->plan_magic = d[0]; a->plan_width = d[1]; a->plan_waves = d[2]; a->plan_items = d[3]; a->plan_inner = d[4];
    }
    a->plan = device_plan;
    return BSMM_OK;
}

size_t bsmm_workspace_bytes(int op, const bsmm_args* a) {
    if (!a) return 0;
    const bool xprop_op = op == BSMM_OP_FPROP || op == BSMM_OP_BPROP;
    // locked reference-policy tables on the per-segment kernels with a 16-bit type: fp32 image of the output (see xprop_typed);
    // asked for whenever the call COULD take that path (no plan, a gate, or the size heuristic)
    if (a->bsize == 64) {   // 'BS64' plans: [the quadrant copy of W, unless prepared][the quadrants' gates][what the nested bsize-32 call needs]
        if (a->axis != 1 || !a->plan || a->plan_magic != B64PLAN_MAGIC || a->plan_inner != (xprop_op ? 0 : 1)) return 0;
        bsmm_args b = b64_inner(a, !xprop_op);
        if (!xprop_op) { b.flags |= BSMM_FLAG_DW_SUMS; return bsmm_workspace_bytes(op, &b); }
        return (a->prepared_w ? 0 : b64_w_bytes(a)) + b64_gate_bytes(a) + bsmm_workspace_bytes(op, &b);
    }
    const size_t lock = xprop_op ? lock_acc_bytes(a) : 0;
    if (op == BSMM_OP_UPDAT && updat8_f32_split_applies(a))     // fp32 / bsize 8 through the bf16 streaming kernel: its workspace + the pieces of X and DY
        return updat8_f32_inner_bytes(a) + 6 * ((size_t)a->N * a->C + (size_t)a->N * a->K) + F32_SPLIT_FLAG_BYTES;
    if (a->bsize == 8) {   // 'BSS8' plans: the expanded W (xprop) / the fp32 sums of the super-blocks (updat)
        if (!a->plan || a->plan_magic != S8PLAN_MAGIC || a->plan_width <= 0 || a->dtype == BSMM_F32) return lock;
        const size_t blk = (size_t)a->plan_width * 1024;
        if (op == BSMM_OP_UPDAT && ((a->plan_inner >> 8) & 7) == 2) {      // nested streaming plan: its sums + partial-sum regions
            bsmm_args b = s8_inner(a, true);
            b.flags = BSMM_FLAG_DW_SUMS; b.split = 0;
            return bsmm_workspace_bytes(op, &b);
        }
        return op == BSMM_OP_UPDAT ? blk * sizeof(float) : std::max(round16(blk * elem_size(a->dtype)) + 16, lock);   // (+ the non-finite flag of the call)
    }
    if (op == BSMM_OP_UPDAT && updat16_f32_rows_applies(a) && !updat16_f32_rows_taken(a)) return 0;      // (the per-block fp32 kernel: no workspace)
    if (op == BSMM_OP_UPDAT && updat16_f32_rows_applies(a))     // fp32 / bsize 16 / feature axis 0 on the row-owner kernel: its images + the pieces of X and DY
        return updat16_f32_rows_images_bytes(a) + 6 * ((size_t)a->N * a->C + (size_t)a->N * a->K) + F32_SPLIT_FLAG_BYTES;
    if (op == BSMM_OP_UPDAT && updat16_f32_split_applies(a))    // fp32 / bsize 16 on the windowed kernel: the fp32 sums + the pieces of X and DY
        return updat16_f32_sums_bytes(a) + 6 * ((size_t)a->N * a->C + (size_t)a->N * a->K) + F32_SPLIT_FLAG_BYTES;
    if (op == BSMM_OP_UPDAT && updat_f32_split_applies(a))      // fp32 through the bf16 streaming kernel: its workspace + the pieces of X and DY
        return updat_f32_split_inner_bytes(a) + 6 * ((size_t)a->N * a->C + (size_t)a->N * a->K) + F32_SPLIT_FLAG_BYTES;
    if (op == BSMM_OP_UPDAT && a->plan && (a->bsize == 32 || a->bsize == 16) && a->dtype != BSMM_F32) {
        if (a->plan_magic == U2PLAN_MAGIC) {   // streaming kernel: the fp32 sums + one region of partial sums per (round, workgroup)
            const U2Launch L = updat2_shape(a, true);
            return u2_sums_bytes(a) + (size_t)L.rounds * L.grid * u2_region_bytes() + u2_direct_bytes(L.direct);
        }
        if (a->plan_magic != UPLAN_MAGIC || a->bsize != 16) return 0;     // (a plan check_plan refuses: nothing to size)
        // bsize 16 windowed kernel: one fp32 image of the sums (split-minibatch path); with the 'BSU6' section (feature axis 0) one image per part
        // of the row-owner kernel's minibatch split
        return (size_t)a->blocks * a->bsize * a->bsize * sizeof(float) * (a->plan_inner > 0 && a->axis == 0 ? U6_MAX_SPLIT : 1);
    }
    if (xprop_op && a->dtype == BSMM_F32 && a->bsize == 32 && a->plan && a->plan_magic == XCPLAN_MAGIC)   // bf16 pieces of the activations and
        return std::max(xcols_workspace_bytes(a), (op == BSMM_OP_FPROP ? wt_bytes(a) : 0) + lock);          // (unless prepared) the weights -- or what
                                                                                                           // the kernels without a plan need, if the cost model sends the call there
    // fprop keeps a transposed copy of W (the matrix-core operand wants the contraction index contiguous)
    if (op == BSMM_OP_FPROP) return wt_bytes(a) + lock;
    if (op == BSMM_OP_BPROP) return lock;
    return 0;
}

const char* bsmm_error_string(int code) {
    switch (code) {
        case BSMM_OK: return "ok";
        case BSMM_ERR_ARG: return "bsmm: invalid argument (null pointer, non-positive size, misaligned lut, pcount outside 1..8)";
        case BSMM_ERR_UNSUPPORTED: return "bsmm: unsupported configuration (bsize must be 8/16/32, axis 0/1, dtype f32/f16/bf16, gate NULL)";
        case BSMM_ERR_WORKSPACE: return "bsmm: workspace missing, misaligned or smaller than bsmm_workspace_bytes()";
        default: return code > 0 ? hipGetErrorString(static_cast<hipError_t>(code)) : "bsmm: unknown error";
    }
}

int bsmm_version(void) { return BSMM_VERSION; }

}  // extern "C"
