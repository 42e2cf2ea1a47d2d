// bsmm_api.hip -- C-ABI entry points (include/bsmm.h) and kernel dispatch for gfx950.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "bsmm.h"
#include "bsmm_plan.h"
#include "bsmm_l2norm.h"
#include "bsmm_sparse_proj.h"
#include "bsmm_updat.h"
#include "bsmm_updat_tr.h"
#include "bsmm_updat_win.h"
#include "bsmm_super8.h"
#include "bsmm_xcols.h"
#include "bsmm_xcol.h"
#include "bsmm_xcol16.h"
#include "bsmm_xprop.h"

using namespace bsmm;

namespace {

std::atomic<int> g_variant{0};

inline size_t elem_size(int dtype) { return dtype == BSMM_F32 ? 4 : 2; }

int check_common(const bsmm_args* a) {
    if (!a || !a->lut) return BSMM_ERR_ARG;
    if (a->blocks <= 0 || a->N <= 0 || a->C <= 0 || a->K <= 0) return BSMM_ERR_ARG;
    if (a->bsize != 8 && a->bsize != 16 && a->bsize != 32) return BSMM_ERR_UNSUPPORTED;
    if (a->axis != 0 && a->axis != 1) return BSMM_ERR_UNSUPPORTED;
    if (a->dtype != BSMM_F32 && a->dtype != BSMM_F16 && a->dtype != BSMM_BF16) return BSMM_ERR_UNSUPPORTED;
    if (a->C % a->bsize || a->K % a->bsize) return BSMM_ERR_ARG;
    if (reinterpret_cast<uintptr_t>(a->lut) & 15) return BSMM_ERR_ARG;
    return BSMM_OK;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// ---------------------------------------------------------------------------------------------
// xprop
// ---------------------------------------------------------------------------------------------
template <class DT, int BS, int AXIS, bool FPROP>
int launch_xprop_valu(const void* X, const void* W, void* Y, const bsmm_args* a, hipStream_t st) {
    typedef typename DT::T T;
    dim3 grid((a->N + 255) / 256, a->segments);
    xprop_valu_kernel<DT, BS, AXIS, FPROP><<<grid, 256, 0, st>>>(static_cast<const T*>(X), static_cast<const T*>(W),
                                                                 static_cast<T*>(Y), a->lut, a->N, a->C, a->K, a->gate);
    return (int)hipGetLastError();
}

template <class DT, int BS, int AXIS>
int launch_xprop_mfma(const void* X, const void* Wsel, void* Y, const bsmm_args* a, hipStream_t st) {
    typedef typename DT::T T;
    const int N = a->N;
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
                                                                               static_cast<T*>(Y), a->lut, m, N, a->C, a->K, a->gate);
            else
                xprop16_kernel<DT, AXIS, NSUB, true><<<m.grid(), 256, 0, st>>>(static_cast<const T*>(X), static_cast<const T*>(Wsel),
                                                                               static_cast<T*>(Y), a->lut, m, N, a->C, a->K, a->gate);
            return;
        }
        if constexpr (BS == 32)
            xprop32_kernel<DT, AXIS, NSUB><<<m.grid(), 256, 0, st>>>(static_cast<const T*>(X), static_cast<const T*>(Wsel),
                                                                     static_cast<T*>(Y), a->lut, m, N, a->C, a->K);
        else
            xprop16_kernel<DT, AXIS, NSUB><<<m.grid(), 256, 0, st>>>(static_cast<const T*>(X), static_cast<const T*>(Wsel),
                                                                     static_cast<T*>(Y), a->lut, m, N, a->C, a->K);
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
// fp32, bsize 32: the exact three-piece bf16 kernel (bsmm_xcols.h) instead of the fp32-MFMA kernel xcol32f
// (BSMM_F32_SPLIT=0, read once, selects the latter for A/B runs).  Decides the plan format too ('BSXC' G = 16 / 'BSXF').
inline bool f32_split(int axis) {
    static const int on = [] { const char* e = getenv("BSMM_F32_SPLIT"); return e ? atoi(e) : 1; }();
    return (axis == 0 || axis == 1) && on;
}
inline int xc16_group() {   // output blocks per workgroup of the bsize-16 xcol kernel: 16, or 32 ("wide")
    static const int wide = [] { const char* e = getenv("BSMM_XC16_WIDE"); return e ? atoi(e) : 1; }();
    return wide ? 32 : XC16_G;
}
// Output blocks per workgroup of the 16-bit bsize-32 xcol kernels: 16 (the wide <16, 4> variants, bsmm_xcol.h; measured
// 5-10 % faster than <8, 2> from N = 3072 up, 8 % slower at N = 2048 where it fills only half the CUs).  The plan is
// built for the same width; BSMM_XC_WIDE=0 (read once) selects the narrow kernels for A/B runs.
inline int xc_group(int /*axis*/) {
    static const int wide = [] { const char* e = getenv("BSMM_XC_WIDE"); return e ? atoi(e) : 1; }();
    return wide ? 16 : XC_G;
}

template <class DT, int AXIS, int NW, int PH>
int launch_xcol16_g(const void* X, const void* Wsel, void* Y, const bsmm_args* a, hipStream_t st) {
    typedef typename DT::T T;
    const int n_out = a->K / 16;
    XMap m;
    m.ntiles = (a->N + XC_R - 1) / XC_R;
    m.segments = (n_out + 2 * NW - 1) / (2 * NW);
    m.P = m.ntiles >= 8 ? 1 : (8 + m.ntiles - 1) / m.ntiles;
    if (m.P > m.segments) m.P = m.segments;
    m.SP = (m.segments + m.P - 1) / m.P;
    constexpr int LDS = xc_lds_bytes(NW, PH);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&xcol16_kernel<DT, AXIS, NW, PH>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        attr_set = true;
    }
    xcol16_kernel<DT, AXIS, NW, PH><<<m.grid(), 64 * NW, LDS, st>>>(static_cast<const T*>(X), static_cast<const T*>(Wsel), static_cast<T*>(Y), a->plan, m,
                                                                   a->N, a->C, a->K);
    return (int)hipGetLastError();
}

template <class DT, int AXIS>
int launch_xcol16(const void* X, const void* Wsel, void* Y, const bsmm_args* a, hipStream_t st) {
    if (xc16_group() == 32) return launch_xcol16_g<DT, AXIS, 16, BSMM_XC_WIDE_PH>(X, Wsel, Y, a, st);
    return launch_xcol16_g<DT, AXIS, 8, XC_PH>(X, Wsel, Y, a, st);
}

template <int AXIS>
int launch_xcol32f(const void* X, const void* Wsel, void* Y, const bsmm_args* a, hipStream_t st) {
    const int n_out = a->K / 32;
    XMap m;
    m.ntiles = (a->N + XF_R - 1) / XF_R;
    m.segments = (n_out + XC_G - 1) / XC_G;
    m.P = m.ntiles >= 8 ? 1 : (8 + m.ntiles - 1) / m.ntiles;
    if (m.P > m.segments) m.P = m.segments;
    m.SP = (m.segments + m.P - 1) / m.P;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&xcol32f_kernel<AXIS>), hipFuncAttributeMaxDynamicSharedMemorySize, XF_LDS);
        attr_set = true;
    }
    xcol32f_kernel<AXIS><<<m.grid(), 512, XF_LDS, st>>>(static_cast<const float*>(X), static_cast<const float*>(Wsel), static_cast<float*>(Y),
                                                        a->plan, m, a->N, a->C, a->K);
    return (int)hipGetLastError();
}

// fp32 on the bf16 matrix cores: split pre-passes into the workspace ([3][N*C] activation pieces, then [3][blocks*1024]
// weight pieces), then the wide xcol kernel with three slabs per step.
inline size_t xcols_workspace_bytes(const bsmm_args* a) {
    return 6 * ((size_t)a->N * a->C + (size_t)a->blocks * 1024);
}
template <int AXIS>
int launch_xcol32s(bool fprop, const void* X, const void* W, void* Y, const bsmm_args* a, hipStream_t st) {
    const size_t nx = (size_t)a->N * a->C;
    if (!a->workspace || a->workspace_bytes < xcols_workspace_bytes(a) || !aligned16(a->workspace)) return BSMM_ERR_WORKSPACE;
    uint16_t* xp = static_cast<uint16_t*>(a->workspace);
    uint16_t* wp = xp + 3 * nx;
    split3_x_kernel<<<(unsigned)((nx / 8 + 255) / 256), 256, 0, st>>>(static_cast<const float*>(X), xp, nx);
    if (fprop) split3_w_kernel<true><<<a->blocks, 256, 0, st>>>(static_cast<const float*>(W), wp, a->blocks);
    else       split3_w_kernel<false><<<a->blocks, 256, 0, st>>>(static_cast<const float*>(W), wp, a->blocks);
    const int n_out = a->K / 32;
    XMap m;
    m.ntiles = (a->N + XC_R - 1) / XC_R;
    m.segments = (n_out + XS_G - 1) / XS_G;
    m.P = m.ntiles >= 8 ? 1 : (8 + m.ntiles - 1) / m.ntiles;
    if (m.P > m.segments) m.P = m.segments;
    m.SP = (m.segments + m.P - 1) / m.P;
    static bool attr_set = false;   // benign race: idempotent
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&xcol32s_kernel<AXIS>), hipFuncAttributeMaxDynamicSharedMemorySize, XS_LDS);
        attr_set = true;
    }
    xcol32s_kernel<AXIS><<<m.grid(), 64 * XS_G, XS_LDS, st>>>(xp, wp, static_cast<float*>(Y), a->plan, m, a->N, a->C, a->K, a->blocks);
    return (int)hipGetLastError();
}

template <class DT, bool TRANSW, int G, int PH>
void launch_xcol0_g(const void* X, const void* Wsel, void* Y, const bsmm_args* a, hipStream_t st) {
    typedef typename DT::T T;
    const int n_out = a->K / 32;
    XMap m;
    m.ntiles = (a->N + XC_R - 1) / XC_R;
    m.segments = (n_out + G - 1) / G;
    m.P = m.ntiles >= 8 ? 1 : (8 + m.ntiles - 1) / m.ntiles;
    if (m.P > m.segments) m.P = m.segments;
    m.SP = (m.segments + m.P - 1) / m.P;
    constexpr int LDS = 2 * PH * XC0_SLAB;
    static bool attr_set = false;   // benign race: idempotent
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&xcol32_a0_kernel<DT, TRANSW, G, PH>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        attr_set = true;
    }
    xcol32_a0_kernel<DT, TRANSW, G, PH><<<m.grid(), 64 * G, LDS, st>>>(static_cast<const T*>(X), static_cast<const T*>(Wsel), static_cast<T*>(Y), a->plan, m,
                                                                     a->N, a->C, a->K);
}

template <class DT, bool TRANSW>
void launch_xcol0(const void* X, const void* Wsel, void* Y, const bsmm_args* a, hipStream_t st) {
    if (xc_group(0) == 16) launch_xcol0_g<DT, TRANSW, 16, BSMM_XC_WIDE_PH>(X, Wsel, Y, a, st);
    else                   launch_xcol0_g<DT, TRANSW, XC_G, XC_PH>(X, Wsel, Y, a, st);
}

template <class DT, bool TRANSW, int G, int PH>
void launch_xcol_g(const void* X, const void* Wsel, void* Y, const bsmm_args* a, hipStream_t st) {
    typedef typename DT::T T;
    const int n_out = a->K / 32;
    XMap m;
    m.ntiles = (a->N + XC_R - 1) / XC_R;
    m.segments = (n_out + G - 1) / G;
    m.P = m.ntiles >= 8 ? 1 : (8 + m.ntiles - 1) / m.ntiles;
    if (m.P > m.segments) m.P = m.segments;
    m.SP = (m.segments + m.P - 1) / m.P;
    constexpr int LDS = xc_lds_bytes(G, PH);
    static bool attr_set = false;   // benign race: idempotent
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&xcol32_a1_kernel<DT, TRANSW, G, PH>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        attr_set = true;
    }
    xcol32_a1_kernel<DT, TRANSW, G, PH><<<m.grid(), 64 * G, LDS, st>>>(static_cast<const T*>(X), static_cast<const T*>(Wsel), static_cast<T*>(Y), a->plan, m,
                                                                     a->N, a->C, a->K);
}

template <class DT, bool TRANSW>
void launch_xcol(const void* X, const void* Wsel, void* Y, const bsmm_args* a, hipStream_t st) {
    if (xc_group(1) == 16) launch_xcol_g<DT, TRANSW, 16, BSMM_XC_WIDE_PH>(X, Wsel, Y, a, st);
    else                   launch_xcol_g<DT, TRANSW, XC_G, XC_PH>(X, Wsel, Y, a, st);
}

template <class DT, int AXIS>
int launch_xgroup32(const void* X, const void* Wsel, void* Y, const bsmm_args* a, hipStream_t st, bool transw) {
    if constexpr (AXIS == 1) {
        if (transw) launch_xcol<DT, true>(X, Wsel, Y, a, st);
        else        launch_xcol<DT, false>(X, Wsel, Y, a, st);
    } else {
        if (transw) launch_xcol0<DT, true>(X, Wsel, Y, a, st);
        else        launch_xcol0<DT, false>(X, Wsel, Y, a, st);
    }
    return (int)hipGetLastError();
}

template <class DT, int BS>
int launch_transpose(const void* W, void* Wt, int blocks, hipStream_t st) {
    typedef typename DT::T T;
    transpose_blocks_kernel<DT, BS><<<blocks, 256, 0, st>>>(static_cast<const T*>(W), static_cast<T*>(Wt), blocks);
    return (int)hipGetLastError();
}

template <class DT, int BS, int AXIS>
int xprop_typed(bool fprop, const void* X, const void* W, void* Y, const bsmm_args* a) {
    hipStream_t st = static_cast<hipStream_t>(a->stream);
    const int variant = g_variant.load(std::memory_order_relaxed);
    const bool vec_ok = aligned16(X) && aligned16(W) && aligned16(Y);
    const bool use_valu = (BS == 8) || variant == 1 || !vec_ok;
    // grouped kernels need enough (row tile x group) workgroups to fill 256 CUs; below that the per-segment kernel,
    // which has segments x tiles workgroups, is faster
    bool enough = false;
    if (BS == 16 && a->plan != nullptr && DT::is16) {
        enough = (long)((a->N + XC_R - 1) / XC_R) * ((a->K / 16 + XC16_G - 1) / XC16_G) >= 224;
        if (AXIS == 0 && (a->N % 8 != 0)) enough = false;
        if (variant == 3 && !(AXIS == 0 && (a->N % 8 != 0))) enough = true;
    }
    if (BS == 32 && a->plan != nullptr && !DT::is16) {   // fp32: xcol32f (axis 0 needs 16-byte aligned row pieces: N % 4 == 0)
        enough = use_xcol() && (long)((a->N + XF_R - 1) / XF_R) * ((a->K / 32 + XC_G - 1) / XC_G) >= 224;
        if (variant == 3 && use_xcol()) enough = true;
        if (AXIS == 0 && (a->N % 4 != 0)) enough = false;
    }
    if (BS == 32 && a->plan != nullptr && DT::is16) {
        // Cost model fitted to the measurements in profiles/r01_sweeps.md (4096^2 / 20 % and 8192^2 / 5 %, N = 512 .. 8192):
        // the grouped kernel pays ~0.48 us per pair step of a group plus ~0.045 us per block, once per round of 256
        // workgroups, whatever the density; the per-segment kernel pays ~1.04e-5 us per (block, minibatch row).
        const int G = xc_group(AXIS);
        const double CB = a->C / 32.0, KB = a->K / 32.0;
        const double ngroups = (double)((a->K / 32 + G - 1) / G), ntiles = (double)((a->N + XC_R - 1) / XC_R);
        const double rounds = std::max(1.0, std::ceil(ntiles * ngroups / 256.0));
        const double dens = std::min(1.0, a->blocks / std::max(1.0, CB * KB));
        const double steps = std::ceil(CB / 2.0) * (1.0 - std::pow(1.0 - dens, 2.0 * G));
        const double t_group = rounds * (0.48 * steps + 0.045 * a->blocks / ngroups) + 8.0;
        const double t_segment = 17.0 + 1.04e-5 * (double)a->blocks * a->N;
        enough = t_group < t_segment;
        if (AXIS == 0 && use_xcol() && (a->N % 8 != 0)) enough = false;   // axis-0 xcol needs 16-byte aligned row pieces
    }
    if (variant == 3 && a->plan != nullptr && !(AXIS == 0 && use_xcol() && (a->N % 8 != 0))) enough = true;   // test hook
    const bool use_group = !use_valu && (BS == 32 || (BS == 16 && DT::is16)) && a->plan != nullptr && (variant == 0 || variant == 3) && enough &&
                           a->gate == nullptr;   // gated calls take the per-segment kernels
    if constexpr (BS == 8 && DT::is16) {
        // bsize 8 on the matrix cores: expand W into the 32x32 super-blocks of the 'BSS8' plan and run the bsize-32 kernel
        const bool shape_ok = a->C % 32 == 0 && a->K % 32 == 0 && !(AXIS == 0 && (a->N % 8 != 0));
        if (a->plan != nullptr && a->plan_aux > 0 && a->gate == nullptr && vec_ok && shape_ok && (variant == 0 || variant == 3)) {
            const bool fill = (long)((a->N + XC_R - 1) / XC_R) * ((a->K / 32 + XC_G - 1) / XC_G) >= 224;
            if (fill || variant == 3) {
                const int ns = a->plan_aux;
                const size_t need = (size_t)ns * 1024 * elem_size(a->dtype);
                if (!a->workspace || a->workspace_bytes < need || !aligned16(a->workspace)) return BSMM_ERR_WORKSPACE;
                typedef typename DT::T T;
                if (fprop) expand8_kernel<DT, true><<<ns, 256, 0, st>>>(static_cast<const T*>(W), a->plan, static_cast<T*>(a->workspace));
                else       expand8_kernel<DT, false><<<ns, 256, 0, st>>>(static_cast<const T*>(W), a->plan, static_cast<T*>(a->workspace));
                bsmm_args b = *a;
                b.bsize = 32; b.blocks = ns; b.plan = a->plan + s8_off_nested(ns); b.plan_aux = 0;
                return launch_xgroup32<DT, AXIS>(X, a->workspace, Y, &b, st, false);
            }
        }
    }
    if (a->locks > 0 && !use_group) {   // several segments accumulate into the same output block: start from zero
        hipError_t e = hipMemsetAsync(Y, 0, (size_t)a->N * a->K * elem_size(a->dtype), st);
        if (e != hipSuccess) return (int)e;
    }
    if (use_valu) {
        return fprop ? launch_xprop_valu<DT, BS, AXIS, true>(X, W, Y, a, st)
                     : launch_xprop_valu<DT, BS, AXIS, false>(X, W, Y, a, st);
    }
    // (xcol can gather the fprop operand transposed itself -- launch_xgroup32(..., transw = true), no workspace and no
    //  pre-pass -- but that measured SLOWER than the 6 us transpose kernel + contiguous fragment loads: 140 vs 127 us, also with
    //  the kernel held at 128 VGPRs.)
    if constexpr (BS == 32 && !DT::is16) {
        // (axis 0 needs 16-byte aligned bf16 row pieces: N % 8 == 0; otherwise the per-segment kernel below)
        if (use_group && f32_split(AXIS) && a->C % 32 == 0 && !(AXIS == 0 && a->N % 8 != 0)) return launch_xcol32s<AXIS>(fprop, X, W, Y, a, st);
    }
    if constexpr (BS != 8) {
        const void* Wsel = W;
        if (fprop) {
            const size_t need = (size_t)a->blocks * BS * BS * elem_size(a->dtype);
            if (!a->workspace || a->workspace_bytes < need || !aligned16(a->workspace)) return BSMM_ERR_WORKSPACE;
            int rc = launch_transpose<DT, BS>(W, a->workspace, a->blocks, st);
            if (rc) return rc;
            Wsel = a->workspace;
        }
        if constexpr (BS == 32 && DT::is16) {
            if (use_group) return launch_xgroup32<DT, AXIS>(X, Wsel, Y, a, st, false);
        }
        if constexpr (BS == 16 && DT::is16) {
            if (use_group) return launch_xcol16<DT, AXIS>(X, Wsel, Y, a, st);
        }
        if constexpr (BS == 32 && !DT::is16) {
            if (use_group && !f32_split(AXIS)) return launch_xcol32f<AXIS>(X, Wsel, Y, a, st);     // plan is 'BSXF' only then
        }
        return launch_xprop_mfma<DT, BS, AXIS>(X, Wsel, Y, a, st);
    }
    return BSMM_ERR_UNSUPPORTED;
}

template <class DT>
int xprop_dt(bool fprop, const void* X, const void* W, void* Y, const bsmm_args* a) {
#define BSMM_CASE(BS, AX) \
    if (a->bsize == BS && a->axis == AX) return xprop_typed<DT, BS, AX>(fprop, X, W, Y, a);
    BSMM_CASE(32, 0) BSMM_CASE(32, 1) BSMM_CASE(16, 0) BSMM_CASE(16, 1) BSMM_CASE(8, 0) BSMM_CASE(8, 1)
#undef BSMM_CASE
    return BSMM_ERR_UNSUPPORTED;
}

int xprop(bool fprop, const void* X, const void* W, void* Y, const bsmm_args* a) {
    int rc = check_common(a);
    if (rc) return rc;
    if (!X || !W || !Y || a->segments <= 0) return BSMM_ERR_ARG;
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
// Windowed bsize-32 kernels (bsmm_updat_win.h).  raw_sums: leave the fp32 sums of every block in a->workspace (zeroed
// here) and apply no alpha / beta -- the bsize-8 super-block path finishes them itself; DW is not touched then.
template <class DT, int AXIS>
int launch_updat32_win(const PtrList8& xs, const PtrList8& es, void* DW, const bsmm_args* a, bool raw_sums) {
    typedef typename DT::T T;
    hipStream_t st = static_cast<hipStream_t>(a->stream);
    const int N = a->N, nitems = a->plan_items;
    if (nitems <= 0) return BSMM_ERR_ARG;
    static bool attr_set = false;   // benign race: idempotent
    if (!attr_set) {
        if constexpr (AXIS == 0)
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&updat32_a0_win_kernel<DT>), hipFuncAttributeMaxDynamicSharedMemorySize, UW0_LDS);
        else {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&updat32_a1_win_kernel<DT, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, UWN_LDS);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&updat32_a1_win_kernel<DT, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, UWN_LDS);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&updat32_a1_win_kernel<DT, 16, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, UWN_LDS);
        }
        attr_set = true;
    }
    // plan_aux = window side of the plan (+ 256 when it was built for 16 waves per workgroup): 16x16 windows use 32-row chunks
    const bool wide_win = AXIS == 1 && (a->plan_aux & 255) == 16, waves16 = wide_win && (a->plan_aux & 256);
    const int nchunks = wide_win ? (N + 31) / 32 : (N + 63) / 64;
    int split = 1;
    while (nitems * split < 256 && split * 2 <= nchunks / 8 && split < 8) split *= 2;   // one workgroup per CU, >= 8 chunks each
    const char* senv = getenv("BSMM_UPDAT_SPLIT");
    if (senv) split = atoi(senv) > 0 ? atoi(senv) : split;
    float* scratch = nullptr;
    if (split > 1 || raw_sums) {
        const size_t need = (size_t)a->blocks * 1024 * sizeof(float);
        if (!a->workspace || a->workspace_bytes < need || !aligned16(a->workspace)) return BSMM_ERR_WORKSPACE;
        scratch = static_cast<float*>(a->workspace);
        hipError_t e = hipMemsetAsync(scratch, 0, need, st);
        if (e != hipSuccess) return (int)e;
    }
    if constexpr (AXIS == 0)
        updat32_a0_win_kernel<DT><<<dim3(nitems, split), 512, UW0_LDS, st>>>(xs, es, static_cast<T*>(DW), scratch, a->plan, N, a->C, a->K, a->pcount,
                                                                            a->alpha, a->beta);
    else if (waves16)
        updat32_a1_win_kernel<DT, 16, 16><<<dim3(nitems, split), 1024, UWN_LDS, st>>>(xs, es, static_cast<T*>(DW), scratch, a->plan, N, a->C, a->K,
                                                                                     a->pcount, a->alpha, a->beta);
    else if (wide_win)
        updat32_a1_win_kernel<DT, 16><<<dim3(nitems, split), 512, UWN_LDS, st>>>(xs, es, static_cast<T*>(DW), scratch, a->plan, N, a->C, a->K,
                                                                                a->pcount, a->alpha, a->beta);
    else
        updat32_a1_win_kernel<DT, 8><<<dim3(nitems, split), 512, UWN_LDS, st>>>(xs, es, static_cast<T*>(DW), scratch, a->plan, N, a->C, a->K, a->pcount,
                                                                               a->alpha, a->beta);
    if (scratch && !raw_sums) {
        const size_t n = (size_t)a->blocks * 1024;
        updat_finalize_kernel<DT><<<(unsigned)((n / 4 + 255) / 256), 256, 0, st>>>(scratch, static_cast<T*>(DW), n, a->alpha, a->beta);
    }
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
    const int variant = g_variant.load(std::memory_order_relaxed);
    const bool use_valu = (BS == 8) || variant == 1 || !vec_ok;
    const float* ug = (a->flags & BSMM_FLAG_GATED_DW) ? a->gate : nullptr;   // gated dw: per-block kernels only
    const bool gated = ug != nullptr;
    if constexpr (BS == 8 && DT::is16) {
        // bsize 8 on the matrix cores: fp32 sums of whole 32x32 super-blocks ('BSS8' plan) into the workspace, then the
        // present 8x8 parts get alpha / beta and are rounded once
        bool al = aligned16(DW) && a->C % 32 == 0 && a->K % 32 == 0 && !(AXIS == 0 && (N % 8 != 0));
        for (int p = 0; p < a->pcount; ++p) al = al && aligned16(xs.p[p]) && aligned16(es.p[p]);
        if (a->plan != nullptr && a->plan_aux > 0 && a->plan_items > 0 && !gated && al && (variant == 0 || variant == 3)) {
            const int ns = a->plan_aux;
            bsmm_args b = *a;
            b.bsize = 32; b.blocks = ns; b.plan_aux = 0; b.flags = 0; b.gate = nullptr;
            b.lut = a->plan + s8_off_lut32(ns);
            b.plan = a->plan + s8_off_nested(ns);
            const int rc = launch_updat32_win<DT, AXIS>(xs, es, nullptr, &b, true);
            if (rc) return rc;
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
        if (!use_valu && !gated && al && (variant == 0 || variant == 3) && a->plan != nullptr && a->plan_items > 0) {   // windowed kernels (plan = bsmm_updat_plan_build)
            // Sparse layouts at small minibatch (BASELINE configs[3]'s per-GPU shard: 8192^2, 5 %, N = 512): a window holds ~3
            // blocks, so the windowed kernel streams 64 KiB per chunk for almost nothing, while the per-block transposing-read
            // kernel moves 128 bytes per (block, row).  Fitted to measurements (us): windowed 8 + rounds * chunks * 0.9 (1.8x that
            // per chunk with 16x16 windows); per block 8 + rounds of 512 blocks * N * 0.0065 .. 0.0105.
            bool windowed = true;
            if (AXIS == 1 && variant == 0) {
                const bool w16 = (a->plan_aux & 255) == 16;
                const double chunks = std::ceil(N / 64.0) * a->pcount;                 // 64-row units per window
                int split = 1;                                                          // as launch_updat32_win chooses it
                while (a->plan_items * split < 256 && split * 2 <= (w16 ? (N + 31) / 32 : (N + 63) / 64) / 8 && split < 8) split *= 2;
                const double rounds = std::max(1.0, std::ceil(a->plan_items * (double)split / 256.0));
                const double t_win = 8.0 + rounds * (chunks / split) * 0.9 * (w16 ? 1.8 : 1.0) + (split > 1 ? 6.0 : 0.0);
                // per-block kernel: two workgroups per CU, each walks the whole minibatch for ONE block
                const double rounds_b = std::max(1.0, std::ceil(a->blocks / 512.0));
                const double t_blk = 8.0 + rounds_b * (double)N * a->pcount * (N > 1024 ? 0.0105 : 0.0065);
                windowed = t_win <= t_blk;
            }
            if (windowed) return launch_updat32_win<DT, AXIS>(xs, es, DW, a, false);
        }
    }
    if constexpr (BS == 32 && AXIS == 1 && DT::is16) {
        bool al = aligned16(DW);
        for (int p = 0; p < a->pcount; ++p) al = al && aligned16(xs.p[p]) && aligned16(es.p[p]);
        if (!use_valu && !gated && al && variant != 1) {   // LDS-DMA + transposing-read kernel
            static bool attr_set = false;
            if (!attr_set) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&updat32_a1_tr_kernel<DT>), hipFuncAttributeMaxDynamicSharedMemorySize, UT_LDS);
                attr_set = true;
            }
            const int grid = 8 * ((a->blocks + 7) / 8);
            updat32_a1_tr_kernel<DT><<<grid, 256, UT_LDS, st>>>(xs, es, static_cast<T*>(DW), a->lut, a->blocks, N, a->C, a->K, a->pcount,
                                                              a->alpha, a->beta);
            return (int)hipGetLastError();
        }
    }
    if constexpr (BS == 16 && DT::is16) {
        bool al16 = aligned16(DW) && (AXIS == 1 || N % 8 == 0);
        for (int p = 0; p < a->pcount; ++p) al16 = al16 && aligned16(xs.p[p]) && aligned16(es.p[p]);
        if (!use_valu && !gated && al16 && (variant == 0 || variant == 3) && a->plan != nullptr && a->plan_items > 0) {   // windowed, 16x16 blocks
            static bool attr_set_w16 = false;
            if (!attr_set_w16) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&updat16_win_kernel<DT, AXIS>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * UWN_SLOT);
                attr_set_w16 = true;
            }
            const int nitems = a->plan_items;
            const int nchunks = (N + 63) / 64;
            int split = 1;
            while (nitems * split < 256 && split * 2 <= nchunks / 8 && split < 8) split *= 2;
            const char* senv = getenv("BSMM_UPDAT_SPLIT");
            if (senv) split = atoi(senv) > 0 ? atoi(senv) : split;
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
            static bool attr_set16 = false;
            if (!attr_set16) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&updat16_a1_tr_kernel<DT>), hipFuncAttributeMaxDynamicSharedMemorySize, UT16_LDS);
                attr_set16 = true;
            }
            const int grid = 8 * ((a->blocks + 7) / 8);
            updat16_a1_tr_kernel<DT><<<grid, 256, UT16_LDS, st>>>(xs, es, static_cast<T*>(DW), a->lut, a->blocks, N, a->C, a->K, a->pcount,
                                                                a->alpha, a->beta);
            return (int)hipGetLastError();
        }
    }
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
    if (!X || !DY || !DW) return BSMM_ERR_ARG;
    if (a->pcount < 1 || a->pcount > 8) return BSMM_ERR_ARG;
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
    dim3 grid(std::min((N + 255) / 256, 64), K);
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
    dim3 grid(std::min((N + 255) / 256, 64), K);
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

#ifdef BSMM_XC_TRACE
int bsmm_debug_trace_copy(void* host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(bsmm::g_xc_trace), sizeof(bsmm::g_xc_trace)); }
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

long bsmm_xprop_plan_words(const int32_t* host_lut, int32_t segments, int32_t blocks, int32_t n_out_blocks, int32_t bsize,
                           int32_t dtype, int32_t axis) {
    if (axis != 0 && axis != 1) return 0;
    if (bsize == 8) return dtype == BSMM_F32 ? 0 : build_super8_xprop_plan(host_lut, segments, blocks, n_out_blocks, nullptr, xc_group(axis));   // 'BSS8'
    if (bsize != 32 && bsize != 16) return 0;   // plan kernels: bsize 32 (any dtype) / 16 and 8 (16-bit)
    if (dtype == BSMM_F32) {
        if (bsize != 32 || !use_xcol()) return 0;
        return f32_split(axis) ? build_xcol_plan(host_lut, segments, blocks, n_out_blocks, nullptr, XS_G)
                               : build_xcolf_plan(host_lut, segments, blocks, n_out_blocks, nullptr);
    }
    if (bsize == 16) return build_xcol16_plan(host_lut, segments, blocks, n_out_blocks, nullptr, xc16_group());
    return build_xcol_plan(host_lut, segments, blocks, n_out_blocks, nullptr, xc_group(axis));
}

int bsmm_xprop_plan_build(const int32_t* host_lut, int32_t segments, int32_t blocks, int32_t n_out_blocks, int32_t bsize,
                          int32_t dtype, int32_t axis, int32_t* host_plan_out) {
    if (!host_plan_out) return BSMM_ERR_ARG;
    if (bsize == 8 && dtype != BSMM_F32 && (axis == 0 || axis == 1))
        return build_super8_xprop_plan(host_lut, segments, blocks, n_out_blocks, host_plan_out, xc_group(axis)) > 0 ? BSMM_OK : BSMM_ERR_ARG;
    if ((bsize != 32 && bsize != 16) || (axis != 0 && axis != 1)) return BSMM_ERR_UNSUPPORTED;
    if (dtype == BSMM_F32) {
        if (bsize != 32 || !use_xcol()) return BSMM_ERR_UNSUPPORTED;
        if (f32_split(axis)) return build_xcol_plan(host_lut, segments, blocks, n_out_blocks, host_plan_out, XS_G) > 0 ? BSMM_OK : BSMM_ERR_ARG;
        return build_xcolf_plan(host_lut, segments, blocks, n_out_blocks, host_plan_out) > 0 ? BSMM_OK : BSMM_ERR_ARG;
    }
    if (bsize == 16) return build_xcol16_plan(host_lut, segments, blocks, n_out_blocks, host_plan_out, xc16_group()) > 0 ? BSMM_OK : BSMM_ERR_ARG;
    return build_xcol_plan(host_lut, segments, blocks, n_out_blocks, host_plan_out, xc_group(axis)) > 0 ? BSMM_OK : BSMM_ERR_ARG;
}

// bsize 32, axis 1: 16x16-block windows when they hold <= 16 blocks on average (sparse layouts), else 8x8 (bsmm_updat_win.h)
static int updat_window(int32_t blocks, int32_t CB, int32_t KB, int32_t axis) {   // 8, 16, or 1616 (16x16 windows, 16 waves)
    static const int force = [] { const char* e = getenv("BSMM_UPDAT_WINDOW"); return e ? atoi(e) : 0; }();   // A/B runs
    if (axis != 1) return UW;
    if (force == 8 || force == 16 || force == 1616) return force;
    const double windows = (double)((CB + 15) / 16) * ((KB + 15) / 16);
    return blocks <= 16.0 * windows ? 16 : UW;     // <= 2 block slots per wave on average (measured: 13 per window 2.1x faster, 26 per window 20 % slower)
}
static long updat32_plan(const int32_t* lut, int32_t blocks, int32_t CB, int32_t KB, int32_t axis, int32_t* out) {
    const int w = updat_window(blocks, CB, KB, axis);
    return build_updat_plan(lut, blocks, CB, KB, w == 8 ? 8 : 16, UP_MAXB, out, w == 1616 ? 16 : 8);
}

long bsmm_updat_plan_words(const int32_t* host_updat_lut, int32_t blocks, int32_t CB, int32_t KB, int32_t bsize, int32_t dtype,
                           int32_t axis) {
    if (dtype == BSMM_F32 || (axis != 0 && axis != 1)) return 0;   // windowed kernels: 16-bit types
    if (bsize == 8) return build_super8_updat_plan(host_updat_lut, blocks, CB, KB, nullptr);   // 'BSS8'
    if (bsize != 32 && bsize != 16) return 0;
    return bsize == 32 ? updat32_plan(host_updat_lut, blocks, CB, KB, axis, nullptr)
                       : build_updat_plan(host_updat_lut, blocks, CB, KB, UW16, UP16_MAXB, nullptr);
}

int bsmm_updat_plan_build(const int32_t* host_updat_lut, int32_t blocks, int32_t CB, int32_t KB, int32_t bsize, int32_t dtype,
                          int32_t axis, int32_t* host_plan_out) {
    if (!host_plan_out) return BSMM_ERR_ARG;
    if (bsize == 8 && dtype != BSMM_F32 && (axis == 0 || axis == 1))
        return build_super8_updat_plan(host_updat_lut, blocks, CB, KB, host_plan_out) > 0 ? BSMM_OK : BSMM_ERR_ARG;
    if ((bsize != 32 && bsize != 16) || dtype == BSMM_F32 || (axis != 0 && axis != 1)) return BSMM_ERR_UNSUPPORTED;
    const long n = bsize == 32 ? updat32_plan(host_updat_lut, blocks, CB, KB, axis, host_plan_out)
                               : build_updat_plan(host_updat_lut, blocks, CB, KB, UW16, UP16_MAXB, host_plan_out);
    return n > 0 ? BSMM_OK : BSMM_ERR_ARG;
}

size_t bsmm_workspace_bytes(int op, const bsmm_args* a) {
    if (!a) return 0;
    if (a->bsize == 8) {   // 'BSS8' plans: the expanded W (xprop) / the fp32 sums of the super-blocks (updat)
        if (!a->plan || a->plan_aux <= 0 || a->dtype == BSMM_F32) return 0;
        const size_t blk = (size_t)a->plan_aux * 1024;
        return op == BSMM_OP_UPDAT ? blk * sizeof(float) : blk * elem_size(a->dtype);
    }
    if (op == BSMM_OP_UPDAT && a->plan && (a->bsize == 32 || a->bsize == 16) && a->dtype != BSMM_F32)
        return (size_t)a->blocks * a->bsize * a->bsize * sizeof(float);   // fp32 partial sums of the split-minibatch path
    if ((op == BSMM_OP_FPROP || op == BSMM_OP_BPROP) && a->dtype == BSMM_F32 && a->bsize == 32 && a->plan && f32_split(a->axis))
        return xcols_workspace_bytes(a);   // bf16 pieces of the activations and the weights (bsmm_xcols.h)
    // fprop keeps a transposed copy of W (the matrix-core operand wants the contraction index contiguous)
    if (op == BSMM_OP_FPROP && a->bsize != 8) return (size_t)a->blocks * a->bsize * a->bsize * elem_size(a->dtype);
    return 0;
}

void bsmm_set_kernel_variant(int variant) { g_variant.store((variant >= 1 && variant <= 3) ? variant : 0); }
int bsmm_get_kernel_variant(void) { return g_variant.load(); }

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
