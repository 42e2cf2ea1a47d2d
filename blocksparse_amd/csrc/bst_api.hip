// bst_api.hip -- C ABI (include/bst.h) of the block-sparse attention path: argument checks and kernel dispatch.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>

#include "bst.h"
#include "bst_kernels.h"

using namespace bsmm;

namespace {

int check_common(const bst_args* a) {
    if (!a || !a->lut) return BSMM_ERR_ARG;
    if (a->blocks <= 0 || a->batch <= 0 || a->heads <= 0 || a->ctx_blks_q <= 0 || a->ctx_blks_k <= 0 || a->lut_dim <= 0) return BSMM_ERR_ARG;
    if (a->lut_heads != 1 && a->lut_heads != a->heads) return BSMM_ERR_ARG;          // src/bst_op.cc:209
    if (a->bsize != 8 && a->bsize != 16 && a->bsize != 32 && a->bsize != 64) return BSMM_ERR_UNSUPPORTED;
    if (reinterpret_cast<uintptr_t>(a->lut) & 7) return BSMM_ERR_ARG;
    return BSMM_OK;
}
int check_mm(const bst_args* a) {
    if (int rc = check_common(a)) return rc;
    if (a->head_state <= 0 || (a->head_state & 7)) return BSMM_ERR_ARG;               // src/bst_op.cc:208
    if (a->dtype != BSMM_F32 && a->dtype != BSMM_F16 && a->dtype != BSMM_BF16) return BSMM_ERR_UNSUPPORTED;
    if (a->score_dtype != BSMM_F16 && a->score_dtype != BSMM_BF16) return BSMM_ERR_UNSUPPORTED;
    if ((unsigned long long)a->batch * a->heads * a->blocks * a->bsize * a->bsize >= (1ull << 40)) return BSMM_ERR_ARG;
    return BSMM_OK;
}
inline int lut_stride(const bst_args* a) { return a->lut_heads > 1 ? 2 * a->lut_dim : 0; }

// dispatch a generic lambda on (activation type, score type, block size)
template <class F>
int by_types(int act, int score, F&& f) {
    auto with_score = [&](auto ta) {
        if (score == BSMM_BF16) return f(ta, DTbf16{});
        return f(ta, DTf16{});
    };
    if (act == BSMM_F32) return with_score(DTf32{});
    if (act == BSMM_F16) return with_score(DTf16{});
    return with_score(DTbf16{});
}
template <class F>
int by_bsize(int bs, F&& f) {
    switch (bs) {
        case 8: return f(std::integral_constant<int, 8>{});
        case 16: return f(std::integral_constant<int, 16>{});
        case 32: return f(std::integral_constant<int, 32>{});
        default: return f(std::integral_constant<int, 64>{});
    }
}

}  // namespace

extern "C" {

int bst_nt(const void* a_, const void* b_, void* s_, const bst_args* a) {
    if (int rc = check_mm(a)) return rc;
    if (!a_ || !b_ || !s_) return BSMM_ERR_ARG;
    if (a->lut_dim != a->blocks) return BSMM_ERR_ARG;
    hipStream_t st = static_cast<hipStream_t>(a->stream);
    return by_types(a->dtype, a->score_dtype, [&](auto ta, auto ts) {
        typedef decltype(ta) TA;
        typedef decltype(ts) TS;
        return by_bsize(a->bsize, [&](auto bs_tag) {
            constexpr int BS = decltype(bs_tag)::value;
            const auto* A = static_cast<const typename TA::T*>(a_);
            const auto* B = static_cast<const typename TA::T*>(b_);
            auto* S = static_cast<typename TS::T*>(s_);
            const int rq = a->ctx_blks_q * BS, rk = a->ctx_blks_k * BS;
            if constexpr (BS >= 32) {
                constexpr int SUB = BS / 32;
                const int ntiles = a->blocks * SUB * SUB;
                auto staged = [&](auto ch_tag) {
                    constexpr int CH = decltype(ch_tag)::value;
                    const int il = 1;     // heads of an XCD interleaved workgroup by workgroup: 2 / 4 / 8 made no difference
                    const int grid = xcd_head_grid((ntiles + NT_NB - 1) / NT_NB, a->heads, a->batch, il);
                    const bool split = !(a->flags & BST_FLAG_FP32_MFMA);
                    if constexpr (!TA::is16) {
                        if (split) {     // fp32 activations: exact bf16 piece products on the 16-bit matrix core
                            bst_nt_mfma_kernel<TA, TS, BS, CH, true><<<grid, 64, 0, st>>>(A, B, S, a->lut, lut_stride(a), a->blocks, a->heads, a->batch, a->head_state, rq, rk, il);
                            return;
                        }
                    }
                    bst_nt_mfma_kernel<TA, TS, BS, CH><<<grid, 64, 0, st>>>(A, B, S, a->lut, lut_stride(a), a->blocks, a->heads, a->batch, a->head_state, rq, rk, il);
                };
                // LDS-DMA kernel: whole 32-feature chunks only (1, 2 or 4 of them); other head sizes take the direct kernel
                if (a->head_state == 32) staged(std::integral_constant<int, 1>{});
                else if (a->head_state == 64) staged(std::integral_constant<int, 2>{});
                else if (a->head_state == 128) staged(std::integral_constant<int, 4>{});
                else {
                    const int grid = xcd_head_grid((ntiles + 3) / 4, a->heads, a->batch);
                    bst_nt_mfma_direct_kernel<TA, TS, BS><<<grid, 256, 0, st>>>(A, B, S, a->lut, lut_stride(a), a->blocks, a->heads, a->batch, a->head_state, rq, rk);
                }
            } else {
                dim3 grid(a->blocks, a->heads, a->batch);
                bst_nt_valu_kernel<TA, TS, BS><<<grid, BS * BS, 0, st>>>(A, B, S, a->lut, lut_stride(a), a->blocks, a->heads, a->head_state, rq, rk);
            }
            return (int)hipGetLastError();
        });
    });
}

static int bst_xn(const void* s_, const void* b_, void* c_, const bst_args* a, bool trans) {
    if (int rc = check_mm(a)) return rc;
    if (!s_ || !b_ || !c_) return BSMM_ERR_ARG;
    const int ctx_c = trans ? a->ctx_blks_k : a->ctx_blks_q, ctx_b = trans ? a->ctx_blks_q : a->ctx_blks_k;
    if (a->lut_dim != ctx_c + a->blocks) return BSMM_ERR_ARG;
    hipStream_t st = static_cast<hipStream_t>(a->stream);
    return by_types(a->dtype, a->score_dtype, [&](auto tb, auto ts) {
        typedef decltype(tb) TB;
        typedef decltype(ts) TS;
        return by_bsize(a->bsize, [&](auto bs_tag) {
            constexpr int BS = decltype(bs_tag)::value;
            const auto* S = static_cast<const typename TS::T*>(s_);
            const auto* B = static_cast<const typename TB::T*>(b_);
            auto* C = static_cast<typename TB::T*>(c_);
            const int rb = ctx_b * BS, rc_ = ctx_c * BS;
            auto launch = [&](auto tr) {
                constexpr bool TR = decltype(tr)::value;
                if constexpr (BS >= 32 && TB::is16 && std::is_same<TB, TS>::value) {   // 16-bit scores and activations of one type: native 16-bit MFMA
                    if (a->head_state % 32 == 0) {
                        constexpr int SUB16 = BS / 32;
                        const int grid16 = xcd_head_grid(ctx_c * SUB16 * (a->head_state / 32), a->heads, a->batch);
                        bst_xn_mfma16_kernel<TB, BS, TR><<<grid16, 256, 0, st>>>(S, B, C, a->lut, lut_stride(a), a->blocks, a->heads, a->batch, a->head_state, ctx_c, rb, rc_);
                        return;
                    }
                }
                if constexpr (BS >= 32 && std::is_same<TB, DTf32>::value && std::is_same<TS, DTbf16>::value) {
                    // fp32 activations x bf16 scores: exact three-way bf16 split on the 16-bit matrix core (BST_FLAG_FP32_MFMA: fp32 MFMA)
                    const bool split = !(a->flags & BST_FLAG_FP32_MFMA);
                    if (split) {
                        constexpr int SUBS = BS / 32;
                        const int grids = xcd_head_grid(ctx_c * SUBS * ((a->head_state + 31) / 32), a->heads, a->batch);
                        bst_xn_split_kernel<BS, TR><<<grids, 256, 0, st>>>(S, B, C, a->lut, lut_stride(a), a->blocks, a->heads, a->batch, a->head_state, ctx_c, rb, rc_);
                        return;
                    }
                }
                if constexpr (BS >= 32) {
                    constexpr int SUB = BS / 32;
                    const int nct = (a->head_state + 31) / 32;
                    const int grid = xcd_head_grid(ctx_c * SUB * nct, a->heads, a->batch);
                    bst_xn_mfma_kernel<TS, TB, BS, TR><<<grid, 256, 0, st>>>(S, B, C, a->lut, lut_stride(a), a->blocks, a->heads, a->batch, a->head_state, ctx_c, rb, rc_);
                } else {
                    dim3 grid(ctx_c, a->heads, a->batch);
                    bst_xn_valu_kernel<TS, TB, BS, TR><<<grid, 256, 0, st>>>(S, B, C, a->lut, lut_stride(a), a->blocks, a->heads, a->head_state, rb, rc_);
                }
            };
            if (trans) launch(std::true_type{});
            else       launch(std::false_type{});
            return (int)hipGetLastError();
        });
    });
}
int bst_nn(const void* s, const void* b, void* c, const bst_args* a) { return bst_xn(s, b, c, a, false); }
int bst_tn(const void* s, const void* b, void* c, const bst_args* a) { return bst_xn(s, b, c, a, true); }

int bst_masked_softmax(const void* x, void* y, const void* mask, int32_t mask_heads, float scale, int32_t x_dtype, int32_t y_dtype,
                       const bst_args* a) {
    if (int rc = check_common(a)) return rc;
    if (!x || !y) return BSMM_ERR_ARG;
    if (a->lut_dim != a->ctx_blks_q + a->blocks) return BSMM_ERR_ARG;
    if (mask && mask_heads != 1 && mask_heads != a->heads) return BSMM_ERR_ARG;      // src/bst_op.cc:411
    if ((x_dtype != BSMM_F16 && x_dtype != BSMM_BF16) || (y_dtype != BSMM_F16 && y_dtype != BSMM_BF16)) return BSMM_ERR_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(a->stream);
    return by_types(x_dtype, y_dtype, [&](auto tx, auto ty) {
        typedef decltype(tx) TX;
        typedef decltype(ty) TY;
        if constexpr (!TX::is16) {
            return (int)BSMM_ERR_UNSUPPORTED;
        } else {
            return by_bsize(a->bsize, [&](auto bs_tag) {
                constexpr int BS = decltype(bs_tag)::value;
                typedef typename MaskT<BS>::T MT;
                dim3 grid(a->ctx_blks_q, a->heads, a->batch);
                const int mstride = (mask && mask_heads > 1) ? a->blocks * BS : 0;
                bst_softmax_kernel<TX, TY, BS><<<grid, BS * 8, 0, st>>>(static_cast<const typename TX::T*>(x), static_cast<typename TY::T*>(y), a->lut,
                                                                        lut_stride(a), static_cast<const MT*>(mask), mstride, a->blocks, a->heads, scale);
                return (int)hipGetLastError();
            });
        }
    });
}

// scores + softmax in one launch (round 6, bsize 32): see bst_nt_softmax_kernel.  args->lut = nn_lut; max_row_blocks = the longest query row of the layout.
// probs != NULL: the backward pair -- dx = softmax_grad(round(e . v^T), probs) -- with q_ = e, k_ = v (bst_nt_softmax_grad below)
static int nt_softmax(const void* q_, const void* k_, void* y_, const void* mask, int32_t mask_heads, float scale, int32_t max_row_blocks, const void* probs,
                      const bst_args* a) {
    if (int rc = check_mm(a)) return rc;
    if (!q_ || !k_ || !y_ || max_row_blocks <= 0) return BSMM_ERR_ARG;
    if (a->lut_dim != a->ctx_blks_q + a->blocks) return BSMM_ERR_ARG;
    if (mask && mask_heads != 1 && mask_heads != a->heads) return BSMM_ERR_ARG;
    // what the fused kernel serves; everything else: bst_nt, then bst_masked_softmax
    if (a->bsize != 32 || max_row_blocks > 4 * NTS_MAXT || (a->head_state != 32 && a->head_state != 64 && a->head_state != 128) || (a->flags & BST_FLAG_FP32_MFMA))
        return BSMM_ERR_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(a->stream);
    return by_types(a->dtype, a->score_dtype, [&](auto ta, auto ts) {
        typedef decltype(ta) TA;
        typedef decltype(ts) TS;
        const auto* Q = static_cast<const typename TA::T*>(q_);
        const auto* K = static_cast<const typename TA::T*>(k_);
        auto* Y = static_cast<typename TS::T*>(y_);
        const int rq = a->ctx_blks_q * 32, rk = a->ctx_blks_k * 32;
        const int mstride = (mask && mask_heads > 1) ? a->blocks * 32 : 0;
        const int grid = xcd_head_grid(a->ctx_blks_q, a->heads, a->batch);
        auto go = [&](auto ch_tag) {
            constexpr int CH = decltype(ch_tag)::value;
            if (probs)
                bst_nt_softmax_kernel<TA, TS, CH, !TA::is16, true><<<grid, 256, 0, st>>>(Q, K, Y, a->lut, lut_stride(a), nullptr, 0, a->blocks, a->heads, a->batch,
                                                                                        a->head_state, rq, rk, a->ctx_blks_q, scale, static_cast<const typename TS::T*>(probs));
            else
                bst_nt_softmax_kernel<TA, TS, CH, !TA::is16><<<grid, 256, 0, st>>>(Q, K, Y, a->lut, lut_stride(a), static_cast<const uint32_t*>(mask), mstride, a->blocks,
                                                                                  a->heads, a->batch, a->head_state, rq, rk, a->ctx_blks_q, scale);
        };
        if (a->head_state == 32) go(std::integral_constant<int, 1>{});
        else if (a->head_state == 64) go(std::integral_constant<int, 2>{});
        else go(std::integral_constant<int, 4>{});
        return (int)hipGetLastError();
    });
}

int bst_nt_softmax(const void* q, const void* k, void* y, const void* mask, int32_t mask_heads, float scale, int32_t max_row_blocks, const bst_args* a) {
    return nt_softmax(q, k, y, mask, mask_heads, scale, max_row_blocks, nullptr, a);
}
int bst_nt_softmax_grad(const void* e, const void* v, const void* probs, void* dx, float scale, int32_t max_row_blocks, const bst_args* a) {
    if (!probs) return BSMM_ERR_ARG;
    return nt_softmax(e, v, dx, nullptr, 1, scale, max_row_blocks, probs, a);
}

int bst_softmax_grad(const void* dy, const void* y, void* dx, float scale, int32_t dtype16, const bst_args* a) {
    if (int rc = check_common(a)) return rc;
    if (!dy || !y || !dx) return BSMM_ERR_ARG;
    if (a->lut_dim != a->ctx_blks_q + a->blocks) return BSMM_ERR_ARG;
    if (dtype16 != BSMM_F16 && dtype16 != BSMM_BF16) return BSMM_ERR_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(a->stream);
    auto go = [&](auto t) {
        typedef decltype(t) T16;
        return by_bsize(a->bsize, [&](auto bs_tag) {
            constexpr int BS = decltype(bs_tag)::value;
            dim3 grid(a->ctx_blks_q, a->heads, a->batch);
            bst_softmax_grad_kernel<T16, BS><<<grid, BS * 8, 0, st>>>(static_cast<const typename T16::T*>(dy), static_cast<const typename T16::T*>(y),
                                                                      static_cast<typename T16::T*>(dx), a->lut, lut_stride(a), a->blocks, a->heads, scale);
            return (int)hipGetLastError();
        });
    };
    return dtype16 == BSMM_BF16 ? go(DTbf16{}) : go(DTf16{});
}

int bst_partial_autoregressive_mask(const void* mask_in, void* mask_out, const int32_t* nt_lut, int32_t bsize, int32_t blocks, int32_t lut_heads,
                                    int32_t autoregress_at_k, void* stream) {
    if (!mask_in || !mask_out || !nt_lut || blocks <= 0 || lut_heads <= 0 || autoregress_at_k < 0) return BSMM_ERR_ARG;
    if (bsize != 8 && bsize != 16 && bsize != 32 && bsize != 64) return BSMM_ERR_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    return by_bsize(bsize, [&](auto bs_tag) {
        constexpr int BS = decltype(bs_tag)::value;
        typedef typename MaskT<BS>::T MT;
        dim3 grid((blocks + 63) / 64, BS, lut_heads);
        bst_partial_ar_mask_kernel<BS><<<grid, 64, 0, st>>>(static_cast<const MT*>(mask_in), static_cast<MT*>(mask_out), nt_lut, blocks, autoregress_at_k);
        return (int)hipGetLastError();
    });
}

}  // extern "C"
