/* bst.h -- C ABI of the block-sparse attention path (BlocksparseTransformer: scores, row softmax, weighted values)
 * in libbsmm_hip.so.  SURVEY.md section 8 row a13 / BASELINE.json configs[4].
 *
 * Each entry point replaces one host launcher of the reference (declared in src/bst_op.cc, defined in the .cu files):
 *
 *   bst_nt                          <- bst_sgemm_nt / bst_hgemm_nt      src/bst_op.cc:139-143, src/bst_sgemm_op_gpu.cu:418-446
 *   bst_nn, bst_tn                  <- bst_sgemm_xn / bst_hgemm_xn (op = NN_OP / TN_OP)
 *                                                                       src/bst_op.cc:141-144, src/bst_sgemm_op_gpu.cu:449-497
 *   bst_masked_softmax              <- BlocksparseMaskedSoftmax         src/bst_op.cc:330-340, src/bst_softmax_op_gpu.cu:316-392
 *   bst_softmax_grad                <- BlocksparseMaskedSoftmaxGrad     src/bst_op.cc:434-443, src/bst_softmax_op_gpu.cu:395-458
 *   bst_partial_autoregressive_mask <- BstPartialAutoregressiveMask     src/bst_op.cc:500-502, src/bst_softmax_op_gpu.cu:461-520
 *
 * Tensors (all device pointers owned by the caller; nothing is allocated here; launches are asynchronous on `stream`):
 *   activations  a, b, c of nt / nn / tn:  [batch][ctx_blks * bsize][heads * head_state], dtype `dtype`
 *                (the reference: fp32 or fp16; bf16 is accepted as well)
 *   scores       [batch][heads][blocks][bsize][bsize], dtype `score_dtype` (BSMM_BF16 or BSMM_F16; the reference
 *                stores bf16 scores next to fp32 activations and fp16 next to fp16, src/bst_op.cc:76-78,100-101)
 *   lut          int32 [lut_heads][lut_dim][2], lut_heads == heads or 1 (one table shared by all heads):
 *                  nt_lut: lut_dim = blocks,               entry b = (q block, k block)     blocksparse/transformer.py:109
 *                  nn_lut: lut_dim = ctx_blks_q + blocks,  header q = (offset, count), entries (block id, k block)
 *                  tn_lut: lut_dim = ctx_blks_k + blocks,  header k = (offset, count), entries (block id, q block)
 *                                                                                      blocksparse/transformer.py:141-165
 *   mask         unsigned integers of bsize bits, kernel layout [mask_heads][bsize (query row)][blocks], bit k = key k
 *                of the block visible (blocksparse/transformer.py:146-159); NULL = no mask.
 * Arithmetic: fp32 accumulation (MFMA f32 for bsize 32/64, VALU for 8/16), one rounding to the output type.
 * Softmax: y = exp((x - max) * scale) / sum over the visible keys of all blocks of the query's row-block; a row with no
 *   visible key comes out uniform over its stored entries, as in the reference's NumPy oracle
 *   (blocksparse/transformer.py:259-296); the reference KERNEL yields NaN there.
 * Return value: 0 ok, >0 hipError_t, <0 BSMM_ERR_* (bsmm.h).  Thread-safe per distinct stream; no global state.
 */
#ifndef BST_H
#define BST_H
#include "bsmm.h"

#ifdef __cplusplus
extern "C" {
#endif

enum { BST_FLAG_FP32_MFMA = 1 };   /* fp32 activations: the fp32 matrix-core kernels instead of the exact bf16 three-piece split (A/B) */

typedef struct bst_args {
    const int32_t* lut;   /* device: the table this entry point walks (nt: nt_lut, nn: nn_lut, tn: tn_lut, softmax: nn_lut) */
    int32_t lut_heads;    /* heads or 1                                                                                      */
    int32_t lut_dim;      /* entries per head (see above)                                                                     */
    int32_t blocks;       /* blocks per head                                                                                  */
    int32_t bsize;        /* 8, 16, 32 or 64                                                                                  */
    int32_t batch;
    int32_t heads;
    int32_t head_state;   /* features per head; must be a multiple of 8 (src/bst_op.cc:208)                                   */
    int32_t ctx_blks_q;   /* query  blocks (rows of the layout)                                                               */
    int32_t ctx_blks_k;   /* key    blocks (columns of the layout)                                                            */
    int32_t dtype;        /* activations: BSMM_F32 / BSMM_F16 / BSMM_BF16                                                     */
    int32_t score_dtype;  /* scores: BSMM_BF16 / BSMM_F16                                                                     */
    int32_t flags;        /* BST_FLAG_* (0 = none); kernel choice is a function of the arguments only, never of the environment */
    void* stream;         /* hipStream_t                                                                                      */
} bst_args;

/* scores[n][h][b] = a[n][q-block rows][h] . b[n][k-block rows][h]^T          a: queries (ctx_blks_q), b: keys (ctx_blks_k) */
int bst_nt(const void* a, const void* b, void* scores, const bst_args* args);
/* c[n][q-block rows][h] = sum over the row's blocks  scores[n][h][b] . b[n][k-block rows][h]         (args->lut = nn_lut) */
int bst_nn(const void* scores, const void* b, void* c, const bst_args* args);
/* c[n][k-block rows][h] = sum over the column's blocks  scores[n][h][b]^T . b[n][q-block rows][h]    (args->lut = tn_lut) */
int bst_tn(const void* scores, const void* b, void* c, const bst_args* args);

/* x: scores of dtype x_dtype (the reference: always bf16), y: dtype y_dtype (bf16 / fp16).  args->lut = nn_lut.
 * Only lut, lut_heads, lut_dim, blocks, bsize, batch, heads, ctx_blks_q, stream of args are read.                          */
int bst_masked_softmax(const void* x, void* y, const void* mask, int32_t mask_heads, float scale, int32_t x_dtype,
                       int32_t y_dtype, const bst_args* args);
/* y[n][h][b] = softmax over the query row of (scale * round_to_score_dtype(q . k^T) + mask): bst_nt followed by bst_masked_softmax as ONE launch
 * (round 6) -- the raw scores never reach memory.  Replaces the pair bst_sgemm_nt / bst_hgemm_nt + BlocksparseMaskedSoftmax where a caller
 * composes them (blocksparse/transformer.py:364-409: query_key_op, masked_softmax).  args->lut = nn_lut, args->score_dtype = type of y;
 * max_row_blocks = the longest query row of the layout (the nn_max of the host tables).  Serves bsize 32, head_state 32 / 64 / 128, rows of
 * up to 20 blocks; anything else returns BSMM_ERR_UNSUPPORTED and the caller runs the two entry points.                                      */
int bst_nt_softmax(const void* q, const void* k, void* y, const void* mask, int32_t mask_heads, float scale, int32_t max_row_blocks,
                   const bst_args* args);
/* The backward pair of an attention layer as one launch: dx = (dp - sum_row(dp * probs)) * probs * scale with dp = round_to_score_dtype(e . v^T)
 * -- bst_nt(e, v) followed by bst_softmax_grad(dp, probs) (blocksparse/transformer.py:446-509: the nn gradient w.r.t. the scores, then
 * blocksparse_softmax_grad); dp never reaches memory.  e: gradient of the layer's output (query rows), v: values (key rows), probs and dx of
 * args->score_dtype.  Same limits and the same BSMM_ERR_UNSUPPORTED contract as bst_nt_softmax.                                              */
int bst_nt_softmax_grad(const void* e, const void* v, const void* probs, void* dx, float scale, int32_t max_row_blocks, const bst_args* args);
/* dx = (dy - sum_row(dy * y)) * y * scale; dy, y, dx share `dtype16` (bf16 / fp16).  args->lut = nn_lut.                   */
int bst_softmax_grad(const void* dy, const void* y, void* dx, float scale, int32_t dtype16, const bst_args* args);
/* mask_out = mask_in with keys >= autoregress_at_k made causal (see the kernel cited above).  nt_lut [lut_heads][blocks][2] */
int bst_partial_autoregressive_mask(const void* mask_in, void* mask_out, const int32_t* nt_lut, int32_t bsize, int32_t blocks,
                                    int32_t lut_heads, int32_t autoregress_at_k, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* BST_H */
