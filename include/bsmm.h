/* bsmm.h -- C ABI of the MI355X-native block-sparse matmul engine (libbsmm_hip.so).
 *
 * This is the drop-in boundary for the BlocksparseMatMul hot path of openai/blocksparse.  Every entry
 * point replaces one piece of the reference's TensorFlow custom-op layer (paths are relative to
 * /root/reference):
 *
 *   bsmm_args            <- struct bsmm_params                       src/gpu_types.h:172-194
 *                           (+ axis/dtype, which the reference carries as op attrs / template args,
 *                            src/blocksparse_matmul_op.cc:356-382)
 *   bsmm_fprop           <- op "BlocksparseMatmul"   -> BsmmXprop_CN<true ,T>  /  hgemm_blocksparse_xn_sdd(OP_T) / nx_dsd(OP_N)
 *                           src/blocksparse_matmul_op.cc:120-222, src/blocksparse_matmul_op_gpu.cu:2895-2941
 *   bsmm_bprop           <- op "BlocksparseMatmulDX" -> BsmmXprop_CN<false,T>  (same kernel family, C/K swapped by the
 *                           caller exactly as blocksparse/matmul.py:506-510 does)
 *   bsmm_updat           <- ops "BlocksparseMatmulDW" / "BlocksparseMatmulDWA" -> BsmmUpdat_CN<T>
 *                           src/blocksparse_matmul_op.cc:223-311, src/blocksparse_matmul_op_gpu.cu:2944-2984
 *                           (up to 8 (x,dy) pairs = Plist<T,8>, src/gpu_types.h:167-170; beta!=0 = the DWA in-place form)
 *   bsmm_identity_init   <- op "BlocksparseMatmulIdentityInit" -> IdentityInitCK  src/blocksparse_matmul_op_gpu.cu:2988-3028
 *   bsmm_workspace_bytes <- the scratch the op allocates as output 1 ("temp"/lock scratch), src/blocksparse_matmul_op.cc:150-159
 *
 * Conventions (same as the reference boundary):
 *   - every pointer is a DEVICE pointer owned by the caller; the library never allocates or frees device memory;
 *   - every call only enqueues work on `stream` (a hipStream_t) and returns immediately; no host sync;
 *   - return value: 0 = ok; >0 = a hipError_t from the launch; <0 = BSMM_ERR_* (bad arguments); nothing throws;
 *   - no global mutable state: the library reads no environment variables and keeps no process-wide switches; kernel
 *     choice is a function of the arguments only (bsmm_args.flags, the plan descriptor fields, the problem size), so
 *     concurrent calls on distinct streams are safe.
 *
 * Tensor layouts:
 *   W  [blocks][bsize][bsize]  W[w][ci][ki] = Wdense[c*bsize+ci][k*bsize+ki] with (c,k) = updat_lut[w]
 *   axis 0: X (C,N)  Y (K,N)  row-major, N contiguous          fprop Y = Wd^T X   bprop DX = Wd DY   updat DW[w] = X[c] DY[k]^T
 *   axis 1: X (N,C)  Y (N,K)  row-major, features contiguous   fprop Y = X Wd     bprop DX = DY Wd^T updat DW[w] = X[:,c]^T DY[:,k]
 *   xprop lut  int32[4*segments + 2*blocks]: header (entry_offset/2, n_entries, out_block, lock_id), entry (in_block, w)
 *   updat lut  int32[blocks][2] = (c, k)
 * Accumulation is always fp32; 16-bit outputs are rounded once, round-to-nearest-even.
 */
#ifndef BSMM_H_
#define BSMM_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI version: bumped whenever struct bsmm_args, a plan format or an option bit range changes.  Bindings compare it with the
 * header they were written against (blocksparse_amd/_lib.py does at load time).
 *   100 rounds 1-2;  110 round 3 (bsmm_args.prepared_w, BSMM_PLAN_UPDAT_SETS_SHIFT moved to bits 12..15, 'BSX7' plans v2, composite
 *   'BSS8' / 'BS64' descriptors);  120 round 4 (see DESIGN.md "Round 4");  121 round 4: kernels retired -- 'BSX6' (bsize 16, round 1), 'BSXF' (fp32
 *   matrix-core instruction) and bsize-32 'BSUP' plans are no longer built or accepted, the options that named them are aliases;
 *   122 round 5: 'BSX5' plans (BSMM_PLAN_XCOL_ROWS), BSMM_K_XCOL32_ROWS;  123 round 5: 'BSUP' plans v6 (12 header words; bsize 16 on feature
 *   axis 0 carries a 'BSU6' section for the row-owner weight-gradient kernel), BSMM_K_UPDAT16_ROWS, BSMM_PLAN_UPDAT16_WINDOWED;
 *   124 round 6: the row-split xprop kernel of round 5 retired ('BSX5' plans are no longer built or accepted, BSMM_PLAN_XCOL_ROWS is ignored,
 *   trace code 13 is not emitted; source and measurements: profiles/r05_xrows.patch);  125 round 6: 'BSU2' plans version 3 (32 header words; direct
 *   blocks), BSMM_PLAN_UPDAT_NO_DIRECT;  126 round 6: bsmm_gate_weights;  127 round 6: 'BSX4' plans version 4 (per-group output-block table; unbalanced
 *   layouts regrouped), BSMM_PLAN_FLOW_CONSECUTIVE;  128 round 6: 'BSX2' / 'BSX7' plans version 3 (16 header words, output-block table
 *   per group; unbalanced layouts regrouped: 'BSX2' on feature axis 0, 'BSX7' on both) */
#define BSMM_VERSION 128

enum { BSMM_F32 = 0, BSMM_F16 = 1, BSMM_BF16 = 2 };
enum {
    BSMM_FLAG_GATED_DW = 1,     /* updat: scale dw by the gate (op attr gated_dw, src/blocksparse_matmul_op.cc:363,403)  */
    BSMM_FLAG_FORCE_VALU = 2,   /* per call: plain V_FMA kernels for every bsize (independent second implementation)      */
    BSMM_FLAG_NO_PLAN = 4,      /* per call: ignore bsmm_args.plan (per-segment / per-block matrix-core kernels)          */
    BSMM_FLAG_FORCE_PLAN = 8,   /* per call: take the plan kernels whenever a plan is given, whatever the size heuristic   */
    BSMM_FLAG_DW_SUMS = 16,     /* updat: leave the raw fp32 sums sum_p X_p DY_p^T of every block in the workspace
                                   ([blocks][bsize][bsize] floats at its start) and do NOT write DW -- the data-parallel path
                                   all-reduces those sums in fp32 and then calls bsmm_updat_finalize().  Kernels that cannot
                                   (no plan / not the streaming kernel) answer BSMM_ERR_UNSUPPORTED                          */
    BSMM_FLAG_FORCE_MID = 32    /* per call, xprop: the medium-minibatch kernel (bsmm_xmid.h: bsize 32, 16-bit, feature axis 1, no gate,
                                   no locks) whenever it can run, whatever the cost model says (tests)                       */
};

/* bsmm_args.trace: which kernel family a call dispatched to (tests assert that the intended kernel ran): the BSMM_K_* code in
 * bits 0..7; bits 8..15 name a variant inside the family (BSMM_KV_*, 0 = the plain one) */
enum { BSMM_KV_ONE_WAVE = 1 /* BSMM_K_UPDAT_BLOCK_TR: the small-minibatch form, one wave per weight block (round 4) */,
       BSMM_KV_FLOW_HALF_UNITS = 2 /* BSMM_K_XCOL32_FLOW: units of 64 rows (minibatches that do not fill the chip with 128-row units); 0 = 128 rows */ };
enum {
    BSMM_K_NONE = 0,
    /* (4 BSMM_K_XCOL16, 6 BSMM_K_XCOL32_F32MFMA and 19 BSMM_K_UPDAT_WIN belonged to kernels retired in round 4: never reported any more) */
    BSMM_K_XPROP_VALU = 1, BSMM_K_XPROP_SEGMENT = 2, BSMM_K_XCOL32 = 3, BSMM_K_XCOL16 = 4, BSMM_K_XCOL32_F32SPLIT = 5,
    BSMM_K_XCOL32_F32MFMA = 6, BSMM_K_XPROP_SUPER8 = 7, BSMM_K_XCOL32_STAGED = 8, BSMM_K_XCOL16_STAGED = 9, BSMM_K_XCOL32_FLOW = 10, BSMM_K_XPROP_SMALL = 11, BSMM_K_XPROP_MID = 12, BSMM_K_XCOL32_ROWS = 13 /* retired, never emitted */,
    BSMM_K_UPDAT_VALU = 16, BSMM_K_UPDAT_BLOCK = 17, BSMM_K_UPDAT_BLOCK_TR = 18, BSMM_K_UPDAT_WIN = 19, BSMM_K_UPDAT16_WIN = 20,
    BSMM_K_UPDAT_SUPER8 = 21, BSMM_K_UPDAT_STREAM = 22, BSMM_K_UPDAT16_ROWS = 23
};

/* options of the plan builders (0 = the library's default for the layout) */
enum {
    BSMM_PLAN_XCOL_NARROW = 1,      /* xprop bsize 32 / 16: 8 (16) output blocks per workgroup instead of 16 (32)            */
    BSMM_PLAN_F32_MFMA = 2,         /* (retired in round 4: the fp32 matrix-core kernel xcol32f, 0.65 ms against 0.48 for the exact bf16 split
                                       at the bench shape; accepted and ignored)                                                 */
    BSMM_PLAN_XCOL_UNSTAGED = 4,    /* xprop bsize 32, 16-bit: the round-1 kernel (weights by register loads, bsmm_xcol.h) instead of the
                                       staged one (weights through LDS as well, bsmm_xcol_v2.h).  bsize 16: its round-1 kernel was
                                       retired in round 4 -- ignored there, as is _NARROW                                        */
    BSMM_PLAN_XCOL_FLOW = 8,        /* xprop bsize 32, 16-bit, feature axis 1: the barrier-free persistent kernel (bsmm_xflow.h, 'BSX4' plans) */
    BSMM_PLAN_WINDOW_8 = 0x10,      /* updat bsize 32: (the windowed kernels of round 1 were retired in round 4: now aliases)  = BSMM_PLAN_STREAM_8  */
    BSMM_PLAN_WINDOW_16 = 0x20,     /*                                                                                  = BSMM_PLAN_STREAM_16 */
    BSMM_PLAN_WINDOW_16W = 0x30,    /*                                                                                  = BSMM_PLAN_STREAM_16 */
    BSMM_PLAN_STREAM_16 = 0x40,     /*                 streaming kernel (bsmm_updat_v2.h, axis 1), 16x16-block windows      */
    BSMM_PLAN_STREAM_8 = 0x50,      /*                 streaming kernel, 8x8-block windows (dense layouts)                  */
    BSMM_PLAN_STREAM_32 = 0x60,     /*                 streaming kernel, 32x32-block windows (sparse layouts, feature axis 1)    */
    BSMM_PLAN_WINDOW_MASK = 0xf0,   /* (0: bsize 32, either feature axis -> streaming kernel, window side by density; bsize 16 -> 16x16 windows) */
    /* experiment knobs of the builders (0 = the builder's own choice); disjoint bit ranges, one meaning each: */
    BSMM_PLAN_XPROP_PH_SHIFT = 8,   /* bits  8..10  xprop staged plans ('BSX2'): steps per phase (2, 3, 4)                      */
    BSMM_PLAN_UPDAT_SETS_SHIFT = 12,/* bits 12..15  updat streaming plan ('BSU2'): item sets (1, 2, 4, 8)                       */
    BSMM_PLAN_XCOL_ROWS = 0x20000,  /* retired in round 6 (named the row-split xprop kernel of round 5, slower than the flow kernel): ignored       */
    BSMM_PLAN_UPDAT16_WINDOWED = 0x40000, /* updat bsize 16, feature axis 0: do NOT append the 'BSU6' section (row-owner kernel, bsmm_updat16_rows.h,
                                       round 5: 512 x 512-feature windows, X rows straight into registers) -- the call then always runs the
                                       windowed kernel (256 x 256-feature windows through LDS); comparison / tests                        */
    BSMM_PLAN_UPDAT_NO_DIRECT = 0x80000, /* updat bsize 32, feature axis 1 ('BSU2' plans): no DIRECT blocks (round 6: the blocks a window's 16 waves cannot hold
                                       get workgroups of their own behind the schedule's) -- overflow items in a sliced last round, as before; comparison / tests */
    BSMM_PLAN_FLOW_CONSECUTIVE = 0x100000, /* 'BSX4' / 'BSX2' / 'BSX7' plans: always groups of CONSECUTIVE output blocks (round 6: an unbalanced layout -- hubs -- is
                                       regrouped by default: pairs of output blocks dealt to the groups heaviest first; same results) */
    BSMM_PLAN_FLOW_SCHEDULED = 0x10000 /* 'BSX4' plans, experiment: steps in the order the builder's list scheduling picks instead of ascending
                                       input blocks (the same sums in another fp32 summation order; measured no faster, see bsmm_plan.h) */
};

enum {
    BSMM_OK = 0,
    BSMM_ERR_ARG = -1,         /* NULL pointer / non-positive size / pcount out of 1..8 / plan descriptor mismatch */
    BSMM_ERR_UNSUPPORTED = -2, /* bsize not in {8,16,32}, axis not in {0,1}, unknown dtype, gating */
    BSMM_ERR_WORKSPACE = -3    /* workspace pointer NULL or smaller than bsmm_workspace_bytes()     */
};

enum { BSMM_OP_FPROP = 0, BSMM_OP_BPROP = 1, BSMM_OP_UPDAT = 2 };

typedef struct bsmm_args {
    const int32_t* lut;     /* device: fprop_lut (fprop) / bprop_lut (bprop) / updat_lut (updat)                    */
    const float* gate;      /* optional per-block fp32 gate [blocks] (reference: Gate, src/gpu_types.h:175).  xprop: block
                               w contributes gate[w] * (its product), gate 0 = skipped; updat: only read when
                               flags & BSMM_FLAG_GATED_DW, then DW[w] = alpha * gate[w] * sum + beta * DW[w].  Gated calls run
                               the plan kernels for bsize 32 with 16-bit types (staged xprop kernel: exact two-piece split
                               of gate * w; streaming updat kernel: gate applied in its reduce pass), else the per-segment /
                               per-block kernels.                                                                        */
    void* workspace;        /* device scratch of >= bsmm_workspace_bytes(op, args) bytes (may be NULL when that is 0) */
    size_t workspace_bytes;
    const int32_t* plan;    /* optional device copy of the schedule built by bsmm_xprop_plan_build() (fprop/bprop) or
                               bsmm_updat_plan_build() (updat) for THIS lut (NULL = generic kernels).  Like the luts
                               it is a constant of the layout.                                                       */
    int32_t plan_magic;     /* plan descriptor, filled by bsmm_plan_attach() from the HOST copy of the plan: format tag    */
    int32_t plan_width;     /*   output blocks per workgroup (xprop) / window side (updat); bsize 8: number of super-blocks */
    int32_t plan_waves;     /*   waves per workgroup the schedule was dealt for                                          */
    int32_t plan_items;     /*   updat: number of work items (= grid size)                                               */
    int32_t plan_inner;     /*   staged xprop plan: steps per phase; streaming updat plan: item sets | 16 if all equally long | longest set << 8;
                                 bsize 8: width / window side of the nested bsize-32 plan | its format << 8 | its own word of this kind << 11;
                                 bsize 16 updat plan ('BSUP'): word offset of its 'BSU6' section (row-owner kernel, feature axis 0; 0 = none) --
                                 the section's window width and item count then ride in bits 8.. of plan_width / plan_waves;
                                 bsize 64: 0 = xprop plan, 1 = updat plan (the nested plan is described in plan_width / plan_items and
                                 plan_waves = its waves | its format << 5 | its own word of this kind << 8)                       */
                            /* The launchers check the descriptor against the kernel they are about to launch and return
                               BSMM_ERR_ARG on a mismatch (a plan built with other options, or for another pass).        */
    int32_t flags;          /* BSMM_FLAG_* (0 = none)                                                                 */
    int32_t split;          /* updat with a plan: workgroups per work item (each takes a slice of the minibatch; > 1 or a
                               gate: fp32 partial sums in the workspace + a summing pass); 0 = library chooses          */
    int32_t blocks;         /* nonzero blocks                                                                        */
    int32_t bsize;          /* 8, 16 or 32; 64 on feature axis 1 (the reference's second axis-1 block size, blocksparse/matmul.py:84-89,
                               src/blocksparse_hgemm_nc_op_gpu.cu:38-281) WITH a plan built for bsize 64: the library runs the four 32x32
                               quadrants of every block on the bsize-32 kernels (see bsmm_xprop_plan_build)                        */
    int32_t segments;       /* xprop: number of lut headers (incl. empty output blocks)                              */
    int32_t locks;          /* xprop: number of output blocks written by more than one segment                       */
    int32_t C;              /* input features  of this call (bprop: caller passes the layer's K here)                */
    int32_t K;              /* output features of this call (bprop: caller passes the layer's C here)                */
    int32_t N;              /* minibatch = product of all non-feature dims                                           */
    int32_t shared;         /* reference's LDS lut-cache size in bytes; accepted, unused                             */
    int32_t pcount;         /* updat: number of (x,dy) pairs, 1..8                                                   */
    int32_t axis;           /* feature axis: 0 => (C,N) activations, 1 => (N,C)                                      */
    int32_t dtype;          /* BSMM_F32 / BSMM_F16 / BSMM_BF16: type of X, W, Y, DW                                  */
    float alpha;            /* updat: DW = alpha * sum_p X_p DY_p^T + beta * DW                                      */
    float beta;
    void* stream;           /* hipStream_t                                                                           */
    int32_t* trace;         /* optional HOST pointer: receives the BSMM_K_* code of the kernel this call dispatched to  */
    const void* prepared_w; /* optional: the result of bsmm_prepare_weights() for THIS op, W and plan (fp32 / bsize 32 with a plan:
                               the bf16 pieces of W).  W is constant across the calls of a forward / backward pass; without this
                               pointer every fprop / bprop re-splits it (one extra launch) into the workspace.  NULL = do that. */
} bsmm_args;

/* Y = fprop(X, W).  args->lut = fprop_lut.  Workspace: a transposed copy of W (not read by the staged bsize-32 kernel, which
 * transposes in LDS); bsize 8 with a plan: the expanded W; fp32 / bsize 32 with a plan: the bf16 pieces of X and W (6 bytes per
 * element) -- ask bsmm_workspace_bytes(). */
int bsmm_fprop(const void* X, const void* W, void* Y, const bsmm_args* args);

/* DX = bprop(DY, W).  args->lut = bprop_lut, args->C/K swapped by the caller.  Workspace only for bsize 8 with a plan
 * (the expanded W) and fp32 / bsize 32 with a plan (bf16 pieces): ask bsmm_workspace_bytes(BSMM_OP_BPROP, args). */
int bsmm_bprop(const void* DY, const void* W, void* DX, const bsmm_args* args);

/* DW = alpha * sum_{p<pcount} updat(X[p], DY[p]) + beta * DW.  args->lut = updat_lut.
 * X and DY are HOST arrays of pcount device pointers.
 * fp32 with a streaming plan ('BSU2', bsize 32): on feature axis 1 with one pair the call splits X and DY into bf16 pieces (workspace) and
 * runs the six significant piece products as six pairs of one launch of the bf16 streaming kernel (fp32 accuracy; ask
 * bsmm_workspace_bytes); bsize 8 with its 'BSS8' plan likewise on either feature axis (super-block sums, fp32 gather) and bsize 16 with its
 * 'BSUP' plan on feature axis 1 (windowed kernel, fp32 sums + finalize); every other fp32 call ignores the plan. */
int bsmm_updat(const void* const* X, const void* const* DY, void* DW, const bsmm_args* args);

/* Per-weights preparation that the xprop kernels would otherwise repeat on every call (fp32, bsize 32, with a plan: the exact
 * three-piece bf16 split of W, transposed per block for fprop).  bsmm_prepared_bytes() > 0 means: allocate that many bytes, call
 * bsmm_prepare_weights() once whenever W changes, and pass the buffer in args->prepared_w of the fprop (op BSMM_OP_FPROP) or bprop
 * (BSMM_OP_BPROP) calls; their workspace then holds only the pieces of the activations.  0 = nothing to prepare for these args. */
size_t bsmm_prepared_bytes(int op, const bsmm_args* args);
int bsmm_prepare_weights(int op, const void* W, void* prepared, const bsmm_args* args);

/* Second half of an updat issued with BSMM_FLAG_DW_SUMS: DW[w] = alpha * [gate[w] *] sums[w] + beta * DW[w], one rounding.
 * sums: fp32 [blocks][bsize][bsize] (the workspace of that call, possibly all-reduced in between); gate may be NULL. */
int bsmm_updat_finalize(const float* sums, void* DW, const float* gate, int32_t blocks, int32_t bsize, int32_t dtype, float alpha,
                        float beta, void* stream);

/* W[w] = scale * I if (c % KB) == (k % CB) else 0, (c,k) = updat_lut[w];  dtype as BSMM_* */
int bsmm_identity_init(void* W, const int32_t* updat_lut, int32_t CB, int32_t KB, int32_t blocks,
                       int32_t bsize, float scale, int32_t dtype, void* stream);

/* Gate gradient (BlocksparseMatmulDG, src/blocksparse_matmul_op.cc:492-540; blocksparse_gate_grad,
 * src/blocksparse_hgemm_cn_64_op_gpu.cu:1339-1412):  dw_out[w] = dw[w] * gate[w],  dg[w] = sum(dw[w] * W[w]).
 * dw_out may alias dw.  dg is fp32 [blocks]. */
int bsmm_gate_grad(void* dw_out, float* dg, const void* dw, const void* W, const float* gate, int32_t blocks,
                   int32_t bsize, int32_t dtype, void* stream);

/* Gated weight images (round 6): how a gated fprop / bprop call reaches the fast UNGATED kernels.  The reference applies the gate to the weight
 * fragments inside its tensor-core kernels (src/blocksparse_hgemm_cn_64_op_gpu.cu:54-66, 96-124: mul.rn.f16x2, one rounding); here
 *   out[0][w] = round(gate[w] * W[w]),   pieces == 2:  out[1][w] = round(gate[w] * W[w] - out[0][w])      (gate 0: zeros)
 * out: [pieces][blocks][bsize][bsize] of the storage type (16-bit types only).  A 0 / 1 gate (pruning mask) needs ONE piece and the
 * ungated call over `out` is exact; any other gate takes both pieces and the ungated call over the DOUBLED lookup table (every entry (c, w)
 * followed by (c, w + blocks); blocks' = 2 * blocks): the sum is g * w to ~2^-17.  blocksparse_amd/matmul.py composes the two. */
int bsmm_gate_weights(const void* W, const float* gate, void* out, int32_t blocks, int32_t bsize, int32_t dtype, int32_t pieces,
                      void* stream);

/* Block-sparse L2 normalisation of W over each output feature (L2NormalizeCK / L2NormalizeGainCK and their gradients,
 * src/blocksparse_l2_norm_op_gpu.cu:396-426,911-934; Python: blocksparse/matmul.py:421-453):
 *   y = gain * x / sqrt(max(sum_sqr, eps)),  sum_sqr[k] = sum of x^2 over the rows of every block in k's column block.
 * l2_lut: device table, header (offset, size, K, 0) per column block then weight ids (blocksparse/matmul.py:254-268);
 * cols = number of headers (= K / bsize).  gain: fp32 [K] or NULL.  sum_sqr: fp32 [K], written by the forward call and
 * read by the gradient.  x_dtype: type of x / dx, y_dtype: type of y / dy.  dgain may be NULL when gain is NULL. */
int bsmm_l2_normalize(void* y, float* sum_sqr, const void* x, const float* gain, const int32_t* l2_lut, int32_t cols,
                      int32_t bsize, int32_t x_dtype, int32_t y_dtype, float epsilon, void* stream);
int bsmm_l2_normalize_grad(void* dx, float* dgain, const void* dy, const void* x, const float* gain, const float* sum_sqr,
                           const int32_t* l2_lut, int32_t cols, int32_t bsize, int32_t x_dtype, int32_t y_dtype,
                           float epsilon, void* stream);

/* SparseProj row gather / scatter on (rows, N) tensors, N contiguous (GatherScatterOp / ScatterAddMulOp / ScatterMulGradOp,
 * launcher SparseOp / SparseMulGrad, src/layer_norm_cn_op_gpu.cu:718-830; Python: blocksparse/matmul.py:835-921).
 *   op 0 gather      z[k] = x[lut[k]]                      k < K = rows of z
 *   op 1 scatter     z[k] = lut[k] >= 0 ? x[lut[k]] : 0    k < K = rows of z
 *   op 2 scatter_add z = x, z[lut[k]] += y[k]              k < K = rows of y; rows_z = rows of x and z (z may alias x)
 *   op 3 scatter_mul z[k] = lut[k] >= 0 ? x[k] * y[lut[k]] : x[k]     k < K = rows of x and z
 * bsmm_sparse_mul_grad: dx = dz with dx[lut[k]] = dz[lut[k]] * y[k], dy[k] = dz[lut[k]] * x[lut[k]], k < K = rows of y
 * (dx may alias dz; rows_x = rows of x, dz, dx). */
int bsmm_sparse_op(void* z, const void* x, const void* y, const int32_t* lut, int32_t op, int32_t K, int32_t rows_z, int32_t N,
                   int32_t dtype, void* stream);
int bsmm_sparse_mul_grad(void* dx, void* dy, const void* dz, const void* x, const void* y, const int32_t* lut, int32_t K,
                         int32_t rows_x, int32_t N, int32_t dtype, void* stream);

/* Host-only: derive the grouped-kernel schedule ("plan") from a reference-format xprop lut that lives in HOST
 * memory (the luts are constants of the layout: the reference builds them in NumPy, blocksparse/matmul.py:137-138).
 * n_out_blocks = K / bsize of the pass the lut belongs to; axis = feature axis the plan will be used with.  bsmm_xprop_plan_words returns the number of int32 words
 * (0 if this (bsize, dtype, axis) has no grouped kernel, <0 on malformed input); bsmm_xprop_plan_build fills host_plan_out
 * (that many words).  The caller uploads the words to the device and attaches both copies with bsmm_plan_attach().
 * options: BSMM_PLAN_* (0 = default).
 * bsize 8 (16-bit types, n_out_blocks % 4 == 0): the result is a composite 'BSS8' plan -- the 8x8 blocks grouped into 32x32
 * super-blocks, the bsize-32 plan of that super layout nested at word [5]; fprop / bprop then need workspace
 * (bsmm_workspace_bytes).
 * bsize 64 (feature axis 1, any dtype; replaces hgemm_blocksparse_64x64x64_*, src/blocksparse_hgemm_nc_op_gpu.cu:38-281,949-1083): a composite
 * 'BS64' plan -- the lookup table of the quadrant view (a 64x64 block = four 32x32 blocks of kron(layout, ones(2,2))) and the bsize-32
 * plan built from it.  bsmm_fprop / bsmm_bprop with bsize = 64 REQUIRE this plan; they run the bsize-32 kernels on a quadrant-ordered copy
 * of W, made per call into the workspace or, better, once per weights version by bsmm_prepare_weights (args->prepared_w; one image per op). */
long bsmm_xprop_plan_words(const int32_t* host_lut, int32_t segments, int32_t blocks, int32_t n_out_blocks,
                           int32_t bsize, int32_t dtype, int32_t axis, int32_t options);
int bsmm_xprop_plan_build(const int32_t* host_lut, int32_t segments, int32_t blocks, int32_t n_out_blocks,
                          int32_t bsize, int32_t dtype, int32_t axis, int32_t options, int32_t* host_plan_out);

/* Host-only: work items of the windowed weight-gradient kernel for an updat lut in HOST memory (CB/KB = block rows /
 * columns of the layout).  Same conventions as the xprop plan (options: BSMM_PLAN_WINDOW_* or 0).
 * With a plan, bsmm_updat needs workspace (the fp32 sums and, for the streaming kernel, one 256 KiB region of partial sums per
 * round and workgroup: 141 MB at 4096^2 / 20 % / N = 8192): ask bsmm_workspace_bytes(BSMM_OP_UPDAT, args).
 * bsize 8 (16-bit types, CB % 4 == 0 and KB % 4 == 0): composite 'BSS8' plan as above.
 * bsize 64 (feature axis 1, 16-bit types): composite 'BS64' plan around the streaming bsize-32 plan of the quadrant view; bsmm_updat then
 * leaves the fp32 sums of the quadrants in the workspace and one pass writes DW (64x64 blocks) with alpha / beta / gate and ONE rounding. */
long bsmm_updat_plan_words(const int32_t* host_updat_lut, int32_t blocks, int32_t CB, int32_t KB, int32_t bsize,
                           int32_t dtype, int32_t axis, int32_t options);
int bsmm_updat_plan_build(const int32_t* host_updat_lut, int32_t blocks, int32_t CB, int32_t KB, int32_t bsize,
                          int32_t dtype, int32_t axis, int32_t options, int32_t* host_plan_out);

/* Host-only: point args at a plan.  host_plan = the words produced by one of the builders above (host memory, `words` of
 * them), device_plan = the caller's device copy of the same words.  Fills args->plan and the descriptor fields
 * (plan_magic, plan_width, plan_waves, plan_items, plan_inner); BSMM_ERR_ARG if the words are not a plan of this library
 * version.  device_plan == NULL detaches (generic kernels). */
int bsmm_plan_attach(bsmm_args* args, const int32_t* host_plan, long words, const int32_t* device_plan);

/* Bytes of device scratch the given op (BSMM_OP_*) needs for these args. */
size_t bsmm_workspace_bytes(int op, const bsmm_args* args);

const char* bsmm_error_string(int code);
int bsmm_version(void);

#ifdef __cplusplus
}
#endif
#endif /* BSMM_H_ */
