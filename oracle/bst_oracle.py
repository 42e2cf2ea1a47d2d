"""CPU restatement of the reference's block-sparse attention path (SURVEY.md section 8, row a13 / BASELINE configs[4]).

TEST INFRASTRUCTURE ONLY.  Nothing under blocksparse_amd/ may import this module; only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg do, and only as the checker.

Pinned: tests/test_bst_oracle_golden.py compares every function here with tests/golden/bst.npz, which
tests/golden/make_golden_bst.py produced by running the reference's own NumPy builder / ``*_test`` methods
(/root/reference/blocksparse/transformer.py) in the authoring container.

What is restated (reference file:line):
  * lookup tables   BlocksparseTransformer.__init__ / xn_lut          blocksparse/transformer.py:61-165
  * mask packing    init_softmax_mask                                  blocksparse/transformer.py:129-159
  * nt / nn / tn    nt_test, nn_test, tn_test                          blocksparse/transformer.py:186-248
  * softmax         masked_softmax_test, masked_softmax_grad_test      blocksparse/transformer.py:251-318
  * partial autoregressive mask   bst_partial_autoregressive_mask      src/bst_softmax_op_gpu.cu:461-503
Kernel-level facts that the Python does not show: scores are stored as bf16 (fp32 inputs) or fp16 (fp16 inputs)
(src/bst_op.cc:76-78,143-144), the softmax reads bf16 and writes bf16/fp16 (src/bst_op.cc:342-348).
"""
import numpy as np


# ----------------------------------------------------------------------------------------------------
# lookup tables
# ----------------------------------------------------------------------------------------------------
def xn_lut(ys, xs, blocks, ctx_blks):
    """Header [ctx_blks] of (offset, count) followed by (block id, x) entries grouped by y, in block order.
    blocksparse/transformer.py:141-165."""
    per_y = [[] for _ in range(ctx_blks)]
    for b in range(blocks):
        per_y[ys[b]].append((b, xs[b]))
    lut = np.empty((ctx_blks + blocks, 2), dtype=np.int32)
    off, mx = ctx_blks, 0
    for y, ent in enumerate(per_y):
        lut[y] = (off, len(ent))
        mx = max(mx, len(ent))
        for e in ent:
            lut[off] = e
            off += 1
    return lut, per_y, mx


def build_luts(layout):
    """layout [heads_l, Qb, Kb] (nonzero = block present).  Blocks are numbered row-major (sorted by (q, k)),
    'contiguous along the rows' (blocksparse/transformer.py:103-107).  Returns a dict with nt_lut [H,blocks,2],
    nn_lut [H,Qb+blocks,2], tn_lut [H,Kb+blocks,2], nn_max, tn_max, blocks and the python lists."""
    layout = np.asarray(layout)
    if layout.ndim == 2:
        layout = layout[None]
    H, Qb, Kb = layout.shape
    nt_luts, nn_luts, tn_luts, nt_lists, nn_lists, tn_lists = [], [], [], [], [], []
    nn_max = tn_max = 0
    blocks = None
    for h in range(H):
        qs, ks = np.nonzero(layout[h])                 # row-major == sorted by (q, k)
        if blocks is None:
            blocks = len(qs)
        elif blocks != len(qs):
            raise ValueError("number of layout blocks must be equal across heads")
        nt = np.stack([qs, ks], axis=1).astype(np.int32)
        nn, nn_list, m1 = xn_lut(qs, ks, blocks, Qb)
        tn, tn_list, m2 = xn_lut(ks, qs, blocks, Kb)
        nt_luts.append(nt); nn_luts.append(nn); tn_luts.append(tn)
        nt_lists.append([tuple(int(v) for v in e) for e in nt]); nn_lists.append(nn_list); tn_lists.append(tn_list)
        nn_max, tn_max = max(nn_max, m1), max(tn_max, m2)
    return dict(nt_lut=np.array(nt_luts, dtype=np.int32), nn_lut=np.array(nn_luts, dtype=np.int32),
                tn_lut=np.array(tn_luts, dtype=np.int32), nn_max=nn_max, tn_max=tn_max, blocks=blocks,
                nt_list=nt_lists, nn_list=nn_lists, tn_list=tn_lists, lut_heads=H, ctx_blks_q=Qb, ctx_blks_k=Kb)


def mask_dtype(bsize):
    return {64: np.uint64, 32: np.uint32, 16: np.uint16, 8: np.uint8}[bsize]


def pack_mask(mask):
    """bool [bs, bs] -> one unsigned integer per query row, bit k = mask[q, k] (blocksparse/transformer.py:146-149)."""
    bs = mask.shape[0]
    m = np.asarray(mask, dtype=bool)
    w = (np.uint64(1) << np.arange(bs, dtype=np.uint64))
    return (m.astype(np.uint64) * w[None, :]).sum(axis=1, dtype=np.uint64).astype(mask_dtype(bs))


def build_masks(luts, bsize, mask_callback):
    """softmax_mask_np [H, blocks, bs] and the kernel layout [H, bs, blocks] (blocksparse/transformer.py:139-159)."""
    H = luts["lut_heads"]
    out = np.empty((H, luts["blocks"], bsize), dtype=mask_dtype(bsize))
    for h in range(H):
        for b, (q, k) in enumerate(luts["nt_list"][h]):
            out[h, b] = pack_mask(mask_callback((bsize, bsize), h, q, k, b))
    return out, np.ascontiguousarray(out.transpose(0, 2, 1))


def partial_autoregressive_mask(mask_kernel_layout, nt_lut, bsize, autoregress_at_k):
    """mask [H, bs, blocks] -> same shape, src/bst_softmax_op_gpu.cu:461-503."""
    H, bs, blocks = mask_kernel_layout.shape
    out = mask_kernel_layout.copy()
    ones = (1 << bsize) - 1
    for h in range(H):
        for b in range(blocks):
            Q, K = int(nt_lut[h, b, 0]) * bsize, int(nt_lut[h, b, 1]) * bsize
            shift_a = bsize - min(max(autoregress_at_k - K, 0), bsize)
            for qi in range(bs):
                shift_b = min(max(bsize - 1 + K - (Q + qi), 0), bsize)
                out[h, qi, b] = int(out[h, qi, b]) & (ones >> min(shift_a, shift_b))
    return out


# ----------------------------------------------------------------------------------------------------
# matmuls (float64 accumulation of the given inputs; callers round inputs / outputs as the kernels do)
# ----------------------------------------------------------------------------------------------------
def _split(X, ctx_blks, bsize, heads):
    n, ctx, state = X.shape
    assert ctx == ctx_blks * bsize and state % heads == 0
    return X.reshape(n, ctx_blks, bsize, heads, state // heads)


def nt(luts, A, B, bsize, heads):
    """C[n,h,b] = A[n, q-block, :, h, :] . B[n, k-block, :, h, :]^T       (blocksparse/transformer.py:186-203)"""
    A5 = _split(np.asarray(A, dtype=np.float64), luts["ctx_blks_q"], bsize, heads)
    B5 = _split(np.asarray(B, dtype=np.float64), luts["ctx_blks_k"], bsize, heads)
    N = A5.shape[0]
    C = np.empty((N, heads, luts["blocks"], bsize, bsize), dtype=np.float64)
    for h in range(heads):
        lut = luts["nt_lut"][h if luts["lut_heads"] > 1 else 0]
        for n in range(N):
            C[n, h] = np.einsum("bik,bjk->bij", A5[n, lut[:, 0], :, h, :], B5[n, lut[:, 1], :, h, :])
    return C


def _xn(luts, W, B, bsize, heads, trans):
    ctx_b = luts["ctx_blks_q"] if trans else luts["ctx_blks_k"]
    ctx_c = luts["ctx_blks_k"] if trans else luts["ctx_blks_q"]
    B5 = _split(np.asarray(B, dtype=np.float64), ctx_b, bsize, heads)
    W = np.asarray(W, dtype=np.float64)
    N = B5.shape[0]
    C = np.zeros((N, ctx_c, bsize, heads, B5.shape[4]), dtype=np.float64)
    for h in range(heads):
        lut = luts["nt_lut"][h if luts["lut_heads"] > 1 else 0]
        src, dst = (lut[:, 0], lut[:, 1]) if trans else (lut[:, 1], lut[:, 0])
        for n in range(N):
            Wb = W[n, h].transpose(0, 2, 1) if trans else W[n, h]
            prod = np.einsum("bij,bjk->bik", Wb, B5[n, src, :, h, :])
            acc = np.zeros((ctx_c, bsize, B5.shape[4]))
            np.add.at(acc, dst, prod)
            C[n, :, :, h, :] = acc
    return C.reshape(N, ctx_c * bsize, -1)


def nn(luts, W, B, bsize, heads):
    """C[n, q-block, :, h, :] = sum_k W[n,h,b(q,k)] . B[n, k-block, :, h, :]   (blocksparse/transformer.py:205-225)"""
    return _xn(luts, W, B, bsize, heads, False)


def tn(luts, W, B, bsize, heads):
    """C[n, k-block, :, h, :] = sum_q W[n,h,b(q,k)]^T . B[n, q-block, :, h, :] (blocksparse/transformer.py:227-248)"""
    return _xn(luts, W, B, bsize, heads, True)


# ----------------------------------------------------------------------------------------------------
# softmax over the blocks of one query row-block
# ----------------------------------------------------------------------------------------------------
def unpack_mask(bits, bsize):
    """[..., bs] unsigned -> bool [..., bs(q), bs(k)]"""
    b = np.asarray(bits).astype(np.uint64)
    return ((b[..., None] >> np.arange(bsize, dtype=np.uint64)) & np.uint64(1)).astype(bool)


def masked_softmax(luts, x, bsize, scale=1.0, mask_np=None):
    """x [N, heads, blocks, bs, bs] -> same; per (n, h, query row): softmax over the unmasked keys of all blocks in the
    row-block (blocksparse/transformer.py:251-300).  mask_np [H_m, blocks, bs] (bit k of entry q) or None.
    Masked entries are filled with -FLT_MAX before the max, exactly as the reference oracle does (a fully masked row
    therefore comes out uniform)."""
    x = np.asarray(x, dtype=np.float64)
    N, heads = x.shape[:2]
    y = np.empty_like(x)
    neg = -float(np.finfo(np.float32).max)
    for h in range(heads):
        hl = h if luts["lut_heads"] > 1 else 0
        for ent in luts["nn_list"][hl]:
            if not ent:
                continue
            bs_ids = [b for b, _ in ent]
            xb = x[:, h, bs_ids] * scale                               # [N, nb, q, k]
            if mask_np is not None:
                hm = hl if mask_np.shape[0] > 1 else 0
                keep = unpack_mask(mask_np[hm, bs_ids], bsize)         # [nb, q, k]
                xb = np.where(keep[None], xb, neg)
            mx = xb.max(axis=(1, 3), keepdims=True)
            ex = np.exp(xb - mx)
            y[:, h, bs_ids] = ex / ex.sum(axis=(1, 3), keepdims=True)
    return y


def masked_softmax_grad(luts, dy, y, scale=1.0):
    """dx = (dy - sum_row(dy * y)) * y * scale          (blocksparse/transformer.py:303-318)"""
    dy = np.asarray(dy, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    dx = np.empty_like(dy)
    for h in range(dy.shape[1]):
        hl = h if luts["lut_heads"] > 1 else 0
        for ent in luts["nn_list"][hl]:
            if not ent:
                continue
            bs_ids = [b for b, _ in ent]
            d, v = dy[:, h, bs_ids], y[:, h, bs_ids]
            dx[:, h, bs_ids] = (d - (d * v).sum(axis=(1, 3), keepdims=True)) * v * scale
    return dx


# ----------------------------------------------------------------------------------------------------
# the layout and mask of BASELINE configs[4] (cfg 5)
# ----------------------------------------------------------------------------------------------------
def local_strided_layout(ctx_blks, local=4, stride=8):
    q, k = np.indices((ctx_blks, ctx_blks))
    return ((k <= q) & ((q - k < local) | ((q - k) % stride == 0))).astype(np.int32)


def causal_mask_callback(blk_shape, head, q, k, b):
    m = np.ones(blk_shape, dtype=bool)
    if q == k:
        m = np.tril(m)
    return m
