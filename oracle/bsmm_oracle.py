"""TEST INFRASTRUCTURE ONLY: NumPy oracle for the block-sparse matmul hot path.

This file is an independent, loop-level CPU restatement of what the reference computes on the path
``BlocksparseMatMul`` fprop / bprop / updat.  It is *not* product code: the product is the HIP library
behind ``include/bsmm.h``; the product package never imports this module.

Parity pin: every function here is checked in ``tests/test_oracle_golden.py`` against fixtures under
``tests/golden/`` that were produced by importing the reference's own Python module
(``/root/reference/blocksparse/matmul.py``) behind a TensorFlow stub -- see
``tests/golden/make_golden.py`` (the generating script) -- so parity is *pinned*, not assumed.

Reference lines each function follows (paths relative to /root/reference):
  z_order_2d         blocksparse/utils.py:95-103
  segment_policy     blocksparse/matmul.py:94-105
  build_layout_luts  blocksparse/matmul.py:107-159   (ctor)  + :172-270 (xprop_lut)
  fprop / bprop / updat  blocksparse/matmul.py:353-375 / :377-399 / :401-419
  updat alpha/beta and multi-pair sum   src/blocksparse_matmul_op_gpu.cu:2684-2814,2865
  identity_init      src/blocksparse_matmul_op_gpu.cu:2988-3015 (rule at :3010)
  round_bf16/round_fp16  src/ew_op_gpu.h:238-251, src/gpu_hmma.h:22-63 (we use RNE)

The reference relies on ``scipy.sparse.find`` returning entries column-major (SURVEY TRAP 3); this
restatement sorts explicitly instead so that it does not depend on the SciPy version.
"""
import numpy as np

SEG_MAX = (1 << 63) - 1


def ceil_div(x, y):
    return -(-x // y)


def z_order_2d(x, y):
    """Morton interleave: bit i of x -> bit 2i, bit i of y -> bit 2i+1 (utils.py:95-103)."""
    x = int(x)
    y = int(y)
    out = 0
    bit = 0
    while (x >> bit) or (y >> bit):
        out |= ((x >> bit) & 1) << (2 * bit)
        out |= ((y >> bit) & 1) << (2 * bit + 1)
        bit += 1
    return out


def segment_policy(layout):
    """(max_seg, min_seg) from the per-output-column block counts (matmul.py:94-105).

    The reference uses the k-column sums for *both* the fprop and the bprop table ("assume symmetrical
    transpose"); so do we."""
    counts = np.asarray(layout).astype(np.int64).sum(axis=0)
    hi = int(counts.max())
    lo = int(counts[counts > 0].min())
    if hi / lo > 2.0:
        max_seg = max(ceil_div(hi, 4), lo * 2)
    else:
        max_seg = SEG_MAX
    min_seg = max(ceil_div(max_seg, 4), 4)
    return max_seg, min_seg


def _xprop_table(n_out, ins, outs, wids, order, max_seg, min_seg):
    """One xprop lookup table (matmul.py:172-270).

    ``order`` walks the blocks grouped by output block index ``outs`` (ascending), ``ins`` is the input
    block index of each block, ``wids`` its weight-block id.  Returns
    (cols, lut int32[4*S+2*B], l2_lut, shared_bytes, l2_shared_bytes, segments, locks)."""
    per_out = {}
    seen_order = []
    for i in order:
        o = int(outs[i])
        if o not in per_out:
            per_out[o] = []
            seen_order.append(o)
        per_out[o].append((int(ins[i]), int(wids[i])))

    segs = []   # (out block, [(in block, w), ...])
    cols = []   # unsegmented columns
    lock_of = {}
    locks = 0
    for o in seen_order:
        entries = per_out[o]
        cols.append((o, list(entries)))
        left = len(entries)
        cur = []
        nseg = 0
        for ent in entries:
            cur.append(ent)
            left -= 1
            # close a segment only if what remains is still worth its own segment (:218)
            if len(cur) >= max_seg and left >= min_seg:
                segs.append((o, cur))
                cur = []
                nseg += 1
        if cur:
            segs.append((o, cur))
            nseg += 1
        if nseg > 1:            # several writers of one output block -> needs a lock id (:206-208,228-230)
            locks += 1
            lock_of[o] = locks
    for o in range(n_out):       # empty output blocks go last (:233-236)
        if o not in per_out:
            segs.append((o, []))
            cols.append((o, []))

    nblk = len(wids)
    S = len(segs)
    lut = np.empty(4 * S + 2 * nblk, dtype=np.int32)
    pos = 4 * S
    longest = 0
    for s, (o, entries) in enumerate(segs):
        lut[4 * s:4 * s + 4] = (pos // 2, len(entries), o, lock_of.get(o, 0))
        longest = max(longest, len(entries))
        for (i_blk, w) in entries:
            lut[pos] = i_blk
            lut[pos + 1] = w
            pos += 2

    # weight-norm table: whole columns, one int per entry, padded to an even int32 count (:254-268)
    Cn = len(cols)
    size = 4 * Cn + nblk
    size += size & 1
    l2 = np.zeros(size, dtype=np.int32)
    pos = 4 * Cn
    l2_longest = 0
    for s, (o, entries) in enumerate(cols):
        l2[4 * s:4 * s + 4] = (pos, len(entries), o, 0)
        l2_longest = max(l2_longest, len(entries))
        for (_, w) in entries:
            l2[pos] = w
            pos += 1
    return cols, lut, l2, longest * 8, l2_longest * 4, S, locks


def build_layout_luts(layout, block_size=32, z_order=True):
    """All host tables of a ``BlocksparseMatMul`` (matmul.py:82-162) as a dict."""
    layout = (np.asarray(layout) != 0)
    CB, KB = layout.shape
    max_seg, min_seg = segment_policy(layout)

    cs, ks = np.nonzero(layout)
    col_major = np.lexsort((cs, ks))          # k major, c minor -- what the reference assumes of find()
    cs = cs[col_major]
    ks = ks[col_major]
    B = len(cs)
    fwd = list(range(B))
    bwd = sorted(fwd, key=lambda i: cs[i])    # stable: c major, k minor (the "transpose view", :117)

    wids = np.empty(B, dtype=np.int64)
    if z_order:
        keyed = sorted((z_order_2d(cs[i], ks[i]), i) for i in range(B))
        updat_list = []
        for blk, (_, i) in enumerate(keyed):
            wids[i] = blk
            updat_list.append((int(cs[i]), int(ks[i])))
    else:
        updat_list = [(int(c), int(k)) for c, k in zip(cs, ks)]
        wids[:] = np.arange(B)

    f = _xprop_table(KB, cs, ks, wids, fwd, max_seg, min_seg)
    b = _xprop_table(CB, ks, cs, wids, bwd, max_seg, min_seg)
    return dict(
        CB=CB, KB=KB, C=CB * block_size, K=KB * block_size, bsize=block_size, blocks=B,
        updat_list=updat_list, updat_lut=np.array(updat_list, dtype=np.int32).reshape(B, 2),
        fprop_list=f[0], fprop_lut=f[1], l2_lut=f[2], fprop_shared=f[3], l2_shared=f[4],
        fprop_segments=f[5], fprop_locks=f[6],
        bprop_list=b[0], bprop_lut=b[1], bprop_shared=b[3], bprop_segments=b[5], bprop_locks=b[6],
    )


# ----------------------------------------------------------------------------------------------
# math (float64 accumulation); W is (blocks, bs, bs) with W[w] = Wdense[c*bs:(c+1)*bs, k*bs:(k+1)*bs]
# ----------------------------------------------------------------------------------------------

def _gated(W, gate):
    """Per-block gate: block w contributes gate[w] * (its product), gate 0 = skipped (matmul.py:367-373,391-397).
    The gate multiplies the exact block product, so scaling the float64 weights is the same thing."""
    W = np.asarray(W, dtype=np.float64)
    if gate is None:
        return W
    return W * np.asarray(gate, dtype=np.float64)[:, None, None]


def fprop(t, I, W, axis, gate=None):
    """axis 0: Y(K,N) = Wd^T X ; axis 1: Y(N,K) = X Wd   (matmul.py:353-375)."""
    bs, CB, KB = t["bsize"], t["CB"], t["KB"]
    I = np.asarray(I, dtype=np.float64)
    W = _gated(W, gate)
    if axis:
        n = I.shape[0]
        X = I.reshape(n, CB, bs)
        Y = np.zeros((n, KB, bs))
        for k, col in t["fprop_list"]:
            for c, w in col:
                Y[:, k, :] += X[:, c, :] @ W[w]
        return Y.reshape(n, KB * bs)
    n = I.size // (CB * bs)
    X = I.reshape(CB, bs, n)
    Y = np.zeros((KB, bs, n))
    for k, col in t["fprop_list"]:
        for c, w in col:
            Y[k] += W[w].T @ X[c]
    return Y.reshape(KB * bs, n)


def bprop(t, E, W, axis, gate=None):
    """axis 0: DX(C,N) = Wd DY ; axis 1: DX(N,C) = DY Wd^T   (matmul.py:377-399)."""
    bs, CB, KB = t["bsize"], t["CB"], t["KB"]
    E = np.asarray(E, dtype=np.float64)
    W = _gated(W, gate)
    if axis:
        n = E.shape[0]
        D = E.reshape(n, KB, bs)
        B = np.zeros((n, CB, bs))
        for c, row in t["bprop_list"]:
            for k, w in row:
                B[:, c, :] += D[:, k, :] @ W[w].T
        return B.reshape(n, CB * bs)
    n = E.size // (KB * bs)
    D = E.reshape(KB, bs, n)
    B = np.zeros((CB, bs, n))
    for c, row in t["bprop_list"]:
        for k, w in row:
            B[c] += W[w] @ D[k]
    return B.reshape(CB * bs, n)


def updat(t, Is, Es, axis, alpha=1.0, beta=0.0, dw_in=None, gate=None):
    """DW[w] = alpha * sum_p X_p[c] DY_p[k]^T + beta * DW_in[w]   (matmul.py:401-419; kernel
    semantics for alpha/beta/pairs: src/blocksparse_matmul_op_gpu.cu:2684-2814,2865).  gate (with dw_gated=True in the
    reference, matmul.py:412-418): the sum of block w is scaled by gate[w]."""
    bs, CB, KB = t["bsize"], t["CB"], t["KB"]
    if isinstance(Is, np.ndarray):
        Is, Es = [Is], [Es]
    U = np.zeros((t["blocks"], bs, bs))
    for I, E in zip(Is, Es):
        I = np.asarray(I, dtype=np.float64)
        E = np.asarray(E, dtype=np.float64)
        if axis:
            X = I.reshape(-1, CB, bs)
            D = E.reshape(-1, KB, bs)
            for w, (c, k) in enumerate(t["updat_list"]):
                U[w] += X[:, c, :].T @ D[:, k, :]
        else:
            X = I.reshape(CB, bs, -1)
            D = E.reshape(KB, bs, -1)
            for w, (c, k) in enumerate(t["updat_list"]):
                U[w] += X[c] @ D[k].T
    if gate is not None:
        U *= np.asarray(gate, dtype=np.float64)[:, None, None]
    U *= alpha
    if beta != 0.0:
        U += beta * np.asarray(dw_in, dtype=np.float64)
    return U


def gate_grad(dw, W, gate):
    """(dw * gate, dg) with dg[w] = sum(dw[w] * W[w])   (blocksparse_gate_grad,
    src/blocksparse_hgemm_cn_64_op_gpu.cu:1339-1392)."""
    dw = np.asarray(dw, dtype=np.float64)
    W = np.asarray(W, dtype=np.float64)
    g = np.asarray(gate, dtype=np.float64)
    return dw * g[:, None, None], (dw * W).sum(axis=(1, 2))


def l2_normalize(t, W, gain=None, epsilon=1e-12):
    """(Y, sum_sqr): Y = gain * W / sqrt(max(sum_sqr, eps)), sum_sqr[k] over all rows of all blocks in k's column block
    (l2_normalize_test, blocksparse/matmul.py:421-429; gain and the returned sums: l2_normalize_CK_32,
    src/blocksparse_l2_norm_op_gpu.cu:151-235)."""
    bs = t["bsize"]
    W = np.asarray(W, dtype=np.float64)
    Y = np.zeros_like(W)
    S = np.zeros(t["K"])
    for k, col in t["fprop_list"]:
        ws = [w for _, w in col]
        if not ws:
            continue
        W2 = W[ws].reshape(-1, bs)
        ss = np.square(W2).sum(axis=0)
        S[k * bs:(k + 1) * bs] = ss
        g = 1.0 if gain is None else np.asarray(gain, dtype=np.float64)[k * bs:(k + 1) * bs]
        Y[ws] = W[ws] * (g / np.sqrt(np.maximum(ss, epsilon)))
    return Y, S


def l2_normalize_grad(t, W, U, gain=None, epsilon=1e-12):
    """(dW, dgain) (l2_normalize_grad_test, blocksparse/matmul.py:431-445; gain terms: the formula block at
    src/blocksparse_l2_norm_op_gpu.cu:705-708)."""
    bs = t["bsize"]
    W = np.asarray(W, dtype=np.float64)
    U = np.asarray(U, dtype=np.float64)
    D = np.zeros_like(W)
    DG = np.zeros(t["K"])
    for k, col in t["fprop_list"]:
        ws = [w for _, w in col]
        if not ws:
            continue
        W2, U2 = W[ws].reshape(-1, bs), U[ws].reshape(-1, bs)
        g = np.ones(bs) if gain is None else np.asarray(gain, dtype=np.float64)[k * bs:(k + 1) * bs]
        ss = np.square(W2).sum(axis=0)
        mx = np.maximum(ss, epsilon)
        red = (-(U2 * g) * W2 / mx).sum(axis=0) * (ss >= epsilon)
        D[ws] = (((U2 * g) + W2 * red) / np.sqrt(mx)).reshape(-1, bs, bs)
        DG[k * bs:(k + 1) * bs] = (U2 * W2 / np.sqrt(mx)).sum(axis=0)
    return D, DG


def to_dense(t, W):
    bs = t["bsize"]
    Wd = np.zeros((t["C"], t["K"]), dtype=np.float64)
    for w, (c, k) in enumerate(t["updat_list"]):
        Wd[c * bs:(c + 1) * bs, k * bs:(k + 1) * bs] = W[w]
    return Wd


def identity_init(t, scale=1.0):
    """W[w] = scale*I iff (c % KB) == (k % CB) else 0 (src/blocksparse_matmul_op_gpu.cu:3010)."""
    bs = t["bsize"]
    W = np.zeros((t["blocks"], bs, bs), dtype=np.float32)
    for w, (c, k) in enumerate(t["updat_list"]):
        if (c % t["KB"]) == (k % t["CB"]):
            W[w] = np.eye(bs, dtype=np.float32) * scale
    return W


# ----------------------------------------------------------------------------------------------
# storage rounding helpers (round-to-nearest-even; see SURVEY A.3 for the reference's bf16 mode)
# ----------------------------------------------------------------------------------------------

def round_bf16(x):
    """float32 -> bfloat16 (RNE) -> float32, NaN-free inputs assumed."""
    u = np.asarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def round_fp16(x):
    return np.asarray(x, dtype=np.float32).astype(np.float16).astype(np.float32)


def round_to(x, dtype):
    if dtype in ("bf16", "bfloat16"):
        return round_bf16(x)
    if dtype in ("f16", "fp16", "float16"):
        return round_fp16(x)
    return np.asarray(x, dtype=np.float32)


# ----------------------------------------------------------------------------------------------
# fast (batched-BLAS) variants of the same math, used for big parity cases and the CPU baseline
# ----------------------------------------------------------------------------------------------

def fprop_fast(t, I, W, axis, dtype=np.float32):
    """Same result as ``fprop`` but one batched matmul per output block column (for 4096^2 cases)."""
    bs, CB, KB = t["bsize"], t["CB"], t["KB"]
    I = np.asarray(I, dtype=dtype)
    W = np.asarray(W, dtype=dtype)
    if axis:
        n = I.shape[0]
        X = I.reshape(n, CB, bs)
        Y = np.zeros((n, KB, bs), dtype=dtype)
        for k, col in t["fprop_list"]:
            if not col:
                continue
            c = [e[0] for e in col]
            w = [e[1] for e in col]
            Y[:, k, :] = X[:, c, :].reshape(n, -1) @ W[w].reshape(-1, bs)
        return Y.reshape(n, KB * bs)
    n = I.size // (CB * bs)
    X = I.reshape(CB, bs, n)
    Y = np.zeros((KB, bs, n), dtype=dtype)
    for k, col in t["fprop_list"]:
        if not col:
            continue
        c = [e[0] for e in col]
        w = [e[1] for e in col]
        Y[k] = W[w].reshape(-1, bs).T @ X[c].reshape(-1, n)
    return Y.reshape(KB * bs, n)


def bprop_fast(t, E, W, axis, dtype=np.float32):
    bs, CB, KB = t["bsize"], t["CB"], t["KB"]
    E = np.asarray(E, dtype=dtype)
    W = np.asarray(W, dtype=dtype)
    if axis:
        n = E.shape[0]
        D = E.reshape(n, KB, bs)
        B = np.zeros((n, CB, bs), dtype=dtype)
        for c, row in t["bprop_list"]:
            if not row:
                continue
            k = [e[0] for e in row]
            w = [e[1] for e in row]
            Wt = np.transpose(W[w], (0, 2, 1)).reshape(-1, bs)       # rows = (k-block, k-in-block)
            B[:, c, :] = D[:, k, :].reshape(n, -1) @ Wt
        return B.reshape(n, CB * bs)
    n = E.size // (KB * bs)
    D = E.reshape(KB, bs, n)
    B = np.zeros((CB, bs, n), dtype=dtype)
    for c, row in t["bprop_list"]:
        if not row:
            continue
        k = [e[0] for e in row]
        w = [e[1] for e in row]
        Wc = np.transpose(W[w], (1, 0, 2)).reshape(bs, -1)           # (c-in-block, (k-block,k-in-block))
        B[c] = Wc @ D[k].reshape(-1, n)
    return B.reshape(CB * bs, n)


def updat_fast(t, I, E, axis, dtype=np.float32):
    bs, CB, KB = t["bsize"], t["CB"], t["KB"]
    I = np.asarray(I, dtype=dtype)
    E = np.asarray(E, dtype=dtype)
    ul = t["updat_lut"]
    if axis:
        X = np.ascontiguousarray(I.reshape(-1, CB, bs).transpose(1, 2, 0))   # CB, bs, n
        D = np.ascontiguousarray(E.reshape(-1, KB, bs).transpose(1, 2, 0))
    else:
        X = I.reshape(CB, bs, -1)
        D = E.reshape(KB, bs, -1)
    return np.matmul(X[ul[:, 0]], np.transpose(D[ul[:, 1]], (0, 2, 1)))


# ----------------------------------------------------------------------------------------------
# sampled variants for BASELINE-size parity tests: the same sums as fprop / bprop / updat, in float64, restricted to a
# chosen set of output block columns / rows / weight blocks (the full float64 product of a 4096^2 layout at minibatch
# 8192 would take minutes; a sample of every workgroup class takes seconds)
# ----------------------------------------------------------------------------------------------

def fprop_cols(t, I, W, axis, ks):
    """{k: output block column k of fprop(I, W)} (axis 1: (n, bs) arrays, axis 0: (bs, n))  -- blocksparse/matmul.py:353-375."""
    bs = t["bsize"]
    cols = dict(t["fprop_list"])
    I = np.asarray(I)
    out = {}
    for k in ks:
        n = I.shape[0] if axis else I.shape[1]
        ref = np.zeros((n, bs) if axis else (bs, n), dtype=np.float64)
        for c, w in cols[k]:
            Wb = np.asarray(W[w], dtype=np.float64)
            if axis:
                ref += I[:, c * bs:(c + 1) * bs].astype(np.float64) @ Wb
            else:
                ref += Wb.T @ I[c * bs:(c + 1) * bs, :].astype(np.float64)
        out[k] = ref
    return out


def bprop_rows(t, E, W, axis, cs):
    """{c: input block c of bprop(E, W)}  -- blocksparse/matmul.py:377-399."""
    bs = t["bsize"]
    rows = dict(t["bprop_list"])
    E = np.asarray(E)
    out = {}
    for c in cs:
        n = E.shape[0] if axis else E.shape[1]
        ref = np.zeros((n, bs) if axis else (bs, n), dtype=np.float64)
        for k, w in rows[c]:
            Wb = np.asarray(W[w], dtype=np.float64)
            if axis:
                ref += E[:, k * bs:(k + 1) * bs].astype(np.float64) @ Wb.T
            else:
                ref += Wb @ E[k * bs:(k + 1) * bs, :].astype(np.float64)
        out[c] = ref
    return out


def updat_blocks(t, I, E, axis, ws):
    """{w: weight-gradient block w of updat(I, E)}  -- blocksparse/matmul.py:401-419."""
    bs = t["bsize"]
    ul = t["updat_lut"]
    I = np.asarray(I); E = np.asarray(E)
    out = {}
    for w in ws:
        c, k = int(ul[w][0]), int(ul[w][1])
        if axis:
            out[w] = I[:, c * bs:(c + 1) * bs].astype(np.float64).T @ E[:, k * bs:(k + 1) * bs].astype(np.float64)
        else:
            out[w] = I[c * bs:(c + 1) * bs, :].astype(np.float64) @ E[k * bs:(k + 1) * bs, :].astype(np.float64).T
    return out
