"""Host mirror of the reference's ``BlocksparseTransformer`` (blocksparse/transformer.py:49-443) over the C ABI in
include/bst.h: block-sparse attention scores (nt), row softmax with bit masks, weighted values (nn) and their gradients
(tn), on torch CUDA tensors.  Same constructor, attributes, op names and NumPy ``*_test`` methods as the reference, so a
reference test ports line by line.  There is no CPU fallback for the ops.

Shapes: activations [batch, ctx_blks * blk_size, heads * head_state]; scores [batch, heads, blocks, blk_size, blk_size].
"""
import ctypes

import numpy as np

from . import _lib

try:
    import torch
except Exception:  # pragma: no cover
    torch = None

_MASK_DTYPE = {64: np.uint64, 32: np.uint32, 16: np.uint16, 8: np.uint8}
_MASK_TORCH = {64: "int64", 32: "int32", 16: "int16", 8: "uint8"}


def _code(dt):
    if dt == torch.float32:
        return _lib.F32
    if dt == torch.float16:
        return _lib.F16
    if dt == torch.bfloat16:
        return _lib.BF16
    raise TypeError("blocksparse_amd: unsupported dtype %s" % dt)


def build_bst_tables(layout):
    """layout [heads_l, Qb, Kb] -> nt_lut [H, blocks, 2], nn_lut [H, Qb+blocks, 2], tn_lut [H, Kb+blocks, 2], nn_max,
    tn_max.  Blocks are numbered row-major ("contiguous along the rows", blocksparse/transformer.py:103-109); nn/tn
    tables = header (offset, count) per output block followed by (block id, other block) entries in block order
    (xn_lut, blocksparse/transformer.py:141-165).  Vectorised; bit-identical to the reference's tables."""
    H, Qb, Kb = layout.shape
    nts, nns, tns = [], [], []
    nn_max = tn_max = 0
    blocks = None
    for h in range(H):
        qs, ks = np.nonzero(layout[h])                       # row-major: sorted by (q, k)
        if blocks is None:
            blocks = qs.size
        elif qs.size != blocks:
            raise AssertionError("number of layout blocks must be equal across heads")
        ids = np.arange(blocks, dtype=np.int32)
        nts.append(np.stack([qs, ks], axis=1).astype(np.int32))

        def xn(ys, xs, ctx):
            cnt = np.bincount(ys, minlength=ctx).astype(np.int32)
            off = ctx + np.concatenate([[0], np.cumsum(cnt)[:-1]]).astype(np.int32)
            order = np.argsort(ys, kind="stable")
            lut = np.empty((ctx + blocks, 2), dtype=np.int32)
            lut[:ctx, 0], lut[:ctx, 1] = off, cnt
            lut[ctx:, 0], lut[ctx:, 1] = ids[order], xs[order]
            return lut, int(cnt.max()) if cnt.size else 0

        nn, m1 = xn(qs, ks, Qb)
        tn, m2 = xn(ks, qs, Kb)
        nns.append(nn)
        tns.append(tn)
        nn_max, tn_max = max(nn_max, m1), max(tn_max, m2)
    return np.array(nts, dtype=np.int32), np.array(nns, dtype=np.int32), np.array(tns, dtype=np.int32), nn_max, tn_max, blocks


class BlocksparseTransformer(object):
    """Same surface as the reference class (blocksparse/transformer.py:49-139)."""

    def __init__(self, layout, block_size=64, heads=None, mask_callback=None, name=None):
        layout = np.asarray(layout)
        if layout.ndim == 2:
            assert heads is not None, "heads must be explicitly specified when using shared layouts per head"
            layout = layout[None]
        if heads is None:
            heads = layout.shape[0]
        assert block_size in (8, 16, 32, 64), "Block sizes of 8, 16, 32 and 64 currently supported"
        assert layout.ndim == 3, "bad layout shape: " + str(layout.shape)
        self.blk_size = block_size
        self.name = name
        self.heads = heads
        self.lut_heads = layout.shape[0]
        self.ctx_blks_q = layout.shape[1]
        self.ctx_blks_k = layout.shape[2]
        self.blk_shape = (block_size, block_size)
        self.softmax_dtype = None
        assert self.lut_heads in (1, heads), "layout heads must be 1 or heads"
        lay = (layout != 0)
        self.nt_lut, self.nn_lut, self.tn_lut, self.nn_max, self.tn_max, self.blocks = build_bst_tables(lay)
        self.nt_list = [[(int(q), int(k)) for q, k in t] for t in self.nt_lut]
        self._dev = {}
        if mask_callback is not None:
            self.init_softmax_mask(mask_callback)
        else:
            self.softmax_mask = None
            self.softmax_mask_np = None

    # python lists of the reference (built on demand: only the NumPy test methods use them)
    def _xn_list(self, lut, ctx):
        return [[(int(b), int(x)) for b, x in lut[off:off + cnt]] for off, cnt in lut[:ctx]]

    @property
    def nn_list(self):
        return [self._xn_list(t, self.ctx_blks_q) for t in self.nn_lut]

    @property
    def tn_list(self):
        return [self._xn_list(t, self.ctx_blks_k) for t in self.tn_lut]

    def init_softmax_mask(self, mask_callback):
        """One unsigned integer of blk_size bits per (block, query row), bit k = key k visible
        (blocksparse/transformer.py:129-159)."""
        bs = self.blk_size
        dtype = _MASK_DTYPE[bs]
        w = np.uint64(1) << np.arange(bs, dtype=np.uint64)
        out = np.empty((self.lut_heads, self.blocks, bs), dtype=dtype)
        for h in range(self.lut_heads):
            for b, (q, k) in enumerate(self.nt_list[h]):
                m = np.asarray(mask_callback(self.blk_shape, h, q, k, b)).astype(bool)
                out[h, b] = (m.astype(np.uint64) * w[None, :]).sum(axis=1, dtype=np.uint64).astype(dtype)
        self.softmax_mask_np = out                                             # heads, blocks, blk_size
        self.softmax_mask = np.ascontiguousarray(out.transpose(0, 2, 1))       # heads, blk_size, blocks (kernel layout)
        self._dev = {k: v for k, v in self._dev.items() if k[0] != "mask"}

    def block_coord(self, block, head=0):
        return self.nt_list[head][block]

    # ------------------------------------------------------------------ device side
    def _table(self, name, device):
        key = (name, str(device))
        t = self._dev.get(key)
        if t is None:
            if name == "mask":
                arr = self.softmax_mask.view({64: np.int64, 32: np.int32, 16: np.int16, 8: np.uint8}[self.blk_size])
            else:
                arr = getattr(self, name)
            t = torch.from_numpy(np.ascontiguousarray(arr)).to(device)
            self._dev[key] = t
        return t

    def _args(self, lut_t, batch, head_state, dtype, score_dtype):
        a = _lib.BstArgs()
        a.lut = lut_t.data_ptr()
        a.lut_heads, a.lut_dim = lut_t.shape[0], lut_t.shape[1]
        a.blocks, a.bsize, a.batch, a.heads, a.head_state = self.blocks, self.blk_size, batch, self.heads, head_state
        a.ctx_blks_q, a.ctx_blks_k = self.ctx_blks_q, self.ctx_blks_k
        a.dtype, a.score_dtype = dtype, score_dtype
        a.stream = _lib.raw_stream(lut_t.device)
        return a

    def _check_act(self, t, ctx_blks, what):
        if not (t.is_cuda and t.is_contiguous() and t.dim() == 3):
            raise ValueError("%s: expected a contiguous 3-d CUDA tensor" % what)
        if t.shape[1] != ctx_blks * self.blk_size:
            raise ValueError("%s: bad context length %d (expected %d)" % (what, t.shape[1], ctx_blks * self.blk_size))
        if t.shape[2] % self.heads:
            raise ValueError("state_dim not evenly divisible by number of heads")
        if (t.shape[2] // self.heads) % 8:
            raise ValueError("Head state dim must be multiple of 8")

    def _check_scores(self, w, batch):
        want = (batch, self.heads, self.blocks, self.blk_size, self.blk_size)
        if not (w.is_cuda and w.is_contiguous() and tuple(w.shape) == want and w.dtype in (torch.bfloat16, torch.float16)):
            raise ValueError("scores: expected a contiguous %s bf16/fp16 CUDA tensor, got %s %s" % (want, tuple(w.shape), w.dtype))

    @staticmethod
    def _score_dtype(act_dtype):
        # bf16 scores next to fp32 (and bf16) activations, fp16 next to fp16 (blocksparse/transformer.py:367,441)
        return torch.float16 if act_dtype == torch.float16 else torch.bfloat16

    def _nt(self, a, b, score_dtype):
        self._check_act(a, self.ctx_blks_q, "a")
        self._check_act(b, self.ctx_blks_k, "b")
        if a.dtype != b.dtype or a.shape[0] != b.shape[0] or a.shape[2] != b.shape[2]:
            raise ValueError("Mismatched Shapes: a,b")
        lut = self._table("nt_lut", a.device)
        c = torch.empty((a.shape[0], self.heads, self.blocks, self.blk_size, self.blk_size), dtype=score_dtype, device=a.device)
        args = self._args(lut, a.shape[0], a.shape[2] // self.heads, _code(a.dtype), _code(score_dtype))
        _lib.check(_lib.load().bst_nt(a.data_ptr(), b.data_ptr(), c.data_ptr(), ctypes.byref(args)), "bst_nt")
        return c

    def _xn(self, w, b, trans):
        ctx_b, ctx_c = (self.ctx_blks_q, self.ctx_blks_k) if trans else (self.ctx_blks_k, self.ctx_blks_q)
        self._check_act(b, ctx_b, "b")
        self._check_scores(w, b.shape[0])
        lut = self._table("tn_lut" if trans else "nn_lut", b.device)
        c = torch.empty((b.shape[0], ctx_c * self.blk_size, b.shape[2]), dtype=b.dtype, device=b.device)
        args = self._args(lut, b.shape[0], b.shape[2] // self.heads, _code(b.dtype), _code(w.dtype))
        fn = _lib.load().bst_tn if trans else _lib.load().bst_nn
        _lib.check(fn(w.data_ptr(), b.data_ptr(), c.data_ptr(), ctypes.byref(args)), "bst_tn" if trans else "bst_nn")
        return c

    def _softmax_fwd(self, x, scale, mask_t, y_dtype):
        self._check_scores(x, x.shape[0])
        lut = self._table("nn_lut", x.device)
        y = torch.empty(x.shape, dtype=y_dtype, device=x.device)
        args = self._args(lut, x.shape[0], 8, _lib.F32, _code(x.dtype))
        mp = mask_t.data_ptr() if mask_t is not None else None
        mh = mask_t.shape[0] if mask_t is not None else 1
        _lib.check(_lib.load().bst_masked_softmax(x.data_ptr(), y.data_ptr(), mp, mh, float(scale), _code(x.dtype), _code(y_dtype),
                                                  ctypes.byref(args)), "bst_masked_softmax")
        return y

    def _nt_softmax(self, q, k, scale, mask_t, y_dtype):
        """probabilities = softmax(scale * scores(q, k) + mask) as ONE launch (bst_nt_softmax, round 6); None where the fused kernel does not
        serve the configuration (the caller composes the two ops then)."""
        self._check_act(q, self.ctx_blks_q, "q")
        self._check_act(k, self.ctx_blks_k, "k")
        if q.dtype != k.dtype or q.shape[0] != k.shape[0] or q.shape[2] != k.shape[2]:
            raise ValueError("Mismatched Shapes: q,k")
        hs = q.shape[2] // self.heads
        if self.blk_size != 32 or self.nn_max > 20 or hs not in (32, 64, 128):
            return None
        lut = self._table("nn_lut", q.device)
        y = torch.empty((q.shape[0], self.heads, self.blocks, self.blk_size, self.blk_size), dtype=y_dtype, device=q.device)
        args = self._args(lut, q.shape[0], hs, _code(q.dtype), _code(y_dtype))
        mp = mask_t.data_ptr() if mask_t is not None else None
        mh = mask_t.shape[0] if mask_t is not None else 1
        rc = _lib.load().bst_nt_softmax(q.data_ptr(), k.data_ptr(), y.data_ptr(), mp, mh, float(scale), int(self.nn_max), ctypes.byref(args))
        if rc == -2:                                                           # BSMM_ERR_UNSUPPORTED
            return None
        _lib.check(rc, "bst_nt_softmax")
        return y

    def _nt_softmax_grad(self, e, v, y, scale):
        """dx = softmax_grad(scores(e, v) rounded to y's type, y) as ONE launch (bst_nt_softmax_grad); None where the fused kernel does not serve."""
        self._check_act(e, self.ctx_blks_q, "e")
        self._check_act(v, self.ctx_blks_k, "v")
        self._check_scores(y, e.shape[0])
        hs = e.shape[2] // self.heads
        if self.blk_size != 32 or self.nn_max > 20 or hs not in (32, 64, 128) or e.dtype != v.dtype:
            return None
        lut = self._table("nn_lut", e.device)
        dx = torch.empty_like(y)
        args = self._args(lut, e.shape[0], hs, _code(e.dtype), _code(y.dtype))
        rc = _lib.load().bst_nt_softmax_grad(e.data_ptr(), v.data_ptr(), y.data_ptr(), dx.data_ptr(), float(scale), int(self.nn_max), ctypes.byref(args))
        if rc == -2:
            return None
        _lib.check(rc, "bst_nt_softmax_grad")
        return dx

    def _softmax_bwd(self, dy, y, scale):
        self._check_scores(y, y.shape[0])
        dy = dy.contiguous().to(y.dtype)
        lut = self._table("nn_lut", y.device)
        dx = torch.empty_like(y)
        args = self._args(lut, y.shape[0], 8, _lib.F32, _code(y.dtype))
        _lib.check(_lib.load().bst_softmax_grad(dy.data_ptr(), y.data_ptr(), dx.data_ptr(), float(scale), _code(y.dtype), ctypes.byref(args)),
                   "bst_softmax_grad")
        return dx

    def partial_autoregressive_mask(self, autoregress_at_key, device):
        """softmax mask with keys from ``autoregress_at_key`` on made causal (bst_partial_autoregressive_mask)."""
        if not (0 <= autoregress_at_key < self.ctx_blks_k * self.blk_size):
            raise ValueError("autoregress_at_key out of range")                 # src/bst_op.cc:533
        m = self._table("mask", device)
        out = torch.empty_like(m)
        lut = self._table("nt_lut", device)
        st = _lib.raw_stream(device)
        _lib.check(_lib.load().bst_partial_autoregressive_mask(m.data_ptr(), out.data_ptr(), lut.data_ptr(), self.blk_size, self.blocks,
                                                               self.lut_heads, int(autoregress_at_key), st), "bst_partial_autoregressive_mask")
        return out

    # ------------------------------------------------------------------ ops (with the reference's registered gradients)
    def nt_op(self, a, b, name=None, bench=0):
        return _NT.apply(self, a, b, torch.bfloat16)                             # CT=tf.bfloat16, blocksparse/transformer.py:326-333

    def nn_op(self, a, b, name=None, bench=0):
        return _XN.apply(self, a, b, False)

    def tn_op(self, a, b, name=None, bench=0):
        return _XN.apply(self, a, b, True)

    def query_key_op(self, q, k, name=None, bench=0):
        self.softmax_dtype = self._score_dtype(q.dtype)
        return _NT.apply(self, q, k, torch.bfloat16)                             # blocksparse/transformer.py:364-374

    def query_key_softmax(self, q, k, scale=1.0, autoregress_at_key=None, dtype=None):
        """``masked_softmax(query_key_op(q, k), scale, autoregress_at_key, dtype)`` as one operator (round 6): where the fused kernel serves the
        configuration (block size 32, head states of 32 / 64 / 128, query rows of up to 20 blocks) the raw scores never reach memory -- one
        launch, a third of the bytes; elsewhere the two operators run as the reference composes them (blocksparse/transformer.py:364-409).  Same
        values either way (the scores are rounded to the score type before the softmax in both), same gradients (blocksparse_softmax_grad, then
        blocksparse_transformer_nt_grad)."""
        self.softmax_dtype = self._score_dtype(q.dtype)
        if self.softmax_mask is None:
            if autoregress_at_key is not None:
                raise ValueError("autoregress_at_key only applies to ops with mask_callback defined.")
            mask_t = None
        elif autoregress_at_key is not None:
            mask_t = self.partial_autoregressive_mask(autoregress_at_key, q.device)
        else:
            mask_t = self._table("mask", q.device)
        if dtype is None:
            dtype = self.softmax_dtype
        return _NTSoftmax.apply(self, q, k, float(scale), mask_t, dtype)

    def weight_value_op(self, w, v, name=None, bench=0):
        return _XN.apply(self, w, v, False)

    def attention(self, q, k, v, scale=1.0, autoregress_at_key=None):
        """``weight_value_op(masked_softmax(query_key_op(q, k), scale, autoregress_at_key), v)`` as one operator (round 6): two launches forward
        (scores + softmax, weighted values) and four backward (dv; [scores of (dy, v) + softmax gradient] as one; dq; dk) where the fused kernels
        serve the configuration, the reference's composition elsewhere.  The probabilities are kept for the backward pass, the raw scores and
        the gradient of the probabilities never reach memory."""
        self.softmax_dtype = self._score_dtype(q.dtype)
        if self.softmax_mask is None:
            if autoregress_at_key is not None:
                raise ValueError("autoregress_at_key only applies to ops with mask_callback defined.")
            mask_t = None
        elif autoregress_at_key is not None:
            mask_t = self.partial_autoregressive_mask(autoregress_at_key, q.device)
        else:
            mask_t = self._table("mask", q.device)
        return _Attention.apply(self, q, k, v, float(scale), mask_t, self.softmax_dtype)

    def masked_softmax(self, x, scale=1.0, autoregress_at_key=None, dtype=None):
        if self.softmax_mask is None:
            if autoregress_at_key is not None:
                raise ValueError("autoregress_at_key only applies to ops with mask_callback defined.")
            return self.softmax(x, scale, dtype)
        if autoregress_at_key is not None:
            mask_t = self.partial_autoregressive_mask(autoregress_at_key, x.device)
        else:
            mask_t = self._table("mask", x.device)
        if dtype is None:
            dtype = self.softmax_dtype or x.dtype
        return _Softmax.apply(self, x, float(scale), mask_t, dtype)

    def softmax(self, x, scale=1.0, dtype=None):
        if dtype is None:
            dtype = self.softmax_dtype or x.dtype
        return _Softmax.apply(self, x, float(scale), None, dtype)

    # ------------------------------------------------------------------ NumPy reference methods (same names as the reference)
    def _split(self, X, ctx_blks):
        s = list(X.shape)
        return X.reshape(s[0], ctx_blks, self.blk_size, self.heads, s[2] // self.heads)

    def nt_test(self, A, B):
        A5, B5 = self._split(A, self.ctx_blks_q), self._split(B, self.ctx_blks_k)
        C = np.empty([A5.shape[0], self.heads, self.blocks, self.blk_size, self.blk_size], dtype=np.float32)
        for h in range(self.heads):
            lut = self.nt_lut[h if self.lut_heads > 1 else 0]
            for n in range(A5.shape[0]):
                C[n, h] = np.matmul(A5[n, lut[:, 0], :, h, :], B5[n, lut[:, 1], :, h, :].transpose(0, 2, 1))
        return C

    def _xn_test(self, A, B, trans):
        ctx_b, ctx_c = (self.ctx_blks_q, self.ctx_blks_k) if trans else (self.ctx_blks_k, self.ctx_blks_q)
        B5 = self._split(B, ctx_b)
        C = np.zeros([B5.shape[0], ctx_c, self.blk_size, self.heads, B5.shape[4]], dtype=np.float32)
        for h in range(self.heads):
            lut = self.nt_lut[h if self.lut_heads > 1 else 0]
            src, dst = (lut[:, 0], lut[:, 1]) if trans else (lut[:, 1], lut[:, 0])
            for n in range(B5.shape[0]):
                Wb = A[n, h].transpose(0, 2, 1) if trans else A[n, h]
                acc = np.zeros((ctx_c, self.blk_size, B5.shape[4]), dtype=np.float32)
                np.add.at(acc, dst, np.matmul(Wb, B5[n, src, :, h, :]))
                C[n, :, :, h, :] = acc
        return C.reshape(B5.shape[0], ctx_c * self.blk_size, -1)

    def nn_test(self, A, B):
        return self._xn_test(A, B, False)

    def tn_test(self, A, B):
        return self._xn_test(A, B, True)

    def masked_softmax_test(self, x, scale=1.0, autoregress_at_key=None):
        bs = self.blk_size
        m = self.softmax_mask_np
        y = np.empty_like(x)
        neg = -np.finfo(np.float32).max
        for h in range(x.shape[1]):
            hl = h if self.lut_heads > 1 else 0
            for ent in self.nn_list[hl]:
                if not ent:
                    continue
                ids = [b for b, _ in ent]
                xb = x[:, h, ids] * scale
                if m is not None:
                    bits = m[hl, ids].astype(np.uint64)                                  # [nb, q]
                    if autoregress_at_key is not None:
                        ones = (1 << bs) - 1
                        for i, (b, k) in enumerate(ent):
                            Q, K = self.nt_list[hl][b][0] * bs, k * bs
                            sa = bs - min(max(autoregress_at_key - K, 0), bs)
                            for q in range(bs):
                                sb = min(max(bs - 1 + K - (Q + q), 0), bs)
                                bits[i, q] = int(bits[i, q]) & (ones >> min(sa, sb))
                    keep = ((bits[..., None] >> np.arange(bs, dtype=np.uint64)) & np.uint64(1)).astype(bool)
                    xb = np.where(keep[None], xb, neg)
                ex = np.exp(xb - xb.max(axis=(1, 3), keepdims=True))
                y[:, h, ids] = ex / ex.sum(axis=(1, 3), keepdims=True)
        return y

    def masked_softmax_grad_test(self, dy, y, scale=1.0):
        dx = np.empty_like(dy)
        for h in range(dy.shape[1]):
            hl = h if self.lut_heads > 1 else 0
            for ent in self.nn_list[hl]:
                if not ent:
                    continue
                ids = [b for b, _ in ent]
                d, v = dy[:, h, ids], y[:, h, ids]
                dx[:, h, ids] = (d - (d * v).sum(axis=(1, 3), keepdims=True)) * v * scale
        return dx


if torch is not None:

    class _NT(torch.autograd.Function):
        """scores = a . b^T;  gradients as blocksparse_transformer_nt_grad (blocksparse/transformer.py:411-438)."""

        @staticmethod
        def forward(ctx, bst, a, b, score_dtype):
            ctx.bst = bst
            ctx.save_for_backward(a, b)
            return bst._nt(a.contiguous(), b.contiguous(), score_dtype)

        @staticmethod
        def backward(ctx, dw):
            a, b = ctx.saved_tensors
            dw = dw.contiguous()
            db = ctx.bst._xn(dw, a.contiguous(), True)
            da = ctx.bst._xn(dw, b.contiguous(), False)
            return None, da, db, None

    class _NTSoftmax(torch.autograd.Function):
        """probabilities = softmax(scale * (q . k^T) + mask): forward fused where the kernel serves it, else nt then softmax; backward =
        blocksparse_softmax_grad followed by the nt gradients (blocksparse/transformer.py:411-438, 479-509)."""

        @staticmethod
        def forward(ctx, bst, q, k, scale, mask_t, y_dtype):
            q, k = q.contiguous(), k.contiguous()
            y = bst._nt_softmax(q, k, scale, mask_t, y_dtype)
            if y is None:
                y = bst._softmax_fwd(bst._nt(q, k, torch.bfloat16), scale, mask_t, y_dtype)
            ctx.bst, ctx.scale = bst, scale
            ctx.save_for_backward(q, k, y)
            return y

        @staticmethod
        def backward(ctx, dy):
            q, k, y = ctx.saved_tensors
            bst = ctx.bst
            dw = bst._softmax_bwd(dy, y, ctx.scale)
            dk = bst._xn(dw, q, True)
            dq = bst._xn(dw, k, False)
            return None, dq, dk, None, None, None

    class _Attention(torch.autograd.Function):
        """y = nn(softmax(scale * nt(q, k) + mask), v) with the gradients of the three registered ops composed
        (blocksparse/transformer.py:411-509): dv = tn(a, dy); dx = softmax_grad(nt(dy, v), a); dq = nn(dx, k); dk = tn(dx, q)."""

        @staticmethod
        def forward(ctx, bst, q, k, v, scale, mask_t, a_dtype):
            q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
            a = bst._nt_softmax(q, k, scale, mask_t, a_dtype)
            if a is None:
                a = bst._softmax_fwd(bst._nt(q, k, torch.bfloat16), scale, mask_t, a_dtype)
            ctx.bst, ctx.scale = bst, scale
            ctx.save_for_backward(q, k, v, a)
            return bst._xn(a, v, False)

        @staticmethod
        def backward(ctx, dy):
            q, k, v, a = ctx.saved_tensors
            bst = ctx.bst
            dy = dy.contiguous()
            dv = bst._xn(a, dy, True)
            dx = bst._nt_softmax_grad(dy, v, a, ctx.scale)
            if dx is None:
                dx = bst._softmax_bwd(bst._nt(dy, v, a.dtype), a, ctx.scale)
            dq = bst._xn(dx, k, False)
            dk = bst._xn(dx, q, True)
            return None, dq, dk, dv, None, None, None

    class _XN(torch.autograd.Function):
        """c = w . b (nn) or w^T . b (tn); nn gradients as blocksparse_transformer_nn_grad
        (blocksparse/transformer.py:446-476); tn gets the symmetric pair."""

        @staticmethod
        def forward(ctx, bst, w, b, trans):
            ctx.bst, ctx.trans = bst, trans
            ctx.save_for_backward(w, b)
            return bst._xn(w.contiguous(), b.contiguous(), trans)

        @staticmethod
        def backward(ctx, dc):
            w, b = ctx.saved_tensors
            bst = ctx.bst
            dc = dc.contiguous()
            db = bst._xn(w, dc, not ctx.trans)
            dw = bst._nt(b, dc, w.dtype) if ctx.trans else bst._nt(dc, b, w.dtype)
            return None, dw, db, None

    class _Softmax(torch.autograd.Function):
        """blocksparse_masked_softmax / blocksparse_softmax with blocksparse_softmax_grad
        (blocksparse/transformer.py:479-509)."""

        @staticmethod
        def forward(ctx, bst, x, scale, mask_t, y_dtype):
            y = bst._softmax_fwd(x.contiguous(), scale, mask_t, y_dtype)
            ctx.bst, ctx.scale, ctx.x_dtype = bst, scale, x.dtype
            ctx.save_for_backward(y)
            return y

        @staticmethod
        def backward(ctx, dy):
            (y,) = ctx.saved_tensors
            return None, ctx.bst._softmax_bwd(dy, y, ctx.scale).to(ctx.x_dtype), None, None, None
