"""Host mirror of the reference's ``SparseProj`` (blocksparse/matmul.py:835-921): gather a subset of the feature rows of a
(features, minibatch) activation tensor, scatter them back, scatter-add / scatter-multiply a small tensor into a large one --
with the registered gradients (``gather_scatter_grad``, ``scatter_add_mul_grad``, blocksparse/matmul.py:893-909).
Device ops go through libbsmm_hip.so (bsmm_sparse_op / bsmm_sparse_mul_grad); there is no CPU fallback."""
import numpy as np

from . import _lib

try:
    import torch
except Exception:  # pragma: no cover
    torch = None

OP_GAT, OP_SCT, OP_ADD, OP_MUL = 0, 1, 2, 3


def _code(dt):
    return {torch.float32: _lib.F32, torch.float16: _lib.F16, torch.bfloat16: _lib.BF16}[dt]


class SparseProj(object):
    def __getstate__(self):
        return (self.nhidden, self.nproj, self.gather_lut, self.name)

    def __setstate__(self, state):
        self.__init__(state[0], nproj=state[1], gather_lut=state[2], name=state[3])

    def __init__(self, nhidden, nproj=None, proj_stride=None, block_size=32, gather_lut=None, name=None):
        if gather_lut is None:
            gather_lut = np.arange(nhidden, dtype=np.int32)
            if nproj is not None:
                assert nproj <= nhidden
                np.random.shuffle(gather_lut)                                  # as the reference: global NumPy RNG
                gather_lut = np.sort(gather_lut[0:nproj])
            elif proj_stride is not None:
                assert proj_stride <= nhidden
                gather_max = ((nhidden // proj_stride) // block_size) * block_size * proj_stride   # trim to a multiple of block_size
                gather_lut = gather_lut[:gather_max:proj_stride].copy()
                nproj = gather_lut.size
            else:
                raise ValueError("missing nproj, proj_stride or gather_lut")
        gather_lut = np.ascontiguousarray(gather_lut, dtype=np.int32)
        nproj = int(gather_lut.size)
        scatter_lut = np.full(nhidden, -1, dtype=np.int32)                     # reverse mapping
        scatter_lut[gather_lut] = np.arange(nproj, dtype=np.int32)
        self.name = name or "SparseProj"
        self.gather_lut, self.scatter_lut = gather_lut, scatter_lut
        self.nhidden, self.nproj = int(nhidden), nproj
        self._dev = {}

    def _luts(self, device):
        key = str(device)
        if key not in self._dev:
            self._dev[key] = (torch.from_numpy(self.gather_lut).to(device), torch.from_numpy(self.scatter_lut).to(device))
        return self._dev[key]

    def _check(self, t, rows, what):
        if not (t.is_cuda and t.dim() >= 1 and t.shape[0] == rows):
            raise ValueError("%s: expected a CUDA tensor with %d rows (feature axis 0)" % (what, rows))
        return t.contiguous()

    def _op(self, op, x, y, lut, K, rows_z):
        N = x.numel() // x.shape[0]
        z = torch.empty((rows_z,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        st = _lib.raw_stream(x.device)
        _lib.check(_lib.load().bsmm_sparse_op(z.data_ptr(), x.data_ptr(), y.data_ptr() if y is not None else None, lut.data_ptr(), op, K,
                                              rows_z, N, _code(x.dtype), st), "bsmm_sparse_op")
        return z

    def _mul_grad(self, dz, x, y):
        g, _ = self._luts(x.device)
        N = x.numel() // x.shape[0]
        dx = torch.empty_like(x)
        dy = torch.empty_like(y)
        st = _lib.raw_stream(x.device)
        _lib.check(_lib.load().bsmm_sparse_mul_grad(dx.data_ptr(), dy.data_ptr(), dz.contiguous().data_ptr(), x.data_ptr(), y.data_ptr(),
                                                    g.data_ptr(), self.nproj, self.nhidden, N, _code(x.dtype), st), "bsmm_sparse_mul_grad")
        return dx, dy

    def gather(self, x):
        return _Gather.apply(self, self._check(x, self.nhidden, "x"), False)

    def scatter(self, x):
        return _Gather.apply(self, self._check(x, self.nproj, "x"), True)

    def scatter_add(self, x, y):
        return _AddMul.apply(self, self._check(x, self.nhidden, "x"), self._check(y, self.nproj, "y"), OP_ADD)

    def scatter_mul(self, x, y):
        return _AddMul.apply(self, self._check(x, self.nhidden, "x"), self._check(y, self.nproj, "y"), OP_MUL)

    # NumPy statements of the four ops (host-side helpers, as the reference class ships its *_test methods for matmul)
    def gather_test(self, x):
        return np.asarray(x)[self.gather_lut]

    def scatter_test(self, x):
        z = np.zeros((self.nhidden,) + np.asarray(x).shape[1:], dtype=np.asarray(x).dtype)
        z[self.gather_lut] = x
        return z

    def scatter_add_test(self, x, y):
        z = np.array(x)
        z[self.gather_lut] += y
        return z

    def scatter_mul_test(self, x, y):
        z = np.array(x)
        z[self.gather_lut] *= y
        return z


if torch is not None:

    class _Gather(torch.autograd.Function):
        @staticmethod
        def forward(ctx, sp, x, scatter):
            ctx.sp, ctx.scatter = sp, scatter
            g, s = sp._luts(x.device)
            if scatter:
                return sp._op(OP_SCT, x, None, s, sp.nhidden, sp.nhidden)
            return sp._op(OP_GAT, x, None, g, sp.nproj, sp.nproj)

        @staticmethod
        def backward(ctx, dy):                               # the gradient of gather is scatter and vice versa
            sp = ctx.sp
            g, s = sp._luts(dy.device)
            dy = dy.contiguous()
            if ctx.scatter:
                return None, sp._op(OP_GAT, dy, None, g, sp.nproj, sp.nproj), None
            return None, sp._op(OP_SCT, dy, None, s, sp.nhidden, sp.nhidden), None

    class _AddMul(torch.autograd.Function):
        @staticmethod
        def forward(ctx, sp, x, y, op):
            ctx.sp, ctx.op = sp, op
            g, s = sp._luts(x.device)
            if op == OP_MUL:
                ctx.save_for_backward(x, y)
                return sp._op(OP_MUL, x, y, s, sp.nhidden, sp.nhidden)
            return sp._op(OP_ADD, x, y, g, sp.nproj, sp.nhidden)

        @staticmethod
        def backward(ctx, dz):
            sp = ctx.sp
            dz = dz.contiguous()
            if ctx.op == OP_ADD:
                g, _ = sp._luts(dz.device)
                return None, dz, sp._op(OP_GAT, dz, None, g, sp.nproj, sp.nproj), None
            x, y = ctx.saved_tensors
            dx, dy = sp._mul_grad(dz, x, y)
            return None, dx, dy, None
