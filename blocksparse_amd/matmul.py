"""``BlocksparseMatMul`` -- host-side mirror of the reference operator interface.

Same constructor, attributes and call semantics as /root/reference/blocksparse/matmul.py:74-483
(``BlocksparseMatMul``), with the TensorFlow custom ops replaced by the C ABI of libbsmm_hip.so
(include/bsmm.h) and TF autodiff (matmul.py:485-527) replaced by a ``torch.autograd.Function``.
PyTorch is used for device memory, streams and autograd plumbing only; all arithmetic on the path
(fprop / bprop / updat) runs in the hand-written gfx950 kernels.  There is no CPU fallback.

    bsmm = BlocksparseMatMul(layout, block_size=32, feature_axis=0)
    w = torch.randn(bsmm.w_shape, device="cuda") * 0.01
    y = bsmm(x, w)              # x: (C, N) for feature_axis=0, (N, C) for feature_axis=1
    y.backward(dy)              # dx via bprop lut, dw via updat lut

Differences from the reference that a user can observe:
  * (axis, block_size) combinations: the reference admits axis 0 x {8,16,32} and axis 1 x {32,64}
    (matmul.py:84-89); we admit {8,16,32} on both axes (north_star) and 64 on axis 1 (a 64x64 block is
    addressed as four 32x32 blocks of the layout kron(layout, ones(2,2)); see ``_split64``).
  * By default the device walks an *unsegmented* lookup table (one segment per output block, no locks:
    deterministic, fp32-accumulated, single rounding).  ``segmented=True`` feeds the device the
    reference-policy tables; output blocks shared by several segments then meet in an fp32 image of the
    output and are rounded ONCE (the reference rounds every partial sum to the storage type).  The public
    attributes ``fprop_lut`` etc. always hold the reference-policy tables (bit-identical to the reference builder).
  * ``gate=`` / ``gate_grad`` / ``dw_gated`` follow the reference (matmul.py:455-527).  Gated calls run the plan
    kernels for block_size 32 with 16-bit types (staged xprop kernel: exact two-piece split of gate * w; streaming
    updat kernel: the gate is applied in its summing pass), the per-segment / per-block kernels otherwise.
"""
import ctypes
import weakref

import numpy as np

from . import _lib
from . import lut as _lut

try:  # torch is plumbing (device memory, streams, autograd); import lazily-tolerant for pure-host use
    import torch
except Exception:  # pragma: no cover
    torch = None


def _dtype_code(dt):
    if dt == torch.float32:
        return _lib.F32
    if dt == torch.float16:
        return _lib.F16
    if dt == torch.bfloat16:
        return _lib.BF16
    raise TypeError("blocksparse_amd: unsupported dtype %s (float32, float16, bfloat16)" % dt)


def _host_plan(lut, segments, blocks, n_out_blocks, bsize, dtype_code, axis, options=0):
    """Grouped-kernel schedule for one xprop lut (host call into the library: bsmm_xprop_plan_build)."""
    lib = _lib.load()
    lut = np.ascontiguousarray(lut, dtype=np.int32)
    ip = ctypes.POINTER(ctypes.c_int32)
    words = lib.bsmm_xprop_plan_words(lut.ctypes.data_as(ip), segments, blocks, n_out_blocks, bsize, dtype_code, axis, options)
    if words < 0:
        raise RuntimeError("bsmm_xprop_plan_words rejected the lookup table")
    if words == 0:
        return None
    out = np.empty(words, dtype=np.int32)
    _lib.check(lib.bsmm_xprop_plan_build(lut.ctypes.data_as(ip), segments, blocks, n_out_blocks, bsize, dtype_code, axis, options,
                                         out.ctypes.data_as(ip)), "bsmm_xprop_plan_build")
    return out


def _host_updat_plan(updat_lut, blocks, CB, KB, bsize, dtype_code, axis, options=0):
    """Work items of the windowed updat kernel (host call into the library: bsmm_updat_plan_build)."""
    lib = _lib.load()
    lut = np.ascontiguousarray(updat_lut, dtype=np.int32)
    ip = ctypes.POINTER(ctypes.c_int32)
    words = lib.bsmm_updat_plan_words(lut.ctypes.data_as(ip), blocks, CB, KB, bsize, dtype_code, axis, options)
    if words < 0:
        raise RuntimeError("bsmm_updat_plan_words rejected the lookup table")
    if words == 0:
        return None
    out = np.empty(words, dtype=np.int32)
    _lib.check(lib.bsmm_updat_plan_build(lut.ctypes.data_as(ip), blocks, CB, KB, bsize, dtype_code, axis, options,
                                         out.ctypes.data_as(ip)), "bsmm_updat_plan_build")
    return out


class _Plan(object):
    """A schedule of the library: the host words (the library reads its descriptor from them, bsmm_plan_attach) and the
    device copy the kernels walk."""

    def __init__(self, host, device):
        self.host = np.ascontiguousarray(host, dtype=np.int32)
        self.dev = torch.from_numpy(self.host).to(device)

    def attach(self, args):
        ip = ctypes.POINTER(ctypes.c_int32)
        _lib.check(_lib.load().bsmm_plan_attach(ctypes.byref(args), self.host.ctypes.data_as(ip), self.host.size, self.dev.data_ptr()),
                   "bsmm_plan_attach")


class _DeviceTables(object):
    """int32 lookup tables resident on one device (the reference keeps them as TF variables, matmul.py:33-53),
    plus the derived schedules ("plans") of the grouped kernels."""

    def __init__(self, tables, device, bsize, axis, plan_options=0, xprop_only=False):
        def up(a):
            return torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).to(device)

        def plan(words):
            return _Plan(words, device) if words is not None else None
        self.fprop = up(tables["fprop"]["lut"])
        self.bprop = up(tables["bprop"]["lut"])
        self.updat = up(tables["updat_lut"])
        CB, KB, B = tables["CB"], tables["KB"], tables["blocks"]
        f, b = tables["fprop"], tables["bprop"]
        # plans exist only for 16-bit types (they do not depend on which of the two) ...
        self.fprop_plan = plan(_host_plan(f["lut"], f["segments"], B, KB, bsize, _lib.BF16, axis, plan_options))
        self.bprop_plan = plan(_host_plan(b["lut"], b["segments"], B, CB, bsize, _lib.BF16, axis, plan_options))
        # bsize 32 on feature axis 1: the barrier-free persistent kernel ('BSX4' plans, csrc/bsmm_xflow.h) for ungated calls; the staged
        # kernel's plans above stay for gated calls and as the comparison path (BlocksparseMatMul.flow = False).  A caller who names
        # another kernel family in plan_options gets exactly that.
        self.fprop_flow = self.bprop_flow = None
        # (bsize 64 runs on the bsize-32 kernels: its composite plan nests whichever bsize-32 plan the options name)
        if bsize in (32, 64) and axis == 1 and not (plan_options & (_lib.PLAN_XCOL_UNSTAGED | _lib.PLAN_XCOL_NARROW | _lib.PLAN_XCOL_FLOW | (7 << _lib.PLAN_XPROP_PH_SHIFT))):
            self.fprop_flow = plan(_host_plan(f["lut"], f["segments"], B, KB, bsize, _lib.BF16, axis, plan_options | _lib.PLAN_XCOL_FLOW))
            self.bprop_flow = plan(_host_plan(b["lut"], b["segments"], B, CB, bsize, _lib.BF16, axis, plan_options | _lib.PLAN_XCOL_FLOW))
        self.fprop_plan_f32 = self.bprop_plan_f32 = self.updat_plan = self.updat_plan_long = None
        if xprop_only:        # the doubled tables of gated calls (lut.double_tables): 16-bit fprop / bprop only
            return
        # ... and fp32 has its own (xprop-only) plan kernels for bsize 32; the schedule format is the library's business
        self.fprop_plan_f32 = plan(_host_plan(f["lut"], f["segments"], B, KB, bsize, _lib.F32, axis, plan_options))
        self.bprop_plan_f32 = plan(_host_plan(b["lut"], b["segments"], B, CB, bsize, _lib.F32, axis, plan_options))
        self.updat_plan = plan(_host_updat_plan(tables["updat_lut"], B, CB, KB, bsize, _lib.BF16, axis, plan_options))
        # round 6: 32 x 32-block windows for LONG minibatches on grids of >= 64 such windows at 3.7 .. 5 % density (BASELINE configs[3]: 91 against 100 us at
        # N = 4096; at N <= 2048 the 16 x 16 windows' direct stores win) -- a plan does not know the minibatch, BlocksparseMatMul.updat picks per call
        self.updat_plan_long = None
        w32 = (-(-CB // 32)) * (-(-KB // 32))
        if bsize == 32 and axis == 1 and not (plan_options & _lib.PLAN_WINDOW_MASK) and w32 >= 64 and 38 * w32 < B <= 52 * w32:
            self.updat_plan_long = plan(_host_updat_plan(tables["updat_lut"], B, CB, KB, bsize, _lib.BF16, axis, plan_options | _lib.PLAN_STREAM_32))


class BlocksparseMatMul(object):
    LONG_MINIBATCH = 3072     # rows x pairs from which the 32 x 32-window updat plan (where one was built) is used: between 2048 (loses) and 4096 (wins)

    def __getstate__(self):
        return (self.layout, self.bsize, self.axis, self.z_order, self.name, self.segmented, self.plan_options, self.updat_split)

    def __setstate__(self, state):
        self.__init__(*state)

    def __init__(self, layout, block_size=32, feature_axis=0, z_order=True, name=None, segmented=False, plan_options=0,
                 updat_split=0):
        """``plan_options``: BSMM_PLAN_* bits for the library's schedule builders (0 = its defaults); ``updat_split``: minibatch
        split of the windowed updat kernels (0 = the library chooses).  Both are tuning / test knobs, not semantics."""
        # the reference allows axis 0 with 8 / 16 / 32 and axis 1 with 32 / 64 (blocksparse/matmul.py:84-89); here 8 / 16 / 32 run
        # natively on both axes and 64 (axis 1) runs on the bsize-32 kernels: a 64x64 block is four 32x32 blocks
        if feature_axis not in (0, 1) or not (block_size in (8, 16, 32) or (block_size == 64 and feature_axis == 1)):
            raise ValueError("Unsupported block size with this feature axis")
        layout = np.asarray(layout)
        assert len(layout.shape) == 2
        self.axis = feature_axis
        self.bsize = block_size
        self.z_order = bool(z_order)
        self.segmented = bool(segmented)
        self.plan_options = int(plan_options)
        self.updat_split = int(updat_split)
        self.name = name if name is not None else "BlocksparseMatMul"

        ref = _lut.build_tables(layout, z_order=z_order, segmented=True)      # reference-policy tables
        self._ref_tables = ref
        self._dev_tables = ref if segmented else _lut.build_tables(layout, z_order=z_order, segmented=False)

        CB, KB = ref["CB"], ref["KB"]
        blocks = ref["blocks"]
        self.updat_lut = ref["updat_lut"]
        self.updat_list = [tuple(r) for r in ref["updat_lut"].tolist()]
        f, b = ref["fprop"], ref["bprop"]
        self.fprop_list, self.fprop_lut, self.l2_lut = f["cols"], f["lut"], f["l2_lut"]
        self.fprop_shared, self.l2_shared = f["shared"], f["l2_shared"]
        self.fprop_segments, self.fprop_locks = f["segments"], f["locks"]
        self.bprop_list, self.bprop_lut = b["cols"], b["lut"]
        self.bprop_shared, self.bprop_segments, self.bprop_locks = b["shared"], b["segments"], b["locks"]

        self.flops = blocks * block_size * block_size * 2
        self.blocks = blocks
        self.w_shape = (blocks, block_size, block_size)
        self.g_shape = (blocks,)
        self.count = 0
        self.CB, self.KB = CB, KB
        self.C, self.K = CB * block_size, KB * block_size
        self.sparsity = round(float(blocks) / float(CB * KB), 3)
        self.layout = ref["layout"]
        self._device_cache = {}
        self._workspaces = {}
        self._args_cache = {}
        self._prepared_w = {}             # op -> (weakref(w), (op, w.data_ptr, w._version, stream), buffer): bsmm_prepare_weights results
        self.cache_prepared = True        # False: prepare on every call (no per-weights cache at all)
        self._inner = None
        self._split64_hit = None
        self.native64 = True          # bsize 64: call the library with bsize = 64 (False: always the host-side quadrant view)
        self.flow = True              # bsize 32, feature axis 1, 16-bit, no gate: the barrier-free xprop kernel (False: the staged one)
        # round 6: gated 16-bit fprop / bprop calls on the fast UNGATED kernels over gated weight images (bsmm_gate_weights; _gated_xprop);
        # False: the GATED instantiations of the staged kernels (gate applied per fragment inside the kernel)
        self.gate_images = True
        self.gate_kind = "auto"       # "auto": look at a gate once per (tensor object, version); "binary" / "general": the caller's promise
        self._gate_kind_hit = None
        self._dbl = None
        self._xprop_only = False
        if block_size == 64:
            # same weights, cut into 32x32 blocks: inner block n is quadrant (i, j) of outer block b
            self._inner = BlocksparseMatMul(np.kron(self.layout, np.ones((2, 2), dtype=self.layout.dtype)), block_size=32, feature_axis=feature_axis,
                                            z_order=z_order, name=self.name + "/32", plan_options=plan_options, updat_split=updat_split)
            where = {ck: b for b, ck in enumerate(self.updat_list)}
            self._perm64 = np.array([where[(c // 2, k // 2)] * 4 + (c & 1) * 2 + (k & 1) for c, k in self._inner.updat_list], dtype=np.int64)
            self._inv64 = np.argsort(self._perm64)
            self._perm64_dev = {}

    # ---- bsize 64 on the bsize-32 kernels --------------------------------------------------------
    def _idx64(self, device):
        key = (device.type, device.index)
        if key not in self._perm64_dev:
            self._perm64_dev[key] = (torch.from_numpy(self._perm64).to(device), torch.from_numpy(self._inv64).to(device))
        return self._perm64_dev[key]

    def _split64(self, w):
        perm, _ = self._idx64(w.device)
        return w.contiguous().view(self.blocks, 2, 32, 2, 32).permute(0, 1, 3, 2, 4).reshape(4 * self.blocks, 32, 32).index_select(0, perm)

    @staticmethod
    def _same_weights(entry, w, key):
        """A cached per-weights image is valid only for the SAME tensor object (weak reference still alive and identical -- a new
        tensor that the allocator put at the old address is a different object), at the same version, with the same key.
        Consequence: the cache is PER OBJECT -- pass the parameter itself.  A fresh view of it on every call (``w.detach()``,
        ``w.data``, an AMP cast, a non-contiguous ``w`` that fprop makes contiguous) never hits, and the per-weights preparation (fp32 pieces,
        bsize-64 quadrant gather) then runs in every fprop / bprop; the results are the same, only slower."""
        return entry is not None and entry[0]() is w and entry[1] == key

    def invalidate_weights(self):
        """Forget every per-weights image (prepared pieces, quadrant views).  Needed only after a mutation PyTorch's version counter
        cannot see (``w.data.add_()``, a raw-pointer write, another library writing into the storage)."""
        self._prepared_w.clear()
        self._split64_hit = None
        if self._inner is not None:
            self._inner.invalidate_weights()

    def _split64_cached(self, w):
        """The quadrant view of a weight tensor for fprop / bprop, made once per (tensor object, version): W is constant across the
        calls of a pass; an in-place update (``w._version``) or ANOTHER tensor (even at the same address: the entry holds a weak
        reference and compares identity) invalidates it.  Not used under autograd (the gather is part of the graph there)."""
        if w.requires_grad and torch.is_grad_enabled():
            return self._split64(w)
        key = (w.data_ptr(), w._version, w.dtype, w.device)
        if not self._same_weights(self._split64_hit, w, key):
            self._split64_hit = (weakref.ref(w), key, self._split64(w))
        return self._split64_hit[2]

    def _merge64(self, w32):
        _, inv = self._idx64(w32.device)
        return w32.index_select(0, inv).view(self.blocks, 2, 2, 32, 32).permute(0, 1, 3, 2, 4).reshape(self.blocks, 64, 64)

    def _gate64(self, gate):
        if gate is None:
            return None
        perm, _ = self._idx64(gate.device)
        return gate.index_select(0, perm // 4)

    # ---- shapes ----------------------------------------------------------------------------------
    def i_shape(self, N):
        return (N, self.C) if self.axis else (self.C, N)

    def o_shape(self, N):
        return (N, self.K) if self.axis else (self.K, N)

    def block_coord(self, block):
        return self.updat_list[block]

    # ---- device plumbing -------------------------------------------------------------------------
    def _tables_on(self, device):
        key = (device.type, device.index)
        t = self._device_cache.get(key)
        if t is None:
            t = _DeviceTables(self._dev_tables, device, self.bsize, self.axis, self.plan_options, xprop_only=self._xprop_only)
            self._device_cache[key] = t
        return t

    def _check_tensor(self, t, what):
        if torch is None:
            raise RuntimeError("blocksparse_amd needs PyTorch-ROCm for device memory")
        if not isinstance(t, torch.Tensor) or t.device.type != "cuda":
            raise RuntimeError("blocksparse_amd: %s must be a tensor on a ROCm device (no CPU fallback)" % what)

    def _n_of(self, x, feat):
        if self.axis == 0:
            if x.shape[0] != feat:
                raise ValueError("expected %d features on axis 0, got shape %s" % (feat, tuple(x.shape)))
            return int(x.numel() // feat)
        if x.shape[-1] != feat:
            raise ValueError("expected %d features on the last axis, got shape %s" % (feat, tuple(x.shape)))
        return int(x.numel() // feat)

    def _args(self, lut_t, side, N, Cin, Kout, dtype, pcount=1, alpha=1.0, beta=0.0, plan=None):
        a = _lib.BsmmArgs()
        a.lut = lut_t.data_ptr()
        if plan is not None:
            plan.attach(a)
        a.gate = None
        a.workspace, a.workspace_bytes = None, 0
        a.flags = _lib.call_flags()
        a.split = self.updat_split
        a.trace = ctypes.pointer(_lib._last_kernel)
        a.blocks, a.bsize = self.blocks, self.bsize
        if side is not None:
            a.segments, a.locks, a.shared = side["segments"], side["locks"], side["shared"]
        a.C, a.K, a.N = Cin, Kout, N
        a.pcount, a.axis, a.dtype = pcount, self.axis, _dtype_code(dtype)
        a.alpha, a.beta = alpha, beta
        a.stream = _lib.raw_stream(lut_t.device)
        return a

    def _call_args(self, op, tabs, lut_t, side, N, Cin, Kout, dtype, plan, slot=0, pcount=1, flags=0, gated=False):
        """The argument block of a call and its workspace, made once per (op, minibatch, dtype, stream, plan, ...) and reused: filling the
        ctypes struct, attaching the plan and asking the library for the workspace size cost more host time than a small-minibatch
        kernel takes on the device (profiles/r04_smalln.txt: 14-15 us per eager call before this cache).  Fields that change from call to
        call (gate, alpha / beta, prepared_w) are set by the caller -- on a COPY of the cached block (one mutable struct shared by
        every call of a key is not safe from two threads, and a gate-dependent workspace term needs the gate in the key: ``gated``)."""
        stream = _lib.raw_stream(lut_t.device)
        key = (op, N, dtype, lut_t.device.index, stream, id(plan), slot, pcount, _lib.call_flags() | flags, self.updat_split, bool(gated))
        hit = self._args_cache.get(key)
        if hit is None:
            a = self._args(lut_t, side, N, Cin, Kout, dtype, pcount=pcount, plan=plan)
            a.flags |= flags
            a.gate = 16 if gated else None          # (sizing only: the library asks whether there is a gate, it does not read it here)
            need_prep = bool(op != _lib.OP_UPDAT and _lib.load().bsmm_prepared_bytes(op, ctypes.byref(a)))
            need_ws = 0 if need_prep else int(_lib.load().bsmm_workspace_bytes(op, ctypes.byref(a)))   # (prepared calls size theirs after prepared_w is set)
            hit = (a, need_ws, need_prep)
            if len(self._args_cache) > 256:
                self._args_cache.clear()
            self._args_cache[key] = hit
        cached, need_ws, need_prep = hit
        a = _lib.BsmmArgs.from_buffer_copy(cached)
        a.gate = None
        a.alpha, a.beta = 1.0, 0.0
        ws = None
        if need_ws:
            # the scratch is shared by every call of this op on this stream (grown on demand): point the cached block at the current one
            wkey = (lut_t.device.index, stream, op, slot)
            ws = self._workspaces.get(wkey)
            if ws is None or ws.numel() < need_ws:
                ws = torch.empty(max(need_ws, 16), dtype=torch.uint8, device=lut_t.device)
                self._workspaces[wkey] = ws
            a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel()
        return a, ws, need_prep

    def _workspace(self, a, op, device, slot=0):
        """Device scratch for one call, kept per (device, stream, op) and grown on demand: the entry points never allocate, and
        consecutive calls on a stream reuse the same bytes in stream order (nothing inside the timed region of a step)."""
        need = _lib.load().bsmm_workspace_bytes(op, ctypes.byref(a))
        if not need:
            return None
        key = (device.index, a.stream, op, slot)
        ws = self._workspaces.get(key)
        if ws is None or ws.numel() < need:
            ws = torch.empty(max(need, 16), dtype=torch.uint8, device=device)
            self._workspaces[key] = ws
        a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel()
        return ws

    def _prepared(self, a, op, w):
        """The per-weights preparation of the library (bsmm_prepare_weights: fp32 / bsize 32 with a plan = the bf16 pieces of W), made
        once per (op, tensor OBJECT, version) and handed to the call in ``a.prepared_w`` -- W is constant across the calls of a
        pass; an in-place update (optimizer step: ``w._version`` changes) or any other tensor invalidates it.  The entry keeps a weak
        reference to ``w`` and compares identity: a temporary (``w_master.to(bf16)``, ``W * mask``, an ``l2_normalize`` output) that the
        allocator places at a freed tensor's address with the same version is a different object and is prepared again.
        A mutation the version counter cannot see (``w.data.add_()``) needs ``invalidate_weights()``; ``cache_prepared = False``
        prepares on every call."""
        lib = _lib.load()
        need = lib.bsmm_prepared_bytes(op, ctypes.byref(a))
        if not need:
            return
        key = (op, w.data_ptr(), w._version, a.stream)
        hit = self._prepared_w.get(op)
        if not (self.cache_prepared and self._same_weights(hit, w, key)):
            buf = hit[2] if (hit is not None and hit[2].numel() >= need) else torch.empty(need, dtype=torch.uint8, device=w.device)
            _lib.check(lib.bsmm_prepare_weights(op, w.data_ptr(), buf.data_ptr(), ctypes.byref(a)), "bsmm_prepare_weights")
            hit = self._prepared_w[op] = (weakref.ref(w), key, buf)
        a.prepared_w = hit[2].data_ptr()

    def _out_shape(self, x, feat_out):
        shp = list(x.shape)
        if self.axis == 0:
            shp[0] = feat_out
        else:
            shp[-1] = feat_out
        return shp

    def _check_gate(self, gate, device):
        if gate is None:
            return None
        if not (isinstance(gate, torch.Tensor) and gate.is_cuda and gate.dtype == torch.float32 and gate.numel() == self.blocks):
            raise ValueError("gate: expected a float32 CUDA tensor with one entry per block (%d)" % self.blocks)
        if gate.device != device:
            raise ValueError("gate lives on another device")
        return gate.contiguous()

    # ---- gated calls on the ungated kernels (round 6) ----------------------------------------------
    # minibatch from which a gated call takes the weight images (scripts/gpu_gated_smalln.py: below, the per-segment kernel with its gate
    # multiply wins -- the ungated call would run that kernel too, and the image pass is a second launch): bsize 32: 1024, bsize 16: 2048
    GATE_IMAGES_MIN_N = {32: 1024, 16: 2048}

    def _gate_kind_of(self, gate):
        """"binary" (every gate 0 or 1: a pruning mask -- ONE exact weight image) or "general" (two images, g w to ~2^-17).  Looked at once
        per (tensor object, version): one host sync the first time a gate (or a changed gate) is seen, none afterwards; under stream capture
        an unseen gate counts as "general" (always correct).  ``gate_kind = "binary" / "general"`` is the caller's promise instead."""
        if self.gate_kind != "auto":
            return self.gate_kind
        key = (gate.data_ptr(), gate._version)
        hit = self._gate_kind_hit
        if hit is not None and hit[0]() is gate and hit[1] == key:
            return hit[2]
        if torch.cuda.is_current_stream_capturing():
            return "general"
        kind = "binary" if bool(((gate == 0) | (gate == 1)).all().item()) else "general"
        self._gate_kind_hit = (weakref.ref(gate), key, kind)
        return kind

    def _doubled(self):
        """This operator over lut.double_tables(): 2 x blocks weight images [hi ; lo], fprop / bprop only, never gated itself."""
        d = self._dbl
        if d is None:
            d = object.__new__(BlocksparseMatMul)
            d.axis, d.bsize, d.z_order, d.segmented = self.axis, self.bsize, self.z_order, self.segmented
            d.plan_options, d.updat_split, d.name = self.plan_options, self.updat_split, self.name + "/gated2"
            d._dev_tables = d._ref_tables = _lut.double_tables(self._dev_tables)
            d.blocks = 2 * self.blocks
            d.w_shape, d.g_shape = (d.blocks, self.bsize, self.bsize), (d.blocks,)
            d.CB, d.KB, d.C, d.K = self.CB, self.KB, self.C, self.K
            d._device_cache, d._workspaces, d._args_cache, d._prepared_w = {}, {}, {}, {}
            d.cache_prepared, d._inner, d._split64_hit, d.native64 = True, None, None, True
            d.gate_images, d.gate_kind, d._gate_kind_hit, d._dbl, d._xprop_only = False, "general", None, None, True
            self._dbl = d
        d.flow = self.flow
        return d

    def _gated_xprop(self, which, x, w, gate, N):
        """A gated 16-bit fprop / bprop as [bsmm_gate_weights -> the UNGATED call]: the flow / list kernels carry no gate logic (their
        registers are spoken for), and forming g w once per call (6.7 MB at the bench shape, ~3 us) replaces ~80 vector instructions per block
        and row tile inside the staged GATED kernels.  0 / 1 gates: one image, the plain tables (exact); other gates: bf16 two images over the
        doubled tables, fp16 one image (the reference's rounding).  None = not taken (fp32, bsize 8 / 64, locked tables, short minibatches): the caller runs the GATED kernels."""
        if not (self.gate_images and gate is not None and x.dtype in (torch.float16, torch.bfloat16) and self.bsize in (16, 32)
                and self._inner is None and N >= self.GATE_IMAGES_MIN_N[self.bsize]
                and self._dev_tables["fprop"]["locks"] == 0 and self._dev_tables["bprop"]["locks"] == 0):
            return None
        # fp16: ONE image for any gate -- round(g w) to fp16 is the reference's own arithmetic (mul.rn.f16x2 on the weight fragments,
        # src/blocksparse_hgemm_cn_64_op_gpu.cu:104-110) and costs ~3e-4 against the float64 product; a bf16 rounding of g w would cost
        # ~2e-3 (above the 1e-3 bar), hence the second image there unless the gate is a 0 / 1 mask
        pieces = 1 if (w.dtype == torch.float16 or self._gate_kind_of(gate) == "binary") else 2
        img = torch.empty((pieces * self.blocks, self.bsize, self.bsize), dtype=w.dtype, device=w.device)
        stream = _lib.raw_stream(w.device)
        _lib.check(_lib.load().bsmm_gate_weights(w.data_ptr(), gate.data_ptr(), img.data_ptr(), self.blocks, self.bsize, _dtype_code(w.dtype), pieces, stream),
                   "bsmm_gate_weights")
        op = self if pieces == 1 else self._doubled()
        return op.fprop(x, img) if which == "fprop" else op.bprop(x, img)

    # ---- the three passes ------------------------------------------------------------------------
    def _xprop_plan(self, tabs, which, N, n_out_features, dtype, gate):
        """The schedule an fprop / bprop call runs with: fp32 -> its own plans; 16-bit, ungated, feature axis 1, bsize 32 -> the flow
        kernel (which switches to 64-row units itself when 128-row units do not fill the chip);
        gated calls and everything else -> the staged / per-bsize plans."""
        if dtype == torch.float32:
            return getattr(tabs, which + "_plan_f32")
        if gate is None:
            flow = getattr(tabs, which + "_flow")
            if self.flow and flow is not None:
                return flow
        return getattr(tabs, which + "_plan")

    def fprop(self, x, w, gate=None):
        """Y = fprop(X, W): axis 0 Y(K,N) = Wd^T X, axis 1 Y(N,K) = X Wd (op BlocksparseMatmul).
        ``gate`` (float32 [blocks]): block w contributes gate[w] times its product, gate 0 = skipped."""
        self._check_tensor(x, "x"); self._check_tensor(w, "w")
        gate = self._check_gate(gate, x.device)
        if x.dtype != w.dtype:
            raise TypeError("x and w must have the same dtype")
        if self._inner is not None and not self.native64:
            return self._inner.fprop(x, self._split64_cached(w), gate=self._gate64(gate))
        x = x.contiguous(); w = w.contiguous()
        N = self._n_of(x, self.C)
        if gate is not None:
            y = self._gated_xprop("fprop", x, w, gate, N)
            if y is not None:
                return y
        lib = _lib.load()
        tabs = self._tables_on(x.device)
        y = torch.empty(self._out_shape(x, self.K), dtype=x.dtype, device=x.device)
        plan = self._xprop_plan(tabs, "fprop", N, self.K, x.dtype, gate)
        a, _, need_prep = self._call_args(_lib.OP_FPROP, tabs, tabs.fprop, self._dev_tables["fprop"], N, self.C, self.K, x.dtype, plan, gated=gate is not None)
        a.gate = gate.data_ptr() if gate is not None else None
        if need_prep:
            a.prepared_w = None
            self._prepared(a, _lib.OP_FPROP, w)
            self._workspace(a, _lib.OP_FPROP, x.device)
        _lib.check(lib.bsmm_fprop(x.data_ptr(), w.data_ptr(), y.data_ptr(), ctypes.byref(a)), "bsmm_fprop")
        return y

    def bprop(self, dy, w, gate=None):
        """DX = bprop(DY, W) (op BlocksparseMatmulDX; C and K swapped as in matmul.py:506-510); ``gate`` as in fprop."""
        self._check_tensor(dy, "dy"); self._check_tensor(w, "w")
        gate = self._check_gate(gate, dy.device)
        if dy.dtype != w.dtype:
            raise TypeError("dy and w must have the same dtype")
        if self._inner is not None and not self.native64:
            return self._inner.bprop(dy, self._split64_cached(w), gate=self._gate64(gate))
        dy = dy.contiguous(); w = w.contiguous()
        N = self._n_of(dy, self.K)
        if gate is not None:
            dx = self._gated_xprop("bprop", dy, w, gate, N)
            if dx is not None:
                return dx
        lib = _lib.load()
        tabs = self._tables_on(dy.device)
        dx = torch.empty(self._out_shape(dy, self.C), dtype=dy.dtype, device=dy.device)
        plan = self._xprop_plan(tabs, "bprop", N, self.C, dy.dtype, gate)
        a, _, need_prep = self._call_args(_lib.OP_BPROP, tabs, tabs.bprop, self._dev_tables["bprop"], N, self.K, self.C, dy.dtype, plan, gated=gate is not None)
        a.gate = gate.data_ptr() if gate is not None else None
        if need_prep:
            a.prepared_w = None
            self._prepared(a, _lib.OP_BPROP, w)
            self._workspace(a, _lib.OP_BPROP, dy.device)
        _lib.check(lib.bsmm_bprop(dy.data_ptr(), w.data_ptr(), dx.data_ptr(), ctypes.byref(a)), "bsmm_bprop")
        return dx

    def updat(self, xs, dys, alpha=1.0, beta=0.0, dw=None, gate=None, sums_only=False, slot=0):
        """DW = alpha * sum_p updat(X_p, DY_p) + beta * DW  (ops BlocksparseMatmulDW / ...DWA).

        ``xs``/``dys``: one tensor each or equally long lists of up to 8 tensors (the reference's Plist).
        ``gate``: gated dw (op attr gated_dw): the sum of block w is scaled by gate[w].
        ``sums_only``: return the raw fp32 sums [blocks, bs, bs] (a view of the call's workspace, valid until the next updat on
        this stream with the same ``slot``) instead of DW -- the data-parallel path reduces them over the ranks in fp32
        (``dist.DwReduce``) or calls ``updat_finalize``; only the streaming kernel (bsize 32: 16-bit types on either feature axis, fp32 on
        feature axis 1) can, other configurations raise BsmmError(-2).  ``gate`` is NOT applied to the sums: pass it to the finalize step.
        ``slot``: which of the op's workspaces to use -- alternate 0 / 1 when the sums of one step are still being reduced while
        the next step's updat runs."""
        if isinstance(xs, torch.Tensor):
            xs, dys = [xs], [dys]
        if len(xs) != len(dys) or not 1 <= len(xs) <= 8:
            raise ValueError("updat takes 1..8 (x, dy) pairs")
        for t in list(xs) + list(dys):
            self._check_tensor(t, "x/dy")
            if t.dtype != xs[0].dtype:
                raise TypeError("all x/dy tensors must share one dtype")
        xs = [t.contiguous() for t in xs]
        dys = [t.contiguous() for t in dys]
        N = self._n_of(xs[0], self.C)
        for x, dy in zip(xs, dys):
            if self._n_of(x, self.C) != N or self._n_of(dy, self.K) != N:
                raise ValueError("all pairs must share the minibatch size")
        # bsize 64: the library takes it for 16-bit types (bsmm_args.bsize = 64 with a 'BS64' plan: quadrant sums + one finalize pass);
        # fp32 and the raw-sums form go through the host-side quadrant view
        native64 = self._inner is not None and self.native64 and xs[0].dtype != torch.float32 and not sums_only
        if native64:
            # what the library's composite bsize-64 updat cannot take runs on the host-side quadrant view instead of raising
            # (ADVICE r3): operands beyond the streaming kernel's 32-bit offsets, unaligned operands, plans forced to another
            # kernel family, the NO_PLAN / FORCE_VALU test variants
            native64 = ((N * max(self.C, self.K) < (1 << 30)) and all(t.data_ptr() % 16 == 0 for t in xs + dys)
                        and not (_lib.call_flags() & (_lib.FLAG_NO_PLAN | _lib.FLAG_FORCE_VALU))
                        and (self.plan_options & 0xf0) in (0, _lib.PLAN_STREAM_16, _lib.PLAN_STREAM_8, _lib.PLAN_STREAM_32))
        if self._inner is not None and not native64:
            if sums_only:                # the fp32 sums of the quadrants (a view of the inner call's workspace), put together as a copy
                if gate is not None:
                    raise ValueError("updat(sums_only=True) returns the ungated sums: pass the gate to updat_finalize / DwReduce.start")
                return self._merge64(self._inner.updat(xs, dys, sums_only=True, slot=slot))
            if beta != 0.0 and dw is None:
                raise ValueError("beta != 0 needs dw")
            dw32 = self._inner.updat(xs, dys, alpha=alpha, beta=beta, dw=self._split64(dw) if (dw is not None and beta != 0.0) else None,
                                     gate=self._gate64(self._check_gate(gate, xs[0].device)))
            out = self._merge64(dw32)
            if dw is not None:
                dw.copy_(out)
                return dw
            return out
        lib = _lib.load()
        dev = xs[0].device
        tabs = self._tables_on(dev)
        if sums_only:
            dw = None
        elif dw is None:
            if beta != 0.0:
                raise ValueError("beta != 0 needs dw")
            dw = torch.empty(self.w_shape, dtype=xs[0].dtype, device=dev)
        else:
            self._check_tensor(dw, "dw")
            if tuple(dw.shape) != self.w_shape or dw.dtype != xs[0].dtype or not dw.is_contiguous():
                raise ValueError("dw must be a contiguous %s tensor of dtype %s" % (self.w_shape, xs[0].dtype))
        gate = self._check_gate(gate, dev)
        if sums_only and gate is not None:
            raise ValueError("updat(sums_only=True) returns the ungated sums: pass the gate to updat_finalize / DwReduce.start")
        flags = (_lib.FLAG_GATED_DW if (gate is not None and not sums_only) else 0) | (_lib.FLAG_DW_SUMS if sums_only else 0)
        # fp32: the library has a plan path for bsize 32 on feature axis 1 and for bsize 8 / 16 (the six significant bf16 piece products as six
        # pairs of one launch of the 16-bit kernel, round 4; bsize 16 on feature axis 0: round 5, where the row-owner kernel pays -- the library
        # falls back to the kernels without a plan by itself); other fp32 configurations run the kernels without a plan
        use_plan = xs[0].dtype != torch.float32 or (len(xs) == 1 and ((self.bsize == 32 and self.axis == 1) or self.bsize in (8, 16)))
        uplan = tabs.updat_plan
        if tabs.updat_plan_long is not None and N * len(xs) >= self.LONG_MINIBATCH and xs[0].dtype != torch.float32:
            uplan = tabs.updat_plan_long
        a, ws, _ = self._call_args(_lib.OP_UPDAT, tabs, tabs.updat, None, N, self.C, self.K, xs[0].dtype,
                                   uplan if use_plan else None, slot=slot, pcount=len(xs), flags=flags,
                                   gated=gate is not None and not sums_only)
        a.alpha, a.beta = alpha, beta
        if gate is not None and not sums_only:
            a.gate = gate.data_ptr()
        arr = ctypes.c_void_p * len(xs)
        xp = arr(*[t.data_ptr() for t in xs])
        ep = arr(*[t.data_ptr() for t in dys])
        rc = lib.bsmm_updat(xp, ep, dw.data_ptr() if dw is not None else None, ctypes.byref(a))
        if rc == -2 and self._inner is not None and not sums_only:      # BSMM_ERR_UNSUPPORTED from the composite path: quadrant view
            dw32 = self._inner.updat(xs, dys, alpha=alpha, beta=beta, dw=self._split64(dw) if beta != 0.0 else None, gate=self._gate64(gate))
            dw.copy_(self._merge64(dw32))
            return dw
        _lib.check(rc, "bsmm_updat")
        if sums_only:
            n = self.blocks * self.bsize * self.bsize
            return ws[:4 * n].view(torch.float32).view(self.w_shape)
        return dw

    def updat_finalize(self, sums, alpha=1.0, beta=0.0, dw=None, gate=None, dtype=None):
        """DW = alpha * [gate *] sums + beta * DW, rounded once: the second half of ``updat(..., sums_only=True)``."""
        self._check_tensor(sums, "sums")
        if sums.dtype != torch.float32 or tuple(sums.shape) != self.w_shape or not sums.is_contiguous():
            raise ValueError("sums must be the contiguous float32 %s tensor updat(sums_only=True) returned" % (self.w_shape,))
        if dw is None:
            if beta != 0.0:
                raise ValueError("beta != 0 needs dw")
            dw = torch.empty(self.w_shape, dtype=dtype or torch.bfloat16, device=sums.device)
        gate = self._check_gate(gate, sums.device)
        st = _lib.raw_stream(sums.device)
        blocks, bsize = self.blocks, self.bsize
        if bsize == 64:      # elementwise: a 64x64 block is four contiguous quarters of 1024 elements with the block's gate
            blocks, bsize = 4 * blocks, 32
            gate = gate.repeat_interleave(4) if gate is not None else None
        _lib.check(_lib.load().bsmm_updat_finalize(sums.data_ptr(), dw.data_ptr(), gate.data_ptr() if gate is not None else None, blocks,
                                                   bsize, _dtype_code(dw.dtype), alpha, beta, st), "bsmm_updat_finalize")
        return dw

    def updat_grouped(self, xs, dys, group_size=8, dw=None, alpha=1.0):
        """dw (+)= alpha * sum over ALL (x, dy) pairs, issued as one DW launch followed by chained DWA launches of up
        to ``group_size`` (<= 8) pairs each -- what the reference's ``group_param_grads`` graph rewrite produces for
        weight-tied / recurrent uses (blocksparse/matmul.py:612-731).  ``dw`` given: accumulate into it (beta = 1)."""
        if not 1 <= group_size <= 8:
            raise ValueError("group_size must be in 1..8")
        if len(xs) != len(dys) or len(xs) == 0:
            raise ValueError("need equally many x and dy tensors")
        beta = 0.0 if dw is None else 1.0
        for i in range(0, len(xs), group_size):
            dw = self.updat(list(xs[i:i + group_size]), list(dys[i:i + group_size]), alpha=alpha, beta=beta, dw=dw)
            beta = 1.0
        return dw

    # ---- operator interface (matmul.py:455-483) ---------------------------------------------------
    def __call__(self, I, W, gate=None, gate_grad=False, dw_gated=False, name=None, bench=0):
        """y = bsmm(x, w[, gate]).  With a gate: blocks are scaled by it in fprop / bprop; ``dw_gated`` scales dw by the
        gate as well; ``gate_grad`` returns dg = sum(dw * w) per block and gates dw (blocksparse_matmul_grad,
        blocksparse/matmul.py:485-527)."""
        self.count += 1
        if gate is None:
            return _BsmmFunction.apply(I, W, self)
        return _BsmmGatedFunction.apply(I, W, gate, self, bool(gate_grad), bool(dw_gated))

    def gate_grad(self, dw, w, gate):
        """(dw * gate, dg) with dg[b] = sum(dw[b] * w[b])  (op BlocksparseMatmulDG)."""
        self._check_tensor(dw, "dw"); self._check_tensor(w, "w")
        gate = self._check_gate(gate, dw.device)
        if dw.dtype != w.dtype or tuple(dw.shape) != self.w_shape or tuple(w.shape) != self.w_shape:
            raise ValueError("dw and w must be %s tensors of one dtype" % (self.w_shape,))
        if self._inner is not None:      # bsize 64: per quadrant, dg summed over the four quadrants of a block
            o32, dg32 = self._inner.gate_grad(self._split64(dw), self._split64(w), self._gate64(gate))
            perm, _ = self._idx64(dw.device)
            dg = torch.zeros(self.blocks, dtype=torch.float32, device=dw.device).index_add_(0, perm // 4, dg32)
            return self._merge64(o32), dg
        dw = dw.contiguous(); w = w.contiguous()
        out = torch.empty_like(dw)
        dg = torch.empty(self.blocks, dtype=torch.float32, device=dw.device)
        st = _lib.raw_stream(dw.device)
        _lib.check(_lib.load().bsmm_gate_grad(out.data_ptr(), dg.data_ptr(), dw.data_ptr(), w.data_ptr(), gate.data_ptr(), self.blocks,
                                              self.bsize, _dtype_code(dw.dtype), st), "bsmm_gate_grad")
        return out, dg

    # ---- block-sparse L2 weight norm (blocksparse/matmul.py:421-453) ------------------------------
    def l2_normalize(self, W, gain=None, epsilon=1e-12, dtype=None):
        """y = gain * W / sqrt(max(sum of W^2 over each output feature, epsilon)) on the device, differentiable
        (ops L2NormalizeCK / L2NormalizeGainCK and their registered gradients, blocksparse/matmul.py:447-453,529-553)."""
        if self._inner is not None:      # bsize 64: the quadrant view keeps every output feature's column
            return self._merge64(self._inner.l2_normalize(self._split64(W), gain=gain, epsilon=epsilon, dtype=dtype))
        return _L2NormFunction.apply(W, gain, self, float(epsilon), dtype or W.dtype)

    def _l2_tables(self, device):
        key = ("l2", str(device))
        t = self._l2_dev.get(key) if hasattr(self, "_l2_dev") else None
        if t is None:
            if not hasattr(self, "_l2_dev"):
                self._l2_dev = {}
            t = torch.from_numpy(np.ascontiguousarray(self.l2_lut)).to(device)
            self._l2_dev[key] = t
        return t

    def _l2_fwd(self, W, gain, epsilon, y_dtype):
        self._check_tensor(W, "W")
        if tuple(W.shape) != self.w_shape:
            raise ValueError("W must have shape %s" % (self.w_shape,))
        W = W.contiguous()
        if gain is not None:
            if not (gain.is_cuda and gain.dtype == torch.float32 and gain.numel() == self.K):
                raise ValueError("gain: expected a float32 CUDA tensor with K = %d entries" % self.K)
            gain = gain.contiguous()
        lut = self._l2_tables(W.device)
        y = torch.empty(self.w_shape, dtype=y_dtype, device=W.device)
        ss = torch.empty(self.K, dtype=torch.float32, device=W.device)
        st = _lib.raw_stream(W.device)
        _lib.check(_lib.load().bsmm_l2_normalize(y.data_ptr(), ss.data_ptr(), W.data_ptr(), gain.data_ptr() if gain is not None else None,
                                                 lut.data_ptr(), self.KB, self.bsize, _dtype_code(W.dtype), _dtype_code(y_dtype), epsilon, st),
                   "bsmm_l2_normalize")
        return y, ss

    def _l2_bwd(self, dy, W, gain, ss, epsilon):
        dy = dy.contiguous()
        lut = self._l2_tables(W.device)
        dx = torch.empty_like(W)
        dg = torch.empty(self.K, dtype=torch.float32, device=W.device) if gain is not None else None
        st = _lib.raw_stream(W.device)
        _lib.check(_lib.load().bsmm_l2_normalize_grad(dx.data_ptr(), dg.data_ptr() if dg is not None else None, dy.data_ptr(), W.data_ptr(),
                                                      gain.data_ptr() if gain is not None else None, ss.data_ptr(), lut.data_ptr(), self.KB,
                                                      self.bsize, _dtype_code(W.dtype), _dtype_code(dy.dtype), epsilon, st),
                   "bsmm_l2_normalize_grad")
        return dx, dg

    def l2_normalize_test(self, W, epsilon=1e-12):
        W = np.array(W, dtype=np.float64)
        for k, col in self.fprop_list:
            ws = [w for _, w in col]
            if ws:
                W2 = W[ws].reshape(-1, self.bsize)
                W[ws] = W[ws] / np.sqrt(np.maximum(np.square(W2).sum(axis=0, keepdims=True), epsilon))
        return W

    def l2_normalize_grad_test(self, W, U, epsilon=1e-12):
        W = np.asarray(W, dtype=np.float64)
        U = np.array(U, dtype=np.float64)
        for k, col in self.fprop_list:
            ws = [w for _, w in col]
            if ws:
                W2, U2 = W[ws].reshape(-1, self.bsize), U[ws].reshape(-1, self.bsize)
                ss = np.square(W2).sum(axis=0, keepdims=True)
                mx = np.maximum(ss, epsilon)
                g = (U2 + W2 * (ss >= epsilon) * (-U2 * W2 / mx).sum(axis=0, keepdims=True)) / np.sqrt(mx)
                U[ws] = g.reshape(-1, self.bsize, self.bsize)
        return U

    def prune(self, param, gate):
        """Drop the blocks whose gate is 0: returns (new_param, new_gate) and clears those blocks in ``self.layout``
        (blocksparse/matmul.py:272-290; as there, build a new BlocksparseMatMul from the pruned layout afterwards)."""
        is_t = torch is not None and isinstance(gate, torch.Tensor)
        gate_np = gate.detach().cpu().numpy() if is_t else np.asarray(gate)
        keep = gate_np != 0.0
        if int(keep.sum()) != self.blocks:
            for w_id, (c, k) in enumerate(self.updat_list):
                if not keep[w_id]:
                    self.layout[c, k] = 0
            if torch is not None and isinstance(param, torch.Tensor):     # device tensors stay on their device
                param = param[torch.from_numpy(keep).to(param.device)]
            else:
                param = np.asarray(param)[keep]
        n = int(keep.sum())
        new_gate = torch.ones(n, dtype=gate.dtype, device=gate.device) if is_t else np.ones((n,), dtype=gate_np.dtype)
        return param, new_gate

    def matmul(self, I, W, gate=None, gate_grad=False, dw_gated=False, name=None, bench=0):
        return self.__call__(I, W, gate=gate, gate_grad=gate_grad, dw_gated=dw_gated, name=name, bench=bench)

    # ---- initialisers ----------------------------------------------------------------------------
    def identity_init(self, scale=1.0):
        """Returns ``init(shape=None, dtype=torch.float32, device="cuda") -> W`` computed on the device
        (op BlocksparseMatmulIdentityInit, matmul.py:55-72,321-323)."""
        def _initializer(shape=None, dtype=None, device="cuda"):
            dtype = dtype or torch.float32
            if shape is not None:
                assert tuple(shape) == self.w_shape
            if self._inner is not None:  # bsize 64: the identity of a diagonal block is the identity of its two diagonal quadrants
                return self._merge64(self._inner.identity_init(scale)(dtype=dtype, device=device))
            dev = torch.device(device)
            if dev.type != "cuda":
                raise RuntimeError("blocksparse_amd: identity_init runs on a ROCm device only")
            if dev.index is None:
                dev = torch.device("cuda", torch.cuda.current_device())
            lib = _lib.load()
            tabs = self._tables_on(dev)
            W = torch.empty(self.w_shape, dtype=dtype, device=dev)
            st = _lib.raw_stream(dev)
            _lib.check(lib.bsmm_identity_init(W.data_ptr(), tabs.updat.data_ptr(), self.CB, self.KB, self.blocks,
                                              self.bsize, float(scale), _dtype_code(dtype), st), "bsmm_identity_init")
            return W
        return _initializer

    def ortho_init(self):
        """NumPy initializer (host), same construction as matmul.py:291-319."""
        def _initializer(shape=None, dtype=np.float32, partition_info=None):
            W = np.empty(self.w_shape, dtype=dtype)
            bs = self.bsize
            if self.sparsity < 1.0:
                for k, col in self.fprop_list:
                    if not col:
                        continue
                    shp = (len(col) * bs, bs)
                    a = np.random.normal(0.0, 1.0, shp).astype(dtype)
                    u, _, v = np.linalg.svd(a, full_matrices=False)
                    if u.shape != shp:
                        u = v
                    for i, (c, w) in enumerate(col):
                        W[w, :, :] = u[i * bs:(i + 1) * bs, :]
            else:
                shp = (self.C, self.K)
                a = np.random.normal(0.0, 1.0, shp).astype(dtype)
                u, _, v = np.linalg.svd(a, full_matrices=False)
                if u.shape != shp:
                    u = v
                for w, (c, k) in enumerate(self.updat_list):
                    W[w, :, :] = u[c * bs:(c + 1) * bs, k * bs:(k + 1) * bs]
            return W
        return _initializer

    def checker_init(self):
        def _initializer(shape=None, dtype=np.float32, partition_info=None):
            ul = self.updat_lut
            return (((ul[:, 0] & 1) ^ (ul[:, 1] & 1)) ^ 1).astype(dtype)
        return _initializer

    # ---- NumPy reference functions shipped with the class (matmul.py:353-419) ----------------------
    # Host-side API parity only (the reference exposes them as methods); the device path never calls them.
    def fprop_test(self, I, W, gate=None):
        bs = self.bsize
        I = np.asarray(I); W = np.asarray(W)
        if gate is not None:          # the gate scales the block product (matmul.py:367-373); both axes here
            W = W * np.asarray(gate, dtype=W.dtype)[:, None, None]
        if self.axis:
            n = I.shape[0]
            X = I.reshape(n, self.CB, bs)
            O = np.zeros((n, self.KB, bs))
            for k, col in self.fprop_list:
                if col:
                    cs = [e[0] for e in col]; ws = [e[1] for e in col]
                    O[:, k, :] = X[:, cs, :].reshape(n, -1) @ W[ws].reshape(-1, bs)
            return O.reshape(n, -1)
        n = I[0].size
        X = I.reshape(self.CB, bs, n)
        O = np.zeros((self.KB, bs, n))
        for k, col in self.fprop_list:
            if col:
                cs = [e[0] for e in col]; ws = [e[1] for e in col]
                O[k] = W[ws].reshape(-1, bs).T @ X[cs].reshape(-1, n)
        return O.reshape(-1, n)

    def bprop_test(self, E, W, gate=None):
        bs = self.bsize
        E = np.asarray(E); W = np.asarray(W)
        if gate is not None:
            W = W * np.asarray(gate, dtype=W.dtype)[:, None, None]
        if self.axis:
            n = E.shape[0]
            D = E.reshape(n, self.KB, bs)
            B = np.zeros((n, self.CB, bs))
            for c, row in self.bprop_list:
                if row:
                    ks = [e[0] for e in row]; ws = [e[1] for e in row]
                    B[:, c, :] = D[:, ks, :].reshape(n, -1) @ np.transpose(W[ws], (0, 2, 1)).reshape(-1, bs)
            return B.reshape(n, -1)
        n = E[0].size
        D = E.reshape(self.KB, bs, n)
        B = np.zeros((self.CB, bs, n))
        for c, row in self.bprop_list:
            if row:
                ks = [e[0] for e in row]; ws = [e[1] for e in row]
                B[c] = np.transpose(W[ws], (1, 0, 2)).reshape(bs, -1) @ D[ks].reshape(-1, n)
        return B.reshape(-1, n)

    def updat_test(self, I, E, gate=None, dw_gated=False):
        bs = self.bsize
        I = np.asarray(I, dtype=np.float64); E = np.asarray(E, dtype=np.float64)
        ul = self.updat_lut
        if self.axis:
            X = I.reshape(-1, self.CB, bs).transpose(1, 2, 0)
            D = E.reshape(-1, self.KB, bs).transpose(1, 2, 0)
        else:
            X = I.reshape(self.CB, bs, -1)
            D = E.reshape(self.KB, bs, -1)
        U = np.matmul(X[ul[:, 0]], np.transpose(D[ul[:, 1]], (0, 2, 1)))
        if dw_gated and gate is not None:
            U = U * np.asarray(gate, dtype=np.float64)[:, None, None]
        return U


if torch is not None:
    class _BsmmFunction(torch.autograd.Function):
        """y = bsmm(x, w); backward = (bprop, updat), the registered gradient of matmul.py:485-527."""

        @staticmethod
        def forward(ctx, x, w, bsmm):
            ctx.bsmm = bsmm
            ctx.save_for_backward(x, w)
            return bsmm.fprop(x, w)

        @staticmethod
        def backward(ctx, dy):
            x, w = ctx.saved_tensors
            bsmm = ctx.bsmm
            dx = bsmm.bprop(dy, w) if ctx.needs_input_grad[0] else None
            dw = bsmm.updat(x, dy) if ctx.needs_input_grad[1] else None
            return dx, dw, None

    class _L2NormFunction(torch.autograd.Function):
        @staticmethod
        def forward(ctx, W, gain, bsmm, epsilon, y_dtype):
            y, ss = bsmm._l2_fwd(W, gain, epsilon, y_dtype)
            ctx.bsmm, ctx.epsilon = bsmm, epsilon
            ctx.save_for_backward(W, gain, ss)
            return y

        @staticmethod
        def backward(ctx, dy):
            W, gain, ss = ctx.saved_tensors
            dx, dg = ctx.bsmm._l2_bwd(dy, W, gain, ss, ctx.epsilon)
            return dx, dg, None, None, None

    class _BsmmGatedFunction(torch.autograd.Function):
        """y = bsmm(x, w, gate) with the registered gradient of the gated op (blocksparse/matmul.py:485-527): dx through the
        gated bprop, dw optionally gated (``dw_gated``), and with ``gate_grad`` (dw, dg) = blocksparse_matmul_dg(dw, w, gate)."""

        @staticmethod
        def forward(ctx, x, w, gate, bsmm, gate_grad, dw_gated):
            ctx.bsmm, ctx.gate_grad, ctx.dw_gated = bsmm, gate_grad, dw_gated
            ctx.save_for_backward(x, w, gate)
            return bsmm.fprop(x, w, gate=gate)

        @staticmethod
        def backward(ctx, dy):
            x, w, gate = ctx.saved_tensors
            bsmm = ctx.bsmm
            dx = bsmm.bprop(dy, w, gate=gate) if ctx.needs_input_grad[0] else None
            dw = dg = None
            if ctx.needs_input_grad[1] or (ctx.gate_grad and ctx.needs_input_grad[2]):
                dw = bsmm.updat(x, dy, gate=gate if ctx.dw_gated else None)
                if ctx.gate_grad:
                    dw, dg = bsmm.gate_grad(dw, w, gate)
            return dx, dw, dg, None, None, None
