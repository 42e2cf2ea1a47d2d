"""quick parity check of updat (bsize 32, axis 1) for the library selected by BSMM_LIB: ragged N, pairs, alpha/beta"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import _parity as P
from oracle import bsmm_oracle as orc
from blocksparse_amd import BlocksparseMatMul, _lib
_lib.set_kernel_variant(3)
worst = 0.0
for li, (lay, Ns) in enumerate([(P.random_layout(40, 40, 0.15, 2), (392, 100, 8, 1)), (P.random_layout(33, 17, 0.3, 3), (200, 72)), (P.ba_layout(40, 3, 1), (264,))]):
    for opt in (0, _lib.PLAN_STREAM_8):
        for split in (0, 1, 3):
            b = BlocksparseMatMul(lay, block_size=32, feature_axis=1, plan_options=opt, updat_split=split)
            t = orc.build_layout_luts(lay, 32)
            for N in Ns:
                Xs, Es = [], []
                for p in range(2):
                    _, X, E = P.make_inputs(b.w_shape, b.i_shape(N), b.o_shape(N), "bf16", seed=li * 7 + N + p)
                    Xs.append(X); Es.append(E)
                dw0 = orc.round_bf16(np.random.RandomState(1).normal(size=b.w_shape).astype(np.float32) * 0.1)
                ref = orc.updat(t, Xs, Es, 1, alpha=0.5, beta=2.0, dw_in=dw0)
                dw = P.to_dev(dw0, "bf16", torch)
                out = b.updat([P.to_dev(x, "bf16", torch) for x in Xs], [P.to_dev(e, "bf16", torch) for e in Es], alpha=0.5, beta=2.0, dw=dw)
                assert _lib.last_kernel() == _lib.K_UPDAT_STREAM
                l2, mx = P.errors(P.to_host(out), orc.round_bf16(ref))
                worst = max(worst, l2)
                assert l2 <= 1e-3, (li, opt, split, N, l2)
print(os.path.basename(os.environ.get("BSMM_LIB", "default")), "parity ok, worst L2 %.2e" % worst)
