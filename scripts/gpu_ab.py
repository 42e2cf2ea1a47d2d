"""A/B timing of kernel variants on the headline shapes (bf16/f16, bs 32): variant 0 = plan/grouped, 2 = per-segment."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import _parity as P
from blocksparse_amd import BlocksparseMatMul, _lib


def timeit(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    L = _lib.load()
    axes = [int(a) for a in os.environ.get("AXES", "1,0").split(",")]
    dens = [float(d) for d in os.environ.get("DENS", "0.2,0.1,0.5").split(",")]
    Ns = [int(n) for n in os.environ.get("NS", "512,8192").split(",")]
    variants = [int(v) for v in os.environ.get("VARIANTS", "0,2").split(",")]
    # quick parity first (BA layout + 4096^2 sample)
    lay = P.ba_layout(40, 3, seed=1)
    for axis in axes:
        for N in (64, 100, 300):
            r = P.run_case(torch, BlocksparseMatMul, lay, 32, axis, "bf16", N, seed=3, passes=("Y", "DX"))
            print("parity a%d N%d" % (axis, N), " ".join("%s=%.1e" % (k, v[0]) for k, v in r.items()),
                  "FAIL" if any(v[0] > 1e-3 for v in r.values()) else "ok")
    big = P.random_layout(128, 128, 0.2, seed=1234)
    for axis in axes:
        r = P.run_case(torch, BlocksparseMatMul, big, 32, axis, "bf16", 384, seed=4, passes=("Y", "DX"), fast_oracle=True)
        print("parity4096 a%d" % axis, " ".join("%s=%.1e" % (k, v[0]) for k, v in r.items()),
              "FAIL" if any(v[0] > 1e-3 for v in r.values()) else "ok")
    td = torch.bfloat16
    for d in dens:
        layout = P.random_layout(128, 128, d, seed=1234)
        for axis in axes:
            b = BlocksparseMatMul(layout, block_size=32, feature_axis=axis)
            for N in Ns:
                w = (torch.randn(b.w_shape, device="cuda") * 0.01).to(td)
                x = (torch.randn(b.i_shape(N), device="cuda") * 0.1).to(td)
                dy = (torch.randn(b.o_shape(N), device="cuda") * 0.1).to(td)
                fl = 2.0 * b.blocks * 1024 * N
                line = "d%.2f a%d N%-5d" % (d, axis, N)
                for v in variants:
                    L.bsmm_set_kernel_variant(v)
                    tf = timeit(lambda: b.fprop(x, w)); tb = timeit(lambda: b.bprop(dy, w))
                    line += " | v%d fprop %.3f ms %6.1f TF bprop %.3f ms %6.1f TF" % (v, tf, fl / tf / 1e9, tb, fl / tb / 1e9)
                L.bsmm_set_kernel_variant(0)
                if "UPDAT" in os.environ:
                    tu = timeit(lambda: b.updat(x, dy))
                    line += " | updat %.3f ms %6.1f TF" % (tu, fl / tu / 1e9)
                print(line, flush=True)


if __name__ == "__main__":
    main()
