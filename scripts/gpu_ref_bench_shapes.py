"""The shapes of the reference's own benchmark (test/blocksparse_matmul_bench.py:37-78): hidden = k * 2560, Barabasi-Albert(n, m) + I with a dense
m x m corner at the listed sparsities, block sizes 32 / 16 / 8 on feature axis 0, minibatch 64, bf16 -- fprop / bprop / updat as hipGraph replays (us)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import _parity as P
from blocksparse_amd import BlocksparseMatMul, _lib

def graph_us(fn, K=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2): fn()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(K): fn()
    for _ in range(5): g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 20 / K * 1e6

def main():
  mults = [int(v) for v in os.environ.get("MULTS", "1,2,3,4,6,8").split(",")]
  spars = {1: 100.0, 2: 25.62, 3: 11.25, 4: 6.56, 5: 4.25, 6: 2.71, 7: 1.96, 8: 1.41}
  N = int(os.environ.get("N", "64"))
  for k in mults:
      hsize = k * 2560
      for bs in (32, 16, 8):
          n = hsize // bs
          if spars[k] == 100.0:
              lay = np.ones((n, n), dtype=np.int32)
          else:
              for m in range(1, n // 2):
                  blks = 2 * m * (n - m) + m * m + n - m
                  if 100.0 * blks / n ** 2 >= spars[k]:
                      break
              lay = P.ba_layout(n, m, seed=1)
          try:
              b = BlocksparseMatMul(lay, block_size=bs, feature_axis=0)
              g = torch.Generator(device="cuda").manual_seed(1)
              w = (torch.randn(b.w_shape, device="cuda", generator=g) * 0.05).bfloat16()
              x = (torch.randn(b.i_shape(N), device="cuda", generator=g) * 0.1).bfloat16()
              dy = (torch.randn(b.o_shape(N), device="cuda", generator=g) * 0.1).bfloat16()
              dw = torch.empty(b.w_shape, dtype=torch.bfloat16, device="cuda")
              _lib.set_kernel_variant(int(os.environ.get("VARIANT", "0")))
              b.fprop(x, w); kf = _lib.last_kernel(); b.updat(x, dy, dw=dw); ku = _lib.last_kernel()
              f, bp, u = graph_us(lambda: b.fprop(x, w)), graph_us(lambda: b.bprop(dy, w)), graph_us(lambda: b.updat(x, dy, dw=dw))
              fl = 2.0 * b.blocks * bs * bs * N
              wbytes = b.blocks * bs * bs * 2
              print("hidden %5d bs %2d blocks %7d (%.2f %%, W %.1f MB): fprop %.1f bprop %.1f updat %.1f us (k%d / k%d) | %.1f TF, W-stream %.0f GB/s" %
                    (hsize, bs, b.blocks, 100.0 * b.blocks / n ** 2, wbytes / 1e6, f, bp, u, kf, ku, 3 * fl / (f + bp + u) / 1e6, 3 * wbytes / (f + bp + u) / 1e3), flush=True)
              del b, w, x, dy, dw
          except Exception as e:
              print("hidden %5d bs %2d: %s" % (hsize, bs, str(e)[:120]), flush=True)


if __name__ == "__main__":
    main()
