#!/usr/bin/env python
"""bench.py -- block-sparse matmul hot path on MI355X: effective TFLOP/s (+ GB/s, roofline, CPU baseline).

    python bench.py [--gpus N --steps K --warmup W]          (N > 1: launched by torch.distributed.run)

Workload (BASELINE.json metric: bsmm 4096x4096 bs=32 @ 20% density): one STEP = fprop + bprop + updat of
one minibatch of N_LOCAL columns on every rank, bf16 storage / fp32 accumulate, synthetic data resident in
HBM before the timed region.  Multi-GPU = data parallel: tables and W replicated, minibatch sharded
(weak scaling: N_LOCAL fixed per GPU), one RCCL all-reduce of dw per step overlapped with bprop.
Effective FLOPs per pass = 2 * blocks * bs^2 * N (nonzero blocks only; the reference's own definition,
src/gpu_types.cc:48, src/blocksparse_matmul_op.cc:102,182).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_MFMA = {"bf16": 2500.0, "f16": 2500.0, "f32": 157.3}   # dense TFLOP/s, MI355X_MICROARCH.md
PEAK_HBM = 8000.0                                            # GB/s (spec)


def random_layout(CB, KB, density, seed):
    """rng.random < density with at least one block per row and column (same generator as tests/_parity.py)."""
    rng = np.random.default_rng(seed)
    lay = (rng.random((CB, KB)) < density).astype(np.int32)
    for r in np.nonzero(lay.sum(axis=1) == 0)[0]:
        lay[r, rng.integers(0, KB)] = 1
    for c in np.nonzero(lay.sum(axis=0) == 0)[0]:
        lay[rng.integers(0, CB), c] = 1
    return lay


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=200)
    p.add_argument("--warmup", type=int, default=20)
    p.add_argument("--prewarm-seconds", type=float, default=0.5,
                   help="untimed steps run for this long before the warmup steps: the GPU needs a few hundred ms of sustained load "
                        "to reach its boost clock (20 steps measure 0.41 ms/step, 400 steps 0.34 ms/step on the same box)")
    p.add_argument("--hidden", type=int, default=4096)
    p.add_argument("--bsize", type=int, default=32)
    p.add_argument("--density", type=float, default=0.2)
    p.add_argument("--axis", type=int, default=1)
    p.add_argument("--dtype", default="bf16", choices=["bf16", "f16", "f32"])
    p.add_argument("--n-local", type=int, default=8192, help="minibatch columns per GPU")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-seconds", type=float, default=12.0)
    p.add_argument("--no-attention", action="store_true", help="skip the BASELINE configs[4] (block-sparse attention) extra")
    p.add_argument("--sweep", action="store_true", help="also time 10%% and 50%% density (extra JSON fields)")
    return p.parse_args()


def alg_bytes_xprop(b, N, s):
    segs = b._dev_tables["fprop"]["segments"]
    return s * (b.C * N + b.K * N + b.blocks * b.bsize ** 2) + 4 * (4 * segs + 2 * b.blocks)


def alg_bytes_updat(b, N, s):
    return s * (b.C * N + b.K * N) + s * b.blocks * b.bsize ** 2 + 8 * b.blocks


def attention_extra(a):
    """BASELINE configs[4]: block-sparse attention, batch 4, 16 heads x 64, ctx 4096, bsize 32, local(4)+strided(8) causal
    layout (1466 blocks per head), fp32 activations / bf16 scores (the reference's fp32 pathway).  Reported per op:
    ms, effective TFLOP/s (2 * blocks * 32 * 32 * 64 per head and batch entry), algorithmic GB/s, and the bound
    max(flops / 157.3 TF, algorithmic bytes / 8 TB/s).  CPU baseline: the oracle (NumPy, fp32) on one batch entry and two heads."""
    import torch
    from blocksparse_amd import BlocksparseTransformer
    B, H, HS, BS, CTX = 4, 16, 64, 32, 128
    qi, ki = np.indices((CTX, CTX))
    lay = ((ki <= qi) & ((qi - ki < 4) | ((qi - ki) % 8 == 0))).astype(np.int32)      # local 4 blocks + every 8th, causal

    def causal(blk_shape, head, q, k, b):                                            # diagonal blocks: lower triangle
        m = np.ones(blk_shape, dtype=bool)
        return np.tril(m) if q == k else m

    bst = BlocksparseTransformer(lay, block_size=BS, heads=H, mask_callback=causal)
    g = torch.Generator(device="cuda").manual_seed(7)
    q, k, v = (torch.rand(B, CTX * BS, H * HS, device="cuda", generator=g) * 2 - 1 for _ in range(3))
    sd = torch.bfloat16
    mask = bst._table("mask", "cuda")
    scale = 1.0 / np.sqrt(HS)
    w = bst._nt(q, k, sd)
    p = bst._softmax_fwd(w, scale, mask, sd)
    dp = torch.randn(p.shape, device="cuda", generator=g).to(sd)

    def timeit(fn, reps=100):
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    flops = 2.0 * B * H * bst.blocks * BS * BS * HS
    sbytes = B * H * bst.blocks * BS * BS * 2
    abytes = q.numel() * 4
    ops = [("nt", lambda: bst._nt(q, k, sd), flops, 2 * abytes + sbytes),
           ("masked_softmax", lambda: bst._softmax_fwd(w, scale, mask, sd), 0.0, 2 * sbytes),
           ("nn", lambda: bst._xn(p, v, False), flops, 2 * abytes + sbytes),
           ("tn", lambda: bst._xn(p, q, True), flops, 2 * abytes + sbytes),
           ("softmax_grad", lambda: bst._softmax_bwd(dp, p, scale), 0.0, 3 * sbytes)]
    res, total = {}, 0.0
    for name, fn, fl, by in ops:
        ms = timeit(fn)
        total += ms
        bound = max(fl / (PEAK_MFMA["f32"] * 1e12), by / (PEAK_HBM * 1e9)) * 1e3
        res[name] = {"ms": round(ms, 4), "tflops": round(fl / ms / 1e9, 2), "gbps": round(by / ms / 1e6, 1),
                     "bound": "mfma" if fl / (PEAK_MFMA["f32"] * 1e12) > by / (PEAK_HBM * 1e9) else "hbm",
                     "bound_ms": round(bound, 4), "frac": round(bound / ms, 4)}
    # forward + backward of one attention layer = nt, softmax, nn | tn(dv), nt(dp), softmax_grad, nn(dq), tn(dk)
    fb = res["nt"]["ms"] * 2 + res["masked_softmax"]["ms"] + res["softmax_grad"]["ms"] + res["nn"]["ms"] * 2 + res["tn"]["ms"] * 2
    # the same operators with bf16 activations (native 16-bit MFMA, HBM-bound): not the BASELINE configuration, for reference
    qb, kb, vb = q.bfloat16(), k.bfloat16(), v.bfloat16()
    res16 = {}
    for name, fn in (("nt", lambda: bst._nt(qb, kb, sd)), ("nn", lambda: bst._xn(p, vb, False)), ("tn", lambda: bst._xn(p, qb, True))):
        ms = timeit(fn)
        by = 2 * qb.numel() * 2 + sbytes
        res16[name] = {"ms": round(ms, 4), "tflops": round(flops / ms / 1e9, 2), "gbps": round(by / ms / 1e6, 1), "bound": "hbm",
                       "bound_ms": round(by / (PEAK_HBM * 1e9) * 1e3, 4), "frac": round(by / (PEAK_HBM * 1e9) * 1e3 / ms, 4)}
    out = {"workload": "BASELINE configs[4]: block-sparse attention batch %d heads %d x %d ctx %d bsize %d, %d blocks/head, fp32 activations, bf16 scores"
                       % (B, H, HS, CTX * BS, BS, bst.blocks),
           "ops": res, "fwd_bwd_ms": round(fb, 4), "fwd_bwd_tflops": round(6 * flops / fb / 1e9, 2), "ops_bf16_activations": res16}
    if not a.no_cpu_baseline:
        from oracle import bst_oracle as O            # the oracle is only the timed CPU baseline here
        L = O.build_luts(lay)
        qc, kc, vc = (t[:1, :, :2 * HS].float().cpu().numpy() for t in (q, k, v))
        t0 = time.perf_counter()
        W = O.nt(L, qc, kc, BS, 2)
        P = O.masked_softmax(L, W, BS, scale, bst.softmax_mask_np)
        O.nn(L, P, vc, BS, 2)
        el = time.perf_counter() - t0
        fl_s = 2 * 2.0 * 2 * bst.blocks * BS * BS * HS          # nt + nn of 1 batch entry x 2 heads
        out["cpu_baseline"] = {"value": round(fl_s / el / 1e9, 3), "unit": "GFLOP/s (nt + softmax + nn, forward)", "cores": 1, "kind": "port",
                               "sample": "1 of 4 batch entries, 2 of 16 heads, float64 NumPy oracle, %.2f s" % el}
    return out


def cpu_baseline(layout, bs, axis, seconds):
    """The oracle's batched-BLAS port (fp32) of the same three passes on the host cores, bounded sample."""
    from oracle import bsmm_oracle as orc
    t = orc.build_layout_luts(layout, bs)
    N = 1024
    rng = np.random.default_rng(0)
    CB, KB = layout.shape
    W = rng.normal(0, 0.01, (t["blocks"], bs, bs)).astype(np.float32)
    X = rng.normal(0, 0.1, (N, CB * bs) if axis else (CB * bs, N)).astype(np.float32)
    E = rng.normal(0, 0.1, (N, KB * bs) if axis else (KB * bs, N)).astype(np.float32)
    flops_step = 3 * 2.0 * t["blocks"] * bs * bs * N
    orc.fprop_fast(t, X, W, axis)                       # warm up BLAS threads
    t0 = time.perf_counter()
    steps = 0
    while True:
        orc.fprop_fast(t, X, W, axis)
        orc.bprop_fast(t, E, W, axis)
        orc.updat_fast(t, X, E, axis)
        steps += 1
        el = time.perf_counter() - t0
        if el >= seconds or steps >= 50:
            break
    try:
        from threadpoolctl import threadpool_info
        cores = max([i.get("num_threads", 1) for i in threadpool_info()] or [os.cpu_count()])
    except Exception:
        cores = os.cpu_count()
    return {"value": round(flops_step * steps / el / 1e12, 4), "unit": "TFLOP/s", "cores": int(cores), "kind": "port",
            "sample": "oracle batched-BLAS fp32 port (oracle/bsmm_oracle.py *_fast), same layout, minibatch %d, "
                      "%d steps of fprop+bprop+updat in %.1f s" % (N, steps, el)}


def main():
    a = parse()
    import torch
    import torch.distributed as dist
    from blocksparse_amd import BlocksparseMatMul, _lib
    from blocksparse_amd.dist import DwAllReduce

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a ROCm device"
    torch.cuda.set_device(local)
    use_dist = world > 1 or os.environ.get("BSMM_FORCE_DIST") == "1"   # the env forces the RCCL path at world_size 1 (self-test)
    if use_dist:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    _lib.load()

    td = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[a.dtype]
    s = 4 if a.dtype == "f32" else 2
    CB = a.hidden // a.bsize

    def setup(density):
        layout = random_layout(CB, CB, density, seed=1234)
        b = BlocksparseMatMul(layout, block_size=a.bsize, feature_axis=a.axis)
        g = torch.Generator(device="cuda").manual_seed(1 + rank)
        w = (torch.randn(b.w_shape, device="cuda", generator=g) * 0.01).to(td)
        x = (torch.randn(b.i_shape(a.n_local), device="cuda", generator=g) * 0.1).to(td)
        dy = (torch.randn(b.o_shape(a.n_local), device="cuda", generator=g) * 0.1).to(td)
        return layout, b, w, x, dy

    def run(b, w, x, dy, steps, warmup, timed_events):
        red = DwAllReduce(accumulate_fp32=False)
        dw = torch.empty(b.w_shape, dtype=td, device="cuda")

        def step(ev=None):
            if ev: ev[0].record()
            y = b.fprop(x, w)
            if ev: ev[1].record()
            b.updat(x, dy, dw=dw)
            if ev: ev[2].record()
            red.start(dw)                  # overlaps with bprop
            dx = b.bprop(dy, w)
            if ev: ev[3].record()
            red.wait()
            return y, dx

        if a.prewarm_seconds > 0:
            t_pre = time.perf_counter()
            while time.perf_counter() - t_pre < a.prewarm_seconds:
                for _ in range(10):
                    step()
                torch.cuda.synchronize()
        for _ in range(warmup):
            step()
        evs = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(steps)] if timed_events else None
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            step(evs[i] if evs else None)
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if use_dist:
            t = torch.tensor([el], device="cuda", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        per = None
        if evs:
            per = [float(np.mean([e[i].elapsed_time(e[i + 1]) for e in evs])) for i in range(3)]   # ms: fprop, updat, bprop
        return el, per

    layout, b, w, x, dy = setup(a.density)
    el, per = run(b, w, x, dy, a.steps, a.warmup, timed_events=True)
    N = a.n_local
    workload_name = ("bsmm fprop+bprop+updat %dx%d block_size=%d density=%.0f%% feature_axis=%d, minibatch %d per GPU, "
                     "layout default_rng(1234)" % (a.hidden, a.hidden, a.bsize, a.density * 100, a.axis, N))
    flops_pass = 2.0 * b.blocks * a.bsize ** 2 * N
    total_flops = 3 * flops_pass * world * a.steps
    ms_step = el / a.steps * 1e3
    value = total_flops / el / 1e12
    bytes_step = 2 * alg_bytes_xprop(b, N, s) + alg_bytes_updat(b, N, s)
    f_ms, u_ms, b_ms = per

    # roofline of the dominant kernel.  bprop is ONE launch of the xprop kernel (fprop = the same kernel + a
    # small weight-transpose launch); updat is one launch of the updat kernel.  HIP events on the launch stream.
    cand = {
        "bsmm_xprop(bprop)": (b_ms, flops_pass, alg_bytes_xprop(b, N, s)),
        "bsmm_updat": (u_ms, flops_pass, alg_bytes_updat(b, N, s)),
    }
    dom = max(cand, key=lambda k: cand[k][0])
    d_ms, d_flops, d_bytes = cand[dom]
    ai = d_flops / d_bytes
    ridge = PEAK_MFMA[a.dtype] * 1e12 / (PEAK_HBM * 1e9)
    if ai >= ridge:
        roof = {"bound": "mfma", "achieved": round(d_flops / (d_ms * 1e-3) / 1e12, 2), "peak": PEAK_MFMA[a.dtype], "unit": "TFLOP/s"}
    else:
        roof = {"bound": "hbm", "achieved": round(d_bytes / (d_ms * 1e-3) / 1e9, 1), "peak": PEAK_HBM, "unit": "GB/s"}
    roof["frac"] = round(roof["achieved"] / roof["peak"], 4)
    # HBM bytes per launch of that kernel, from the committed rocprofv3 PMC passes (separate runs; gfx950-corrected):
    # only quoted when the profile was taken on exactly this workload
    roof["traffic"] = None
    try:
        tr = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json")))
        if tr.get("workload") == workload_name and dom in tr and world == 1:
            roof["traffic"] = tr[dom]["hbm_bytes"]
    except Exception:
        pass
    roof["kernel"] = dom
    roof["kernel_ms"] = round(d_ms, 4)
    roof["arithmetic_intensity"] = round(ai, 1)

    out = {
        "metric": "bsmm_effective_tflops_%dx%d_bs%d_d%d" % (a.hidden, a.hidden, a.bsize, round(a.density * 100)),
        "value": round(value, 3), "unit": "TFLOP/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(ms_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": a.dtype, "data": "synthetic",
        "config": {"workload": workload_name,
                   "blocks": int(b.blocks), "global_minibatch": N * world,
                   "parallelism": "dp%d (minibatch sharded, dw all-reduce over RCCL)" % world if world > 1 else "single GPU"},
        "gbps_algorithmic": round(bytes_step / (ms_step * 1e-3) / 1e9, 1),
        # what a dense GEMM of the same shapes would have to sustain to take the same time (reported alongside,
        # never the headline)
        "dense_equivalent_tflops": round(3 * 2.0 * a.hidden * a.hidden * N * world * a.steps / el / 1e12, 1),
        "pass_ms": {"fprop": round(f_ms, 4), "bprop": round(b_ms, 4), "updat": round(u_ms, 4)},
        "pass_tflops": {"fprop": round(flops_pass / f_ms / 1e9, 2), "bprop": round(flops_pass / b_ms / 1e9, 2),
                        "updat": round(flops_pass / u_ms / 1e9, 2)},
        "roofline": roof,
    }
    if use_dist:
        # the dw all-reduce on its own (it overlaps with bprop inside the step): time alone, and how much of it the step hides
        red = DwAllReduce(accumulate_fp32=False)
        dw_t = torch.zeros(b.w_shape, dtype=td, device="cuda")
        for _ in range(5):
            red.start(dw_t); red.wait()
        torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
        t_ar = time.perf_counter()
        for _ in range(20):
            red.start(dw_t); red.wait()
        torch.cuda.synchronize()
        ar_ms = (time.perf_counter() - t_ar) / 20 * 1e3
        tt = torch.tensor([ar_ms], device="cuda", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        ar_ms = float(tt.item())
        compute_ms = f_ms + u_ms + b_ms
        out["allreduce"] = {"bytes": int(dw_t.numel() * dw_t.element_size()), "ms_alone": round(ar_ms, 4),
                            "compute_ms": round(compute_ms, 4),
                            "exposed_ms": round(max(0.0, ms_step - compute_ms), 4),
                            "hidden_frac": round(max(0.0, min(1.0, 1.0 - max(0.0, ms_step - compute_ms) / max(ar_ms, 1e-9))), 3)}
    # BASELINE.json configs[1]: same layout, fp32, feature_axis=1, fprop only.  Priced against the fp32 matrix-core peak
    # (157.3 TF; AI 195 > ridge 20) although the kernel computes the fp32 result exactly from bf16 pieces on the 16-bit
    # matrix cores (six MFMAs per product, bsmm_xcols.h), whose ceiling for this formulation is 2500 / 6 = 417 TF.
    if rank == 0 and world == 1:
        b32 = BlocksparseMatMul(layout, block_size=a.bsize, feature_axis=1)
        g32 = torch.Generator(device="cuda").manual_seed(7)
        w32 = torch.randn(b32.w_shape, device="cuda", generator=g32) * 0.01
        x32 = torch.randn(b32.i_shape(N), device="cuda", generator=g32) * 0.1
        for _ in range(20):
            b32.fprop(x32, w32)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            b32.fprop(x32, w32)
        e1.record()
        torch.cuda.synchronize()
        ms32 = e0.elapsed_time(e1) / 50
        tf32 = 2.0 * b32.blocks * a.bsize ** 2 * N / ms32 / 1e9
        out["fp32_fprop_axis1"] = {"workload": "BASELINE configs[1]: %dx%d bs%d d%.0f%% fp32 feature_axis=1 fprop, minibatch %d" %
                                               (a.hidden, a.hidden, a.bsize, a.density * 100, N),
                                   "kernel": "exact three-piece bf16 split on v_mfma_f32_32x32x16_bf16 (incl. the split pre-passes)",
                                   "ms": round(ms32, 4), "tflops": round(tf32, 2), "peak": PEAK_MFMA["f32"],
                                   "frac": round(tf32 / PEAK_MFMA["f32"], 4),
                                   "peak_bf16_six_products": round(PEAK_MFMA["bf16"] / 6, 1),
                                   "frac_bf16_six_products": round(tf32 / (PEAK_MFMA["bf16"] / 6), 4)}
        del b32, w32, x32
    if rank == 0 and world == 1 and not a.no_attention:
        out["attention"] = attention_extra(a)
    if a.sweep:
        sw = {}
        for d in (0.1, 0.5):
            _, b2, w2, x2, dy2 = setup(d)
            el2, per2 = run(b2, w2, x2, dy2, max(3, a.steps // 2), 5, timed_events=True)
            fp = 2.0 * b2.blocks * a.bsize ** 2 * N
            sw["d%d" % round(d * 100)] = {"tflops": round(3 * fp * world * max(3, a.steps // 2) / el2 / 1e12, 2),
                                          "pass_ms": {"fprop": round(per2[0], 4), "updat": round(per2[1], 4), "bprop": round(per2[2], 4)}, "blocks": int(b2.blocks)}
        out["density_sweep"] = sw
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(layout, a.bsize, a.axis, a.cpu_seconds)
    elif rank == 0:
        out["cpu_baseline"] = None
    # The JSON line must be the LAST thing on stdout.  RCCL's version banner (NCCL_DEBUG=VERSION) sits in every rank's C stdio
    # buffer and would otherwise be flushed at process exit, after Python's own output: push it out on ALL ranks first, meet at
    # a barrier, and only then let rank 0 print.
    import ctypes

    def flush_c_stdio():
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()

    flush_c_stdio()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        flush_c_stdio()
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
