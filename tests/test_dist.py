"""N>1 path on CPU: minibatch sharding + ONE all-reduce of dw (gloo, world_size 2).  The per-rank math is done by the
oracle here (no GPU in this tier); on the GPU box the same DwAllReduce object wraps the HIP updat (bench.py)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from blocksparse_amd.dist import DwAllReduce, DwReduce, shard_bounds, shard_minibatch
from oracle import bsmm_oracle as orc


def test_shard_bounds_cover_exactly():
    for N in (1, 7, 64, 8191):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(N, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == N
            for (a, b), (c, d) in zip(spans, spans[1:]):
                assert b == c and b - a >= d - c >= 0 and (b - a) - (d - c) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, axis, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(0)
        layout = (rng.random((5, 6)) < 0.5).astype(np.int32)
        layout[0, :] = 1
        layout[:, 0] = 1
        bs, N = 8, 37
        t = orc.build_layout_luts(layout, bs)
        W = rng.normal(size=(t["blocks"], bs, bs)).astype(np.float32)
        X = rng.normal(size=(N, t["C"]) if axis else (t["C"], N)).astype(np.float32)
        E = rng.normal(size=(N, t["K"]) if axis else (t["K"], N)).astype(np.float32)
        xs = shard_minibatch(torch.from_numpy(X), axis, rank, world).numpy()
        es = shard_minibatch(torch.from_numpy(E), axis, rank, world).numpy()
        # fprop / bprop: no communication, each rank's slice of the full result
        y_local = orc.fprop(t, xs, W, axis)
        y_full = orc.fprop(t, X, W, axis)
        lo, hi = shard_bounds(N, rank, world)
        ref = y_full[lo:hi] if axis else y_full[:, lo:hi]
        ok_y = np.allclose(y_local, ref, atol=1e-10)
        # updat: partial dw per rank, one all-reduce, overlapped with "bprop"
        dw = torch.from_numpy(orc.updat(t, xs, es, axis).astype(np.float32))
        red = DwAllReduce(accumulate_fp32=True)
        red.start(dw)
        dx_local = orc.bprop(t, es, W, axis)          # the work the collective overlaps with
        red.wait()
        ok_dw = np.allclose(dw.numpy(), orc.updat(t, X, E, axis), rtol=1e-5, atol=1e-5)
        dxf = orc.bprop(t, E, W, axis)
        ok_dx = np.allclose(dx_local, dxf[lo:hi] if axis else dxf[:, lo:hi], atol=1e-10)
        # bf16 storage with fp32 accumulation across ranks
        dwb = torch.from_numpy(orc.updat(t, xs, es, axis).astype(np.float32)).bfloat16()
        red2 = DwAllReduce(accumulate_fp32=True)
        red2.start(dwb)
        red2.wait()
        full = torch.from_numpy(orc.updat(t, X, E, axis).astype(np.float32))
        ok_bf = torch.allclose(dwb.float(), full, rtol=2e-2, atol=2e-2)
        # the fused reduction (reduce the raw fp32 sums over the ranks, THEN alpha / beta / gate and one rounding): same object the
        # GPU path uses, CPU tensors go through torch.distributed with the same arithmetic
        class _B(object):                 # what DwReduce needs of a BlocksparseMatMul
            blocks, bsize = t["blocks"], bs
        gate = torch.from_numpy(rng.random(t["blocks"]).astype(np.float32))
        dw_old = torch.from_numpy(rng.normal(size=(t["blocks"], bs, bs)).astype(np.float32)).bfloat16()
        sums = torch.from_numpy(orc.updat(t, xs, es, axis).astype(np.float32))
        dwf = dw_old.clone()
        red3 = DwReduce(_B())
        red3.start(sums, dwf, alpha=0.5, beta=2.0, gate=gate)
        red3.wait()
        want = (0.5 * gate.reshape(-1, 1, 1) * full + 2.0 * dw_old.float()).bfloat16()
        ok_fused = torch.equal(dwf, want) or torch.allclose(dwf.float(), want.float(), rtol=1e-2, atol=1e-2)
        out_q.put((rank, ok_y, ok_dw, ok_dx, bool(ok_bf), bool(ok_fused)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("axis", [0, 1])
def test_two_rank_data_parallel_matches_single_process(axis):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, axis, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == [0, 1]
    for r in res:
        assert all(r[1:]), r


def test_single_process_is_a_noop():
    dw = torch.ones(3, 4)
    red = DwAllReduce()
    assert red.start(dw) is dw
    red.wait()
    assert torch.equal(dw, torch.ones(3, 4))
