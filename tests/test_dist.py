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


def test_fused_dw_layout_arithmetic_covers_every_element_once():
    """bsmm_dist_dw_layout (the shard arithmetic of bsmm_dist_dw_begin) for world in {1, 2, 3, 8, 16}, bsize in {8, 16, 32} and odd
    block counts: 16-byte aligned shards, [lo, hi) of the ranks tile [0, total) exactly (late ranks may own nothing), capacity =
    world * shard >= total; then the three steps emulated in NumPy with exactly these numbers (reduce-scatter of shard-sized pieces
    of PADDED buffers, finalize of [lo, hi), all-gather of shard-sized pieces) equal the single-rank finalize."""
    from blocksparse_amd import _lib
    rng = np.random.default_rng(0)
    for world in (1, 2, 3, 8, 16):
        for bs in (8, 16, 32):
            for blocks in (1, 3, 7, 61, 127):
                total = blocks * bs * bs
                lay = [_lib.dw_layout(world, r, blocks, bs) for r in range(world)]
                shard, cap = lay[0][0], lay[0][3]
                assert shard % 8 == 0 and shard * world == cap >= total and shard >= -(-total // world)
                assert all(l[0] == shard and l[3] == cap for l in lay)
                assert lay[0][1] == 0 and lay[-1][2] == total
                for (_, lo, hi, _), (_, lo2, hi2, _) in zip(lay, lay[1:]):
                    assert lo <= hi == lo2 <= hi2 and hi - lo <= shard
                assert all(lo == min(total, r * shard) for r, (_, lo, _, _) in enumerate(lay))
                # emulate: per-rank padded sums (garbage in the padding, as a workspace tail would hold)
                sums = rng.normal(size=(world, cap)).astype(np.float32)
                gate = rng.random(blocks).astype(np.float32)
                old = rng.normal(size=total).astype(np.float32)
                alpha, beta = 0.5, 2.0
                staging = np.full((world, cap), np.nan, dtype=np.float32)
                for r, (_, lo, hi, _) in enumerate(lay):
                    piece = sums[:, r * shard:(r + 1) * shard].sum(axis=0, dtype=np.float32)      # what rank r receives
                    own = piece[:hi - lo]
                    staging[r, lo:hi] = alpha * gate[np.arange(lo, hi) // (bs * bs)] * own + beta * old[lo:hi]
                gathered = np.concatenate([staging[r, r * shard:(r + 1) * shard] for r in range(world)])[:total]
                want = alpha * np.repeat(gate, bs * bs) * sums[:, :total].sum(axis=0, dtype=np.float32) + beta * old
                np.testing.assert_allclose(gathered, want, rtol=1e-6, atol=1e-6)
    assert _lib.load().bsmm_dist_dw_layout(2, 2, 4, 8, None, None, None, None) == -1        # rank out of range


def test_fused_dw_refuses_a_short_sums_buffer_before_touching_the_device():
    """the capacity requirement of include/bsmm_dist.h is CHECKED: world * shard floats, else BSMM_ERR_WORKSPACE (no GPU needed: the
    check precedes every device call)"""
    import ctypes
    from blocksparse_amd import _lib
    L = _lib.load()
    world, blocks, bs = 3, 7, 8
    shard, _, _, cap = _lib.dw_layout(world, 0, blocks, bs)
    assert cap > blocks * bs * bs                      # ragged: an exactly-sized tensor is too short
    arr = (ctypes.c_void_p * world)(*[0x1000 * (i + 1) for i in range(world)])
    rc = L.bsmm_dist_dw_emulate(world, arr, blocks * bs * bs, arr, arr, None, blocks, bs, _lib.BF16, 1.0, 0.0, None)
    assert rc == -3


def _worker(rank, world, port, axis, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(0)
        layout = (rng.random((5, 6)) < 0.5).astype(np.int32)
        layout[0, :] = 1
        layout[:, 0] = 1
        if layout.sum() % 2 == 0:                     # an odd block count: the shards of the fused reduction are ragged for world 3
            layout[1, 1] ^= 1
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
        # ... and it really went shard by shard: what this rank finalized is 1 / world of the elements (bsmm_dist_dw_layout)
        from blocksparse_amd import _lib
        shard, lo, hi, cap = _lib.dw_layout(world, rank, t["blocks"], bs)
        ok_fused = ok_fused and (hi - lo) <= shard and cap >= t["blocks"] * bs * bs
        out_q.put((rank, ok_y, ok_dw, ok_dx, bool(ok_bf), bool(ok_fused)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("axis,world", [(0, 2), (1, 2), (1, 3)])
def test_two_rank_data_parallel_matches_single_process(axis, world):
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
    assert sorted(r[0] for r in res) == list(range(world))
    for r in res:
        assert all(r[1:]), r


def test_single_process_is_a_noop():
    dw = torch.ones(3, 4)
    red = DwAllReduce()
    assert red.start(dw) is dw
    red.wait()
    assert torch.equal(dw, torch.ones(3, 4))
