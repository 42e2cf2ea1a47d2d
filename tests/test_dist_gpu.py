"""GPU tier of the data-parallel path: the library's RCCL handle (include/bsmm_dist.h), the fp32-sums hand-over of the
streaming updat kernel, and -- when the box has more than one GPU -- one process per GPU against the single-GPU result
(the reference's check: all-reduce on real GPUs vs np.dot(A, B) * size, /root/reference/test/nccl_test.py:22-57)."""
import os
import subprocess
import sys

import numpy as np
import pytest

import _parity as P

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def env():
    import torch
    from blocksparse_amd import BlocksparseMatMul, _lib
    assert torch.cuda.is_available()
    _lib.load()
    return torch, BlocksparseMatMul, _lib


def test_sums_only_plus_finalize_equals_updat(env):
    torch, BSMM, lib = env
    layout = P.random_layout(40, 40, 0.15, seed=2)
    b = BSMM(layout, block_size=32, feature_axis=1)
    g = torch.Generator(device="cuda").manual_seed(3)
    N = 1000
    x = (torch.randn(b.i_shape(N), device="cuda", generator=g) * 0.1).bfloat16()
    dy = (torch.randn(b.o_shape(N), device="cuda", generator=g) * 0.1).bfloat16()
    gate = torch.rand(b.blocks, device="cuda", generator=g)
    dw0 = (torch.randn(b.w_shape, device="cuda", generator=g) * 0.1).bfloat16()
    try:
        lib.set_kernel_variant(3)
        want = b.updat(x, dy, alpha=0.5, beta=2.0, dw=dw0.clone(), gate=gate)
        sums = b.updat(x, dy, sums_only=True)
        assert lib.last_kernel() == lib.K_UPDAT_STREAM and sums.dtype == torch.float32 and tuple(sums.shape) == b.w_shape
        got = b.updat_finalize(sums, alpha=0.5, beta=2.0, dw=dw0.clone(), gate=gate)
    finally:
        lib.set_kernel_variant(0)
    # the same kernel and the same single rounding; the fp32 partial sums of the minibatch parts meet through atomics, whose
    # order differs from launch to launch: equal up to an occasional last-place flip of the 16-bit result
    d = (got.float() - want.float()).abs()
    assert (d > 0).float().mean().item() < 0.02 and (d.norm() / want.float().norm()).item() < 1e-4
    # configurations without the streaming kernel say so instead of returning something else
    b0 = BSMM(torch.ones(8, 8).numpy() > 0, block_size=16, feature_axis=0)
    with pytest.raises(lib.BsmmError):
        b0.updat(torch.zeros(b0.i_shape(64), device="cuda").bfloat16(), torch.zeros(b0.o_shape(64), device="cuda").bfloat16(), sums_only=True)


def test_rccl_handle_world_size_one(env):
    """The whole begin / overlap / end machinery on one GPU: the sum over one rank is the identity, ordering is by events."""
    torch, BSMM, lib = env
    from blocksparse_amd.dist import DwAllReduce, RcclComm
    comm = RcclComm()
    assert comm.world == 1
    t = torch.arange(1 << 20, device="cuda", dtype=torch.float32)
    want = t.clone()
    red = DwAllReduce(force=True, comm=comm)
    for _ in range(3):
        t.mul_(2.0); want.mul_(2.0)                            # producer work on the current stream
        red.start(t)
        y = torch.ones(1 << 22, device="cuda").sum()           # something to overlap with
        red.wait()
        assert torch.equal(t, want)
    h = (torch.randn(1000, device="cuda", generator=torch.Generator(device="cuda").manual_seed(61)) * 3).bfloat16()
    red16 = DwAllReduce(force=True, comm=comm, accumulate_fp32=True)
    keep = h.clone()
    red16.start(h); red16.wait()
    torch.cuda.synchronize()
    assert torch.equal(h, keep)
    comm.close()


def test_fused_dw_reduce_world_size_one(env):
    """bsmm_dist_dw_begin / _end on one GPU (reduce-scatter over one rank, shard finalize, all-gather, copy into dw): equals
    updat_finalize of the same sums bit for bit, with alpha / beta / gate, for a block count that does not divide into shards."""
    torch, BSMM, lib = env
    from blocksparse_amd.dist import DwReduce, RcclComm
    layout = P.random_layout(40, 40, 0.15, seed=2)
    b = BSMM(layout, block_size=32, feature_axis=1)
    g = torch.Generator(device="cuda").manual_seed(3)
    N = 1000
    x = (torch.randn(b.i_shape(N), device="cuda", generator=g) * 0.1).bfloat16()
    dy = (torch.randn(b.o_shape(N), device="cuda", generator=g) * 0.1).bfloat16()
    gate = torch.rand(b.blocks, device="cuda", generator=g)
    dw0 = (torch.randn(b.w_shape, device="cuda", generator=g) * 0.1).bfloat16()
    comm = RcclComm()
    red = DwReduce(b, comm=comm, force=True)
    try:
        lib.set_kernel_variant(3)
        for slot, (alpha, beta, gt) in enumerate(((1.0, 0.0, None), (0.5, 2.0, gate))):
            sums = b.updat(x, dy, sums_only=True, slot=slot)
            want = b.updat_finalize(sums, alpha=alpha, beta=beta, dw=dw0.clone(), gate=gt)
            got = dw0.clone()
            red.start(sums, got, alpha=alpha, beta=beta, gate=gt)
            y = b.bprop(dy, dw0)                                   # something to overlap with
            red.wait()
            torch.cuda.synchronize()
            assert torch.equal(got, want), (alpha, beta)
            # an exactly-sized copy of the sums (no workspace tail behind it): DwReduce gives the reduce-scatter its room
            got2 = dw0.clone()
            red.start(sums.clone(), got2, alpha=alpha, beta=beta, gate=gt)
            red.wait()
            torch.cuda.synchronize()
            assert torch.equal(got2, want), ("exactly sized sums", alpha, beta)
        with pytest.raises(ValueError):
            b.updat(x, dy, sums_only=True, gate=gate)              # the sums are ungated: the gate belongs to the finalize step
    finally:
        lib.set_kernel_variant(0)
        comm.close()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_fused_dw_reduce_emulated_ranks_on_one_gpu(env, world):
    """bsmm_dist_dw_emulate: the fused reduction for `world` VIRTUAL ranks on one device -- the library's own shard arithmetic
    (bsmm_dist_dw_layout), its shard-finalize kernel with every rank's bounds, collectives replaced by sums / copies over the same
    regions RCCL would touch.  Every rank's dw must equal the single-rank finalize of the summed sums BIT FOR BIT, for block counts
    whose totals do not divide into shards, every dtype and block size, alpha / beta / gate; the padding of the sums buffers holds
    NaNs (it must never reach dw)."""
    import ctypes
    torch, BSMM, lib = env
    L = lib.load()
    g = torch.Generator(device="cuda").manual_seed(world)
    st = torch.cuda.current_stream().cuda_stream
    for bs, blocks in ((8, 7), (16, 61), (32, 127), (32, 3)):
        for td, code in ((torch.bfloat16, lib.BF16), (torch.float16, lib.F16), (torch.float32, lib.F32)):
            total = blocks * bs * bs
            shard, _, _, cap = lib.dw_layout(world, 0, blocks, bs)
            sums = [torch.full((cap,), float("nan"), device="cuda") for _ in range(world)]
            for s in sums:
                s[:total] = torch.randn(total, device="cuda", generator=g)
            gate = torch.rand(blocks, device="cuda", generator=g)
            dw0 = (torch.randn(blocks, bs, bs, device="cuda", generator=g) * 0.1).to(td)
            for alpha, beta, gt in ((1.0, 0.0, None), (0.5, 2.0, gate)):
                tot = sums[0][:total].clone()
                for s in sums[1:]:
                    tot += s[:total]                                            # the emulation sums the ranks in rank order too
                want = dw0.clone()
                lib.check(L.bsmm_updat_finalize(tot.data_ptr(), want.data_ptr(), gt.data_ptr() if gt is not None else None, blocks, bs, code,
                                                alpha, beta, st), "bsmm_updat_finalize")
                work = [s.clone() for s in sums]
                dws = [dw0.clone() for _ in range(world)]
                stg = [torch.empty(cap, dtype=td, device="cuda") for _ in range(world)]
                arr = ctypes.c_void_p * world
                rc = L.bsmm_dist_dw_emulate(world, arr(*[t.data_ptr() for t in work]), cap, arr(*[t.data_ptr() for t in dws]),
                                            arr(*[t.data_ptr() for t in stg]), gt.data_ptr() if gt is not None else None, blocks, bs, code,
                                            alpha, beta, st)
                assert rc == 0, rc
                torch.cuda.synchronize()
                for r in range(world):
                    assert torch.equal(dws[r], want), (world, bs, blocks, td, alpha, beta, r)
    # an exactly-sized sums buffer is refused (and DwReduce pads such a tensor itself: see test_fused_dw_reduce_world_size_one)
    blocks, bs = 7, 8
    exact = [torch.zeros(blocks * bs * bs, device="cuda") for _ in range(world)]
    arr = ctypes.c_void_p * world
    if lib.dw_layout(world, 0, blocks, bs)[3] > blocks * bs * bs:
        rc = L.bsmm_dist_dw_emulate(world, arr(*[t.data_ptr() for t in exact]), exact[0].numel(), arr(*[t.data_ptr() for t in exact]),
                                    arr(*[t.data_ptr() for t in exact]), None, blocks, bs, lib.F32, 1.0, 0.0, st)
        assert rc == -3


_WORKER = r"""
import os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np, torch, torch.distributed as dist
import _parity as P
from blocksparse_amd import BlocksparseMatMul
from blocksparse_amd.dist import DwAllReduce, DwReduce, shard_minibatch
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(int(os.environ["LOCAL_RANK"]))
dist.init_process_group("gloo")                                 # bootstrap channel only: the data goes through the library's RCCL handle
layout = P.random_layout(40, 40, 0.15, seed=2)
b = BlocksparseMatMul(layout, block_size=32, feature_axis=1)
g = torch.Generator().manual_seed(5)
N = 64 * world + 24
X = (torch.randn(b.i_shape(N), generator=g) * 0.1).bfloat16()
E = (torch.randn(b.o_shape(N), generator=g) * 0.1).bfloat16()
x, e = shard_minibatch(X, 1, rank, world).cuda(), shard_minibatch(E, 1, rank, world).cuda()
red = DwAllReduce()
sums = b.updat(x, e, sums_only=True)
red.start(sums)
dx = b.bprop(e, (torch.randn(b.w_shape, generator=g) * 0.01).bfloat16().cuda())
red.wait()
dw = b.updat_finalize(sums)
ref = b.updat(X.cuda(), E.cuda())                               # the whole minibatch on this GPU
l2 = ((dw.float() - ref.float()).norm() / ref.float().norm()).item()
assert l2 < 1e-3, l2
# the fused form: reduce-scatter of the fp32 sums, finalize of this rank's shard, all-gather of the finished shards
red2 = DwReduce(b)
sums2 = b.updat(x, e, sums_only=True, slot=1)
dw2 = torch.empty(b.w_shape, dtype=torch.bfloat16, device="cuda")
red2.start(sums2, dw2)
dx = b.bprop(e, dw2 * 0)
red2.wait()
torch.cuda.synchronize()
l2b = ((dw2.float() - ref.float()).norm() / ref.float().norm()).item()
assert l2b < 1e-3, l2b
print("rank", rank, "ok", l2, l2b, flush=True)
dist.destroy_process_group()
"""


def test_multi_gpu_dw_allreduce_matches_single_gpu(env, tmp_path):
    torch, BSMM, lib = env
    n = torch.cuda.device_count()
    if n < 2:
        pytest.skip("one GPU on this box: the multi-rank path is covered by the gloo test (tests/test_dist.py) and world size 1 above")
    script = tmp_path / "worker.py"
    script.write_text(_WORKER % {"root": ROOT})
    envv = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(min(n, 8)), "--master-addr", "127.0.0.1",
                        "--master-port", "29577", str(script)], env=envv, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
