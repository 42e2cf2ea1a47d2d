"""Host-side checks of the 'BSX4' plans of the barrier-free xprop kernel (csrc/bsmm_plan.h build_xflow_plan, csrc/bsmm_xflow.h): the
event lists are the whole synchronisation contract of that kernel, so they are validated here WITHOUT a GPU --
  * every lut entry of an output column appears exactly once as a BLOCK of its wave, in step order (default: ascending input
    blocks; BSMM_PLAN_FLOW_SCHEDULED plans: the order the builder's list scheduling picked -- a permutation of the steps), with the fetch of block j + 2
    riding on block j (and blocks 0 / 1 on the two leading NOPs);
  * every (step, part) has exactly one REQ and one ANN, REQ before ANN in the same wave;
  * the vmcnt each BLOCK / ANN waits with equals the number of vector-memory operations its wave issues in between (capped at 15);
  * a discrete simulation of the 16 waves under the kernel's rules (REQ waits for every wave's progress to pass the slab that held the
    ring slot, BLOCK waits for all parts of its slab, progress = step of the wave's next block) runs to completion: no deadlock, and
    no slab is overwritten while a wave still needs it."""
import ctypes

import numpy as np
import pytest

import _parity as P
from blocksparse_amd import _lib as lib
from blocksparse_amd import lut as L
from blocksparse_amd.matmul import _host_plan

NOFETCH = 0x7ffffff


def _plan(layout, which, order="natural"):
    t = L.build_tables(layout, z_order=True, segmented=False)
    side = t[which]
    n_out = t["KB"] if which == "fprop" else t["CB"]
    words = _host_plan(side["lut"], side["segments"], t["blocks"], n_out, 32, lib.BF16, 1,
                       lib.PLAN_XCOL_FLOW | (lib.PLAN_FLOW_SCHEDULED if order == "scheduled" else 0))
    return t, side, n_out, words


def _columns(side):
    lut = np.asarray(side["lut"])
    cols = {}
    for s in range(side["segments"]):
        off, cnt, ob, _ = lut[4 * s:4 * s + 4]
        ent = lut[2 * off:2 * (off + cnt)].reshape(-1, 2)
        cols.setdefault(int(ob), []).extend((int(c), int(w)) for c, w in ent)
    return cols


@pytest.mark.parametrize("name,layout", [("random 40x24", P.random_layout(40, 24, 0.3, seed=2)), ("dense 12x20", np.ones((12, 20), dtype=np.int32)),
                                         ("BA 64", P.ba_layout(64, 5, seed=1)), ("sparse 300x16", P.random_layout(300, 16, 0.05, seed=6)),
                                         ("single", np.ones((1, 1), dtype=np.int32)), ("BA 128 + I", np.maximum(P.ba_layout(128, 14, seed=3), np.eye(128, dtype=np.int32))), ("groups without blocks", np.eye(15, 40, dtype=np.int32)), ("bench 20 %", P.random_layout(128, 128, 0.2, seed=1234))])
@pytest.mark.parametrize("which", ["fprop", "bprop"])
@pytest.mark.parametrize("order", ["natural", "scheduled"])
def test_flow_plan_event_lists(name, layout, which, order):
    t, side, n_out, p = _plan(layout, which, order)
    assert p is not None and p[0] == 0x42535834 and p[2] == 16
    D, DX, PARTS = p[11] & 0xff, (p[11] >> 8) & 0xff, (p[11] >> 16) & 0xff
    assert 1 <= DX < D and PARTS in (1, 2, 4)
    DI = 16 // PARTS
    cols = _columns(side)
    ngroups = p[3]
    assert p[1] == 4
    owned = []                           # every output block belongs to exactly one (group, wave)
    regrouped_any = False
    for g in range(ngroups):
        step_off, nsteps, ob0, nob, list_off, lcap, nblk, regrouped = p[p[5] + 8 * g:p[5] + 8 * g + 8]
        gcols = [int(c) for c in p[p[7] + list_off + 16:p[7] + list_off + 32]]
        assert gcols[0] == ob0 and sum(c >= 0 for c in gcols) == nob
        if not regrouped:                # consecutive output blocks (uniform layouts, BSMM_PLAN_FLOW_CONSECUTIVE)
            assert ob0 % 16 == 0 and gcols == [ob0 + v if v < nob else -1 for v in range(16)]
        else:                            # adjacent pairs stay together (one 128-byte line of an output row)
            regrouped_any = True
            assert all(gcols[v] < 0 or gcols[v] % 2 == 0 for v in range(0, 16, 2))
            assert all(gcols[v + 1] in (-1, gcols[v] + 1) for v in range(0, 16, 2))
        owned += [c for c in gcols if c >= 0]
        pairs = p[p[6] + step_off:p[6] + step_off + nsteps]
        assert len(set(pairs)) == len(pairs)
        if order == "natural":              # ascending input blocks: the summation order of the staged kernel
            assert list(pairs) == sorted(pairs)
        base = p[7] + list_off
        counts = p[base:base + 16]
        base += 16                           # (version 4: the group's output-block table sits between the counts and the lists)
        reqs, anns = {}, {}
        waves = []
        total_blocks = 0
        for wv in range(16):
            ev = np.asarray(p[base + 16 + 2 * lcap * wv:base + 16 + 2 * lcap * wv + 2 * counts[wv]], dtype=np.int64).reshape(-1, 2) & 0xffffffff
            want = sorted(cols.get(gcols[wv], [])) if gcols[wv] >= 0 else []      # (c, w) ascending in c = step order, even half first
            blocks, ops, fetch_seq, req_seq, fetched = [], 0, [], {}, []
            events = []
            for i, (w0, w1) in enumerate(ev):
                ty, hp, step, nxt = int(w0 & 3), int((w0 >> 2) & 3), int((w0 >> 4) & 0xfff), int((w0 >> 16) & 0xfff)
                f, wait = int(w1 & NOFETCH), int(w1 >> 27)
                assert (w0 >> 28) == step % D
                if ty == 0:
                    assert i < 2
                elif ty == 1:
                    j = len(blocks)
                    assert wait == min(15, ops - fetch_seq[j]), (name, g, wv, i)
                    blocks.append((int(pairs[step]) * 2 + hp, step))
                elif ty == 2:
                    assert (step, hp) not in reqs
                    reqs[(step, hp)] = (wv, i)
                    ops += DI
                    req_seq[(step, hp)] = ops
                else:
                    assert (step, hp) in req_seq and (step, hp) not in anns      # same wave, after its REQ
                    assert wait == min(15, ops - req_seq[(step, hp)])
                    anns[(step, hp)] = (wv, i)
                if f != NOFETCH:
                    assert ty in (0, 1)
                    ops += 2
                    fetch_seq.append(ops)
                    fetched.append(f)
                events.append((ty, hp, step, nxt))
            # every lut entry of my column exactly once, with ITS weight block, in the plan's step order (ascending input blocks when natural)
            assert sorted(zip([b[0] for b in blocks], fetched)) == want, (name, g, wv)
            assert [b[1] for b in blocks] == sorted(b[1] for b in blocks)
            if order == "natural":
                assert [b[0] for b in blocks] == [c for c, _ in want]
            # progress words: after a BLOCK / NOP = the step of the next BLOCK (nsteps if none)
            bsteps = [b[1] for b in blocks]
            k = 0
            for ty, hp, step, nxt in events:
                if ty == 1:
                    k += 1
                if ty < 2:
                    assert nxt == (bsteps[k] if k < len(bsteps) else nsteps)
            total_blocks += len(blocks)
            waves.append(events)
        assert total_blocks == nblk
        assert set(reqs) == set(anns) == {(s, q) for s in range(nsteps) for q in range(PARTS)}
        # ---- discrete simulation of the kernel's rules ----
        pc = [0] * 16
        prog = [0] * 16                      # progress words (unit-local steps; the kernel adds the unit's global offset)
        announced = {}                       # step -> parts announced
        in_ring = {}                         # slot -> step whose slab is (being) loaded there
        moved = True
        while moved:
            moved = False
            for wv in range(16):
                while pc[wv] < len(waves[wv]):
                    ty, hp, step, nxt = waves[wv][pc[wv]]
                    if ty == 1:
                        if announced.get(step, 0) < PARTS:
                            break
                        assert in_ring[step % D] == step, "slab overwritten before a block that needs it"
                    elif ty == 2:
                        if step >= D and min(prog) < step - D + 1:
                            break
                        in_ring[step % D] = step
                    elif ty == 3:
                        announced[step] = announced.get(step, 0) + 1
                    if ty < 2:
                        prog[wv] = nxt
                    pc[wv] += 1
                    moved = True
        assert all(pc[wv] == len(waves[wv]) for wv in range(16)), "deadlock in the event lists of %s group %d" % (name, g)
    assert sorted(owned) == list(range(n_out)) and ngroups == (n_out + 15) // 16
    if name == "BA 128 + I":             # the reference's bench layout: hubs in 16 neighbouring columns -> regrouped, and balanced afterwards
        loads = sorted(int(p[p[5] + 8 * g + 6]) for g in range(ngroups))
        assert regrouped_any and loads[-1] <= 1.1 * (sum(loads) / len(loads)), loads
    if name == "bench 20 %":
        assert not regrouped_any


def test_flow_plan_is_refused_where_the_kernel_does_not_exist():
    """feature axis 0 and the option bits of other kernel families fall back to the staged plan ('BSX2'); bsm_plan_attach describes both"""
    lay = P.random_layout(20, 20, 0.3, seed=1)
    t = L.build_tables(lay, z_order=True, segmented=False)
    f = t["fprop"]
    p0 = _host_plan(f["lut"], f["segments"], t["blocks"], t["KB"], 32, lib.BF16, 0, lib.PLAN_XCOL_FLOW)
    p1 = _host_plan(f["lut"], f["segments"], t["blocks"], t["KB"], 32, lib.BF16, 1, lib.PLAN_XCOL_FLOW)
    assert p0[0] == 0x42535832 and p1[0] == 0x42535834
    a = lib.BsmmArgs()
    ip = ctypes.POINTER(ctypes.c_int32)
    host = np.ascontiguousarray(p1, dtype=np.int32)
    assert lib.load().bsmm_plan_attach(ctypes.byref(a), host.ctypes.data_as(ip), host.size, 0x1000) == 0
    assert a.plan_magic == 0x42535834 and a.plan_width == 16
