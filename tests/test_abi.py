"""CPU tier: the C-ABI library loads, exports every symbol include/bsmm.h declares, and rejects bad
arguments before touching the GPU (no compute calls here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    g.build()
    from blocksparse_amd import _lib
    return _lib


def test_header_symbols_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "bsmm.h")).read()
    declared = set(re.findall(r"\b(bsmm_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"bsmm_args", "bsmm_params"}
    assert declared == set(lib.SYMBOLS), declared ^ set(lib.SYMBOLS)
    L = lib.load()
    for s in declared:
        assert hasattr(L, s), s
    raw = ctypes.CDLL(lib.LIB_PATH)
    for s in declared:
        getattr(raw, s)


def test_version_and_error_strings(lib):
    L = lib.load()
    hdr = open(os.path.join(ROOT, "include", "bsmm.h")).read()
    assert L.bsmm_version() == int(re.search(r"#define BSMM_VERSION (\d+)", hdr).group(1))
    assert lib.error_string(0) == "ok"
    for code in (-1, -2, -3):
        assert "bsmm" in lib.error_string(code)


def test_struct_layout_matches_header(lib):
    # field order in the ctypes mirror == field order in the C struct
    hdr = open(os.path.join(ROOT, "include", "bsmm.h")).read()
    body = re.search(r"typedef struct bsmm_args \{(.*?)\} bsmm_args;", hdr, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = [re.search(r"(\w+)\s*;", ln).group(1) for ln in body.splitlines() if ";" in ln]
    assert names == [f[0] for f in lib.BsmmArgs._fields_]
    assert ctypes.sizeof(lib.BsmmArgs) == 5 * 8 + 18 * 4 + 2 * 4 + 3 * 8   # 4 ptr + size_t, 18 int32, 2 float, 3 ptr


def test_bsize64_composite_plans_without_gpu(lib):
    """bsize 64 at the C ABI (feature axis 1, the reference's second axis-1 block size, blocksparse/matmul.py:84-89): 'BS64' plans carry the
    lookup table of the QUADRANT view -- weight block 4 w + 2 i + j for (input half i, output half j) of the call -- and the nested bsize-32 plan
    built from it; bsmm_plan_attach describes the nested plan in the args, the workspace / prepared-weights queries answer for the
    nested call plus the quadrant copy of W; other combinations are refused on the host."""
    import numpy as np
    from blocksparse_amd import lut as LT
    from blocksparse_amd.matmul import _host_plan, _host_updat_plan
    L = lib.load()
    ip = ctypes.POINTER(ctypes.c_int32)
    rng = np.random.default_rng(3)
    for CB, KB, dens in ((6, 10, 0.4), (20, 20, 0.2), (1, 1, 1.0)):
        lay = rng.random((CB, KB)) < dens
        lay[0, :] = True
        t = LT.build_tables(lay)
        B = t["blocks"]
        for side, n_out in (("fprop", KB), ("bprop", CB)):
            f = t[side]
            for dt in (lib.BF16, lib.F32):
                plan = _host_plan(f["lut"], f["segments"], B, n_out, 64, dt, 1)
                assert plan[0] == 0x42533634 and plan[1] == 1 and plan[2] == B and plan[3] == 0 and plan[4] == 8 and plan[6] == plan.size and plan[7] == f["segments"]
                S = f["segments"]
                lut32 = plan[8:8 + 8 * S + 8 * B]
                assert plan[5] == (8 + lut32.size + 3) // 4 * 4
                got = set()
                for s32 in range(2 * S):
                    off, cnt, ob, lock = (int(v) for v in lut32[4 * s32:4 * s32 + 4])
                    lock64 = int(f["lut"][4 * (s32 // 2) + 3])                         # a locked column's halves get lock ids 2 l - 1 and 2 l
                    assert lock == (2 * lock64 - 1 + (s32 & 1) if lock64 > 0 else 0) and ob == 2 * int(f["lut"][4 * (s32 // 2) + 2]) + (s32 & 1)
                    for e in range(cnt):
                        got.add((ob, int(lut32[2 * (off + e)]), int(lut32[2 * (off + e) + 1])))
                want = set()
                for ob, col in f["cols"]:
                    for c, w in col:
                        for i in range(2):
                            for j in range(2):
                                want.add((2 * ob + j, 2 * c + i, 4 * w + 2 * i + j))
                assert got == want
                nested = plan[plan[5]:]
                assert nested[0] in ((0x42535832, 0x42535843) if dt == lib.BF16 else (0x42535843, 0x42535846))       # 'BSX2' / 'BSXC' / 'BSXF'
                # the same plan the library builds for the quadrant lut directly
                assert (nested == _host_plan(lut32, 2 * S, 4 * B, 2 * n_out, 32, dt, 1)).all()
                a = lib.BsmmArgs()
                a.blocks, a.bsize, a.dtype, a.N, a.axis, a.segments = B, 64, dt, 128, 1, S
                a.C, a.K = (CB * 64, KB * 64) if side == "fprop" else (KB * 64, CB * 64)
                assert L.bsmm_plan_attach(ctypes.byref(a), plan.ctypes.data_as(ip), plan.size, ctypes.c_void_p(4096)) == 0
                assert a.plan_magic == 0x42533634 and a.plan_inner == 0
                es = 2 if dt == lib.BF16 else 4
                op = lib.OP_FPROP if side == "fprop" else lib.OP_BPROP
                assert L.bsmm_prepared_bytes(op, ctypes.byref(a)) == B * 4096 * es
                ws = L.bsmm_workspace_bytes(op, ctypes.byref(a))
                assert ws >= B * 4096 * es
                a.prepared_w = 8192
                assert ws - L.bsmm_workspace_bytes(op, ctypes.byref(a)) == B * 4096 * es
        up = _host_updat_plan(t["updat_lut"], B, CB, KB, 64, lib.BF16, 1)
        assert up[0] == 0x42533634 and up[3] == 1 and up[2] == B and up[6] == up.size
        q = up[8:8 + 8 * B].reshape(4 * B, 2)
        for w, (c, k) in enumerate(t["updat_lut"]):
            assert [tuple(r) for r in q[4 * w:4 * w + 4]] == [(2 * c, 2 * k), (2 * c, 2 * k + 1), (2 * c + 1, 2 * k), (2 * c + 1, 2 * k + 1)]
        assert up[up[5]] == 0x42535532                                                          # nested: the streaming plan
        a = lib.BsmmArgs()
        a.blocks, a.bsize, a.dtype, a.N, a.axis, a.C, a.K = B, 64, lib.BF16, 128, 1, CB * 64, KB * 64
        assert L.bsmm_plan_attach(ctypes.byref(a), up.ctypes.data_as(ip), up.size, ctypes.c_void_p(4096)) == 0 and a.plan_inner == 1
        assert L.bsmm_workspace_bytes(lib.OP_UPDAT, ctypes.byref(a)) >= 4 * B * 4096                # the fp32 sums of the quadrants
        assert _host_updat_plan(t["updat_lut"], B, CB, KB, 64, lib.F32, 1) is None
        assert _host_plan(t["fprop"]["lut"], t["fprop"]["segments"], B, KB, 64, lib.BF16, 0) is None       # feature axis 0: not a reference configuration


def test_prepared_weights_entry_points_without_gpu(lib):
    """bsmm_prepared_bytes / bsmm_prepare_weights (the cached split of constant fp32 weights): only fp32 / bsize 32 with the 16-wide
    'BSXC' plan has something to prepare; everything else answers 0 / BSMM_ERR_UNSUPPORTED; bad arguments are refused."""
    import numpy as np
    from blocksparse_amd import lut as LT
    from blocksparse_amd.matmul import _host_plan
    L = lib.load()
    ip = ctypes.POINTER(ctypes.c_int32)
    lay = np.random.default_rng(1).random((8, 8)) < 0.4
    lay[0, :] = True
    t = LT.build_tables(lay)
    f = t["fprop"]
    a = lib.BsmmArgs()
    a.blocks, a.bsize, a.dtype, a.N, a.C, a.K, a.axis = t["blocks"], 32, lib.F32, 256, 256, 256, 1
    assert L.bsmm_prepared_bytes(lib.OP_FPROP, ctypes.byref(a)) == 0                      # no plan
    words = _host_plan(f["lut"], f["segments"], t["blocks"], 8, 32, lib.F32, 1)
    assert L.bsmm_plan_attach(ctypes.byref(a), words.ctypes.data_as(ip), words.size, ctypes.c_void_p(4096)) == 0
    assert L.bsmm_prepared_bytes(lib.OP_FPROP, ctypes.byref(a)) == 6 * t["blocks"] * 1024   # three bf16 pieces of every weight
    assert L.bsmm_prepared_bytes(lib.OP_BPROP, ctypes.byref(a)) == 6 * t["blocks"] * 1024
    assert L.bsmm_prepared_bytes(lib.OP_UPDAT, ctypes.byref(a)) == 0
    # feature axis 1 (round 4): the activations are split inside the kernel, so the workspace holds only W's pieces -- nothing at all once
    # they are prepared (bprop; fprop keeps room for the transposed copy of W the kernels without a plan read, should the call go there)
    wpieces, wt = 6 * t["blocks"] * 1024, 4 * t["blocks"] * 1024
    assert L.bsmm_workspace_bytes(lib.OP_FPROP, ctypes.byref(a)) == wpieces and L.bsmm_workspace_bytes(lib.OP_BPROP, ctypes.byref(a)) == wpieces
    a.prepared_w = 8192
    assert L.bsmm_workspace_bytes(lib.OP_FPROP, ctypes.byref(a)) == wt and L.bsmm_workspace_bytes(lib.OP_BPROP, ctypes.byref(a)) == 0
    # feature axis 0 keeps the pre-pass: the pieces of the activations (6 bytes per element) + W's unless prepared
    a.axis, a.prepared_w = 0, None
    ws_without = L.bsmm_workspace_bytes(lib.OP_BPROP, ctypes.byref(a))
    assert ws_without == 6 * a.N * a.C + wpieces
    a.prepared_w = 8192
    assert ws_without - L.bsmm_workspace_bytes(lib.OP_BPROP, ctypes.byref(a)) == wpieces                   # the call no longer splits W
    a.axis = 1
    one = ctypes.c_void_p(256)
    assert L.bsmm_prepare_weights(lib.OP_UPDAT, one, one, ctypes.byref(a)) == -1
    assert L.bsmm_prepare_weights(lib.OP_FPROP, None, one, ctypes.byref(a)) == -1
    a.dtype = lib.BF16
    assert L.bsmm_prepared_bytes(lib.OP_FPROP, ctypes.byref(a)) == 0
    assert L.bsmm_prepare_weights(lib.OP_FPROP, one, one, ctypes.byref(a)) == -2


def test_argument_validation_without_gpu(lib):
    L = lib.load()
    a = lib.BsmmArgs()
    one = ctypes.c_void_p(256)          # a non-null, 16-byte aligned dummy address: never dereferenced
    assert L.bsmm_fprop(one, one, one, None) == -1
    a.lut = 256
    a.blocks, a.N, a.C, a.K, a.segments = 4, 8, 64, 64, 2
    a.bsize = 7
    assert L.bsmm_fprop(one, one, one, ctypes.byref(a)) == -2
    a.bsize, a.axis = 32, 3
    assert L.bsmm_bprop(one, one, one, ctypes.byref(a)) == -2
    a.axis, a.dtype = 0, 9
    assert L.bsmm_fprop(one, one, one, ctypes.byref(a)) == -2
    a.dtype = lib.BF16
    a.C = 65                                     # not a multiple of bsize
    assert L.bsmm_fprop(one, one, one, ctypes.byref(a)) == -1
    a.C = 64
    a.pcount = 9
    arr = (ctypes.c_void_p * 1)(256)
    assert L.bsmm_updat(arr, arr, one, ctypes.byref(a)) == -1
    # workspace query is pure host arithmetic
    a.blocks, a.bsize, a.dtype = 10, 32, lib.BF16
    assert L.bsmm_workspace_bytes(lib.OP_FPROP, ctypes.byref(a)) == 10 * 32 * 32 * 2
    assert L.bsmm_workspace_bytes(lib.OP_BPROP, ctypes.byref(a)) == 0
    a.bsize = 8
    assert L.bsmm_workspace_bytes(lib.OP_FPROP, ctypes.byref(a)) == 0
    # fp32, bsize 32 with a plan: bf16 pieces of the weights (6 bytes per element) -- and of the activations on feature axis 0 (axis 1
    # splits them inside the kernel since round 4)
    a.bsize, a.dtype, a.axis, a.N, a.C, a.plan, a.plan_magic = 32, lib.F32, 1, 100, 64, 256, 0x42535843
    assert L.bsmm_workspace_bytes(lib.OP_FPROP, ctypes.byref(a)) == 6 * 10 * 1024
    a.axis = 0
    assert L.bsmm_workspace_bytes(lib.OP_FPROP, ctypes.byref(a)) == 6 * (100 * 64 + 10 * 1024)
    a.axis = 1
    assert L.bsmm_workspace_bytes(lib.OP_BPROP, ctypes.byref(a)) == 6 * 10 * 1024
    a.axis = 0
    assert L.bsmm_workspace_bytes(lib.OP_FPROP, ctypes.byref(a)) == 6 * (100 * 64 + 10 * 1024)
    a.plan = None
    assert L.bsmm_workspace_bytes(lib.OP_FPROP, ctypes.byref(a)) == 10 * 32 * 32 * 4      # no plan: the transposed copy of W
    a.axis, a.dtype = 1, lib.BF16
    assert L.bsmm_gate_grad(one, None, one, one, one, 4, 32, lib.F32, None) == -1
    assert L.bsmm_gate_grad(one, one, one, one, one, 4, 64, lib.F32, None) == -2
    assert L.bsmm_gate_weights(one, None, one, 4, 32, lib.BF16, 1, None) == -1            # (round 6) gated weight images: arguments, then support
    assert L.bsmm_gate_weights(one, one, one, 4, 32, lib.F32, 1, None) == -2
    assert L.bsmm_gate_weights(one, one, one, 4, 32, lib.BF16, 3, None) == -2


def _check_xcol_plan(plan, f, t, n_out):
    """xcol plan (build_xcol_plan): per group the union of input PAIRS (possibly rotated), per wave / half / step a weight id or -1."""
    assert plan[0] == 0x42535843 and plan[8] == n_out
    G, ngroups = int(plan[2]), int(plan[3])
    groups = plan[plan[5]:plan[6]].reshape(-1, 4)
    pairs = plan[plan[6]:plan[7]]
    assert len(groups) == ngroups == -(-n_out // G)
    got = set()
    for g, (so, ns, ob0, nob) in enumerate(groups):
        assert ob0 == g * G and nob == min(G, n_out - ob0)
        gp = pairs[so:so + ns]
        assert len(set(gp.tolist())) == ns                      # every pair once
        base = int(plan[7]) + 2 * G * so
        tab = plan[base:base + 2 * G * ns].reshape(2 * G, ns)
        for slot in range(2 * G):
            for tt in range(ns):
                w = int(tab[slot, tt])
                if w >= 0:
                    assert (slot >> 1) < nob
                    got.add((ob0 + (slot >> 1), 2 * int(gp[tt]) + (slot & 1), w))
        assert ns == 0 or (tab >= 0).any(axis=0).all()           # no empty step
    return got


def test_plan_builder_covers_every_block_once(lib):
    """bsmm_xprop_plan_build (host code in the library): every (in_block, w) entry of the lut appears exactly once,
    under the right output block -- for whichever schedule format the library emits for that axis."""
    import numpy as np
    from blocksparse_amd import lut as L
    from blocksparse_amd.matmul import _host_plan
    rng = np.random.default_rng(3)
    for CB, KB, dens, seg in ((40, 52, 0.3, True), (128, 128, 0.2, False), (5, 3, 1.0, False), (1, 1, 1.0, False), (7, 9, 0.5, False)):
        lay = rng.random((CB, KB)) < dens
        lay[0, :] = True
        t = L.build_tables(lay, segmented=seg)
        for axis in (0, 1):
            for side, n_out in (("fprop", KB), ("bprop", CB)):
                f = t[side]
                plan = _host_plan(f["lut"], f["segments"], t["blocks"], n_out, 32, lib.BF16, axis, lib.PLAN_XCOL_UNSTAGED)   # 'BSXC'
                got = _check_xcol_plan(plan, f, t, n_out)
                want = set()
                for ob, col in f["cols"]:
                    for c, w in col:
                        want.add((ob, c, w))
                assert got == want
                # fp32 (bsize 32): the split kernel (bsmm_xcols.h) walks the 16-wide 'BSXC' format of the 16-bit kernels, both axes
                # (the fp32-MFMA kernel xcol32f and its 'BSXF' plans were retired in round 4: BSMM_PLAN_F32_MFMA is ignored)
                assert (_host_plan(f["lut"], f["segments"], t["blocks"], n_out, 32, lib.F32, axis, lib.PLAN_F32_MFMA) == _host_plan(f["lut"], f["segments"], t["blocks"], n_out, 32, lib.F32, axis)).all()
                pf = _host_plan(f["lut"], f["segments"], t["blocks"], n_out, 32, lib.F32, axis)
                assert pf[0] == 0x42535843 and int(pf[2]) == 16
                assert _check_xcol_plan(pf, f, t, n_out) == want
                # bsize 16: the round-1 kernel ('BSX6' plans) was retired in round 4 -- the option that named it is ignored: the 'BSX7' plan
                # of the staged / list kernels either way (its contents: test_staged16_* below)
                p16 = _host_plan(f["lut"], f["segments"], t["blocks"], n_out, 16, lib.BF16, axis, lib.PLAN_XCOL_UNSTAGED)
                d16 = _host_plan(f["lut"], f["segments"], t["blocks"], n_out, 16, lib.BF16, axis)
                assert p16 is not None and p16[0] == 0x42535837 and (p16 == d16).all()
    # no plan kernels for fp32 at bsize 16, nor for bsize-8 grids that are not whole 32-feature blocks (tests/test_super8_plan.py)
    t = L.build_tables(np.ones((2, 2)))
    assert _host_plan(t["fprop"]["lut"], 2, 4, 2, 16, lib.F32, 1) is None
    assert _host_plan(t["fprop"]["lut"], 2, 4, 2, 8, lib.BF16, 1) is None


def test_staged_xcol_plan(lib):
    """'BSX2' plans (the default for bsize 32, 16-bit; bsmm_xcol_v2.h): every lut entry is multiplied exactly once, by the wave that
    owns its output block, from a slot of the phase's ring half that exactly one pair of DMA duties fills with that weight block;
    phases hold <= PH steps and <= WCAP blocks (PH = 2 / 3 / 4 by density: WCAP 23 / 15 / 7), waves <= 3 duties."""
    import numpy as np
    from blocksparse_amd import lut as L
    from blocksparse_amd.matmul import _host_plan
    rng = np.random.default_rng(5)
    import _parity as P
    for CB, KB, dens, ph in ((128, 128, 0.2, 2), (128, 128, 0.08, 3), (128, 128, 0.02, 4), (40, 52, 0.3, 2), (9, 35, 1.0, 2), (1, 1, 1.0, None),
                             (64, 16, 0.6, 2), (64, 48, 0.5, (3 << 8)), (64, 48, 0.5, (4 << 8)), (128, 128, "BA", None)):
        if dens == "BA":            # the reference's bench layout: hubs in neighbouring columns -> regrouped on feature axis 0 (version 3)
            lay = P.ba_layout(128, 14, seed=1) != 0
        else:
            lay = rng.random((CB, KB)) < dens
            lay[0, :] = True
        t = L.build_tables(lay)
        opt = ph if (ph is not None and ph >= 256) else 0
        for axis, side, n_out in ((1, "fprop", KB), (1, "bprop", CB), (0, "fprop", KB), (0, "bprop", CB)):
            f = t[side]
            plan = _host_plan(f["lut"], f["segments"], t["blocks"], n_out, 32, lib.BF16, axis, opt)
            assert plan[0] == 0x42535832 and int(plan[1]) == 3 and int(plan[2]) == 16 and plan[8] == n_out and plan[7] % 4 == 0
            cols = plan[plan[12]:plan[12] + 16 * int(plan[3])].reshape(-1, 16)
            regrouped = int(plan[13])
            assert int(plan[12]) + cols.size == plan.size and sorted(int(c) for c in cols.ravel() if c >= 0) == list(range(n_out))
            assert not (regrouped and axis == 1)                  # (the axis-1 epilogue stores whole rows of 16 adjacent output blocks)
            if dens == "BA":
                assert regrouped == (1 if axis == 0 else 0)
            WCAP, PH = int(plan[9]), int(plan[11])
            assert (PH, WCAP) in ((2, 23), (3, 15), (4, 7))
            if opt:
                assert PH == opt >> 8
            elif ph is not None and CB == 128:
                assert (PH == 2) if ph == 2 else (PH >= 3), (dens, PH)      # sparse layouts get longer phases
            nph_total = int(plan[4])
            groups = plan[plan[5]:plan[5] + 4 * int(plan[3])].reshape(-1, 4)
            px = plan[plan[6]:plan[6] + 2 * nph_total].reshape(-1, 2)
            tab = plan[plan[7]:plan[7] + nph_total * 128].reshape(-1, 16, 8)
            got = set()
            if not regrouped:
                assert sorted(int(x) for x in groups[:, 2]) == [16 * g for g in range(len(groups))]      # every group once ...
            assert all(groups[i, 1] >= groups[i + 1, 1] for i in range(len(groups) - 1))              # ... longest first
            for g, (po, nph, ob0, nob) in enumerate(groups):
                gc = [int(c) for c in cols[g]]
                assert gc[0] == ob0 and sum(c >= 0 for c in gc) == nob
                if not regrouped:
                    assert nob == min(16, n_out - ob0) and gc == [ob0 + v if v < nob else -1 for v in range(16)]
                for phs in range(po, po + nph):
                    pw = [int(px[phs, 0]) & 0xffffffff, int(px[phs, 1]) & 0xffffffff]
                    pairs = [(pw[u >> 1] >> (16 * (u & 1))) & 0xffff for u in range(4)]
                    assert pairs[0] != 0xffff and all(p == 0xffff for p in pairs[PH:])
                    slots = {}
                    for wave in range(16):
                        duties = [int(d) & 0xffffffff for d in tab[phs, wave, 2:5] if d != -1]
                        assert (tab[phs, wave, 5:] == 0).all()
                        for d in duties:
                            blk2, slot2 = d & 0x3ffffff, d >> 26
                            assert blk2 & 1 == slot2 & 1 and slot2 < 2 * WCAP
                            slots.setdefault(slot2 >> 1, []).append(blk2)
                    for sl, halves in slots.items():
                        assert sorted(halves) == [2 * (halves[0] >> 1), 2 * (halves[0] >> 1) + 1]
                    used = set()
                    for wave in range(16):
                        for byte in range(8):
                            sl = (int(tab[phs, wave, byte >> 2]) >> (8 * (byte & 3))) & 0xff
                            if sl == 0xff:
                                continue
                            u, half = byte >> 1, byte & 1
                            assert u < PH and gc[wave] >= 0 and pairs[u] != 0xffff and sl in slots and sl not in used
                            used.add(sl)
                            got.add((gc[wave], 2 * pairs[u] + half, slots[sl][0] >> 1))
                    assert used == set(slots) and len(used) <= WCAP
            want = {(ob, c, w) for ob, col in f["cols"] for c, w in col}
            assert got == want


def test_staged_xcol16_plan(lib):
    """'BSX7' plans (default for bsize 16, 16-bit; bsmm_xcol16_v2.h): every lut entry is multiplied exactly once, by the wave that
    owns its output block, from a slot of the phase's ring half that exactly one DMA duty fills with that weight block (a duty
    fetches two blocks into slots 2j, 2j+1); phases hold <= 2 steps and <= WCAP blocks, waves <= 3 duties.  Version 2: the block
    LIST of every (phase, wave) -- what xcol16_list_kernel walks -- names the same (position, slot) set as the slot bytes, column 0
    first, with the counts in word 10; the phase's request table (what four waves issue for everyone) repeats the duties' slot map."""
    import numpy as np
    from blocksparse_amd import lut as L
    from blocksparse_amd.matmul import _host_plan
    rng = np.random.default_rng(6)
    import _parity as P
    for CB, KB, dens in ((256, 256, 0.1), (40, 52, 0.3), (9, 70, 1.0), (1, 1, 1.0), (64, 16, 0.6), (7, 33, 0.5), (256, 256, "BA")):
        if dens == "BA":
            lay = P.ba_layout(256, 14, seed=1) != 0
        else:
            lay = rng.random((CB, KB)) < dens
            lay[0, :] = True
        t = L.build_tables(lay)
        for axis in (0, 1):
            for side, n_out in (("fprop", KB), ("bprop", CB)):
                f = t[side]
                plan = _host_plan(f["lut"], f["segments"], t["blocks"], n_out, 16, lib.BF16, axis)
                assert plan[0] == 0x42535837 and int(plan[1]) == 3 and int(plan[2]) == 32 and plan[8] == n_out and plan[7] % 4 == 0
                sect = plan[plan[11]:plan[11] + int(plan[4]) * 768].reshape(-1, 768)
                cols = plan[plan[12]:plan[12] + 32 * int(plan[3])].reshape(-1, 32)
                regrouped = int(plan[13])
                assert int(plan[11]) + sect.size == int(plan[12]) and int(plan[12]) + cols.size == plan.size
                assert sorted(int(c) for c in cols.ravel() if c >= 0) == list(range(n_out))
                if dens == "BA":
                    assert regrouped == 1
                lists, reqs = sect[:, :640].reshape(-1, 16, 40), sect[:, 640:].reshape(-1, 64, 2)
                WCAP = int(plan[9])
                groups = plan[plan[5]:plan[5] + 4 * int(plan[3])].reshape(-1, 4)
                px = plan[plan[6]:plan[6] + int(plan[4])]
                tab = plan[plan[7]:plan[7] + int(plan[4]) * 16 * 12].reshape(-1, 16, 12)
                got = set()
                for g, (po, nph, ob0, nob) in enumerate(groups):
                    gc = [int(c) for c in cols[g]]
                    assert gc[0] == ob0 and sum(c >= 0 for c in gc) == nob
                    if not regrouped:
                        assert ob0 == 32 * g and nob == min(32, n_out - ob0)
                    else:                # adjacent pairs stay together: a wave's two columns are neighbours
                        assert all(gc[v + 1] in (-1, gc[v] + 1) and (gc[v] < 0 or gc[v] % 2 == 0) for v in range(0, 32, 2))
                    for ph in range(po, po + nph):
                        quads = (int(px[ph]) & 0xffff, (int(px[ph]) >> 16) & 0xffff)
                        assert quads[0] != 0xffff
                        slots = {}
                        for wave in range(16):
                            for k in range(3):
                                a, b = int(tab[ph, wave, 4 + 2 * k]), int(tab[ph, wave, 5 + 2 * k])
                                if a == -1:
                                    continue
                                a &= 0xffffffff
                                pair = a >> 26
                                assert 2 * pair + 1 < WCAP and 2 * pair not in slots
                                slots[2 * pair], slots[2 * pair + 1] = a & 0x3ffffff, b
                            assert tab[ph, wave, 11] == 0
                        used = set()
                        for wave in range(16):
                            by_bytes = [[], []]
                            for byte in range(16):
                                sl = (int(tab[ph, wave, byte >> 2]) >> (8 * (byte & 3))) & 0xff
                                if sl == 0xff:
                                    continue
                                u, c, sub = byte >> 3, (byte >> 2) & 1, byte & 3
                                col = 2 * wave + c
                                assert gc[col] >= 0 and quads[u] != 0xffff and sl in slots and sl not in used
                                used.add(sl)
                                got.add((gc[col], 4 * quads[u] + sub, slots[sl]))
                                by_bytes[c].append((u, sub, sl))
                            n0, n1 = int(tab[ph, wave, 10]) & 0xff, (int(tab[ph, wave, 10]) >> 8) & 0xff
                            assert (n0, n1) == (len(by_bytes[0]), len(by_bytes[1])) and int(tab[ph, wave, 10]) >> 19 == 0
                            words = lists[ph, wave]
                            ent = [((int(words[2 * e]) >> 14) & 1, (int(words[2 * e]) >> 5) & 3, int(words[2 * e + 1]) // 512) for e in range(n0 + n1)]
                            for e in range(n0 + n1):
                                sub = (int(words[2 * e]) >> 5) & 3
                                assert int(words[2 * e]) == (sub << 5 | sub << 12 | ent[e][0] << 14) and int(words[2 * e + 1]) % 512 == 0
                            assert ent[:n0] == sorted(by_bytes[0]) and ent[n0:] == sorted(by_bytes[1]) and (words[2 * (n0 + n1):38] == 0).all()
                            # the row carries the wave's counts | role word of this phase and of the next one of the group
                            assert words[38] == tab[ph, wave, 10] and words[39] == (tab[ph + 1, wave, 10] if ph + 1 < po + nph else 0)
                        assert len(used) <= WCAP and set(slots) - used <= {max(slots)}      # at most the padded partner of the last pair
                        # four waves issue the next phase's requests: issuers 0..3, each once, none busier than a wave that does not issue
                        load = [(int(tab[ph, w_, 10]) & 0xff) + ((int(tab[ph, w_, 10]) >> 8) & 0xff) for w_ in range(16)]
                        role = [(int(tab[ph, w_, 10]) >> 16) & 0xff for w_ in range(16)]
                        assert sorted(r for r in role if r) == [1, 2, 3, 4]
                        assert max(l for l, r in zip(load, role) if r) <= min(l for l, r in zip(load, role) if not r)
                        # the request table names the same (slot -> weight block) map, pairs in slot order, and the phase's quads
                        npairs = int(reqs[ph, 48, 1])
                        assert int(reqs[ph, 48, 0]) == int(px[ph]) and npairs == (len(slots) + 1) // 2 and npairs <= 47
                        for k in range(npairs):
                            assert int(reqs[ph, k, 0]) == slots[2 * k] * 512 and int(reqs[ph, k, 1]) == slots[2 * k + 1] * 512
                        assert (reqs[ph, npairs:48] == 0).all() and (reqs[ph, 49:] == 0).all()
                want = {(ob, c, w) for ob, col in f["cols"] for c, w in col}
                assert got == want


def _check_rows_section(plan, off, t, CB, KB):
    """The 'BSU6' section behind a bsize-16 'BSUP' plan on feature axis 0 (bsmm_plan.h::build_updat16_rows_section; the row-owner
    weight-gradient kernel, csrc/bsmm_updat16_rows.h): every block in exactly one (item, wave, slot); a wave owns at most 2 block rows
    of its 32-row window and at most 12 blocks, every one of them in a row it owns and a column inside the window; no block row of a
    window has two owners."""
    import numpy as np
    sec = plan[off:]
    assert off % 4 == 0 and int(sec[0]) == 0x42535536 and int(sec[1]) == 3 and int(sec[2]) == 32 and int(sec[3]) in (32, 16)
    WK, nitems = int(sec[3]), int(sec[4])
    WAVES, ROWS, MAXB = int(sec[5]) & 255, (int(sec[5]) >> 8) & 255, int(sec[5]) >> 16
    assert (WAVES, ROWS, MAXB) == (16, 2, 12)
    wave_words = 2 + MAXB // 2 + MAXB
    isz = 4 + WAVES * wave_words
    assert int(sec[6]) == isz and sec.size == 8 + nitems * isz
    items = sec[8:].reshape(nitems, isz)
    seen, windows = set(), set()
    for it in items:
        c0, k0, n = int(it[0]), int(it[1]), int(it[2])
        assert c0 % 32 == 0 and k0 % WK == 0 and (c0, k0) not in windows
        windows.add((c0, k0))
        owned = set()
        cnt, duties, weights = 0, 0, []
        for v in range(WAVES):
            wv = it[4 + v * wave_words:4 + (v + 1) * wave_words]
            rows = [(int(np.uint32(wv[0])) >> (8 * r)) & 255 for r in range(ROWS)]
            live = [r for r in rows if r != 255]
            assert all(r < 32 for r in live) and len(set(live)) == len(live) and not (set(live) & owned)
            assert rows[:len(live)] == live                                  # (empty row slots last)
            owned |= set(live)
            nb, ii0, ni = int(wv[1]) & 255, (int(wv[1]) >> 8) & 255, (int(wv[1]) >> 16) & 255
            assert 0 <= nb <= MAXB and ii0 == duties                                 # the DMA duties tile 0 .. 2 WK - 1 in wave order
            duties += ni
            weights.append((3 * nb + ni, ni))
            used_slots = set()
            for j in range(MAXB):
                m = (int(np.uint32(wv[2 + j // 2])) >> (16 * (j & 1))) & 0xffff
                w = int(wv[2 + MAXB // 2 + j])
                if j >= nb:
                    assert m == 0 and w == 0
                    continue
                kidx, rs = m & 255, m >> 8
                assert kidx < WK and rs < len(live)
                assert tuple(t["updat_lut"][w]) == (c0 + rows[rs], k0 + kidx)
                assert w not in seen
                seen.add(w)
                used_slots.add(rs)
                cnt += 1
            assert used_slots == set(range(len(live)))                       # a row slot is only given to a row with blocks
        assert cnt == n and n > 0 and duties == 2 * WK
        # greedy deal: a wave that got a duty weighs at most one instruction more than the lightest wave (3 per block + 1 per instruction)
        assert max(w for w, ni in weights if ni > 0) <= min(w for w, _ in weights) + 1
    assert seen == set(range(t["blocks"]))


def test_updat_plan_covers_every_block_once(lib):
    """bsmm_updat_plan_build, the windowed format ('BSUP': bsize 16): every weight
    block appears in exactly one (item, wave, slot), inside its window, items are padded to a multiple of 8 (one list per XCD),
    every wave of an item has at most `nslots` blocks."""
    import numpy as np
    from blocksparse_amd import lut as L
    from blocksparse_amd.matmul import _host_updat_plan
    rng = np.random.default_rng(5)
    for CB, KB, dens in ((40, 52, 0.3), (70, 96, 0.1), (128, 128, 0.2), (5, 3, 1.0), (1, 1, 1.0), (16, 16, 1.0)):
        lay = rng.random((CB, KB)) < dens
        lay[0, 0] = True
        t = L.build_tables(lay)
        # bsize 32: BSMM_PLAN_WINDOW_* named the windowed kernels of round 1 (retired in round 4) -- the streaming plan with that window side
        for opt, same in ((lib.PLAN_WINDOW_8, lib.PLAN_STREAM_8), (lib.PLAN_WINDOW_16, lib.PLAN_STREAM_16), (lib.PLAN_WINDOW_16W, lib.PLAN_STREAM_16)):
            for axis in (0, 1):
                pw = _host_updat_plan(t["updat_lut"], t["blocks"], CB, KB, 32, lib.BF16, axis, opt)
                assert pw[0] == 0x42535532 and (pw == _host_updat_plan(t["updat_lut"], t["blocks"], CB, KB, 32, lib.BF16, axis, same)).all()
        for bsize, axis, opt in ((16, 1, 0), (16, 0, 0)):
            plan = _host_updat_plan(t["updat_lut"], t["blocks"], CB, KB, bsize, lib.BF16, axis, opt)
            assert plan[0] == 0x42535550 and plan[5] == t["blocks"]
            assert plan[2] == 16 and plan[7] == 8
            UW, MAXB, nitems, waves = int(plan[2]), int(plan[3]), int(plan[4]), int(plan[7])
            assert nitems % 8 == 0
            isz = 4 + waves * MAXB * 2
            items = plan[plan[6]:plan[6] + nitems * isz].reshape(nitems, isz)
            if axis == 0 and int(plan[8]) > 0:
                _check_rows_section(plan, int(plan[8]), t, CB, KB)
            else:
                assert int(plan[8]) == 0 and plan.size == plan[6] + nitems * isz        # (feature axis 1 / dense layouts: no 'BSU6' section)
            seen = set()
            for it in items:
                c0, k0 = int(it[0]), int(it[1])
                n, nslots = int(it[2]) & 0xffff, int(it[3]) & 0xffff
                cmask, kmask = (int(it[2]) >> 16) & 0xffff, (int(it[3]) >> 16) & 0xffff
                want_c, want_k = 0, 0
                slots = it[4:].reshape(waves, MAXB, 2)
                cnt = 0
                for v in range(waves):
                    per_wave = 0
                    for j in range(MAXB):
                        meta, w = int(slots[v, j, 0]), int(slots[v, j, 1])
                        if meta & 256:
                            c, k = c0 + (meta & 15), k0 + ((meta >> 4) & 15)
                            assert (meta & 15) < UW and ((meta >> 4) & 15) < UW
                            assert tuple(t["updat_lut"][w]) == (c, k)
                            want_c |= 1 << (meta & 15)
                            want_k |= 1 << ((meta >> 4) & 15)
                            assert w not in seen
                            seen.add(w)
                            cnt += 1
                            per_wave += 1
                            assert j < nslots
                            if meta & 512:      # "same X fragment as the previous slot"
                                assert j > 0 and (int(slots[v, j - 1, 0]) & 15) == (meta & 15)
                        else:
                            assert meta == 0
                    assert per_wave <= nslots
                assert cnt == n
                assert (cmask, kmask) == (want_c, want_k)      # the window rows / columns the kernels stage
            assert seen == set(range(t["blocks"]))
    assert _host_updat_plan(t["updat_lut"], t["blocks"], 16, 16, 32, lib.F32, 1) is None
    assert _host_updat_plan(t["updat_lut"], t["blocks"], 16, 16, 8, lib.F32, 0) is None
    assert _host_updat_plan(t["updat_lut"], t["blocks"], 16, 16, 8, lib.BF16, 0)[0] == 0x42535338   # 'BSS8' (tests/test_super8_plan.py)


def test_fp32_updat_takes_the_streaming_plan_on_axis1_without_gpu(lib):
    """fp32 / bsize 32 / feature axis 1 / one pair: bsmm_updat accepts the streaming 'BSU2' plan (the six bf16 piece products run as six
    pairs of one launch) and asks for the bf16 call's workspace plus the pieces of X and DY (6 bytes per element) and 16 bytes of flag; on feature axis 0,
    with two pairs, or without a plan the fp32 call needs no workspace; bsize 16 ('BSUP' plan) and bsize 8 ('BSS8') have the same route."""
    import numpy as np
    from blocksparse_amd import lut as LT
    from blocksparse_amd.matmul import _host_updat_plan
    L = lib.load()
    ip = ctypes.POINTER(ctypes.c_int32)
    lay = np.random.default_rng(2).random((24, 40)) < 0.3
    lay[0, :] = True
    t = LT.build_tables(lay)
    words = _host_updat_plan(t["updat_lut"], t["blocks"], 24, 40, 32, lib.BF16, 1)
    a = lib.BsmmArgs()
    a.blocks, a.bsize, a.dtype, a.N, a.C, a.K, a.axis, a.pcount = t["blocks"], 32, lib.F32, 512, 24 * 32, 40 * 32, 1, 1
    assert L.bsmm_workspace_bytes(lib.OP_UPDAT, ctypes.byref(a)) == 0                                   # no plan
    assert L.bsmm_plan_attach(ctypes.byref(a), words.ctypes.data_as(ip), words.size, ctypes.c_void_p(4096)) == 0
    b = lib.BsmmArgs()
    ctypes.memmove(ctypes.byref(b), ctypes.byref(a), ctypes.sizeof(a))
    b.dtype, b.pcount, b.flags = lib.BF16, 6, lib.FLAG_DW_SUMS
    inner = L.bsmm_workspace_bytes(lib.OP_UPDAT, ctypes.byref(b))
    need = L.bsmm_workspace_bytes(lib.OP_UPDAT, ctypes.byref(a))
    assert inner > 0 and need == (inner + 15) // 16 * 16 + 6 * 512 * (24 * 32 + 40 * 32) + 16    # (+ the non-finite flag)
    a.pcount = 2
    assert L.bsmm_workspace_bytes(lib.OP_UPDAT, ctypes.byref(a)) == 0                                   # two pairs: the kernels without a plan
    a.pcount, a.axis = 1, 0
    assert L.bsmm_workspace_bytes(lib.OP_UPDAT, ctypes.byref(a)) == 0                                   # feature axis 0 likewise
    # a call with a NULL operand list is refused before anything is launched
    a.axis = 1
    assert L.bsmm_updat(None, None, ctypes.c_void_p(256), ctypes.byref(a)) == -1
    # bsize 16 with its windowed 'BSUP' plan (feature axis 1) and bsize 8 with its 'BSS8' plan take the same route: the fp32 sums + the pieces
    w16 = _host_updat_plan(t["updat_lut"], t["blocks"], 24, 40, 16, lib.BF16, 1)
    c = lib.BsmmArgs()
    ctypes.memmove(ctypes.byref(c), ctypes.byref(a), ctypes.sizeof(a))
    c.bsize, c.C, c.K = 16, 24 * 16, 40 * 16
    assert L.bsmm_plan_attach(ctypes.byref(c), w16.ctypes.data_as(ip), w16.size, ctypes.c_void_p(4096)) == 0
    assert L.bsmm_workspace_bytes(lib.OP_UPDAT, ctypes.byref(c)) == (t["blocks"] * 256 * 4 + 15) // 16 * 16 + 6 * 512 * (24 * 16 + 40 * 16) + 16
    c.axis = 0
    assert L.bsmm_workspace_bytes(lib.OP_UPDAT, ctypes.byref(c)) == 0           # (feature axis 0: the per-block fp32 kernel, no workspace)
    c.axis = 1
    arr = (ctypes.c_void_p * 1)(256)
    c.lut = 4096
    assert L.bsmm_updat(arr, arr, ctypes.c_void_p(256), ctypes.byref(c)) == -3          # (accepted: it asks for that workspace)
    w8 = _host_updat_plan(t["updat_lut"], t["blocks"], 24, 40, 8, lib.BF16, 0)
    d = lib.BsmmArgs()
    ctypes.memmove(ctypes.byref(d), ctypes.byref(a), ctypes.sizeof(a))
    d.bsize, d.C, d.K, d.axis = 8, 24 * 8, 40 * 8, 0
    assert L.bsmm_plan_attach(ctypes.byref(d), w8.ctypes.data_as(ip), w8.size, ctypes.c_void_p(4096)) == 0
    assert L.bsmm_workspace_bytes(lib.OP_UPDAT, ctypes.byref(d)) > 6 * 512 * (24 * 8 + 40 * 8)


def test_streaming_updat_plan(lib):
    """'BSU2' plans (bsize 32, either feature axis: the default): every block in exactly one (item, wave, slot); a wave holds <= 4 blocks from
    <= 2 rows of the window, group 0 first; a window side of 16 for layouts up to ~22 % density, 8 above; hub rows split over waves;
    the two halves of the block rows alternate in the item list.  Round 6 (version 3): on feature axis 1 the blocks a window's 16 waves cannot hold
    are DIRECT blocks (own workgroups, no overflow items); feature axis 0 and PLAN_UPDAT_NO_DIRECT keep the overflow items."""
    import numpy as np
    from blocksparse_amd import lut as L
    from blocksparse_amd.matmul import _host_updat_plan
    rng = np.random.default_rng(7)
    cases = [(40, 52, 0.3), (128, 128, 0.2), (128, 128, 0.1), (128, 128, 0.5), (5, 3, 1.0), (1, 1, 1.0), (16, 16, 1.0), (33, 17, 0.05), (256, 256, 0.05)]
    saw_direct = set()
    for CB, KB, dens in cases:
        lay = rng.random((CB, KB)) < dens
        lay[0, 0] = True
        if CB >= 40:
            lay[3, :] = True                                   # a hub row: more blocks than a wave has slots
        t = L.build_tables(lay)
        for opt, axis in ((0, 1), (lib.PLAN_STREAM_16, 1), (lib.PLAN_STREAM_8, 1), (0, 0), (lib.PLAN_STREAM_16, 0)):
            plan = _host_updat_plan(t["updat_lut"], t["blocks"], CB, KB, 32, lib.BF16, axis, opt)
            if axis == 0:       # the same items as feature axis 1 without direct blocks (and with the same window side: 32 x 32 windows are feature axis 1's)
                force = opt or {16: lib.PLAN_STREAM_16, 8: lib.PLAN_STREAM_8}[int(plan[2])]
                assert (plan == _host_updat_plan(t["updat_lut"], t["blocks"], CB, KB, 32, lib.BF16, 1, force | lib.PLAN_UPDAT_NO_DIRECT)).all()
            assert plan[0] == 0x42535532 and plan[1] == 3 and plan[3] == 4 and plan[5] == t["blocks"] and plan[7] == 16 and plan[6] == 32
            WS, nitems = int(plan[2]), int(plan[4])
            if opt == 0:
                w32 = (-(-CB // 32)) * (-(-KB // 32))
                if axis == 1 and w32 >= 16 and t["blocks"] <= 38 * w32:
                    assert WS == 32                                    # very sparse layouts (round 3)
                    continue                                           # (the fifth index bits ride in the id words: decoded by the GPU tests)
                assert WS == (16 if t["blocks"] <= 56 * (-(-CB // 16)) * (-(-KB // 16)) else 8)
            else:
                assert WS == (16 if opt == lib.PLAN_STREAM_16 else 8)
            items = plan[plan[6]:plan[6] + nitems * 84].reshape(nitems, 4 + 16 * 5)
            ndir, offd = int(plan[28]), int(plan[29])
            assert int(plan[26]) == plan[6] + nitems * 84 and int(plan[30]) == 4 and int(plan[31]) == 0
            assert (ndir == 0 and offd == 0 and plan.size == int(plan[26]) + t["blocks"]) or \
                   (0 < ndir <= 64 and axis == 1 and offd % 4 == 0 and 0 <= offd - int(plan[26]) - t["blocks"] < 4 and plan.size == offd + 4 * ndir)
            bmap = plan[int(plan[26]):int(plan[26]) + t["blocks"]]   # block -> item << 8 | wave * 4 + slot (for the summing pass); -(2 + d): direct block d
            seen = set()
            for dd in range(ndir):                            # direct blocks: (block, c, k, 0), each once, none of them in an item
                w, c, k, z = (int(v) for v in plan[offd + 4 * dd:offd + 4 * dd + 4])
                assert z == 0 and tuple(t["updat_lut"][w]) == (c, k) and w not in seen and int(bmap[w]) == -2 - dd
                seen.add(w)
            if ndir:
                assert int(plan[8]) <= 2                      # partial-sum schedules only
                saw_direct.add((CB, KB, dens, opt))
            for ii, it in enumerate(items):
                c0, k0, n = int(it[0]), int(it[1]), int(it[2])
                assert c0 % WS == 0 and k0 % WS == 0 and 1 <= n <= 64
                cnt = 0
                for v in range(16):
                    m = int(it[4 + 5 * v]) & 0xffffffff
                    n0, n1 = m & 15, (m >> 4) & 15
                    assert n0 + n1 <= 4 and (n1 == 0 or n0 > 0)
                    rows = [(m >> 8) & 15] * n0 + [(m >> 12) & 15] * n1
                    if n1:
                        assert rows[0] != rows[-1]
                    for j in range(4):
                        w = int(it[4 + 5 * v + 1 + j])
                        if j < n0 + n1:
                            kidx = (m >> (16 + 4 * j)) & 15
                            assert rows[j] < WS and kidx < WS
                            assert tuple(t["updat_lut"][w]) == (c0 + rows[j], k0 + kidx)
                            assert w not in seen and int(bmap[w]) == (ii << 8 | (4 * v + j))
                            seen.add(w)
                            cnt += 1
                        else:
                            assert w == -1
                assert cnt == n
            assert seen == set(range(t["blocks"]))
            # the item sets (one per XCD group): contiguous lists that cover the items, each a compact patch of the window grid
            nsets = int(plan[8])
            assert nsets in (1, 2, 8)
            firsts = [int(plan[9 + 2 * s_]) for s_ in range(8)]
            counts = [int(plan[10 + 2 * s_]) for s_ in range(8)]
            assert sum(counts) == nitems and all(c == 0 for c in counts[nsets:])
            assert firsts[0] == 0 and all(firsts[i + 1] == firsts[i] + counts[i] for i in range(7))
            assert int(plan[25]) == (counts[0] if len(set(counts[:nsets])) == 1 else 0) and int(plan[27]) == max(counts)
            wc = -(-CB // WS)
            if nsets == 2:                                    # a split row: set 0 above it, set 1 below, item counts as equal as rows allow
                rows0 = set(int(it[0]) // WS for it in items[firsts[0]:firsts[0] + counts[0]])
                rows1 = set(int(it[0]) // WS for it in items[firsts[1]:firsts[1] + counts[1]])
                assert rows0 and rows1 and max(rows0) < min(rows1)
                per_row = np.bincount([int(it[0]) // WS for it in items], minlength=wc)
                best = min(abs(2 * int(per_row[:r].sum()) - nitems) for r in range(1, wc))
                assert abs(counts[0] - counts[1]) == best
        if (CB, KB, dens) == (128, 128, 0.2):
            p16 = _host_updat_plan(t["updat_lut"], t["blocks"], CB, KB, 32, lib.BF16, 1, 0)
            assert int(p16[4]) == 64 and 0 < int(p16[28]) <= 64   # one item per 16x16 window, what does not fit: direct blocks
            p16n = _host_updat_plan(t["updat_lut"], t["blocks"], CB, KB, 32, lib.BF16, 1, lib.PLAN_UPDAT_NO_DIRECT)
            assert 64 < int(p16n[4]) <= 72 and int(p16n[28]) == 0
    assert saw_direct


def test_host_class_surface():
    import numpy as np
    from blocksparse_amd import BlocksparseMatMul
    lay = np.array([[1, 0, 1], [1, 1, 0]])
    b = BlocksparseMatMul(lay, block_size=16, feature_axis=1)
    assert b.w_shape == (4, 16, 16) and b.C == 32 and b.K == 48 and b.blocks == 4
    assert b.i_shape(5) == (5, 32) and b.o_shape(5) == (5, 48) and b.flops == 4 * 16 * 16 * 2
    assert b.block_coord(0) == tuple(b.updat_lut[0])
    for bad in ((64, 0), (32, 2), (4, 0)):
        with pytest.raises(ValueError):
            BlocksparseMatMul(lay, block_size=bad[0], feature_axis=bad[1])
    import pickle
    b2 = pickle.loads(pickle.dumps(b))
    np.testing.assert_array_equal(b2.fprop_lut, b.fprop_lut)
    # host NumPy reference methods agree with the oracle
    from oracle import bsmm_oracle as orc
    t = orc.build_layout_luts(lay, 16)
    rng = np.random.default_rng(0)
    W = rng.normal(size=b.w_shape); X = rng.normal(size=b.i_shape(6)); E = rng.normal(size=b.o_shape(6))
    np.testing.assert_allclose(b.fprop_test(X, W), orc.fprop(t, X, W, 1), atol=1e-12)
    np.testing.assert_allclose(b.bprop_test(E, W), orc.bprop(t, E, W, 1), atol=1e-12)
    np.testing.assert_allclose(b.updat_test(X, E), orc.updat(t, X, E, 1), atol=1e-12)
    b0 = BlocksparseMatMul(lay, block_size=16, feature_axis=0)
    np.testing.assert_allclose(b0.fprop_test(X.T.copy(), W), orc.fprop(t, X.T.copy(), W, 0), atol=1e-12)
    np.testing.assert_allclose(b0.bprop_test(E.T.copy(), W), orc.bprop(t, E.T.copy(), W, 0), atol=1e-12)
    np.testing.assert_allclose(b0.updat_test(X.T.copy(), E.T.copy()), orc.updat(t, X.T.copy(), E.T.copy(), 0), atol=1e-12)


def test_cpu_tensors_are_rejected_loudly():
    import numpy as np
    import torch
    from blocksparse_amd import BlocksparseMatMul
    b = BlocksparseMatMul(np.ones((2, 2)), block_size=8)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        b.fprop(torch.zeros(b.i_shape(4)), torch.zeros(b.w_shape))


def test_plan_attach_descriptor_and_host_side_rejection(lib):
    """bsmm_plan_attach reads the descriptor from the HOST words; a call whose plan does not belong to it is refused on the
    host (BSMM_ERR_ARG) before anything is launched -- no GPU needed to see that."""
    import numpy as np
    from blocksparse_amd import lut as LT
    from blocksparse_amd.matmul import _host_plan, _host_updat_plan
    L = lib.load()
    ip = ctypes.POINTER(ctypes.c_int32)
    lay = np.random.default_rng(1).random((24, 40)) < 0.3
    lay[0, :] = True
    t = LT.build_tables(lay)
    f = t["bprop"]
    dev = ctypes.c_void_p(4096)                     # stand-in for the device copy: never dereferenced on the host

    def attach(words):
        a = lib.BsmmArgs()
        assert L.bsmm_plan_attach(ctypes.byref(a), words.ctypes.data_as(ip), words.size, dev) == 0
        return a
    xp = _host_plan(f["lut"], f["segments"], t["blocks"], 24, 32, lib.BF16, 1)
    a = attach(xp)                                  # default for bsize 32 / 16-bit / axis 1: the staged kernel's 'BSX2' plan
    assert (a.plan_magic, a.plan_width, a.plan_waves, a.plan_items) == (0x42535832, 16, 16, 0) and a.plan_inner in (2, 3, 4) and a.plan == 4096   # plan_inner: steps per phase
    a = attach(_host_plan(f["lut"], f["segments"], t["blocks"], 24, 32, lib.BF16, 1, lib.PLAN_XCOL_UNSTAGED))
    assert (a.plan_magic, a.plan_width, a.plan_waves, a.plan_items, a.plan_inner) == (0x42535843, 16, 16, 0, 0)
    a = attach(_host_plan(f["lut"], f["segments"], t["blocks"], 24, 32, lib.BF16, 0))
    assert a.plan_magic == 0x42535832               # feature_axis 0: the same staged plan
    a = attach(_host_plan(f["lut"], f["segments"], t["blocks"], 24, 32, lib.BF16, 1, lib.PLAN_XCOL_NARROW))
    assert (a.plan_width, a.plan_waves) == (8, 8)
    a = attach(_host_plan(f["lut"], f["segments"], t["blocks"], 24, 32, lib.F32, 1, lib.PLAN_F32_MFMA))
    assert (a.plan_magic, a.plan_width) == (0x42535843, 16)      # fp32: the split kernel's 'BSXC' plan (the fp32-MFMA kernel was retired)
    up = _host_updat_plan(t["updat_lut"], t["blocks"], 24, 40, 16, lib.BF16, 1)
    a = attach(up)
    assert (a.plan_magic, a.plan_width, a.plan_waves, a.plan_items) == (0x42535550, 16, 8, int(up[4]))      # bsize 16: the windowed 'BSUP' plan
    up = _host_updat_plan(t["updat_lut"], t["blocks"], 24, 40, 32, lib.BF16, 1, lib.PLAN_WINDOW_8)
    a = attach(up)
    assert (a.plan_magic, a.plan_width, a.plan_waves, a.plan_items) == (0x42535532, 8, 16, int(up[4]))       # bsize 32: streaming, 8x8 windows
    s8 = _host_updat_plan(t["updat_lut"], t["blocks"], 24, 40, 8, lib.BF16, 0)
    a = attach(s8)
    # bsize 8: the nested plan's width | its format (2 = the streaming 'BSU2' plan, round 3) << 8 | its descriptor word << 11
    assert (a.plan_magic, a.plan_width) == (0x42535338, int(s8[2])) and a.plan_inner & 0xff in (8, 16) and (a.plan_inner >> 8) & 7 == 2 and a.plan_waves == 16 and a.plan_items > 0
    # garbage / truncated / foreign-version words are not a plan
    bad = xp.copy(); bad[1] += 1
    b = lib.BsmmArgs()
    assert L.bsmm_plan_attach(ctypes.byref(b), bad.ctypes.data_as(ip), bad.size, dev) == -1 and not b.plan
    assert L.bsmm_plan_attach(ctypes.byref(b), xp.ctypes.data_as(ip), 4, dev) == -1
    assert L.bsmm_plan_attach(ctypes.byref(b), xp.ctypes.data_as(ip), xp.size, None) == 0 and not b.plan     # detach

    # a call with a plan of the wrong kind is refused before any launch (bprop: no pre-pass, so the check is the first thing)
    one = ctypes.c_void_p(256)
    a = attach(up)                                   # updat plan ...
    a.lut = 256
    a.blocks, a.N, a.C, a.K, a.segments, a.bsize, a.axis, a.dtype = t["blocks"], 256, 40 * 32, 24 * 32, 24, 32, 1, lib.BF16
    a.flags = lib.FLAG_FORCE_PLAN
    assert L.bsmm_bprop(one, one, one, ctypes.byref(a)) == -1          # ... handed to bprop
    a = attach(xp)
    a.lut = 256
    a.blocks, a.N, a.C, a.K, a.segments, a.bsize, a.axis, a.dtype = t["blocks"], 256, 40 * 32, 24 * 32, 24, 32, 1, lib.BF16
    a.flags = lib.FLAG_FORCE_PLAN
    a.plan_width = 5                                 # descriptor of a width no kernel has
    assert L.bsmm_bprop(one, one, one, ctypes.byref(a)) == -1
    a.plan_width, a.bsize, a.C, a.K = 16, 16, 40 * 16, 24 * 16      # bsize-32 plan on a bsize-16 call
    assert L.bsmm_bprop(one, one, one, ctypes.byref(a)) == -1
    a = attach(xp)
    a.lut, a.pcount = 256, 1
    a.blocks, a.N, a.C, a.K, a.bsize, a.axis, a.dtype = t["blocks"], 256, 24 * 32, 40 * 32, 32, 1, lib.BF16
    arr = (ctypes.c_void_p * 1)(256)
    assert L.bsmm_updat(arr, arr, one, ctypes.byref(a)) == -1          # xprop plan handed to updat


def test_library_reads_no_environment_and_keeps_no_switches(lib):
    """include/bsmm.h promises a stateless library: no getenv, no process-wide kernel switch in the sources."""
    src = ""
    for f in ("bsmm_api.hip", "bst_api.hip", "bsmm_dist.hip"):
        src += open(os.path.join(ROOT, "blocksparse_amd", "csrc", f)).read()
    assert "getenv" not in src and "g_variant" not in src and "static bool attr_set" not in src
    assert not hasattr(ctypes.CDLL(lib.LIB_PATH), "bsmm_set_kernel_variant")


def test_dist_header_symbols_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "bsmm_dist.h")).read()
    declared = set(re.findall(r"\b(bsmm_dist_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(lib.DIST_SYMBOLS), declared ^ set(lib.DIST_SYMBOLS)
    raw = ctypes.CDLL(lib.LIB_PATH)
    for s in declared:
        getattr(raw, s)
    L = lib.load()
    assert L.bsmm_dist_unique_id(None) == -1                       # argument checks come before RCCL is touched
    assert L.bsmm_dist_allreduce_begin(None, None, 0, 0, None) == -1
    assert L.bsmm_dist_destroy(None) == 0
