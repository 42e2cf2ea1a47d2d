"""The 'BSU6' section of the bsize-16 weight-gradient plan on feature axis 0 (csrc/bsmm_plan.h::build_updat16_rows_section, the row-owner
kernel's work items) over random layouts, host only: every block in exactly one (item, wave, slot), at most 2 block rows and 12 blocks per
wave, the DMA duties tile the slab, and the section disappears exactly when a 16-column window cannot be dealt (dense layouts) or the option
PLAN_UPDAT16_WINDOWED / feature axis 1 asks for the plain 'BSUP' plan.  The validator is tests/test_abi.py::_check_rows_section."""
import numpy as np
import pytest

from test_abi import _check_rows_section, lib  # noqa: F401  (the `lib` fixture)


def _plan(lib, lay, axis=0, opt=0):
    from blocksparse_amd import lut as L
    from blocksparse_amd.matmul import _host_updat_plan
    t = L.build_tables(lay)
    return t, _host_updat_plan(t["updat_lut"], t["blocks"], lay.shape[0], lay.shape[1], 16, lib.BF16, axis, opt)


def test_rows_section_over_random_layouts(lib):
    rng = np.random.default_rng(2024)
    with_section = without = 0
    for it in range(40):
        CB, KB = int(rng.integers(1, 140)), int(rng.integers(1, 140))
        dens = float(rng.choice([0.02, 0.05, 0.1, 0.15, 0.2, 0.3, 0.6, 1.0]))
        lay = rng.random((CB, KB)) < dens
        lay[rng.integers(0, CB), rng.integers(0, KB)] = True
        t, plan = _plan(lib, lay)
        assert int(plan[0]) == 0x42535550 and int(plan[1]) == 6
        off = int(plan[8])
        if off > 0:
            _check_rows_section(plan, off, t, CB, KB)
            with_section += 1
            # the window is 32 columns wide unless some wave would own more than 12 blocks
            assert int(plan[off + 3]) in (32, 16)
        else:
            without += 1
            # no section: some pair of block rows of some 32 x 16 window holds more than 12 blocks however the rows are dealt -- certainly
            # true when a single block row of a window does
            rows_max = max(int(lay[r, c0:c0 + 16].sum()) for r in range(CB) for c0 in range(0, KB, 16))
            rows_in_window = min(CB, 32)
            assert rows_max > 12 or rows_in_window > 16, (CB, KB, dens, rows_max)
        # the option and the other feature axis give the plain plan
        for axis, opt in ((0, lib.PLAN_UPDAT16_WINDOWED), (1, 0)):
            _, p2 = _plan(lib, lay, axis, opt)
            assert int(p2[8]) == 0 and p2.size == int(p2[6]) + int(p2[4]) * (4 + int(p2[7]) * int(p2[3]) * 2)
            assert np.array_equal(p2[:8], plan[:8]) and np.array_equal(p2[12:], plan[12:p2.size])
    assert with_section >= 20 and without >= 3, (with_section, without)


def test_rows_section_descriptor(lib):
    """bsmm_plan_attach packs the section's window width / item count into bits 8.. of plan_width / plan_waves and its offset into plan_inner;
    the workspace of a 16-bit call with the section holds eight images of the sums (one per part of the minibatch), without it one."""
    import ctypes
    L = lib.load()
    ip = ctypes.POINTER(ctypes.c_int32)
    lay = np.random.default_rng(3).random((70, 96)) < 0.1
    t, plan = _plan(lib, lay)
    off = int(plan[8])
    assert off > 0
    a = lib.BsmmArgs()
    a.blocks, a.bsize, a.dtype, a.N, a.C, a.K, a.axis, a.pcount = t["blocks"], 16, lib.BF16, 2048, 70 * 16, 96 * 16, 0, 1
    assert L.bsmm_plan_attach(ctypes.byref(a), plan.ctypes.data_as(ip), plan.size, ctypes.c_void_p(4096)) == 0
    assert a.plan_inner == off and (a.plan_width & 255, a.plan_width >> 8) == (16, int(plan[off + 3])) and (a.plan_waves & 255, a.plan_waves >> 8) == (8, int(plan[off + 4]))
    assert L.bsmm_workspace_bytes(lib.OP_UPDAT, ctypes.byref(a)) == 8 * t["blocks"] * 256 * 4
    _, p2 = _plan(lib, lay, 0, lib.PLAN_UPDAT16_WINDOWED)
    assert L.bsmm_plan_attach(ctypes.byref(a), p2.ctypes.data_as(ip), p2.size, ctypes.c_void_p(4096)) == 0
    assert a.plan_inner == 0 and a.plan_width == 16 and a.plan_waves == 8
    assert L.bsmm_workspace_bytes(lib.OP_UPDAT, ctypes.byref(a)) == t["blocks"] * 256 * 4
    # a corrupted section offset is refused
    bad = plan.copy(); bad[8] = plan.size - 4
    assert L.bsmm_plan_attach(ctypes.byref(a), bad.ctypes.data_as(ip), bad.size, ctypes.c_void_p(4096)) != 0
