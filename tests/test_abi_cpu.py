"""The C-ABI library loads without a GPU and exports every symbol include/shotvae_hip.h declares."""
import ctypes
import os
import re

from shot_vae_amd import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "shotvae_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sv_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    if not os.path.exists(L.LIB_PATH):
        L.build()
    lib = ctypes.CDLL(L.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 28
    for n in names:
        assert hasattr(lib, n), "missing export " + n
    assert set(names) == set(L.EXPORTS), set(names) ^ set(L.EXPORTS)
    assert L.lib().sv_version() == 1


def test_struct_layout_matches_header():
    assert ctypes.sizeof(L.SvPhase) == 72
    assert ctypes.sizeof(L.SvGeom) == 72 + 4 * 72
    assert ctypes.sizeof(L.SvIgemmArgs) == 17 * 8
    assert ctypes.sizeof(L.SvBnBranch) == 48


def _tile_program(hv):
    items = (ctypes.c_int * 540)()
    waits = (ctypes.c_int * 5)()
    assert L.lib().sv_debug_wgrad_tile_program(hv, items, waits) == 0
    return [[items[3 * i + j] for j in range(3) if items[3 * i + j]] for i in range(180)], list(waits)


def test_wide_wgrad_tile_program_is_consistent():
    """The compile-time schedule of wgrad3x3w_kernel (wgrad3x3.hip, make_wsched): every transform step exactly once and in
    order, vector loads after the vector's last step, VMEM instructions only in gaps without an LDS read and >= 6 MFMAs
    apart, and the hand-counted vmcnt values equal to the number of VMEM instructions issued in between."""
    has_read = lambda g: (g % 5) < 2 or ((g % 5) == 2 and g // 5 < 8)
    for hv in (3, 4):
        gaps, waits = _tile_program(hv)
        flat = [(gi, c) for gi, items in enumerate(gaps) for c in items]
        steps = [c - 1000 for _, c in flat if c // 1000 == 1]
        assert steps == list(range(41 * hv)), "transform steps out of order / missing"
        assert all(gi < 135 for gi, c in flat if c // 1000 in (1, 4)), "transform must finish before the barrier"
        vmem = [(gi, c) for gi, c in flat if c // 1000 in (2, 3)]
        assert sorted(c for _, c in vmem if c // 1000 == 2) == [2000 + k for k in range(11)]
        assert sorted(c for _, c in vmem if c // 1000 == 3) == [3000 + v for v in range(hv)]
        for gi, c in vmem:
            assert not has_read(gi % 45) and gaps[gi] == [c], "a VMEM instruction owns a read-free gap"
        for (g0, _), (g1, _) in zip(vmem, vmem[1:]):
            assert g1 - g0 >= 4
        # D4..D10 land in the other stage before the barrier, D0..D3 start after it
        assert all((gi < 135) == (c - 2000 >= 4) for gi, c in vmem if c // 1000 == 2)
        pos = {c: i for i, (_, c) in enumerate(flat)}
        order = [c for _, c in vmem]
        for v in range(hv):
            assert pos[3000 + v] > pos[1000 + 41 * v + 40], "reload only after the vector's store"
            assert flat[pos[4000 + v] + 1][1] == 1000 + 41 * v, "the wait directly precedes the vector's first step"
            # steady state: VMEM issued after the load (previous iteration) + before the wait (this iteration)
            after = len(order) - 1 - order.index(3000 + v)
            before = sum(1 for _, c in flat[:pos[4000 + v]] if c // 1000 in (2, 3))
            assert waits[v] == after + before, (hv, v, waits[v], after, before)
        last_dma = max(i for i, (gi, c) in enumerate(flat) if c // 1000 == 2 and gi < 135)
        assert waits[4] == sum(1 for gi, c in flat[last_dma + 1:] if gi < 135 and c // 1000 in (2, 3))
        # per gap: one step next to an LDS read, two otherwise
        for gi, items in enumerate(gaps[:135]):
            n = sum(1 for c in items if c // 1000 == 1)
            assert n <= (1 if has_read(gi % 45) else 2)
