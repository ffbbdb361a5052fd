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


def test_wide_conv_chunk_program_is_consistent():
    """The compile-time schedule of conv3x3x_kernel (conv3x3x.hip, make_xsched): the 246 BatchNorm steps exactly once, in
    order and finished before the barrier after tap 7; every VMEM instruction alone in a gap without fragment reads; halo
    registers re-loaded after their vector's store, the coefficients after the last step; the weight slices inside the
    window the six-slot ring allows; and the hand-counted vmcnt values: register-load waits = the younger register loads,
    DMA waits = every younger VMEM instruction (steady state)."""
    items = (ctypes.c_int * (180 * 6))()
    waits = (ctypes.c_int * 10)()
    assert L.lib().sv_debug_conv_chunk_program(items, waits) == 0
    gaps = [[items[6 * i + j] for j in range(6) if items[6 * i + j]] for i in range(180)]
    waits = list(waits)
    has_read = lambda m: (m & 1) == 1 or m in (8, 18)
    flat = [(gi, c) for gi, it in enumerate(gaps) for c in it]
    steps = [c - 1000 for _, c in flat if 1000 <= c < 2000]
    assert steps == list(range(41 * 6))
    assert all(gi < 160 for gi, c in flat if 1000 <= c < 2000 or 4000 <= c < 5000)
    vmem = [(gi, c) for gi, c in flat if 2000 <= c < 4000]
    assert sorted(c for _, c in vmem if c < 3000) == [2000 + k for k in range(27)]
    assert sorted(c for _, c in vmem if 3000 <= c < 3500) == [3000 + v for v in range(6)]
    assert sorted(c for _, c in vmem if c >= 3500) == [3500 + q for q in range(4)]
    for gi, c in vmem:
        assert not has_read(gi % 20), "VMEM next to a fragment read"
        assert sum(1 for x in gaps[gi] if 2000 <= x < 4000) == 1
    for (g0, _), (g1, _) in zip(vmem, vmem[1:]):
        assert g1 - g0 >= 2, "at most one VMEM instruction per two MFMAs"
    pos = {c: i for i, (_, c) in enumerate(flat)}
    gap_of = {c: gi for gi, c in flat}
    # ring windows: slice s of step k = 5 + s (this chunk's taps 5..8, the next chunk's 0..4 = steps 9..13) may be written
    # after the barrier that follows the fragment reads of step k - 6 (those happen during step k - 7), and must be awaited
    # at a barrier before step k - 1
    barrier_after = lambda step: next(b for b in (1, 4, 7, 10, 13) if b >= step)
    for sl in range(9):
        k = 5 + sl
        first_gap = min(gap_of[2000 + 3 * sl + i] for i in range(3))
        assert first_gap // 20 > barrier_after(k - 7) if k - 7 >= 0 else True, (sl, first_gap)
    order = [c for _, c in vmem]
    # waits for register loads count only the younger REGISTER loads (LDS-DMA retires out of order with respect to them)
    reg_since_wrap = lambda code, until: (sum(1 for c in order[order.index(code) + 1:] if c >= 3000) +
                                          sum(1 for gi, c in vmem if pos[c] < until and c >= 3000))
    since = lambda code, until_gap: sum(1 for gi, c in vmem[order.index(code) + 1:] if gi < until_gap)
    for v in range(6):
        assert pos[3000 + v] > pos[1000 + 41 * v + 40], "reload only after the vector's store"
        assert waits[v] == reg_since_wrap(3000 + v, pos[4000 + v])
    assert pos[3500] > pos[1000 + 41 * 6 - 1]
    assert waits[6] == reg_since_wrap(3503, pos[4500])
    assert waits[7] == since(2000 + 3 * 0 + 2, 40)          # (this, 5), issued in tap 0, awaited after tap 1
    assert waits[8] == since(2000 + 3 * 3 + 2, 100)         # (this, 8), tap 3 -> barrier after tap 4
    assert waits[9] == since(2000 + 3 * 6 + 2, 160)         # (next, 2), tap 6 -> barrier after tap 7
    assert gap_of[5000] == 160, "the stage flip precedes the first fragment read of tap 8"
