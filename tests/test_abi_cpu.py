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
    assert L.lib().sv_version() == 3


def test_struct_layout_matches_header():
    assert ctypes.sizeof(L.SvPhase) == 72
    assert ctypes.sizeof(L.SvGeom) == 72 + 4 * 72
    assert ctypes.sizeof(L.SvIgemmArgs) == 18 * 8
    assert ctypes.sizeof(L.SvWgradArgs) == 10 * 8
    assert ctypes.sizeof(L.SvBnBranch) == 48
    assert ctypes.sizeof(L.SvRepackJob) == 64


def _tile_program(hv):
    items = (ctypes.c_int * 540)()
    waits = (ctypes.c_int * 5)()
    assert L.lib().sv_debug_wgrad_tile_program(hv, items, waits) == 0
    return [[items[3 * i + j] for j in range(3) if items[3 * i + j]] for i in range(180)], list(waits)


def test_wide_wgrad_tile_program_is_consistent():
    """The compile-time schedule of wgrad3x3w_kernel (wgrad3x3.hip, make_wsched): every transform step exactly once and in
    order before the barrier, VMEM instructions only in gaps without an LDS read and >= 6 MFMAs apart, NO counted vmcnt
    wait (the halo registers are double-buffered; one vmcnt(0) in front of the barrier covers both kinds of VMEM
    instruction), the loads of the second register set and D4..D10 before that barrier with >= 40 MFMAs to land, and the
    copy of the second set into the first, dword by dword, behind it."""
    has_read = lambda g: (g % 5) < 2 or ((g % 5) == 2 and g // 5 < 8)
    for hv in (3, 4):
        gaps, waits = _tile_program(hv)
        flat = [(gi, c) for gi, items in enumerate(gaps) for c in items]
        steps = [c - 1000 for _, c in flat if c // 1000 == 1]
        assert steps == list(range(41 * hv)), "transform steps out of order / missing"
        assert all(gi < 135 for gi, c in flat if c // 1000 == 1), "transform must finish before the barrier"
        assert not any(c // 1000 == 4 for _, c in flat), "no hand-counted waits"
        assert waits[:4] == [0, 0, 0, 0]
        vmem = [(gi, c) for gi, c in flat if c // 1000 in (2, 3)]
        assert sorted(c for _, c in vmem if c // 1000 == 2) == [2000 + k for k in range(11)]
        assert sorted(c for _, c in vmem if c // 1000 == 3) == [3000 + v for v in range(hv)]
        for gi, c in vmem:
            assert not has_read(gi % 45) and gaps[gi] == [c], "a VMEM instruction owns a read-free gap"
        for (g0, _), (g1, _) in zip(vmem, vmem[1:]):
            assert g1 - g0 >= 4
        # D4..D10 and every register load are issued before the barrier (behind gap 134) that awaits them with vmcnt(0);
        # D0..D3 of the tile after next start after it
        assert all((gi < 135) == (c - 2000 >= 4) for gi, c in vmem if c // 1000 == 2)
        assert all(gi < 135 for gi, c in vmem if c // 1000 == 3)
        last = max(gi for gi, c in vmem if gi < 135)
        assert waits[4] == last and 135 - last >= 40
        # the second register set moves into the first after the barrier: every dword once, after nothing reads the first
        copies = [(gi, c - 5000) for gi, c in flat if c // 1000 == 5]
        assert [c for _, c in copies] == list(range(4 * hv)) and all(gi >= 135 for gi, _ in copies)
        # per gap: one step next to an LDS read, two otherwise
        for gi, items in enumerate(gaps):
            n = sum(1 for c in items if c // 1000 in (1, 5))
            assert n <= (1 if has_read(gi % 45) else 2)


def test_wide_conv_chunk_program_is_consistent():
    """The compile-time schedule of conv3x3x_kernel (conv3x3x.hip, make_xsched): the 246 BatchNorm steps exactly once, in
    order and finished before the barrier after tap 7; every VMEM instruction alone in a gap without fragment reads; halo
    registers re-loaded after their vector's store, the coefficients after the last step; the weight slices inside the
    window the six-slot ring allows; and the SAME-KIND rule for every hand-counted vmcnt: a wait for register loads = the
    younger register loads only, a wait for LDS-DMA (the three barriers) = the younger LDS-DMA only."""
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
    # at a barrier before the fragment reads of step k (which happen during step k - 1)
    barriers = (1, 4, 7, 10, 13)
    barrier_after = lambda step: next(b for b in barriers if b >= step)
    for sl in range(9):
        k = 5 + sl
        first_gap = min(gap_of[2000 + 3 * sl + i] for i in range(3))
        last_gap = max(gap_of[2000 + 3 * sl + i] for i in range(3))
        if k - 7 >= 0:
            assert first_gap // 20 > barrier_after(k - 7), (sl, first_gap)
        assert any(last_gap // 20 <= b < k - 1 for b in barriers), "awaited at a barrier before its first read"
    order = [c for _, c in vmem]
    reg_since_wrap = lambda code, until: (sum(1 for c in order[order.index(code) + 1:] if c >= 3000) +
                                          sum(1 for gi, c in vmem if pos[c] < until and c >= 3000))
    for v in range(6):
        assert pos[3000 + v] > pos[1000 + 41 * v + 40], "reload only after the vector's store"
        assert waits[v] == reg_since_wrap(3000 + v, pos[4000 + v])
    assert pos[3500] > pos[1000 + 41 * 6 - 1]
    assert waits[6] == reg_since_wrap(3503, pos[4500])
    # the barriers after taps 1, 4, 7 (before gaps 40, 100, 160) need every DMA issued before them: zero younger DMAs
    assert waits[7:10] == [0, 0, 0]
    # ... so they drain the register loads in flight as well: none is issued less than 14 MFMAs before a barrier, and none
    # behind the last barrier of the chunk (the compiler is told there that the registers have arrived)
    for gi, c in vmem:
        if c >= 3000:
            nb = min(b for b in (40, 100, 160) if b > gi) if gi < 160 else None
            assert nb is not None and nb - gi >= 14, (c, gi)
    assert gap_of[5000] == 160, "the stage flip precedes the first fragment read of tap 8"
