"""The C-ABI library loads without a GPU and exports every symbol include/shotvae_hip.h declares."""
import ctypes
import os
import re
import sys

import pytest

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
    assert L.lib().sv_version() == L.ABI_VERSION == 8


def test_struct_layout_matches_header():
    assert ctypes.sizeof(L.SvPhase) == 72
    assert ctypes.sizeof(L.SvGeom) == 72 + 4 * 72
    assert ctypes.sizeof(L.SvIgemmArgs) == 28 * 8
    assert ctypes.sizeof(L.SvWgradArgs) == 10 * 8
    assert ctypes.sizeof(L.SvBwd3x3Args) == 28 * 8
    assert ctypes.sizeof(L.SvParamJob) == 14 * 8
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


def test_host_dispatchers_under_asan(tmp_path):
    """`make asan` builds the library with the HOST side under AddressSanitizer (GPU ASan is not available on this pool); a
    child process preloads the sanitizer runtime and drives what runs without a GPU: the dispatcher's dry run
    (sv_igemm_query_blocks: geometry checks, halo / tile configuration, tap tables, grid arithmetic) over every conv-like
    layer of WRN-28-2 and WRN-28-10 forward and backward with 1 and 4 groups, the compile-time tile programs, the option
    setters and the argument-error paths.  Any heap / stack / global overflow in that host code aborts the child."""
    import glob
    import subprocess
    import sys
    rt = sorted(glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so"))
    if not rt:
        import pytest
        pytest.skip("no AddressSanitizer runtime in this image")
    csrc = os.path.join(ROOT, "shot_vae_amd", "csrc")
    r = subprocess.run(["make", "-C", csrc, "-j8", "asan"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    script = tmp_path / "drive.py"
    script.write_text(r'''
import ctypes as C, sys
sys.path.insert(0, %r)
from shot_vae_amd import _lib as L
from shot_vae_amd.engine import Plan
lib = L.lib()
assert lib.sv_version() == L.ABI_VERSION
n_ok = n_err = 0
for net, K in (("wideresnet-28-2", 10), ("wideresnet-28-10", 100), ("wideresnet-10-1", 10)):
    plan = Plan(net, K=K)
    for B in (1, 8):
        for cv in plan.convs:
            for g in (cv.geom_fwd(B), cv.geom_dgrad(B)):
                for groups in (1, 4):
                    for dtype in (L.SV_F32, L.SV_BF16):
                        a = L.SvIgemmArgs()
                        a.x = a.w = a.out = 4096                 # never dereferenced: nothing is launched
                        a.groups, a.replicas = groups, 1
                        blocks = C.c_int(-1)
                        rc = lib.sv_igemm_query_blocks(C.byref(g), dtype, C.byref(a), C.byref(blocks))
                        # (kernels that need the >64 KiB LDS opt-in ask the HIP runtime first: an error without a GPU)
                        if rc == 0:
                            assert blocks.value > 0
                            n_ok += 1
                        else:
                            n_err += 1
items, waits = (C.c_int * (180 * 6))(), (C.c_int * 10)()
assert lib.sv_debug_conv_chunk_program(items, waits) == 0
for hv in (3, 4):
    it3, w5 = (C.c_int * (180 * 3))(), (C.c_int * 5)()
    assert lib.sv_debug_wgrad_tile_program(hv, it3, w5) == 0
for key, val in ((0, 5), (1, 7), (2, 1), (3, 64), (4, 1), (5, 131072)):
    assert lib.sv_set_option(key, val) == 0 and lib.sv_get_option(key) == val
assert lib.sv_set_option(99, 1) != 0 and lib.sv_set_option(3, 1) != 0
order = (C.c_int32 * 4)(0, 2, 2, 3)            # not a permutation: refused before any launch
assert lib.sv_bn_running_update_ex(4096, 4096, 1, 4096, 4096, 1e-5, 0.1, 64, 4, order, None) != 0
assert lib.sv_igemm(None, 1, None, None) != 0 and b"null" in lib.sv_last_error()
# ABI 4: an incomplete BatchNorm fold and an incomplete per-group loss stage are refused before any launch
plan = Plan("wideresnet-28-2", K=10)
g = plan.units[1]["conv1"].geom_fwd(8)
a = L.SvIgemmArgs()
a.x = a.w = a.out = a.pro_scale = a.pro_shift = a.fold_stats = 4096
a.fold_replicas, a.fold_count = 8, 64.0            # gamma / beta / mean / rstd missing
assert lib.sv_igemm(C.byref(g), L.SV_BF16, C.byref(a), None) != 0 and b"fold" in lib.sv_last_error()
blocks = C.c_int(-1)
a.fold_gamma = a.fold_beta = a.fold_mean = a.fold_rstd = 4096
assert lib.sv_igemm_query_blocks(C.byref(g), L.SV_BF16, C.byref(a), C.byref(blocks)) == 0 and blocks.value > 0    # a query launches nothing
gd = plan.units[1]["conv1"].geom_dgrad(8)
# ABI 7: the fused backward's argument checks (nothing is launched on a refusal)
assert lib.sv_bwd3x3(None, L.SV_BF16, None, None) != 0 and b"null" in lib.sv_last_error()
fb = L.SvBwd3x3Args()
fb.dy = fb.x = fb.w = fb.out = fb.dw = fb.ws = fb.bsums = fb.x_scale = fb.x_shift = fb.x_mean = fb.x_rstd = 4096
fb.replicas, fb.groups, fb.ws_elems, fb.x_slope = 4, 4, 1 << 22, 0.01
assert lib.sv_bwd3x3(C.byref(plan.units[1]["conv1"].geom_dgrad(8)), L.SV_F32, C.byref(fb), None) != 0 and b"bf16" in lib.sv_last_error()
assert lib.sv_bwd3x3(C.byref(plan.units[9]["conv1"].geom_dgrad(8)), L.SV_BF16, C.byref(fb), None) != 0 and b"32 -> 32 channels" in lib.sv_last_error()
assert lib.sv_bwd3x3(C.byref(plan.units[1]["conv1"].geom_fwd(8)), L.SV_BF16, C.byref(fb), None) != 0 and b"tap" in lib.sv_last_error()
assert lib.sv_bwd3x3(C.byref(plan.units[1]["conv1"].geom_dgrad(8)), L.SV_BF16, C.byref(fb), None) != 0 and b"deterministic" in lib.sv_last_error()
assert lib.sv_set_option(4, 0) == 0
fb.ws_elems = 100
assert lib.sv_bwd3x3(C.byref(plan.units[1]["conv1"].geom_dgrad(8)), L.SV_BF16, C.byref(fb), None) != 0 and b"workspace" in lib.sv_last_error()
fb.ws_elems, fb.dy2 = 1 << 22, 4096
assert lib.sv_bwd3x3(C.byref(plan.units[1]["conv1"].geom_dgrad(8)), L.SV_BF16, C.byref(fb), None) != 0 and b"two-tensor" in lib.sv_last_error()
assert lib.sv_bn_bwd_affine(None, 1, 32, 64.0, None, None, None, None, None, None, None, None, 1, None) != 0
assert lib.sv_param_gather(L.SV_BF16, None, 3, 3, None, None) != 0 and lib.sv_param_scatter_add(None, 0, 0, None, None) != 0
assert lib.sv_param_gather(L.SV_BF16, 4096, 0, 0, 4096, None) == 0            # an empty table launches nothing
b = L.SvShotLossArgs2()
b.image_l = b.image_u = b.label_l = b.perm_l = b.perm_u = b.terms = b.coef = b.tgt = 4096
b.Bl, b.Bu, b.D, b.K = 4, 6, 128, 10
assert lib.sv_shot_loss_step2(C.byref(b), None) != 0 and b"group" in lib.sv_last_error()
assert lib.sv_repack_strided(L.SV_BF16, 4096, 40, 8, 1, 1, 1, 32, 9, 16, 0, C.byref(g), 4096, None) != 0      # n_real > N
print("ASAN_DRIVE_OK", n_ok, n_err)
''' % ROOT)
    env = dict(os.environ, LD_PRELOAD=rt[-1], ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=66",
               SV_LIB_PATH=os.path.join(ROOT, "shot_vae_amd", "libshotvae_hip_asan.so"))
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "ASAN_DRIVE_OK" in r.stdout, (r.returncode, r.stdout[-500:], r.stderr[-3000:])
    n_ok = int(r.stdout.split("ASAN_DRIVE_OK")[1].split()[0])
    assert n_ok >= 100, r.stdout


def test_asm_mfma_lint_parser():
    """tools/asm_mfma_lint.py on synthetic ISA: a VALU write of an MFMA operand with fewer than two wait states in between is
    reported, the same write followed by `s_nop 1` (or by two other instructions) is not; register RANGES count."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import asm_mfma_lint as lint
    head = "_Zkernel:\n"
    mf = "\t;;#ASMSTART\n\tv_mfma_f32_32x32x16_bf16 a[0:15], v[46:49], v[78:81], a[0:15]\n\t;;#ASMEND\n"
    n, bad = lint.lint_isa(head + "\tv_accvgpr_read_b32 v49, a253\n" + mf)
    assert n == 1 and len(bad) == 1
    n, bad = lint.lint_isa(head + "\tv_accvgpr_read_b32 v49, a253\n\ts_nop 1\n" + mf)
    assert n == 1 and not bad
    n, bad = lint.lint_isa(head + "\tv_mov_b32_e32 v80, v3\n\tds_read_b128 v[10:13], v5\n" + mf)
    assert len(bad) == 1                                   # one instruction in between is one wait state: not enough
    n, bad = lint.lint_isa(head + "\tv_mov_b32_e32 v80, v3\n\tds_read_b128 v[10:13], v5\n\ts_add_u32 s1, s2, s3\n" + mf)
    assert not bad
    n, bad = lint.lint_isa(head + "\tv_mov_b32_e32 v50, v3\n" + mf)           # not an operand
    assert not bad
    n, bad = lint.lint_isa(head + "\tv_accvgpr_write_b32 a7, v3\n" + mf)      # SrcC lives in the AGPR half
    assert len(bad) == 1


@pytest.mark.timeout(900)
def test_inline_asm_mfmas_have_no_valu_hazard():
    """The two kernel files that spell MFMAs in inline assembly (invisible to the compiler's hazard recognizer).  wgrad3x3.hip
    carries no s_nop in front of them (13 % slower with it): its ISA is checked here.  conv3x3x.hip gives every MFMA its own
    `s_nop 1` -- checked at source level (compiling its 15 variants takes minutes; `python tools/asm_mfma_lint.py
    shot_vae_amd/csrc/conv3x3x.hip` is the full check)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import asm_mfma_lint as lint
    n, bad = lint.lint_file(os.path.join(ROOT, "shot_vae_amd", "csrc", "wgrad3x3.hip"))
    assert n > 0 and not bad, bad[:3]
    src = open(os.path.join(ROOT, "shot_vae_amd", "csrc", "conv3x3x.hip")).read()
    assert "#define SV_X3_NOP 1\n" in src and 'asm volatile(SV_X3_PRE "v_mfma' in src
    assert src.count('"v_mfma') == 1                     # every assembly MFMA of the file goes through the prefixed statement
