"""Hazard lint for MFMAs spelled in inline assembly.

The compiler's hazard recognizer does not look inside `asm volatile("v_mfma ...")`: the two wait states gfx9 needs between a
VALU write of a register and an MFMA reading it (SrcA / SrcB / SrcC) are not inserted when the reader is such a statement --
and the writer may be the compiler's own spill re-load or register copy (found in conv3x3x.hip, round 3).  This tool compiles
a kernel file to ISA and reports every inline-assembly MFMA whose operands are written by a VALU instruction fewer than two
wait states before it.

    python tools/asm_mfma_lint.py shot_vae_amd/csrc/wgrad3x3.hip [-DFLAG ...]      # exit code 1 when a hazard is found
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REG = re.compile(r"\b([va])(?:\[(\d+):(\d+)\]|(\d+)\b)")


def regs(tok):
    out = set()
    for m in REG.finditer(tok):
        if m.group(2) is not None:
            out |= {(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)}
        else:
            out.add((m.group(1), int(m.group(4))))
    return out


def lint_isa(text):
    """-> (number of inline-assembly MFMAs, list of (function, line number, mfma, writer))"""
    bad, n_mfma = [], 0
    func, in_asm = "?", False
    window = []                        # the last instructions as (text, wait states they provide)
    for ln, raw in enumerate(text.split("\n"), 1):
        line = raw.strip()
        if raw.startswith("_Z") and line.split(";")[0].strip().endswith(":"):
            func, window = line.split(":")[0], []
            continue
        if line.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if line.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not line or line.startswith(";") or line.startswith(".") or line.endswith(":"):
            if line.endswith(":") and not line.startswith(";"):
                window = []            # a label: predecessors unknown -- the loop bodies are straight-line between MFMAs
            continue
        ins = line.split(";")[0].strip()
        op = ins.split()[0]
        if in_asm and op.startswith("v_mfma"):
            n_mfma += 1
            ops = ins[len(op):].split(",")
            srcs = regs(",".join(ops[1:]))
            ws = 0
            for prev, states in reversed(window):
                if ws >= 2:
                    break
                pop = prev.split()[0]
                if pop.startswith("v_") and not pop.startswith("v_mfma") and not pop.startswith("v_cmp"):
                    dst = regs(prev[len(pop):].split(",")[0])
                    if dst & srcs:
                        bad.append((func, ln, ins, prev))
                        break
                ws += states
        if in_asm and (op.startswith("global_load") or op.startswith("global_store") or op.startswith("buffer_")):
            # VALU write of an SGPR (v_readlane / v_readfirstlane / v_cmp into a scalar pair) -> vector-memory instruction
            # reading that SGPR: five wait states
            srcs = {(m.group(1), int(m.group(2)), int(m.group(3) or m.group(2)))
                    for m in re.finditer(r"\b(s)\[?(\d+)(?::(\d+))?\]?", ins[len(op):])}
            sreg = set()
            for _, lo, hi in srcs:
                sreg |= set(range(lo, hi + 1))
            ws = 0
            for prev, states in reversed(window):
                if ws >= 5:
                    break
                pop = prev.split()[0]
                if pop.startswith("v_readlane") or pop.startswith("v_readfirstlane"):
                    m = re.match(r"\S+\s+s(\d+)", prev)
                    if m and int(m.group(1)) in sreg:
                        bad.append((func, ln, ins, prev))
                        break
                ws += states
        states = 1
        if op == "s_nop":
            states = int(ins.split()[1]) + 1
        window.append((ins, states))
        window = window[-10:]
    return n_mfma, bad


def lint_file(path, flags=()):
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics",
               "-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only", "-o", out, path] + list(flags)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(r.stderr[-2000:])
        return lint_isa(open(out).read())


if __name__ == "__main__":
    n, bad = lint_file(sys.argv[1], sys.argv[2:])
    print("%s: %d inline-assembly MFMAs, %d with a VALU write of an operand less than two wait states before" % (sys.argv[1], n, len(bad)))
    for f, ln, mf, wr in bad[:20]:
        print("  %s line %d\n     %s\n     after: %s" % (f[:70], ln, mf, wr))
    sys.exit(1 if bad else 0)
