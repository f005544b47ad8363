#!/usr/bin/env python3
"""Build-time check of the gfx950 code object: no instruction reads the result of a transcendental VALU op
(v_rcp_f64, v_rsq_f64, v_sqrt_f64, ... ) in the very next issue slot.

Why: gfx950 needs one wait state between a TRANS op and the first use of its result.  The compiler's hazard
recognizer inserts it for the instructions it schedules itself but does not look at the operands of inline-asm
statements, and zd_kernels.hip uses asm `v_fma_f64` for its Horner steps (fma3/fnma3/fmas3).  In the source the
rule is enforced by types (`Trans` values can only become a `double` through a compiler-visible operation); this
script verifies the outcome on the shipped ISA, so a future scheduling change cannot silently re-introduce the
wrong-result hazard at sizes the parity tests do not reach.

    python check_trans_hazard.py build/zd_kernels.o      exit code 0 = clean
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = os.environ.get("ZD_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
TRANS = re.compile(r"^v_(rcp|rsq|sqrt|exp|log|sin|cos)(_iflag|_legacy|_clamp)?_(f64|f32|f16|bf16)")
REG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")


def regs(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def disassemble(obj):
    with tempfile.TemporaryDirectory() as tmp:
        fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "dev.co")
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, obj])
        subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--type=o",
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + fat, "--output=" + co,
                               "--unbundle"])
        return subprocess.check_output([os.path.join(LLVM, "llvm-objdump"), "-d", co], text=True)


def check(asm_text):
    """returns (number of TRANS ops seen, list of violations)"""
    prev = None  # (mnemonic line, dst regs) of a TRANS op in the previous issue slot
    func, ntrans, bad = "?", 0, []
    for line in asm_text.splitlines():
        s = line.strip()
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", s)
        if m:
            func, prev = m.group(1), None
            continue
        if not s or not re.match(r"^[sv]_|^(ds|global|buffer|flat|scratch)_", s):
            continue
        ins = s.split("//")[0].strip()
        mnem, _, ops = ins.partition(" ")
        if prev is not None and not mnem.startswith("s_nop"):
            used = regs(ops)
            if used & prev[1]:
                bad.append((func, prev[0], ins))
        prev = None
        if TRANS.match(mnem):
            ntrans += 1
            dst = ops.split(",")[0]
            prev = (ins, regs(dst))
    return ntrans, bad


def main(argv):
    total, bad = 0, []
    for obj in argv[1:]:
        n, b = check(disassemble(obj))
        total += n
        bad += b
    if bad:
        for func, a, b in bad:
            sys.stderr.write("TRANS-use hazard in %s:\n    %s\n    %s\n" % (func, a, b))
        sys.stderr.write("check_trans_hazard: %d violation(s)\n" % len(bad))
        return 1
    print("check_trans_hazard: %d transcendental ops, none read in the next issue slot" % total)
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
