#!/usr/bin/env python3
"""Build-time check of the 16-byte-store data hazard (csrc/lfg_device.hpp: store_b128_guarded).

A buffer/global store of more than 8 bytes reads its data registers for a few cycles after issue; on gfx950 a VALU
write to them in the following slots was seen to win that race even for the form the ISA manual exempts (scalar
offset in use), as a few wrong pixels per 4K frame on some runs only -- nothing a pixel test catches reliably.
So the rule is checked on the machine code itself, in every gfx950 code object of the library:
  * buffer stores of 12/16 bytes (the hand-issued ones, where the anomaly was seen) must be followed DIRECTLY by an
    s_nop of at least two wait states (s_nop 1), or by BUFFER_WINDOW instructions none of which writes one of their
    data registers;
  * compiler-generated global/flat stores of 12/16 bytes must satisfy the documented rule (one wait state: the next
    instruction does not write their data registers), which the compiler's hazard recogniser is trusted with and this
    script re-checks.

    check_store_hazard.py <liblinuxfg_hip.so>      exit status 1 and a listing on a violation
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile



def find_objdump():
    """llvm-objdump of the ROCm toolchain that built the library: $LLVM_OBJDUMP, next to the clang that hipcc drives
    ($HIPCC --print-prog-name where it answers), under $ROCM_PATH, on PATH.  None if there is none."""
    cand = [os.environ.get("LLVM_OBJDUMP")]
    hipcc = os.environ.get("HIPCC") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    try:
        out = subprocess.run([hipcc, "--print-prog-name=llvm-objdump"], capture_output=True, text=True, timeout=60).stdout.strip()
        cand.append(out if os.path.isabs(out) else None)
    except (OSError, subprocess.SubprocessError):
        pass
    rocm = os.environ.get("ROCM_PATH") or "/opt/rocm"
    cand += [os.path.join(rocm, "lib", "llvm", "bin", "llvm-objdump"), os.path.join(rocm, "llvm", "bin", "llvm-objdump"),
             shutil.which("llvm-objdump")]
    for c in cand:
        if c and os.path.isfile(c) and os.access(c, os.X_OK):
            return c
    return None


OBJDUMP = find_objdump()
BUFFER_WINDOW = 6                 # instructions after a buffer store that must leave its data registers alone
OTHER_WINDOW = 1                  # documented: one wait state
STORE = re.compile(r"\b(buffer|global|flat|scratch)_store_dwordx[34]\b")


def reg_range(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def check(disasm_lines, name):
    insts = []
    for line in disasm_lines:
        line = line.split("//")[0].strip()
        if not line or line.endswith(":") or line.startswith(("Disassembly", "/", "file")):
            continue
        insts.append(line)
    bad = []
    for i, ins in enumerate(insts):
        if not STORE.search(ins):
            continue
        ops = [t.strip() for t in ins.split(None, 1)[1].split(",")]
        data = reg_range(ops[1] if ins.startswith(("global", "flat", "scratch")) else ops[0])
        nxt = insts[i + 1] if i + 1 < len(insts) else ""
        m = re.match(r"s_nop\s+(\d+)", nxt)
        if m and int(m.group(1)) >= 1:
            continue
        window = BUFFER_WINDOW if ins.startswith("buffer") else OTHER_WINDOW
        for later in insts[i + 1:i + 1 + window]:
            # (VALU writers only: data an LDS or memory read returns arrives long after any store has read its operands,
            #  and the compiler waits for the counters where registers are reused)
            if later.startswith("v_") and not later.startswith("v_cmp"):
                operands = [t.strip() for t in later.split(None, 1)[1].split(",")] if " " in later else []
                # the first operand is the destination; v_swap_b32 writes its second as well, and the instructions with a
                # scalar carry / mask output (v_add_co, v_div_scale, v_mad_u64_u32 ...) keep their vector result first
                dsts = operands[:2] if later.startswith("v_swap") else operands[:1]
                if any(reg_range(d) & data for d in dsts):
                    bad.append((name, ins, later))
                    break
    return bad


def main():
    lib = sys.argv[1]
    if OBJDUMP is None:
        print("check_store_hazard: no llvm-objdump (looked at $LLVM_OBJDUMP, `hipcc --print-prog-name`, $ROCM_PATH/lib/llvm/bin and PATH): "
              "the wide-store guard of " + lib + " cannot be checked, so the library is not installed")
        return 1
    tmp = tempfile.mkdtemp(prefix="lfg_hazard_")
    try:
        local = os.path.join(tmp, os.path.basename(lib))
        shutil.copy(lib, local)
        subprocess.run([OBJDUMP, "--offloading", local], check=True, capture_output=True)
        objs = [f for f in sorted(os.listdir(tmp)) if "gfx950" in f]
        if not objs:
            print("check_store_hazard: no gfx950 code object found in", lib)
            return 1
        bad, stores = [], 0
        for f in objs:
            out = subprocess.run([OBJDUMP, "-d", os.path.join(tmp, f)], check=True, capture_output=True, text=True).stdout
            lines = [l.split("\t", 1)[1] if "\t" in l else l for l in out.splitlines()]
            stores += sum(1 for l in lines if STORE.search(l))
            bad += check(lines, f)
        if bad:
            print(f"check_store_hazard: {len(bad)} wide store(s) whose data registers are rewritten too early:")
            for name, st, later in bad[:20]:
                print(f"  {name}: {st}\n      -> {later}")
            return 1
        print(f"check_store_hazard: {stores} stores of more than 8 bytes in {len(objs)} code objects, all guarded")
        return 0
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    sys.exit(main())
