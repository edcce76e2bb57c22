#!/usr/bin/env python3
"""Static opcode histogram of a kernel's VALU instructions x the issue costs of /opt/skills/guides/MI355X_MICROARCH.md
("vector-instruction ISSUE cost": v_exp / v_log / v_rcp / v_rsq / v_sqrt / v_sin / v_cos 8 cycles against 4 for v_add_f32 /
v_fma_f32 one wave alone, i.e. twice a plain op; packed f32 VALU "an anti-lever": two plain ops' worth) -> the mean issue cost
of one VALU wave-instruction of that kernel, in plain-op units.  bench.py multiplies the SQ_INSTS_VALU count of the committed
counter profile by it for a cost-WEIGHTED issue-slot figure.

    tools/valu_cost_histogram.py [lib.so] > profiles/r05_valu_cost_weights.txt

STATIC: every instruction of the code object counts once, whatever the executed mix is (loops, branches not taken); the
figure is a proxy and labelled so.  Assumptions beyond the guide, stated in the output: 64-bit integer multiplies and f64 ops
at a quarter rate (4), v_dot4_u32_u8 / v_sad_u8 / everything else at the plain rate (1)."""
import hashlib
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
TRANS = re.compile(r"^v_(sqrt|rsq|rcp|exp|log|sin|cos)_")
PACKED = re.compile(r"^v_pk_(add|mul|fma)_f32")
QUARTER = re.compile(r"^v_(mul_lo_u32|mul_hi_u32|mul_hi_i32|mad_u64_u32|mad_i64_i32)|^v_\w+_f64")


def cost(op):
    if TRANS.match(op):
        return 2.0
    if PACKED.match(op):
        return 2.0
    if QUARTER.match(op):
        return 4.0
    return 1.0


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "linux-fg_amd", "liblinuxfg_hip.so")
    sha = hashlib.sha256(open(lib, "rb").read()).hexdigest()[:16]
    tmp = tempfile.mkdtemp()
    local = os.path.join(tmp, "lib.so")
    with open(lib, "rb") as f, open(local, "wb") as g:
        g.write(f.read())
    subprocess.run([OBJDUMP, "--offloading", local], check=True, capture_output=True, cwd=tmp)
    kernels = {}
    for fn in sorted(os.listdir(tmp)):
        if "gfx950" not in fn:
            continue
        out = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", os.path.join(tmp, fn)], check=True, capture_output=True, text=True).stdout
        name = None
        for line in out.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
            if m:
                name = m.group(1)
                continue
            m = re.match(r"^\s+(v_[a-z0-9_]+)", line)
            if m and name:
                op = re.sub(r"_(e32|e64|dpp|sdwa)$", "", m.group(1))
                kernels.setdefault(name, {}).setdefault(op, 0)
                kernels[name][op] += 1
    demangle = subprocess.run(["c++filt"], input="\n".join(kernels), capture_output=True, text=True).stdout.splitlines()
    print(f"# lib_sha16 {sha}")
    print("# static VALU opcode histogram x issue cost in plain-op units (transcendental 2, v_pk_*_f32 2, 64-bit multiplies and f64 4 [assumed], all else 1 [v_dot4 / v_sad assumed plain])")
    for mangled, nice in zip(kernels, demangle):
        hist = kernels[mangled]
        n = sum(hist.values())
        if n < 200 or "lfg::" not in nice:
            continue
        w = sum(c * cost(op) for op, c in hist.items())
        short = re.sub(r"\(.*", "", nice).replace("void ", "")
        top = sorted(hist.items(), key=lambda kv: -kv[1] * cost(kv[0]))[:6]
        print(f"kernel {short} valu_static {n} mean_cost {w / n:.4f} transcendental {sum(c for o, c in hist.items() if TRANS.match(o))} "
              f"packed_f32 {sum(c for o, c in hist.items() if PACKED.match(o))} quarter_rate {sum(c for o, c in hist.items() if QUARTER.match(o))}"
              f"   # top by cost: " + ", ".join(f"{o} {c}" for o, c in top))


if __name__ == "__main__":
    main()
