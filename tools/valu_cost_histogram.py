#!/usr/bin/env python3
"""Static opcode histogram of a kernel's VALU instructions x MEASURED issue costs -> the mean issue cost of one VALU
wave-instruction of that kernel, in units of a plain op (v_add_f32).  bench.py multiplies the SQ_INSTS_VALU count of the
committed counter profile by it for a cost-WEIGHTED issue-slot figure.

    tools/valu_cost_histogram.py [lib.so] > profiles/r05_valu_cost_weights.txt

Costs: tools/bench_dpp.hip on the MI355X, four waves per SIMD, ns per wave-instruction and SIMD relative to v_add_f32 (1.08 ns):
v_fma_f32 1.04, v_pk_fma_f32 2.14 (packed f32: two plain ops' worth, as /opt/skills/guides/MI355X_MICROARCH.md says), v_sqrt_f32 3.72
(taken for every transcendental; the guide says twice a plain op for a wave alone), v_dot4_u32_u8 2.07, v_sad_u32 2.13 (taken for
v_sad_u8 too), v_cvt_pk_u8_f32 1.75, v_mov_b32 with a DPP modifier 1.75, arithmetic with a DPP modifier 2.73.  ASSUMED, not
measured: 64-bit integer multiplies and f64 ops 4; everything else 1.
STATIC: every instruction of the code object counts once, whatever the executed mix is (loops, branches not taken); the
figure is a proxy and labelled so."""
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


def cost(op, dpp=False):
    if dpp:
        return 1.75 if op.startswith("v_mov_b32") else 2.73
    if TRANS.match(op):
        return 3.72
    if PACKED.match(op):
        return 2.14
    if op.startswith("v_dot4"):
        return 2.07
    if op.startswith("v_sad_"):
        return 2.13
    if op.startswith("v_cvt_pk_u8_f32"):
        return 1.75
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
                op = re.sub(r"_(e32|e64|sdwa)$", "", m.group(1))          # (a DPP modifier stays in the name: it has a cost of its own)
                kernels.setdefault(name, {}).setdefault(op, 0)
                kernels[name][op] += 1
    demangle = subprocess.run(["c++filt"], input="\n".join(kernels), capture_output=True, text=True).stdout.splitlines()
    print(f"# lib_sha16 {sha}")
    print("# static VALU opcode histogram x issue cost in units of v_add_f32, measured on the MI355X at four waves per SIMD (tools/bench_dpp.hip): transcendental 3.72, v_pk_*_f32 2.14, v_dot4 2.07, v_sad 2.13, v_cvt_pk_u8_f32 1.75, DPP mov 1.75, DPP arithmetic 2.73; assumed: 64-bit multiplies and f64 4, all else 1")
    for mangled, nice in zip(kernels, demangle):
        hist = kernels[mangled]
        n = sum(hist.values())
        if n < 200 or "lfg::" not in nice:
            continue
        def c_of(op):
            return cost(re.sub(r"_dpp$", "", op), op.endswith("_dpp"))
        w = sum(c * c_of(op) for op, c in hist.items())
        short = re.sub(r"\(.*", "", nice).replace("void ", "")
        top = sorted(hist.items(), key=lambda kv: -kv[1] * c_of(kv[0]))[:6]
        print(f"kernel {short} valu_static {n} mean_cost {w / n:.4f} transcendental {sum(c for o, c in hist.items() if TRANS.match(o))} "
              f"packed_f32 {sum(c for o, c in hist.items() if PACKED.match(o))} quarter_rate {sum(c for o, c in hist.items() if QUARTER.match(o))}"
              f"   # top by cost: " + ", ".join(f"{o} {c}" for o, c in top))


if __name__ == "__main__":
    main()
