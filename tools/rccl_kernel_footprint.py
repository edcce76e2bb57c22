#!/usr/bin/env python3
"""What RCCL's device kernels need from a CU on gfx950: registers, LDS, scratch and threads per workgroup of every non-MSCCL kernel in the
gfx950 code object inside librccl.so (why a broadcast needs CUs without a persistent prefilter workgroup).

    tools/rccl_kernel_footprint.py [librccl.so] > profiles/r05_rccl_kernel_footprint.txt

The library's device code is one compressed offload bundle (magic CCOB) in the .hip_fatbin section: it is cut out, unbundled for gfx950
with clang-offload-bundler and read with llvm-readelf --notes.  Needs ~0.7 GB of scratch space under /tmp; runs without a GPU."""
import mmap
import os
import re
import struct
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else "/opt/rocm/lib/librccl.so"
    lib = os.path.realpath(lib)
    tmp = tempfile.mkdtemp(prefix="rcclobj_")
    sections = subprocess.run([f"{LLVM}/llvm-readelf", "-S", lib], capture_output=True, text=True, check=True).stdout
    m = re.search(r"\.hip_fatbin\s+PROGBITS\s+([0-9a-f]+)\s+([0-9a-f]+)\s+([0-9a-f]+)", sections)
    off, size = int(m.group(2), 16), int(m.group(3), 16)
    with open(lib, "rb") as f:
        mm = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
        at = mm.find(b"CCOB", off, off + size)
        version, _method = struct.unpack_from("<HH", mm, at + 4)
        total = struct.unpack_from("<I", mm, at + 8)[0] if version == 2 else struct.unpack_from("<Q", mm, at + 8)[0]
        blob = os.path.join(tmp, "bundle.bin")
        with open(blob, "wb") as g:
            g.write(mm[at:at + total])
    targets = subprocess.run([f"{LLVM}/clang-offload-bundler", "--list", "--type=o", f"--input={blob}"], capture_output=True, text=True, check=True).stdout.split()
    target = [t for t in targets if t.endswith("gfx950")][0]
    co = os.path.join(tmp, "gfx950.co")
    subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={blob}", f"--targets={target}", f"--output={co}"], check=True)
    os.remove(blob)
    notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True, check=True).stdout
    os.remove(co)
    os.rmdir(tmp)
    version_line = subprocess.run(["strings", "-n", "8", lib], capture_output=True, text=True).stdout
    ver = re.search(r"RCCL version[^\n]*|NCCL version [0-9.]+[^\n]*", version_line)
    print(f"# {lib}" + (f"  ({ver.group(0).strip()})" if ver else ""))
    print(f"# bundle target {target}; kernels other than the MSCCL interpreter's and the one-rank reductions")
    print("# kernel | threads per workgroup (max_flat_workgroup_size) | vector registers per lane (of which accumulation registers) | registers its waves take on each SIMD | SGPRs | LDS bytes | scratch bytes per lane")
    for e in re.split(r"\n\s+- \.agpr_count", notes)[1:]:
        e = ".agpr_count" + e
        g = lambda k: (re.search(r"\.%s:\s*(\S+)" % k, e) or [None, "?"])[1]
        name = g("name")
        if "msccl" in name or "oneRank" in name:
            continue
        nice = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        threads, regs = int(g("max_flat_workgroup_size")), int(g("vgpr_count"))
        per_simd = (threads + 255) // 256 * ((regs + 7) // 8 * 8)            # waves per SIMD x the allocation granule of 8
        print(f"{re.sub(r'[(].*', '', nice)} | {threads} | {regs} ({g('agpr_count')}) | {per_simd} | {g('sgpr_count')} | {g('group_segment_fixed_size')} | {g('private_segment_fixed_size')}")
    print("# A SIMD has 512 vector registers per lane; a workgroup's waves are spread over a CU's four SIMDs.  One workgroup of the persistent prefilter")
    print("# kernel takes 256 registers on every SIMD of its CU and stays for the whole launch: a workgroup of a kernel above fits beside it only if its")
    print("# waves take 256 registers or fewer on a SIMD -- none does: a broadcast needs CUs without a persistent workgroup (DESIGN.md section 6).")


if __name__ == "__main__":
    main()
