#!/usr/bin/env python3
"""Registers, scratch and LDS of every kernel in a built libbfsm_hip.so, read from the gfx950 code object's metadata.

usage: kernel_resources.py [lib.so] [--scratch]      (--scratch: only kernels with a private segment, i.e. spills)
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin/"
args = [a for a in sys.argv[1:] if not a.startswith("--")]
lib = args[0] if args else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                        "boltzmann-fourier-spectral-method_amd", "libbfsm_hip.so")
with tempfile.TemporaryDirectory() as d:
    subprocess.check_call([LLVM + "llvm-objcopy", "--dump-section", f".hip_fatbin={d}/fat.bin", lib])
    subprocess.check_call([LLVM + "clang-offload-bundler", "--unbundle", "--type=o", f"--input={d}/fat.bin",
                           "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={d}/dev.co"])
    notes = subprocess.run([LLVM + "llvm-readelf", "--notes", f"{d}/dev.co"], capture_output=True, text=True, check=True).stdout
rows = []
for blk in re.split(r"\n  - \.agpr_count", notes)[1:]:
    g = lambda k: re.search(r"\." + k + r":\s+(\S+)", blk).group(1)
    rows.append((g("name"), int(g("vgpr_count")), int(re.match(r":\s+(\d+)", blk).group(1)), int(g("private_segment_fixed_size")),
                 int(g("group_segment_fixed_size")), int(g("sgpr_count"))))
names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.split("\n")
print(f"{'vgpr':>5s} {'agpr':>5s} {'scratch B':>9s} {'static LDS':>10s} {'sgpr':>5s}  kernel")
for r, n in zip(rows, names):
    if "--scratch" in sys.argv and r[3] == 0:
        continue
    n = re.sub(r"^void bfsm::bfsm_kernel<\(bfsm::(S?K)\)(\d+), (\d+), (\w+),.*", r"\1 \2 N=\3 \4", n)
    print(f"{r[1]:5d} {r[2]:5d} {r[3]:9d} {r[4]:10d} {r[5]:5d}  {n[:100]}")
