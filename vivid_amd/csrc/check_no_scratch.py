#!/usr/bin/env python3
"""Build check: every kernel of OBJECT whose name contains PATTERN must have a zero private (scratch) segment.
  python3 check_no_scratch.py [--arch gfx950] [--llvm-bin DIR] conv_x3.o conv_x3_glds
Reads the AMDGPU code-object metadata (msgpack note) through llvm-readelf; see the Makefile for why it matters.
A toolchain without the LLVM binutils (another ROCm prefix, a stripped install) downgrades the check to a warning: the build
itself does not depend on it."""
import argparse
import os
import re
import shutil
import subprocess
import sys
import tempfile

ap = argparse.ArgumentParser()
ap.add_argument("--arch", default="gfx950")
ap.add_argument("--llvm-bin", default="/opt/rocm/lib/llvm/bin")
ap.add_argument("object")
ap.add_argument("pattern")
a = ap.parse_args()
lib, pattern = a.object, a.pattern


def tool(name):
    p = os.path.join(a.llvm_bin, name)
    return p if os.path.exists(p) else shutil.which(name)


bundler, readelf, objcopy = tool("clang-offload-bundler"), tool("llvm-readelf"), tool("llvm-objcopy")
if not (bundler and readelf and objcopy):
    print(f"check_no_scratch: WARNING - clang-offload-bundler / llvm-readelf / llvm-objcopy not found under {a.llvm_bin} or on PATH; "
          f"scratch usage of {pattern} in {lib} NOT checked", file=sys.stderr)
    sys.exit(0)
with tempfile.TemporaryDirectory() as d:
    fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "dev.co")
    # a hipcc object carries its device code object as an offload bundle in the .hip_fatbin section
    subprocess.check_call([objcopy, f"--dump-section=.hip_fatbin={fat}", lib])
    targets = subprocess.run([bundler, "--list", "--type=o", f"--input={fat}"], capture_output=True, text=True).stdout.split()
    tgt = next((t for t in targets if a.arch in t), None)
    if tgt is None:
        sys.exit(f"check_no_scratch: no {a.arch} code object found in {lib} (targets: {targets})")
    subprocess.check_call([bundler, "--unbundle", "--type=o", f"--input={fat}", f"--targets={tgt}", f"--output={co}"])
    notes = subprocess.run([readelf, "--notes", co], capture_output=True, text=True).stdout
bad, seen = [], 0
for m in re.finditer(r"\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+)", notes, flags=re.S):
    name, scratch = m.group(1), int(m.group(2))
    if pattern in name:
        seen += 1
        if scratch:
            bad.append((name, scratch))
if not seen:
    sys.exit(f"check_no_scratch: no kernel matching {pattern!r} found in the metadata of {lib}")
if bad:
    sys.exit("check_no_scratch: kernels with scratch (spills): " + ", ".join(f"{n}: {s} B" for n, s in bad))
print(f"check_no_scratch: {seen} {pattern} kernels, no scratch")
