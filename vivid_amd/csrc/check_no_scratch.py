#!/usr/bin/env python3
"""Build check: every kernel of OBJECT whose name contains PATTERN must have a zero private (scratch) segment.
  python3 check_no_scratch.py conv_x3.o conv_x3_glds
Reads the AMDGPU code-object metadata (msgpack note) through llvm-readelf; see the Makefile for why it matters."""
import re
import subprocess
import sys

lib, pattern = sys.argv[1], sys.argv[2]
bundler = "/opt/rocm/lib/llvm/bin/clang-offload-bundler"
readelf = "/opt/rocm/lib/llvm/bin/llvm-readelf"
import os, tempfile
objcopy = "/opt/rocm/lib/llvm/bin/llvm-objcopy"
with tempfile.TemporaryDirectory() as d:
    fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "dev.co")
    # a hipcc object carries its device code object as an offload bundle in the .hip_fatbin section
    subprocess.check_call([objcopy, f"--dump-section=.hip_fatbin={fat}", lib])
    targets = subprocess.run([bundler, "--list", "--type=o", f"--input={fat}"], capture_output=True, text=True).stdout.split()
    tgt = next((t for t in targets if "gfx950" in t), None)
    if tgt is None:
        sys.exit(f"check_no_scratch: no gfx950 code object found in {lib} (targets: {targets})")
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
