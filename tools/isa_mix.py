#!/usr/bin/env python3
"""Instruction mix of one kernel out of the gfx950 assembly of the library (hipcc --save-temps).
   python tools/isa_mix.py <mangled-name-prefix> [extra hipcc flags]   e.g. _Z15vio_pass_kernelILi1EE"""
import collections, os, re, subprocess, sys, tempfile
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
name = sys.argv[1]
with tempfile.TemporaryDirectory() as d:
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-Wno-unused-value",
                    "--save-temps", *sys.argv[2:], "-o", os.path.join(d, "x.so"), os.path.join(root, "fast-livo_amd/csrc/fastlivo_hip.hip")],
                   cwd=d, capture_output=True)
    s = open(os.path.join(d, "fastlivo_hip-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
m = re.search(r"^(%s\w*):[^\n]*\n(.*?)\n\.Lfunc_end" % re.escape(name), s, re.S | re.M)
body = m.group(2)
ins = [l.split()[0] for l in body.split("\n") if l.startswith("\t") and l.split() and not l.split()[0].startswith((".", ";"))]
c = collections.Counter(ins)
print(len(ins), "instructions in", m.group(1))
print(", ".join(f"{k} {v}" for k, v in c.most_common(50)))
