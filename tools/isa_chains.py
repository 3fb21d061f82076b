#!/usr/bin/env python3
"""Loads the compiler has serialised: `load ; s_waitcnt vmcnt(0) ; load` chains in the gfx950 assembly of the library, per kernel.
A wait loop that is meant to keep several polls in flight, a block copy or a field-by-field prefetch compiled that way costs one memory
round trip per load (round 6: 1.5 us of every pass of vio_multipass_kernel<1, 1>, 10 us of the camera half) and no profiler counter says so.
    python tools/isa_chains.py [kernel-name-substring] [--show]     (cross-compiles, no GPU needed; ~2 min)
Columns: chains of vector loads <= 8 instructions apart with a vmcnt(0) between them, of which sc1 (polls of hand-off words), scalar-load
chains (lgkmcnt(0) between s_loads), loads in the kernel, instructions. Genuine dependencies (a pointer, then what it points to) show up
too: read the pairs (--show) before believing a number."""
import os, re, subprocess, sys, tempfile
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = [a for a in sys.argv[1:] if not a.startswith("--")]
want = args[0] if args else ""
show = "--show" in sys.argv
with tempfile.TemporaryDirectory() as d:
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-Wno-unused-value",
                    "--save-temps", "-o", os.path.join(d, "x.so"), os.path.join(root, "fast-livo_amd/csrc/fastlivo_hip.hip")], cwd=d, capture_output=True)
    s = open(os.path.join(d, "fastlivo_hip-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
rows = []
for m in re.finditer(r"^(_Z\w+|\w+):\s*; @\S+\n(.*?)\n\.Lfunc_end", s, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if "rocprim" in name or want not in name:
        continue
    ins = [l.strip() for l in body.split("\n")]
    ins = [l for l in ins if l and not l.startswith((";", ".")) and not l.endswith(":")]
    vl = [i for i, l in enumerate(ins) if re.match(r"(buffer|global|flat)_load", l)]
    pairs = [(a, b) for a, b in zip(vl, vl[1:]) if b - a <= 8 and any(re.match(r"s_waitcnt.*vmcnt\(0\)", x) for x in ins[a + 1:b])]
    sc1 = [(a, b) for a, b in pairs if " sc1" in ins[a] and " sc1" in ins[b]]
    sl = [i for i, l in enumerate(ins) if l.startswith("s_load")]
    sp = [(a, b) for a, b in zip(sl, sl[1:]) if b - a <= 8 and any(re.match(r"s_waitcnt.*lgkmcnt\(0\)", x) for x in ins[a + 1:b])]
    if pairs or sp:
        rows.append((len(pairs), len(sc1), len(sp), len(vl), len(ins), name, [(ins[a], ins[b]) for a, b in pairs]))
print(f"{'chains':>6} {'sc1':>4} {'scalar':>6} {'loads':>6} {'instr':>6}  kernel")
for r in sorted(rows, reverse=True):
    print(f"{r[0]:6d} {r[1]:4d} {r[2]:6d} {r[3]:6d} {r[4]:6d}  {r[5][:90]}")
    if show:
        for a, b in r[6]:
            print("           ", a[:70], " ->", b[:70])
