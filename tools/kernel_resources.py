#!/usr/bin/env python3
"""VGPR / AGPR / scratch / occupancy of the pass kernels as hipcc reports them (-Rpass-analysis=kernel-resource-usage).
   python tools/kernel_resources.py [extra hipcc flags, e.g. -DFL_INSTRUMENT] [--src DIR (a checkout of another revision)]"""
import os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
if "--src" in args:
    i = args.index("--src"); root = args[i + 1]; del args[i:i + 2]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-Wno-unused-value",
       *args, "-Rpass-analysis=kernel-resource-usage", "-o", f"/tmp/fl_res_{os.getpid()}.so", os.path.join(root, "fast-livo_amd/csrc/fastlivo_hip.hip")]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
os.path.exists(cmd[-2]) and os.remove(cmd[-2])
cur, rows = None, {}
for ln in out.splitlines():
    m = re.search(r"Function Name: (\S+)", ln)
    if m:
        cur = m.group(1); rows[cur] = {}; continue
    m = re.search(r"remark: .*?\s+(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\d+)", ln)
    if m and cur:
        rows[cur][m.group(1)] = int(m.group(2))
for k, v in rows.items():
    name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip().split("(")[0]
    if any(t in name for t in ("multipass", "pass_kernel", "search_fit")):
        print(f"{name:55s} VGPR {v.get('VGPRs')} AGPR {v.get('AGPRs')} scratch {v.get('ScratchSize [bytes/lane]')} "
              f"occ {v.get('Occupancy [waves/SIMD]')} LDS {v.get('LDS Size [bytes/block]')}")
