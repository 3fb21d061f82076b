#!/bin/bash
# usage: tools/vio_pmc.sh [patches]  (on the GPU box) -- instruction counters of the at-scale VIO pass kernel (one launch per pass)
M=${1:-1000000}
R=$PWD; cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/viopmc
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/viopmc -- python $R/bench.py --only vio_sweep --vio-sweep-patches $M > /dev/null 2>&1
cd $R
python - "$M" <<'PY'
import csv, glob, collections, sys
m = int(sys.argv[1])
f = sorted(glob.glob("gpurun_out/viopmc/**/*counter_collection.csv", recursive=True))[-1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    if "vio_pass_kernel" in r["Kernel_Name"]:
        acc[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
out = {}
for k, v in acc.items():
    vals = sorted(v.values())
    out[k] = vals[len(vals) // 2]
iters = m / 4.0          # wave-iterations: 4 patches per wavefront and iteration
print({k: round(v) for k, v in out.items()})
print({"per_wave_iteration": {k: round(v / iters, 1) for k, v in out.items() if k.startswith("SQ_INSTS")}})
PY
