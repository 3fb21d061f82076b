#!/bin/bash
# usage: tools/vio_pmc.sh [patches]  (on the GPU box) -- instruction and stall counters of the at-scale VIO pass kernels (one launch per
# pass): the one-patch-per-lane producers (vio_pass_kernel<0, 1>) and the 16-lanes-per-patch ones (<0, 0>) side by side. Two counter
# passes (eight SQ counters each), --kernel-trace only beside them.
M=${1:-1000000}
R=$PWD; cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/viopmc
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/viopmc/a -- python $R/bench.py --only vio_sweep --vio-sweep-patches $M > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/viopmc/b -- python $R/bench.py --only vio_sweep --vio-sweep-patches $M > /dev/null 2>&1
# (TA_* / TCP_* / SQ_VMEM_* counter passes hang rocprofv3 on this pool's boxes -- 500 s until the time-out, round 5: leave them out)
cd $R
python - "$M" <<'PY'
import csv, glob, collections, sys
m = int(sys.argv[1])
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
for f in sorted(glob.glob("gpurun_out/viopmc/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        kn = r["Kernel_Name"]
        if "vio_pass_kernel" in kn:
            form = "one_patch_per_lane" if ("<0, 1>" in kn or "Li0ELi1E" in kn) else "16_lanes_per_patch"
            acc[form][r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
for form, ctr in acc.items():
    out = {}
    for k, v in ctr.items():
        vals = sorted(v.values())
        out[k] = vals[len(vals) // 2]
    per = 64.0 if form == "one_patch_per_lane" else 4.0          # patches per wavefront and iteration
    iters = m / per
    print(form, {k: round(v) for k, v in out.items()})
    print(form, {"per_wave_iteration": {k: round(v / iters, 1) for k, v in out.items() if k.startswith("SQ_INSTS")},
                 "per_patch": {k: round(v / m, 2) for k, v in out.items() if k.startswith("SQ_INSTS")}})
    if "SQ_WAVE_CYCLES" in out and "SQ_BUSY_CYCLES" in out:
        print(form, {"waves_in_flight_avg": round(out["SQ_WAVE_CYCLES"] / max(out["SQ_BUSY_CYCLES"], 1), 2)})
    if "SQ_ACTIVE_INST_ANY" in out:
        print(form, {"share_of_wave_cycles": {k: round(out[k] / max(out.get("SQ_WAIT_ANY", 0) + 1, 1), 3) for k in out if k.startswith(("SQ_ACTIVE", "SQ_WAIT"))}})
PY
