#!/bin/bash
# usage: tools/knn_pmc.sh  (on the GPU box) -- per-dispatch counters of the search kernel
R=$PWD; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/knnpmc -- python $R/tools/knn_only.py "$@" 2>&1 | grep search_fit_us
cd $R
python - <<PY
import csv,glob,collections
f=sorted(glob.glob("gpurun_out/knnpmc/**/*counter_collection.csv",recursive=True))[-1]
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    if "search_fit" in r["Kernel_Name"]:
        acc[r["Counter_Name"]][r["Dispatch_Id"]].append(float(r["Counter_Value"]))
for k,v in acc.items():
    vals=[sum(x) for x in v.values()]
    print(k, vals[:2])
PY
