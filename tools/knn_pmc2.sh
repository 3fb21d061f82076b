#!/bin/bash
# usage: tools/knn_pmc2.sh [--points N]  (on the GPU box) -- cache / stall counters of the search kernel, one rocprofv3 pass per group
R=$PWD; cd /tmp && export TMPDIR=/tmp
G1="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum"
G2="TCC_HIT_sum TCC_MISS_sum"
G3="SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
G4="TCC_REQ_sum TCC_EA0_RDREQ_sum"
G5="SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM"
i=0
for G in "$G1" "$G2" "$G3" "$G4" "$G5"; do
  i=$((i+1)); rm -rf $R/gpurun_out/knnpmc2_$i
  rocprofv3 --kernel-trace --pmc $G --output-format csv -d $R/gpurun_out/knnpmc2_$i -- python $R/tools/knn_only.py "$@" > /dev/null 2> $R/gpurun_out/knnpmc2_$i.err || echo "group $i failed: $G"
done
cd $R
python - <<PY
import csv,glob,collections
for d in sorted(glob.glob("gpurun_out/knnpmc2_*/")):
    fs=sorted(glob.glob(d+"**/*counter_collection.csv",recursive=True))
    if not fs: print(d,"no counters"); continue
    acc=collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(fs[-1])):
        if "search_fit" in r["Kernel_Name"]:
            acc[r["Counter_Name"]][r["Dispatch_Id"]]+=float(r["Counter_Value"])
    for k,v in acc.items():
        vals=list(v.values()); print(k, [round(x) for x in vals[-2:]])
PY
