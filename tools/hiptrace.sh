#!/bin/bash
# Host API calls (hipLaunchKernel, hipMemcpyAsync, ...) against the kernels they launch, on one time axis (rocprofv3 --hip-runtime-trace
# --kernel-trace --memory-copy-trace): who waits for whom -- the stream for the host's enqueue, or the host for the stream.
# usage: tools/hiptrace.sh N <command ...>      (the last N records are printed)
N=$1; shift
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/ht_tmp
(cd $R && rocprofv3 --hip-runtime-trace --kernel-trace --memory-copy-trace --output-format csv -d $OUT/ht_tmp -- "$@" > $OUT/ht_tmp.out 2> $OUT/ht_tmp.err)
python - "$OUT/ht_tmp" "$N" <<'PY'
import csv, glob, sys
d, n = sys.argv[1], int(sys.argv[2])
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "    GPU  " + r["Kernel_Name"][:40]))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "    GPU  COPY " + r.get("Direction", "")))
for f in glob.glob(d + "/**/*hip_api_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        fn = r.get("Function", "")
        if fn.startswith(("hipLaunch", "hipMemcpy", "hipMemset", "hipStreamSync", "hipEvent", "hipExtLaunch")):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "HOST " + fn))
rows.sort()
rows = rows[-n:]
t0 = rows[0][0]
for s, e, k in rows:
    print(f"{(s - t0) / 1e3:9.1f} +{(e - s) / 1e3:7.1f} us  {k}")
PY
rm -rf $OUT/ht_tmp
