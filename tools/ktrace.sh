#!/bin/bash
# Kernel time line (rocprofv3 --kernel-trace) of a python tool, last N dispatches printed.  usage: tools/ktrace.sh N tools/x.py [args]
N=$1; shift   # FL_KT_CMD=1: the rest is a command line, not a python tool
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/kt_tmp
if [ -n "$FL_KT_CMD" ]; then
    (cd $R && rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/kt_tmp -- "$@" > $OUT/kt_tmp.out 2> $OUT/kt_tmp.err)
else
    rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/kt_tmp -- python $R/"$@" > $OUT/kt_tmp.out 2> $OUT/kt_tmp.err
fi
tail -n 3 $OUT/kt_tmp.out
python - "$OUT/kt_tmp" "$N" <<'PY'
import csv, glob, sys
d, n = sys.argv[1], int(sys.argv[2])
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:48]))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "")))
rows.sort()
rows = rows[-n:]
t0 = rows[0][0]
for s, e, k in rows:
    print(f"{(s - t0) / 1e3:9.1f} +{(e - s) / 1e3:7.1f} us  {k}")
PY
rm -rf $OUT/kt_tmp
