R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for lib in "" build_ab/lib_noexact.so; do
  n=cur; [ -n "$lib" ] && n=noexact
  rm -rf $OUT/kt_$n
  FL_LIB_PATH=${lib:+$R/$lib} rocprofv3 --kernel-trace --output-format csv -d $OUT/kt_$n -- python $R/tools/computej_breakdown.py > $OUT/kt_$n.out 2>$OUT/kt_$n.err
  f=$(find $OUT/kt_$n -name "*kernel_trace.csv" | head -1)
  cp $f $OUT/kt_${n}_trace.csv
  rm -rf $OUT/kt_$n
done
