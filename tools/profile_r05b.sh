#!/bin/bash
# second measurement pass of round 5 (after the one-patch-per-lane VIO producers)
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT; TAG=r05b
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 > $OUT/${TAG}_gpu_tests.txt
cd /tmp && export TMPDIR=/tmp
stats() { local name=$1; shift; rm -rf $OUT/prof_$name
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$name -- python $R/bench.py "$@" > $OUT/${TAG}_${name}_stdout.json 2> $OUT/prof_$name.err
  local f=$(find $OUT/prof_$name -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${TAG}_${name}_kernel_stats.csv; rm -rf $OUT/prof_$name; }
stats bench_steps20 --steps 20 --warmup 5 --no-cpu-baseline --no-extras
stats vio_sweep_200k --only vio_sweep --vio-sweep-patches 200000
stats vio_sweep_1M --only vio_sweep --vio-sweep-patches 1000000
cd $R
timeout 200 bash tools/vio_pmc.sh 1000000 > $OUT/${TAG}_vio_pmc_1M.txt 2>&1
rm -rf $OUT/viopmc
timeout 900 python bench.py > $OUT/${TAG}_bench_n1.json 2> $OUT/${TAG}_bench_n1.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-extras > $OUT/${TAG}_bench_n1_steps20.json 2> /dev/null
cat $OUT/${TAG}_gpu_tests.txt; ls -la $OUT/${TAG}_*
