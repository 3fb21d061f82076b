#!/bin/bash
# Run on the GPU box from the repo root: every rocprofv3 pass behind the numbers of the bench line, reduced to the small summaries
# that are committed under profiles/ (kernel-trace stats per bench section; FETCH_SIZE / WRITE_SIZE in separate --pmc passes).
# usage: tools/profile_round.sh r02        -> gpurun_out/r02_*.{csv,json}
TAG=${1:-rXX}
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
stats() {   # name, bench args...
  local name=$1; shift
  rm -rf $OUT/prof_$name
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$name -- python $R/bench.py "$@" > $OUT/${TAG}_${name}_stdout.json 2> $OUT/prof_$name.err
  local f=$(find $OUT/prof_$name -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $OUT/${TAG}_${name}_kernel_stats.csv
}
stats bench --no-cpu-baseline --no-extras
stats at_scale_8M --only at_scale --at-scale-points 8000000
stats at_scale_32M --only at_scale --at-scale-points 32000000
stats vio_sweep_2k --only vio_sweep --vio-sweep-patches 2000
stats vio_sweep_200k --only vio_sweep --vio-sweep-patches 200000
stats vio_sweep_1M --only vio_sweep --vio-sweep-patches 1000000
stats mode23 --only mode23
stats frame --only frame
stats restage --only restage
cd $R
bash tools/pmc_traffic.sh gpurun_out/${TAG}_pmc_hbm_traffic.json > /dev/null 2>&1
python bench.py > gpurun_out/${TAG}_bench_n1.json 2> gpurun_out/${TAG}_bench_n1.err
ls -la gpurun_out/${TAG}_* | head -40
