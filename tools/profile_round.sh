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
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$name -- python $R/bench.py "$@" > $OUT/${TAG}_${name}_stdout.json 2> $OUT/prof_$name.err
  local f=$(find $OUT/prof_$name -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $OUT/${TAG}_${name}_kernel_stats.csv
}
stats bench --no-cpu-baseline --no-extras
stats bench_steps20 --steps 20 --warmup 5 --no-cpu-baseline --no-extras      # the driver's round-end command line
stats at_scale_8M --only at_scale --at-scale-points 8000000
stats at_scale_32M --only at_scale --at-scale-points 32000000
stats vio_sweep_2k --only vio_sweep --vio-sweep-patches 2000
stats vio_sweep_200k --only vio_sweep --vio-sweep-patches 200000
stats vio_sweep_1M --only vio_sweep --vio-sweep-patches 1000000
stats mode23 --only mode23
stats frame --only frame
stats restage --only restage
stats config4 --only config4
stats config5 --only config5
stats pipeline --only pipeline
stats map_scale --only map_scale
cd $R
timeout 600 bash tools/pmc_traffic.sh gpurun_out/${TAG}_pmc_hbm_traffic.json > /dev/null 2>&1
timeout 600 python bench.py > gpurun_out/${TAG}_bench_n1.json 2> gpurun_out/${TAG}_bench_n1.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-extras > gpurun_out/${TAG}_bench_n1_steps20.json 2> /dev/null
# the micro-benchmarks the design decisions of the round quote (DESIGN.md / NOTES.md)
{
  echo "== tools/hop_bench.bin"; timeout 120 tools/hop_bench.bin 2>&1 | tail -30
  echo "== tools/clock_bench.bin"; timeout 120 tools/clock_bench.bin 2>&1 | tail -20
  echo "== tools/chain_profile.py"; timeout 120 python tools/chain_profile.py 2>&1 | tail -14
  echo "== tools/mailbox_ab.py"; timeout 120 python tools/mailbox_ab.py 2>&1 | tail -8
  echo "== tools/vio_pass_bench.py"; timeout 120 python tools/vio_pass_bench.py 2>&1 | tail -1
  echo "== tools/multipass_bench.py"; timeout 120 python tools/multipass_bench.py 2>&1 | tail -1
  echo "== tools/ikfom_pass_bench.py"; timeout 120 python tools/ikfom_pass_bench.py 2>&1 | tail -1
  echo "== tools/computej_breakdown.py"; timeout 120 python tools/computej_breakdown.py 2>&1 | tail -2
  echo "== tools/lioframe_breakdown.py"; timeout 120 python tools/lioframe_breakdown.py 2>&1 | tail -2
  echo "== tools/voxel_bench.py"; timeout 120 python tools/voxel_bench.py 2>&1 | tail -1
  echo "== tools/imu_bench.py"; timeout 120 python tools/imu_bench.py 2>&1 | tail -1
  echo "== tools/pipeline_cpp_bench.py --camera"; timeout 200 python tools/pipeline_cpp_bench.py --camera 2>&1 | tail -5
  echo "== tools/pipeline_cpp_bench.py --raw 24000 --camera"; timeout 200 python tools/pipeline_cpp_bench.py --raw 24000 --camera 2>&1 | tail -5
  echo "== tools/ab_computej.sh (speculating accept on / off)"; timeout 300 bash tools/ab_computej.sh 2>&1 | tail -12
} > gpurun_out/${TAG}_microbench.txt 2>&1
timeout 300 python tools/fuzz_vio_spec.py 1 60 > gpurun_out/${TAG}_vio_spec_fuzz.txt 2>&1
FL_KT_CMD=1 timeout 200 bash tools/ktrace.sh 16 python tools/pipeline_cpp_bench.py --reps 3 > gpurun_out/${TAG}_front_timeline.txt 2>&1
ls -la gpurun_out/${TAG}_* | head -40
