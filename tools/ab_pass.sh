#!/bin/bash
# A/B of library builds inside ONE gpurun call (box-to-box variance is +-4 %): per-pass time of the LIO and VIO pass kernels,
# multi-pass and one-launch-per-pass, interleaved twice.  usage: tools/ab_pass.sh build_ab/lib_a.so build_ab/lib_b.so ...
for rep in 1 2; do
  for lib in "" "$@"; do
    echo "== rep $rep lib ${lib:-current}"
    FL_LIB_PATH=$lib python tools/multipass_bench.py
    [ -z "$AB_QUICK" ] && FL_LIB_PATH=$lib FL_NO_MULTIPASS=1 python tools/multipass_bench.py
    FL_LIB_PATH=$lib python tools/vio_pass_bench.py
  done
done
