#!/bin/bash
# A/B of ComputeJ (bench.py --only frame: vio_computej_ms) and of the forced VIO pass inside ONE gpurun call: the current library with and
# without the speculating accept (FL_OPT_VIO_SPECULATE), and any other library builds given.  usage: tools/ab_computej.sh [build_ab/lib_a.so ...]
one() {
    timeout 200 python bench.py --only frame 2>/dev/null | grep "^{" | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read())['result']; print({k: round(d[k],4) for k in ('frame_ms','lio_frame_ms','vio_computej_ms') if k in d})"
    timeout 100 python tools/vio_pass_bench.py 2>/dev/null | tail -1
}
for rep in 1 2 3; do
  echo "== rep $rep current, speculating"; FL_LIB_PATH= one
  echo "== rep $rep current, FL_OPT_VIO_SPECULATE 0"; FL_LIB_PATH= FL_NO_VIO_SPEC=1 one
  for lib in "$@"; do echo "== rep $rep $lib"; FL_LIB_PATH=$lib one; done
done
