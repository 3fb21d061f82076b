#!/bin/bash
# Builds libfastlivo_hip.so for gfx950 (cross-compiles without a GPU).
#   build.sh          the release library
#   build.sh debug    libfastlivo_hip_debug.so: the same sources with -DFL_INSTRUMENT (include/fastlivo_hip_debug.h: phase stamps,
#                     fault injection, test aids) -- what tools/ and the abandon/resume and float-chain tests load
#   FL_OUT=path FL_EXTRA_FLAGS="-D..."   A/B variants (tools/README.md)
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
OUT="$HERE/libfastlivo_hip.so"
FLAGS=""
if [ "$1" = "debug" ]; then OUT="$HERE/libfastlivo_hip_debug.so"; FLAGS="-DFL_INSTRUMENT"; fi
"$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Wno-unused-value \
    $FLAGS ${FL_EXTRA_FLAGS} -o "${FL_OUT:-$OUT}" "$HERE/csrc/fastlivo_hip.hip"
