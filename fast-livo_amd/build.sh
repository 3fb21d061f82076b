#!/bin/bash
# Builds libfastlivo_hip.so for gfx950 (cross-compiles without a GPU).
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
"$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Wno-unused-value \
    ${FL_EXTRA_FLAGS} -o "${FL_OUT:-$HERE/libfastlivo_hip.so}" "$HERE/csrc/fastlivo_hip.hip"
