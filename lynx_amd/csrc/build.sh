#!/bin/bash
# Build liblynxhip.so for gfx950 (cross-compiles without a GPU).  Usage: build.sh [--report]
set -e
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
mkdir -p "$HERE/../_lib"
EXTRA=""
[ "$1" == "--report" ] && EXTRA="-Rpass-analysis=kernel-resource-usage"
hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -ffp-contract=off -fno-slp-vectorize $EXTRA \
  "$HERE/lynx_hip.hip" -o "$HERE/../_lib/liblynxhip.so" -lrccl
