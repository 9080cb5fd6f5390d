#!/bin/bash
# Builds libuw_hip.so for gfx950 in-tree.  Usage: whisper_amd/csrc/build.sh [-j N]
set -e
cd "$(dirname "$0")"
JOBS=4
[ "$1" = "-j" ] && JOBS=$2
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function"
mkdir -p _build
pids=()
for f in uw_api uw_gemm uw_elementwise uw_attention uw_decode uw_mel uw_align; do
  if [ ! -f _build/$f.o ] || [ $f.hip -nt _build/$f.o ] || [ uw_common.h -nt _build/$f.o ] || [ uw_kernels.h -nt _build/$f.o ] || [ ../../include/uw_hip.h -nt _build/$f.o ]; then
    ( hipcc $FLAGS -c $f.hip -o _build/$f.o ) &
    pids+=($!)
    while [ $(jobs -r | wc -l) -ge $JOBS ]; do sleep 0.2; done
  fi
done
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC -o ../libuw_hip.so _build/*.o
echo "built $(cd .. && pwd)/libuw_hip.so"
