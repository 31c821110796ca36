#!/bin/bash
# Developer aid: timing-only variants of one kernel file (wrong results), built next to the product library as
# scripts/ab/lib_<name>.so.   usage: scripts/ab/build_variants.sh <file.hip> name1="-DX" name2="-DY -DZ" ...
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
CS=$ROOT/cosmology_gnn_simulation_amd/csrc
SRC=$1; shift
STEM=$(basename "$SRC" .hip)
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$ROOT/include -Wall -Wno-unused-function"
[ "$STEM" = edge_stream32 ] && [ -z "$NO_S32_FLAGS" ] && FLAGS="$FLAGS -mllvm -amdgpu-mfma-vgpr-form=1"
[ "$STEM" = edge_stream32w ] && FLAGS="$FLAGS -mllvm -amdgpu-mfma-vgpr-form=1 -fno-slp-vectorize"
OTHERS=$(ls $CS/build/*.o | grep -v "/$STEM.o")
for spec in "$@"; do
  name=${spec%%=*}; defs=${spec#*=}
  ( /opt/rocm/bin/hipcc $FLAGS $defs -c $CS/$SRC -o /tmp/ab_$name.o && \
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OTHERS /tmp/ab_$name.o -o $ROOT/scripts/ab/lib_$name.so && echo "built $name" ) &
done
wait
