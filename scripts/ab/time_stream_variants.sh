#!/bin/bash
# Developer aid: scripts/ab_stream.py with every scripts/ab/lib_*.so (one process per library, same box)
for lib in scripts/ab/lib_*.so; do
  n=$(basename $lib .so); n=${n#lib_}
  CGNN_LIB_PATH=$PWD/$lib timeout -k 10 120 python scripts/ab_stream.py "$@" 2>/dev/null | grep -v amdgpu.ids | sed "s/^/$(printf '%-10s' $n) /"
done
