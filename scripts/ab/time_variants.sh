#!/bin/bash
# Developer aid: time one op with every scripts/ab/lib_*.so (same process order each round, 2 rounds)
OP=${1:-edge_stream+enc}; shift
for round in 1 2; do
  for lib in scripts/ab/lib_*.so; do
    n=$(basename $lib .so); n=${n#lib_}
    printf "%-12s " $n
    CGNN_LIB_PATH=$PWD/$lib python scripts/time_ops.py --only "$OP" "$@" 2>/dev/null | tail -1
  done
done
