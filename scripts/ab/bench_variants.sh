#!/bin/bash
# (developer aid: copies each variant over the product library in turn -- run it on the GPU box copy only, and rebuild afterwards)
# bench.py per-kernel averages with every scripts/ab/lib_*.so
for r in 1 2; do
for lib in scripts/ab/lib_*.so; do
  n=$(basename $lib .so); n=${n#lib_}
  cp $lib cosmology_gnn_simulation_amd/libcgnn_hip.so
  python bench.py --no-cpu-baseline --steps 10 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$n', round(d['ms_per_step'],3), {k:v['avg_ms'] for k,v in d['kernels'].items()})"
done; done
