#!/bin/bash
# Developer tool: time the N16 edge kernel with its tile stream kept in L2 (same instructions, no HBM stream),
# to see how far it is from its compute/LDS bound.  Rebuilds the library on the GPU box; run via gpurun.
set -e
cd "$(dirname "$0")/.."
python scripts/time_ops.py --node-precision fp32x3 > gpurun_out/ablate_base.log 2>&1
make -C cosmology_gnn_simulation_amd/csrc clean > /dev/null
make -C cosmology_gnn_simulation_amd/csrc -j16 EXTRA="-DCGNN_ABLATE_STREAM" > gpurun_out/ablate_build.log 2>&1
python scripts/time_ops.py --node-precision fp32x3 > gpurun_out/ablate_l2.log 2>&1
grep -h "edge_block\|enc_edge" gpurun_out/ablate_base.log gpurun_out/ablate_l2.log
