#!/bin/bash
# Round profile for profiles/: rocprofv3 kernel stats of the default bench.py run, PMC traffic passes (FETCH_SIZE and
# WRITE_SIZE in separate passes, kernel-trace only) and SQ counters for the hot kernels.  Run through gpurun:
#   gpurun --timeout 1100 -- 'bash scripts/profile_round.sh r02_x'
# Results land in gpurun_out/profile_<tag>/; scripts/profile_collect.py copies the summaries into profiles/ and
# rewrites profiles/traffic.json (keyed by kernel source hash).
TAG=${1:-r02}
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out/profile_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
echo "== bench under rocprofv3 --kernel-trace --stats"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -- python3 $REPO/bench.py > $OUT/bench.json 2> $OUT/bench.err || echo "bench profile failed"
cp $OUT/bench/*/*kernel_stats.csv $OUT/bench_kernel_stats.csv 2>/dev/null
pmc() {   # name op args... -- then counter sets      (ONLY="edge_stream node_block": just those passes)
  local name=$1 op=$2; shift 2
  if [ -n "$ONLY" ] && ! echo " $ONLY " | grep -q " $name "; then return; fi
  local args=()
  while [ "$1" != "--" ]; do args+=("$1"); shift; done; shift
  local i=0
  for SET in "$@"; do
    i=$((i+1))
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/pmc_$name/p$i -- python3 $REPO/scripts/run_one_op.py $op "${args[@]}" > $OUT/pmc_$name.p$i.log 2>&1 || echo "pmc $name pass $i failed"
  done
}
TRAFFIC=("FETCH_SIZE" "WRITE_SIZE")
SQ=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM SQ_WAVES" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT")
echo "== edge stream (+ encoder)"
pmc edge_stream edge_stream --real-graph --edge-precision bf16 --node-precision fp16x2 -- "${TRAFFIC[@]}" "${SQ[@]}"
echo "== node block (fp16x2, projections fused)"
pmc node_block node_block_proj --edge-precision bf16 --node-precision fp16x2 -- "${TRAFFIC[@]}" "${SQ[@]}"
echo "== aggregation, planned and plain, on the k-NN graph"
pmc aggregate_planned aggregate_planned --real-graph -- "${TRAFFIC[@]}" "TCC_HIT_sum TCC_MISS_sum"
pmc aggregate aggregate --real-graph -- "${TRAFFIC[@]}" "TCC_HIT_sum TCC_MISS_sum"
echo "== general scatter-add (shuffled edge list: one atomic row per edge)"
pmc scatter scatter_shuffled --real-graph -- "${TRAFFIC[@]}"
echo "== f32-accuracy edge block (cfg2 size)"
pmc edge_block_f2 edge_block --particles 262144 --edge-precision fp16x2 --node-precision fp16x2 -- "${TRAFFIC[@]}" "${SQ[0]}"
echo "== 256-wide edge block (cfg5's shape: 1 M particles, k = 32) and the edge stream at cfg4's size (4 M particles)"
pmc edge_block_256 edge_block --particles 1000000 --neighbors 32 --latent 256 --edge-precision bf16 --node-precision fp16x2 -- "${TRAFFIC[@]}" "${SQ[0]}"
pmc edge_stream_4m edge_stream --particles 4000000 --real-graph --edge-precision bf16 --node-precision fp16x2 -- "${TRAFFIC[@]}"
python3 $REPO/scripts/profile_collect.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
