SET="GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS"
CGNN_RUN_ARGS="--edge-precision bf16 --node-precision fp16x2" bash scripts/pmc_sets.sh node_block_proj lds "$SET" > /dev/null 2>&1
CGNN_RUN_ARGS="--particles 262144 --edge-precision fp16x2 --node-precision fp16x2" bash scripts/pmc_sets.sh edge_block lds_f2 "$SET" > /dev/null 2>&1
CGNN_RUN_ARGS="--particles 500000 --neighbors 32 --latent 256 --edge-precision bf16 --node-precision fp16x2" bash scripts/pmc_sets.sh edge_block lds_256 "$SET" > /dev/null 2>&1
CGNN_RUN_ARGS="--real-graph" bash scripts/pmc_sets.sh aggregate_planned lds "$SET" > /dev/null 2>&1
CGNN_RUN_ARGS="--edge-precision bf16 --node-precision fp16x2" bash scripts/pmc_sets.sh enc_edge lds "$SET" > /dev/null 2>&1
CGNN_RUN_ARGS="--particles 500000 --latent 256 --edge-precision bf16 --node-precision fp16x2" bash scripts/pmc_sets.sh node_block_proj lds_256 "$SET" > /dev/null 2>&1
for f in gpurun_out/pmc_*_lds*/summary.txt; do echo "=== $f"; cat $f; done
