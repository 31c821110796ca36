#!/bin/bash
# Developer tool: one bench line each for the other BASELINE configurations / modes (run through gpurun).
p() { python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1:', round(j['ms_per_step'],3), 'ms', round(j['value']/1e9,3), 'G/s', j['roofline']['kernel'], j['roofline']['frac'])"; }
timeout -k 10 300 python bench.py --config cfg2 --no-cpu-baseline 2>/dev/null | p "cfg2 fp16x2"
timeout -k 10 300 python bench.py --config cfg2 --edge-precision fp32 --node-precision fp32 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | p "cfg2 exact f32"
timeout -k 10 600 python bench.py --config cfg5 --scaling weak --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | p "cfg5 shape, one GPU"
timeout -k 10 300 python bench.py --particles 4096 --neighbors 8 --latent 64 --mp-steps 5 --edge-precision fp32 --node-precision fp32 --steps 50 --warmup 5 --no-cpu-baseline --hip-graph 2>/dev/null | p "cfg1 f32 hip-graph"
timeout -k 10 300 python bench.py --particles 4096 --neighbors 8 --latent 64 --mp-steps 5 --edge-precision fp32 --node-precision fp32 --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | p "cfg1 f32 eager"
timeout -k 10 300 python bench.py --message-source edge --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | p "cfg3 edge mode"
