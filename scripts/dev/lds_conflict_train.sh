#!/bin/bash
# Developer aid: one PMC pass (default: the LDS bank-conflict counters; PMC="..." for others) over the training step's kernels (run through gpurun)
REPO=$GRAFT_REPO_ROOT; OUT=$REPO/gpurun_out/pmc_train_lds; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --pmc ${PMC:-GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS} --output-format csv -d $OUT/p1 -- python3 $REPO/scripts/time_train.py --train-precision fp32x3 --iters 2 > $OUT/p1.log 2>&1
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob("$OUT/p1/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        kn = r["Kernel_Name"][:70]
        agg[kn][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(kn, r["Counter_Name"])] += 1
rows = []
for kn, d in agg.items():
    g = d["GRBM_GUI_ACTIVE"]; rows.append((g, kn, d, cnt[(kn, "GRBM_GUI_ACTIVE")]))
for g, kn, d, n in sorted(rows, reverse=True)[:14]:
    print(f"{kn:70s} n={n:4d} gui/8={g/8/n:12.0f} " + ' '.join(f'{k}={v/n:.3e}' for k, v in sorted(d.items()) if k != 'GRBM_GUI_ACTIVE'))
PY
