#!/bin/bash
# Developer tool: PMC passes for one op (counters in separate passes, kernel-trace only). Run through gpurun.
set -e
OP=${1:-edge_block}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$OP
mkdir -p $OUT
REPO=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT TCC_MISS TCC_EA0_RDREQ TCC_EA0_WRREQ"; do
  i=$((i+1))
  echo "pass $i: $SET"
  timeout -k 5 240 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/p$i -- python3 $REPO/scripts/run_one_op.py $OP "${@:2}" > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        kn = r["Kernel_Name"][:60]
        agg[kn][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(kn, r["Counter_Name"])] += 1
with open("$OUT/summary.txt", "w") as o:
    for kn, d in agg.items():
        if "edge_block" in kn or "aggregate" in kn or "node_block" in kn or "mlp_rows" in kn or "knn_search" in kn or "project" in kn:
            o.write(kn + "\n")
            for c, v in sorted(d.items()):
                o.write(f"   {c:40s} {v / cnt[(kn, c)]:18.1f}  (per dispatch, {cnt[(kn, c)]} dispatches)\n")
print(open("$OUT/summary.txt").read())
PY
