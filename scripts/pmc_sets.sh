#!/bin/bash
# Developer tool: PMC passes (one counter set per pass, kernel-trace only) for one op.  Run through gpurun:
#   bash scripts/pmc_sets.sh <op> <tag> "<SET 1>" "<SET 2>" ...
set -e
OP=$1; TAG=$2; shift 2
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_${OP}_$TAG
mkdir -p $OUT
REPO=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "$@"; do
  i=$((i+1))
  echo "pass $i: $SET"
  timeout -k 5 240 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/p$i -- python3 $REPO/scripts/run_one_op.py $OP $CGNN_RUN_ARGS > $OUT/p$i.log 2>&1 || echo "pass $i failed"
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
        if any(t in kn for t in ("edge_block", "edge_stream", "aggregate", "node_block", "mlp_rows", "edge_encode", "knn_search", "project", "f2ring")):
            o.write(kn + "\n")
            for c, v in sorted(d.items()):
                o.write(f"   {c:40s} {v / cnt[(kn, c)]:18.1f}  (per dispatch, {cnt[(kn, c)]} dispatches)\n")
print(open("$OUT/summary.txt").read())
PY
