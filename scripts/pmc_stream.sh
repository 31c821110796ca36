#!/bin/bash
# Developer tool: SQ / TCC counters of the one-launch edge-stream kernels side by side (scripts/ab_stream.py under
# rocprofv3 --pmc, one counter set per pass, kernel-trace only).  Run through gpurun:
#   gpurun --timeout 900 -- 'bash scripts/pmc_stream.sh <tag> "<variants>" [counter sets...]'
TAG=${1:-x}; VARS=${2:-tile32,tile32w:0,tile32w:1}; shift 2
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out/pmc_stream_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
if [ $# -eq 0 ]; then
  set -- "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" \
         "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" \
         "SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM SQ_WAVES" \
         "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_SCA" \
         "GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT" \
         "FETCH_SIZE" "WRITE_SIZE"
fi
rocprofv3 -L > $OUT/counters_available.txt 2>&1
i=0
for SET in "$@"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/p$i -- python3 $REPO/scripts/ab_stream.py --variants $VARS --rounds 2 $CGNN_AB_ARGS > $OUT/p$i.log 2>&1 || echo "pass $i ($SET) failed"
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        kn = r["Kernel_Name"]
        if "edge_stream" not in kn: continue
        kn = kn.split("(")[0][:70]
        agg[kn][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(kn, r["Counter_Name"])] += 1
dur = collections.defaultdict(list)
for f in glob.glob("$OUT/p*/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        kn = r["Kernel_Name"]
        if "edge_stream" not in kn: continue
        dur[kn.split("(")[0][:70]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
with open("$OUT/summary.txt", "w") as o:
    for kn, d in agg.items():
        ds = sorted(dur.get(kn, [0]))
        o.write(f"{kn}   median {ds[len(ds)//2]:.3f} ms under the profiler ({len(ds)} dispatches)\n")
        for c, v in sorted(d.items()):
            o.write(f"   {c:34s} {v / cnt[(kn, c)]:20.1f}  (per dispatch, {cnt[(kn, c)]} dispatches)\n")
print(open("$OUT/summary.txt").read())
PY
