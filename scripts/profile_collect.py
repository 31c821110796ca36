"""Developer tool: summarise the rocprofv3 counter CSVs of scripts/profile_round.sh (per kernel: mean counter value per
dispatch) and print the profiles/traffic.json entries they imply.  HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE)
KiB: FETCH_SIZE under-reports wide coalesced reads by 2 on gfx950 (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is exact."""
import collections
import csv
import glob
import hashlib
import json
import os
import sys

out = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEEP = ("edge_block", "edge_stream", "aggregate", "node_block", "f2ring", "mlp_rows", "project", "knn_search")
summary = {}
for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    for f in glob.glob(os.path.join(d, "p*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            kn = r["Kernel_Name"]
            agg[kn][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[(kn, r["Counter_Name"])] += 1
    name = os.path.basename(d)[4:]
    print(f"==== {name}")
    for kn, cs in agg.items():
        if not any(t in kn for t in KEEP):
            continue
        print(kn[:110])
        vals = {}
        for c, v in sorted(cs.items()):
            vals[c] = v / cnt[(kn, c)]
            print(f"   {c:32s} {vals[c]:18.1f}   per dispatch ({cnt[(kn, c)]} dispatches)")
        if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
            hbm = (2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024
            print(f"   => HBM bytes per launch (2 x FETCH + WRITE) = {hbm / 1e9:.3f} GB")
            summary.setdefault(name, {})[kn] = hbm


def kernel_ms(name, needle):
    """Median duration (ms) of the kernel in the PMC passes' own kernel traces."""
    ds = []
    for f in glob.glob(os.path.join(out, f"pmc_{name}", "p*", "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if needle in r["Kernel_Name"]:
                ds.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    ds.sort()
    return ds[len(ds) // 2] if ds else None


def counters(name, needle):
    for d in [os.path.join(out, f"pmc_{name}")]:
        agg = collections.defaultdict(float)
        cnt = collections.Counter()
        for f in glob.glob(os.path.join(d, "p*", "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if needle in r["Kernel_Name"]:
                    agg[r["Counter_Name"]] += float(r["Counter_Value"])
                    cnt[r["Counter_Name"]] += 1
        return {c: v / cnt[c] for c, v in agg.items()}
    return {}


def sha16(src):
    sys.path.insert(0, root)
    import bench            # one definition of "the source a number was measured on": the .hip file and its headers
    return bench.kernel_source_sha16(src)


def pick(name, needle):
    for kn, v in summary.get(name, {}).items():
        if needle in kn:
            return int(v)
    return None


entries = {}
for key, name, needle, src in (("edge_stream+enc:1000000:16:128:10", "edge_stream", "edge_stream32w", "edge_stream32w.hip"),
                               ("node_block:1000000:128", "node_block", "f2ring", "node_block_f2.hip"),
                               ("aggregate:1000000:16:128", "aggregate_planned", "aggregate_planned", "aggregate_plan.hip"),
                               ("aggregate_plain:1000000:16:128", "aggregate", "aggregate_fixedk", "runtime.hip"),
                               ("scatter_shuffled:1000000:16:128", "scatter", "aggregate_scatter", "runtime.hip"),
                               ("edge_block:262144:16:128:fp16x2", "edge_block_f2", "edge_block_f2ring", "edge_block_f2.hip"),
                               ("edge_block:1000000:32:256:bf16", "edge_block_256", "edge_block_ring256", "edge_block_ring256.hip"),
                               ("edge_stream+enc:4000000:16:128:10", "edge_stream_4m", "edge_stream32w", "edge_stream32w.hip")):
    v = pick(name, needle)
    if v is not None:
        entries[key] = {"hbm_bytes_per_launch": v, "source": src, "source_sha16": sha16(src),
                        "profile": f"profiles/{os.path.basename(out).replace('profile_', '')}_pmc_summary.txt"}
        ms, cs = kernel_ms(name, needle), counters(name, needle)
        if ms and cs.get("GRBM_GUI_ACTIVE"):
            cycles = cs["GRBM_GUI_ACTIVE"] / 8.0
            entries[key]["kernel_ms_under_profiler"] = round(ms, 3)
            entries[key]["clock_ghz_under_profiler"] = round(cycles / (ms * 1e-3) / 1e9, 3)
            if cs.get("SQ_VALU_MFMA_BUSY_CYCLES"):
                entries[key]["mfma_busy_frac"] = round(cs["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / cycles, 3)
print("==== traffic.json entries")
print(json.dumps(entries, indent=1))
json.dump(entries, open(os.path.join(out, "traffic_entries.json"), "w"), indent=1)
