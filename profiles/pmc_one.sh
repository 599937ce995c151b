#!/bin/bash
# one rocprofv3 PMC pass over profiles/syrk_probe.py: pmc_one.sh <tag> <counter> [<counter> ...]
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/pmc1_$TAG
timeout 150 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/pmc1_$TAG/p -o p -- python $R/profiles/syrk_probe.py > $R/gpurun_out/pmc1_$TAG/p.log 2>&1 || echo "pass failed"
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$R/gpurun_out/pmc1_$TAG/p/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if "syrk16" in r["Kernel_Name"]:
            agg[r["Counter_Name"]][0] += float(r["Counter_Value"]); agg[r["Counter_Name"]][1] += 1
    for k, (v, n) in agg.items():
        print(f"{k:36s} total={v:.4g} dispatch_rows={n}")
for f in sorted(glob.glob("$R/gpurun_out/pmc1_$TAG/p/**/*kernel_trace.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "syrk16" in r["Kernel_Name"]:
            print("duration_ms", (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
PY
