#!/bin/bash
# one rocprofv3 PMC pass over profiles/chain_probe.py, aggregated for the chained far-update GEMM launches
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/pmc_chain_$TAG
timeout 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/pmc_chain_$TAG/p -o p -- python $R/profiles/chain_probe.py > $R/gpurun_out/pmc_chain_$TAG/p.log 2>&1 || echo "pass failed"
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob("$R/gpurun_out/pmc_chain_$TAG/p/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm32_kernel<false, 0, false, 0, 128," in r["Kernel_Name"] or "gemm32_chain_full_kernel" in r["Kernel_Name"]:
            agg["far"][r["Counter_Name"]] += float(r["Counter_Value"]); n["far"] += 1
for k, v in agg.items():
    print(k, {a: f"{b:.3g}" for a, b in v.items()})
dur = 0.0; cnt = 0
for f in glob.glob("$R/gpurun_out/pmc_chain_$TAG/p/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm32_kernel<false, 0, false, 0, 128," in r["Kernel_Name"] or "gemm32_chain_full_kernel" in r["Kernel_Name"]:
            dur += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6; cnt += 1
print("far-update launches", cnt, "total ms", dur)
PY
