#!/bin/bash
# one rocprofv3 PMC pass over profiles/chol_probe.py (CS=14336), aggregated per kernel name for the 4 top-level GEMMs
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/pmc_chol_$TAG
CS=14336 timeout 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/pmc_chol_$TAG/p -o p -- python $R/profiles/chol_probe.py > $R/gpurun_out/pmc_chol_$TAG/p.log 2>&1 || echo "pass failed"
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob("$R/gpurun_out/pmc_chol_$TAG/p/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm3" in r["Kernel_Name"] and r["Grid_Size"] == str(14336 * 56):
            k = r["Kernel_Name"].split("(")[0][-36:]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in agg.items():
    print(k, {a: f"{b:.3g}" for a, b in v.items()})
PY
