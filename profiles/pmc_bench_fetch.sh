#!/bin/bash
# roofline.traffic: L2-miss read bytes of the SYRK launches of the bench command itself (rocprofv3 --pmc FETCH_SIZE,
# its own run with --kernel-trace only).  FETCH_SIZE is in KiB-like units of 1 KB and under-reports by 2x on gfx950
# (MI355X_MICROARCH.md): bytes = FETCH_SIZE * 1024 * 2.   usage (GPU box): bash profiles/pmc_bench_fetch.sh [bench args]
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/pmc_fetch
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fetch/p -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $R/gpurun_out/pmc_fetch/p.log 2>&1 || echo "pass failed"
python3 - <<PY
import csv, glob
rows = []
for f in glob.glob("$R/gpurun_out/pmc_fetch/p/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "syrk16" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
            rows.append((int(r["Dispatch_Id"]), int(r["Grid_Size"]), float(r["Counter_Value"])))
rows.sort()
for d, g, v in rows:
    print(f"dispatch {d:6d} grid {g:8d} threads: FETCH_SIZE {v:.4g} -> {v * 1024 * 2 / 1e9:7.2f} GB")
if rows:
    print(f"{len(rows)} launches, average {sum(v for _, _, v in rows) * 2048 / 1e9 / len(rows):.2f} GB per launch")
PY
