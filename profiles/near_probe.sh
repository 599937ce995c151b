#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/r3/near
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r3/near/p -o p -- python $R/profiles/near_probe.py > $R/gpurun_out/r3/near/log.txt 2>&1 || echo "pass failed"
python3 - $(find $R/gpurun_out/r3/near/p -name '*kernel_trace.csv' | head -1) <<'PY'
import csv, sys, collections
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
# last loop only: from the last scale-search launch on
agg = collections.defaultdict(list)
for r in rows[len(rows) // 2:]:
    n = r["Kernel_Name"]
    if "gemm32" in n or "segment" in n:
        agg[(n.split("(")[0][-60:], r.get("Grid_Size_X", r.get("Grid_Size", "")))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(agg.items()):
    print(f"{k[0]:62s} grid {k[1]:>8s}  n {len(v):4d}  avg {sum(v) / len(v) / 1e3:8.2f} us  min {min(v) / 1e3:8.2f}")
PY
rm -rf $R/gpurun_out/r3/near/p
