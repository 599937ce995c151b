#!/bin/bash
# rocprofv3 kernel trace of profiles/chain_probe.py (events off): every kernel of the critical chain, by total time
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/trace_chain $R/gpurun_out/r3
PROF=0 timeout 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_chain/p -o p -- python $R/profiles/chain_probe.py > $R/gpurun_out/trace_chain/p.log 2>&1 || echo "pass failed"
python3 - <<PY
import csv, glob, collections
rows = []
for f in glob.glob("$R/gpurun_out/trace_chain/p/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
i0 = [i for i, r in enumerate(rows) if "col_flags" in r["Kernel_Name"]][-1]
seg = rows[i0:]
agg = collections.defaultdict(lambda: [0, 0])
for r in seg:
    k = r["Kernel_Name"].split("(")[0].replace("void gq::", "").replace("gq::", "")[:64]
    agg[k][0] += 1; agg[k][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
span = int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])
busy = sum(v[1] for v in agg.values())
print(f"last iteration: {len(seg)} kernels, span {span / 1e6:.2f} ms, busy {busy / 1e6:.2f} ms")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:66s} {v[0]:5d} {v[1] / 1e6:7.3f} ms {v[1] / v[0] / 1e3:8.1f} us")
PY
