#!/bin/bash
# rocprofv3 kernel trace of a 3-layer whole-model run (bench.py --workload llama3-8b-model-q4k --layers 3), summarised by
# profiles/trace_model.py.  usage: bash profiles/trace_model.sh [extra bench.py flags, e.g. --fused-forward]
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
tag=$(echo "model$*" | tr -d ' -')
mkdir -p $R/gpurun_out/r3/$tag
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r3/$tag/p -o p -- python $R/bench.py --workload llama3-8b-model-q4k --layers ${LAYERS:-3} --steps 1 --warmup 0 "$@" > $R/gpurun_out/r3/$tag/bench.log 2>&1 || echo "pass failed"
f=$(find $R/gpurun_out/r3/$tag/p -name '*kernel_trace.csv' | head -1)
python3 $R/profiles/trace_model.py $f | tee $R/gpurun_out/r3/$tag/summary.txt
python3 - $f <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: [0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    agg[r["Kernel_Name"][:110]][0] += 1; agg[r["Kernel_Name"][:110]][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{v[1] / 1e6:9.2f} ms {v[0]:7d} {v[1] / v[0] / 1e3:8.1f} us  {k}")
PY
python3 $R/profiles/trace_phases.py $f | tee $R/gpurun_out/r3/$tag/phases.txt
rm -rf $R/gpurun_out/r3/$tag/p
