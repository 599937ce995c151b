#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
GQ_FAR_DMA=2 timeout 900 python -m pytest tests -q -m gpu -x -k "gptq or golden or lookahead or far_update" 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
for v in 0 1 2; do
  d=$R/gpurun_out/r3/far_dma_$v; mkdir -p $d
  if [ $v = 0 ]; then unset GQ_FAR_DMA; else export GQ_FAR_DMA=$v; fi
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $d/p -o p -- python $R/profiles/near_probe.py > $d/log.txt 2>&1 || echo "pass failed"
  python3 - $(find $d/p -name '*kernel_trace.csv' | head -1) "FAR_DMA=$v" <<'PY'
import csv, sys
t = n = 0
for r in csv.DictReader(open(sys.argv[1])):
    if ("chain_full" in r["Kernel_Name"] or "chain_dma" in r["Kernel_Name"]) and int(r.get("Grid_Size_X", r.get("Grid_Size", 0))) >= 4096:
        t += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); n += 1
print(f"{sys.argv[2]:12s} far launches {n}  total {t / 1e6:.3f} ms")
PY
  rm -rf $d
done
