#!/bin/bash
# r06: does the SYRK's fabric traffic move its clock on RANDOM operands?  Clean A/B of the super-tile shape
# an XCD's 32 workgroups share (option syrk_gw: 8 = 4x8, 12 panels per 32 tiles, shipped; 16 = 2x16, 18 panels: +50 % panel
# reads; 32 = 1x32, 33 panels: +175 %; 4 = 8x4, the transposed shipped shape) at C = 14336, 65 536 tokens:
# time from HIP events (no profiler), then L2-miss reads from a separate --pmc FETCH_SIZE pass.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06; mkdir -p $OUT
LOG=$OUT/r06_syrk_traffic_ab.txt; : > $LOG
cd /tmp && export TMPDIR=/tmp
for d in random zeros; do
for gw in 8 16 32 4; do
  t=$(GQ_OPTIONS="syrk_gw=$gw" DATA=$d CS=14336 NSEQ=32 ITERS=6 timeout 200 python $R/profiles/syrk_probe.py | tail -3 | awk '{print $5}' | sort -n | tail -1)
  rm -rf $OUT/pmct
  GQ_OPTIONS="syrk_gw=$gw" DATA=$d CS=14336 NSEQ=32 ITERS=3 timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmct -o p -- python $R/profiles/syrk_probe.py > $OUT/pmct.log 2>&1 || echo "pmc pass failed" >> $LOG
  python3 - <<PY >> $LOG
import csv, glob
v = [float(r["Counter_Value"]) for f in glob.glob("$OUT/pmct/**/*counter_collection.csv", recursive=True)
     for r in csv.DictReader(open(f)) if "syrk16" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
gb = sum(v) * 2048 / 1e9 / max(1, len(v))
print(f"data=$d syrk_gw=$gw: best of 3 = $t TFLOP/s (no profiler)   L2-miss reads {gb:.2f} GB per launch ({gb / 3.52:.2f} x algorithmic)")
PY
  rm -rf $OUT/pmct
done
done
cat $LOG
