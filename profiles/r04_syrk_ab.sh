#!/bin/bash
# r04: SYRK variants side by side: un-profiled throughput of the widest bench input and of the three
# 4096-wide ones, then one rocprofv3 --pmc pass per variant for the clock (GRBM_GUI_ACTIVE / duration) and the matrix pipe's
# busy share (SQ_VALU_MFMA_BUSY_CYCLES).  Variants are library options (GQ_OPTIONS).
# usage (GPU box): bash profiles/r04_syrk_ab.sh "default:" "stagger:syrk_stagger=1" ...   -> gpurun_out/r04/syrk_ab.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04; mkdir -p $OUT
LOG=$OUT/syrk_ab.txt; : > $LOG
cd /tmp && export TMPDIR=/tmp
[ $# -eq 0 ] && set -- "default:" "stagger:syrk_stagger=1"
for V in "$@"; do
  L=${V%%:*}; E=${V#*:}
  for CS in 14336 4096,4096,4096; do
    GQ_OPTIONS=$E CS=$CS NSEQ=32 ITERS=12 python $R/profiles/syrk_probe.py 2>/dev/null | tail -4 | tr '\n' ' ' | sed "s/^/[$L] CS=$CS: /" >> $LOG; echo >> $LOG
  done
  rm -rf $OUT/pmc_$L
  GQ_OPTIONS=$E CS=14336 NSEQ=32 ITERS=2 timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT/pmc_$L -o p -- python $R/profiles/syrk_probe.py > $OUT/pmc_$L.log 2>&1 || echo "[$L] pmc pass failed" >> $LOG
  python3 - <<PY >> $LOG
import csv, glob, collections
agg = collections.defaultdict(float); dur = 0.0; n = 0
for f in glob.glob("$OUT/pmc_$L/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "syrk16" in r["Kernel_Name"]:
            dur += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6; n += 1
for f in glob.glob("$OUT/pmc_$L/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "syrk16" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
if "GRBM_GUI_ACTIVE" in agg and dur > 0:
    cyc = agg["GRBM_GUI_ACTIVE"] / 8
    print(f"[$L] pmc: {n} launches {dur:.2f} ms, clock {cyc / dur / 1e6:.3f} GHz, MFMA busy {agg['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024) * 100:.1f} %, LDS insts {agg['SQ_INSTS_LDS']:.4g}")
PY
  rm -rf $OUT/pmc_$L
done
cat $LOG
