#!/bin/bash
# r04: matrix-pipe occupancy and clock of the two forms of the in-place SYRK (separate rocprofv3 --pmc passes, --kernel-trace
# only) on the widest bench input (C = 14336, 65 536 tokens).  usage (GPU box): bash profiles/r04_syrk_w4_pmc.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04; mkdir -p $OUT
LOG=$OUT/syrk_w4_pmc.txt; : > $LOG
cd /tmp && export TMPDIR=/tmp
pass() {  # $1 = label, $2 = counters, $3 = options
  rm -rf $OUT/pmcw_$1
  GQ_OPTIONS="$3" CS=${CS:-14336} NSEQ=32 ITERS=3 timeout 300 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $OUT/pmcw_$1 -o p -- python $R/profiles/syrk_probe.py > $OUT/pmcw_$1.log 2>&1 || echo "[$1] pass failed ($2)" >> $LOG
  python3 - <<PY >> $LOG
import csv, glob, collections
agg = collections.defaultdict(float); dur = 0.0; n = 0
for f in glob.glob("$OUT/pmcw_$1/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "syrk16" in r["Kernel_Name"]:
            dur += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6; n += 1
for f in glob.glob("$OUT/pmcw_$1/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "syrk16" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
print(f"[$1 '$3'] {n} launches {dur:.2f} ms: " + "  ".join(f"{k}={v:.5g}" for k, v in sorted(agg.items())))
wc = agg.get("SQ_WAVE_CYCLES")
if wc:
    print("[$1]   share of SQ_WAVE_CYCLES: " + "  ".join(f"{k}={v / wc * 100:.1f}%" for k, v in sorted(agg.items()) if k not in ("SQ_WAVE_CYCLES", "GRBM_GUI_ACTIVE")))
g = agg.get("GRBM_GUI_ACTIVE")
if g and dur:
    cyc = g / 8  # summed over the 8 XCDs
    print(f"[$1]   clock = {cyc / (dur * 1e-3) / 1e9:.3f} GHz   matrix pipe busy = {agg.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (cyc * 1024) * 100:.1f} % of SIMD cycles")
PY
  rm -rf $OUT/pmcw_$1
}
for o in "" "syrk_w4=1"; do
  pass busy "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "$o"
  pass waits "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" "$o"
  pass lds "SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM" "$o"
done
cat $LOG
