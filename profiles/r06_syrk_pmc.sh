#!/bin/bash
# r06 (VERDICT r05 next #4 / missing #6): counters for the SHIPPED SYRK (four waves, syrk_ck = 256) -- matrix-pipe busy %, shader
# clock, wave-cycle split, L2-miss reads -- on the widest bench input (C = 14336, 65 536 tokens), random and all-zero operands.
# Separate rocprofv3 --pmc passes with --kernel-trace only.   usage (GPU box): bash profiles/r06_syrk_pmc.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06; mkdir -p $OUT
LOG=$OUT/r06_syrk_pmc.txt; : > $LOG
cd /tmp && export TMPDIR=/tmp
pass() {  # $1 = label, $2 = counters, $3 = DATA, $4 = options
  rm -rf $OUT/pmcs_$1
  GQ_OPTIONS="$4" DATA=$3 CS=${CS:-14336} NSEQ=32 ITERS=3 timeout 300 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $OUT/pmcs_$1 -o p -- python $R/profiles/syrk_probe.py > $OUT/pmcs_$1.log 2>&1 || echo "[$1] pass failed ($2)" >> $LOG
  python3 - <<PY >> $LOG
import csv, glob, collections
agg = collections.defaultdict(float); dur = 0.0; n = 0
for f in glob.glob("$OUT/pmcs_$1/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "syrk16" in r["Kernel_Name"]:
            dur += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6; n += 1
for f in glob.glob("$OUT/pmcs_$1/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "syrk16" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
C = int("${CS:-14336}".split(",")[0]); T = 32 * 2048
fl = n * T * C * (C + 128)
print(f"[$1 data=$3 '$4'] {n} launches {dur:.2f} ms ({fl / (dur * 1e-3) / 1e12 if dur else 0:.0f} TFLOP/s algorithmic under the profiler): " + "  ".join(f"{k}={v:.5g}" for k, v in sorted(agg.items())))
wc = agg.get("SQ_WAVE_CYCLES")
if wc:
    print("[$1]   share of SQ_WAVE_CYCLES: " + "  ".join(f"{k}={v / wc * 100:.1f}%" for k, v in sorted(agg.items()) if k not in ("SQ_WAVE_CYCLES", "GRBM_GUI_ACTIVE")))
g = agg.get("GRBM_GUI_ACTIVE")
if g and dur:
    cyc = g / 8  # summed over the 8 XCDs
    print(f"[$1]   clock = {cyc / (dur * 1e-3) / 1e9:.3f} GHz   matrix pipe busy = {agg.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (cyc * 1024) * 100:.1f} % of SIMD cycles")
fs = agg.get("FETCH_SIZE")
if fs and n:
    alg = (T * C * 2 + 2 * C * C * 4) / 1e9
    print(f"[$1]   L2-miss reads = {fs * 2048 / 1e9 / n:.2f} GB per launch (FETCH_SIZE x 2 KB, gfx950), algorithmic {alg:.2f} GB: ratio {fs * 2048 / 1e9 / n / alg:.2f}")
PY
  rm -rf $OUT/pmcs_$1
}
for d in random zeros; do
  pass busy_$d "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" $d "${OPTS:-}"
  pass waits_$d "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" $d "${OPTS:-}"
  pass fetch_$d "FETCH_SIZE" $d "${OPTS:-}"
done
cat $LOG
