#!/bin/bash
# r06: counters for the north star's second headline, the blocked trailing-update GEMM (fp32 MFMA): matrix-pipe busy % and shader clock of
# the far / near launches of one 4096 x 14336 column loop (profiles/near_probe.py: helper stream off, so every far launch is whole),
# separate rocprofv3 --pmc passes with --kernel-trace only.    usage (GPU box): bash profiles/r06_far_pmc.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06; mkdir -p $OUT
LOG=$OUT/r06_far_pmc.txt; : > $LOG
cd /tmp && export TMPDIR=/tmp
pass() {  # $1 label, $2 counters
  rm -rf $OUT/pmcf
  timeout 300 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $OUT/pmcf -o p -- python $R/profiles/near_probe.py > $OUT/pmcf.log 2>&1 || echo "[$1] pass failed" >> $LOG
  python3 - <<PY >> $LOG
import csv, glob, collections
dur = collections.defaultdict(float); n = collections.Counter(); agg = collections.defaultdict(lambda: collections.defaultdict(float)); fl = collections.defaultdict(float)
def key(r):
    return r["Kernel_Name"].split("(")[0].replace("void gq::", "")[:44]
for f in glob.glob("$OUT/pmcf/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[key(r)] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6; n[key(r)] += 1
for f in glob.glob("$OUT/pmcf/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[key(r)][r["Counter_Name"]] += float(r["Counter_Value"])
for k in sorted(dur, key=lambda k: -dur[k]):
    if "gemm32" not in k and "segment" not in k: continue
    a = agg[k]
    line = f"[$1] {k:46s} n {n[k]:4d} {dur[k]:8.2f} ms"
    cyc = a.get("GRBM_GUI_ACTIVE", 0) / 8
    if cyc > 0:
        line += f"  clock {cyc / dur[k] / 1e6:.3f} GHz  matrix pipe busy {a.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (cyc * 1024) * 100:5.1f} % of SIMD cycles"
    wc = a.get("SQ_WAVE_CYCLES")
    if wc:
        line += "  " + "  ".join(f"{c}={v / wc * 100:.1f}%" for c, v in sorted(a.items()) if c != "SQ_WAVE_CYCLES")
    if a.get("FETCH_SIZE"):
        line += f"  L2-miss reads {a['FETCH_SIZE'] * 2048 / 1e9:.2f} GB"
    print(line)
PY
  rm -rf $OUT/pmcf
}
pass busy "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES"
pass waits "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS"
pass fetch "FETCH_SIZE"
cat $LOG
