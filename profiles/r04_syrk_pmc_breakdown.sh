#!/bin/bash
# r04: where the SYRK's wave cycles go -- separate rocprofv3 --pmc passes (--kernel-trace only) over the
# widest bench input (C = 14336, 65 536 tokens): parked (s_waitcnt / s_barrier), issue-stalled, issuing, per instruction class,
# and the FIFO-full counters of the vector-memory and LDS paths.  usage (GPU box): bash profiles/r04_syrk_pmc_breakdown.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04; mkdir -p $OUT
LOG=$OUT/syrk_pmc_breakdown.txt; : > $LOG
cd /tmp && export TMPDIR=/tmp
pass() {  # $1 = label, $2 = counters
  rm -rf $OUT/pmcb_$1
  CS=${CS:-14336} NSEQ=32 ITERS=2 timeout 300 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $OUT/pmcb_$1 -o p -- python $R/profiles/syrk_probe.py > $OUT/pmcb_$1.log 2>&1 || echo "[$1] pass failed ($2)" >> $LOG
  python3 - <<PY >> $LOG
import csv, glob, collections
agg = collections.defaultdict(float); dur = 0.0; n = 0
for f in glob.glob("$OUT/pmcb_$1/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "syrk16" in r["Kernel_Name"]:
            dur += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6; n += 1
for f in glob.glob("$OUT/pmcb_$1/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "syrk16" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
print(f"[$1] {n} launches {dur:.2f} ms: " + "  ".join(f"{k}={v:.5g}" for k, v in sorted(agg.items())))
wc = agg.get("SQ_WAVE_CYCLES")
if wc:
    print("[$1]   share of SQ_WAVE_CYCLES: " + "  ".join(f"{k}={v / wc * 100:.1f}%" for k, v in sorted(agg.items()) if k != "SQ_WAVE_CYCLES" and k != "GRBM_GUI_ACTIVE"))
PY
  rm -rf $OUT/pmcb_$1
}
pass waits "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS"
pass classes "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"
pass issue "SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VALU SQ_INSTS_VMEM SQ_INSTS_LDS"
pass fifo "SQ_WAVE_CYCLES SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_IDX_ACTIVE"
pass busy "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_LEVEL_WAVES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS"
cat $LOG
