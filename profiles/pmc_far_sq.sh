#!/bin/bash
# SQ counters of the far-update kernel (column loop of one 4096 x 14336 Linear, helper stream off): where do its waves wait?
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
pass() {
  d=$R/gpurun_out/r3/pmc_far_sq_$1; mkdir -p $d; shift
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $d/p -o p -- python $R/profiles/near_probe.py > $d/p.log 2>&1 || { echo "pass failed: $@"; tail -3 $d/p.log; }
  python3 - $d <<'PY'
import csv, glob, collections, sys
d = sys.argv[1]
agg = collections.defaultdict(float); dur = 0.0
for f in glob.glob(d + "/p/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "chain_full_kernel" in r["Kernel_Name"] and int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0) >= 16384:
            dur += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
for f in glob.glob(d + "/p/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "chain_full_kernel" in r["Kernel_Name"] and int(r.get("Grid_Size", 0) or 0) >= 16384 * 32:
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
print(f"  far launches {dur:.2f} ms:", {k: f"{v:.4g}" for k, v in agg.items()})
PY
  rm -rf $d/p
}
pass a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL
pass b SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_WAIT_ANY SQ_IFETCH SQ_INSTS_VMEM SQ_INST_LEVEL_VMEM
pass c SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA
