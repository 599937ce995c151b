#!/bin/bash
# rocprofv3 PMC passes (memory-path view) for the SYRK kernel. Output: gpurun_out/pmc2_<tag>/
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-r1}
cd /tmp && export TMPDIR=/tmp
i=0; mkdir -p $R/gpurun_out/pmc2_$TAG
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_LATENCY_sum" \
           "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TA_BUSY_sum TA_FLAT_READ_LDS_WAVEFRONTS_sum" \
           "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc2_$TAG/p$i -o p -- python $R/profiles/syrk_probe.py > $R/gpurun_out/pmc2_$TAG/p$i.log 2>&1 || echo "pass $i failed"
done
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$R/gpurun_out/pmc2_$TAG/p*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if "syrk16" in r["Kernel_Name"]:
            agg[r["Counter_Name"]][0] += float(r["Counter_Value"]); agg[r["Counter_Name"]][1] += 1
    for k, (v, n) in agg.items():
        print(f"{k:36s} total={v:.4g} dispatch_rows={n}")
PY
