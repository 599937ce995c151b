#!/bin/bash
# rocprofv3 PMC passes for the SYRK kernel (run on the GPU box via gpurun). Output: gpurun_out/pmc_syrk_<tag>/
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-r1}
cd /tmp && export TMPDIR=/tmp
i=0; mkdir -p $R/gpurun_out/pmc_syrk_$TAG
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_LDS_UNALIGNED_STALL" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc_syrk_$TAG/p$i -o p -- python $R/profiles/syrk_probe.py > $R/gpurun_out/pmc_syrk_$TAG/p$i.log 2>&1 || echo "pass $i failed"
done
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$R/gpurun_out/pmc_syrk_$TAG/p*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if "syrk16" in r["Kernel_Name"]:
            agg[r["Counter_Name"]][0] += float(r["Counter_Value"]); agg[r["Counter_Name"]][1] += 1
    for k, (v, n) in agg.items():
        print(f"{k:32s} total={v:.4g} dispatch_rows={n}")
PY
