#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for hb in 128 32; do
mkdir -p $R/gpurun_out/pmc_clk_$hb
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/pmc_clk_$hb/p -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --hessian-batch $hb > $R/gpurun_out/pmc_clk_$hb/p.log 2>&1 || echo "pass failed"
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(float); dur = 0.0; n = 0
disp = {}
for f in glob.glob("$R/gpurun_out/pmc_clk_$hb/p/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "syrk16" in r["Kernel_Name"]:
            dur += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6; n += 1
for f in glob.glob("$R/gpurun_out/pmc_clk_$hb/p/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "syrk16" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
cyc = agg["GRBM_GUI_ACTIVE"] / 8
print(f"hb=$hb: {n} SYRK launches, {dur:.2f} ms, clock {cyc / dur / 1e6:.3f} GHz, MFMA busy {agg['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024) * 100:.1f} %")
PY
done
