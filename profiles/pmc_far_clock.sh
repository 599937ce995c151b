#!/bin/bash
# clock and matrix-pipe occupancy of the trailing-update kernels (column loop of one 4096 x 14336 Linear, helper stream off):
# GRBM_GUI_ACTIVE / duration = shader clock under this load; SQ_VALU_MFMA_BUSY_CYCLES / (cycles x 4 SIMD x 256 CU)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/r3/pmc_far
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/r3/pmc_far/p -o p -- python $R/profiles/near_probe.py > $R/gpurun_out/r3/pmc_far/p.log 2>&1 || echo "pass failed"
python3 - <<PY
import csv, glob, collections
dur = collections.defaultdict(float); n = collections.Counter(); agg = collections.defaultdict(lambda: collections.defaultdict(float))
def key(r):
    k = r["Kernel_Name"].split("(")[0].replace("void gq::", "")[:40]
    g = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)
    return k + (" large" if g >= 16384 and "chain_full" in k else "")
for f in glob.glob("$R/gpurun_out/r3/pmc_far/p/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[key(r)] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6; n[key(r)] += 1
for f in glob.glob("$R/gpurun_out/r3/pmc_far/p/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[key(r)][r["Counter_Name"]] += float(r["Counter_Value"])
for k in sorted(dur):
    if "gemm32" not in k and "segment" not in k and "syrk" not in k: continue
    a = agg[k]; cyc = a["GRBM_GUI_ACTIVE"] / 8
    if cyc <= 0: continue
    print(f"{k:48s} n {n[k]:4d} {dur[k]:8.2f} ms  clock {cyc / dur[k] / 1e6:.3f} GHz  MFMA busy {a['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024) * 100:5.1f} %")
PY
rm -rf $R/gpurun_out/r3/pmc_far/p
