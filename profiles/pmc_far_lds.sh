#!/bin/bash
# LDS bank conflicts / LDS active cycles / MFMA busy of the far kernels, DMA form vs register-staged form
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
d=$R/gpurun_out/r3/pmc_far_lds_$v; mkdir -p $d
if [ $v = 0 ]; then export GQ_FAR_DMA=1; else unset GQ_FAR_DMA; fi
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $d/p -o p -- python $R/profiles/near_probe.py > $d/p.log 2>&1 || echo "pass failed"
python3 - $d <<'PY'
import csv, glob, collections, sys
d = sys.argv[1]
dur = collections.defaultdict(float); n = collections.Counter(); agg = collections.defaultdict(lambda: collections.defaultdict(float))
def key(r):
    k = r["Kernel_Name"].split("(")[0].replace("void gq::", "")[:40]
    g = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)
    return k + (" large" if g >= 16384 and "chain_" in k else "")
for f in glob.glob(d + "/p/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[key(r)] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6; n[key(r)] += 1
for f in glob.glob(d + "/p/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[key(r)][r["Counter_Name"]] += float(r["Counter_Value"])
for k in sorted(dur):
    if "large" not in k and "near256" not in k: continue
    a = agg[k]; cyc = a["GRBM_GUI_ACTIVE"] / 8
    if cyc <= 0: continue
    print(f"{k:44s} n {n[k]:3d} {dur[k]:7.2f} ms clk {cyc / dur[k] / 1e6:.2f} MFMA {a['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024) * 100:5.1f}%  "
          f"LDS active/CU {a['SQ_LDS_IDX_ACTIVE'] / (cyc * 256) * 100:5.1f}% conflict {a['SQ_LDS_BANK_CONFLICT'] / max(a['SQ_LDS_IDX_ACTIVE'], 1) * 100:5.1f}%  "
          f"wait_any/wave_cycles {a['SQ_WAIT_INST_ANY'] / max(a['SQ_WAVE_CYCLES'], 1) * 100:5.1f}%")
PY
rm -rf $d/p
done
