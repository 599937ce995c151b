#!/bin/bash
# Round-6 judged artifacts, one per BASELINE config that fits one GPU (every round since r03):
#   gpurun_out/r06/r06_bench_<workload>.json            the full bench dict (roofline, trailing_update, cpu_baseline, ...)
#   gpurun_out/r06/r06_bench_<workload>_compact.json    the compact headline line (bench.py's LAST stdout line)
#   gpurun_out/r06/r06_bench_<workload>_kernel_stats.csv  rocprofv3 --kernel-trace --stats of the same workload
#   gpurun_out/r06/r06_syrk_traffic.json + r06_syrk_fetch_dispatches.csv   rocprofv3 --pmc FETCH_SIZE pass (default workload)
# usage (GPU box): bash profiles/collect_r06.sh [workload ...]      then copy gpurun_out/r06/* into profiles/
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06
mkdir -p $OUT
WLS=${@:-"llama3-8b-block-q4k llama3-8b-block-mixed tinyllama-block-q4k llama3-70b-block-q4k mixtral-block"}
# L2-miss read bytes of the SYRK launches of the default bench command (its own pass: --pmc with --kernel-trace only)
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-whole-model --no-side-legs > $OUT/pmc_fetch.log 2>&1 || echo "pmc pass failed"
python3 - <<PY
import csv, glob, hashlib, json, os
rows = []
for f in glob.glob("$OUT/pmc_fetch/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "syrk16" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
            rows.append((int(r["Dispatch_Id"]), int(r["Grid_Size"]), float(r["Counter_Value"])))
rows.sort()
with open("$OUT/r06_syrk_fetch_dispatches.csv", "w") as f:
    f.write("dispatch_id,grid_size,FETCH_SIZE,GB_corrected\n")
    for d, g, v in rows:
        f.write(f"{d},{g},{v:.6g},{v * 2048 / 1e9:.3f}\n")
if rows:
    per = sum(v for _, _, v in rows) * 2048 / 1e9 / len(rows)
    T, n4, n14 = 65536, 3, 1
    # algorithmic bytes of a step's four folds (X once, H read + written), over the launches a step makes (r06: one grid per fold)
    lps = len(rows) / 4  # the pass runs 4 steps (1 warm-up + 2 timed + the latency step)
    alg = (4 * (n4 * (T * 4096 * 2 + 2 * 4096 * 4096 * 4)) + 4 * (T * 14336 * 2 + 2 * 14336 * 14336 * 4)) / lps / 1e9
    h = hashlib.sha256()
    for fn in sorted(glob.glob("$R/gptq-gguf-toolkit_amd/csrc/*.h*")):
        h.update(open(fn, "rb").read())
    json.dump({"GB_per_launch": round(per, 2), "algorithmic_GB_per_launch": round(alg, 2), "launches": len(rows),
               "kernel_sources_sha256": h.hexdigest()[:16],
               "measured_on": "rocprofv3 --pmc FETCH_SIZE x 2 KB (gfx950 correction) of bench.py --steps 2 --warmup 1, "
                              "default workload; per-dispatch values: profiles/r06_syrk_fetch_dispatches.csv"},
              open("$OUT/r06_syrk_traffic.json", "w"))
    print(f"{len(rows)} SYRK launches, {per:.2f} GB per launch (algorithmic {alg:.2f})")
PY
rm -rf $OUT/pmc_fetch
# the same pass without the rendezvous inside a tile (option syrk_ck = 0): what the checkpoints save in fabric reads
timeout 600 env GQ_OPTIONS=syrk_ck=0 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch0 -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-whole-model --no-side-legs > $OUT/pmc_fetch0.log 2>&1 || echo "pmc pass (syrk_ck=0) failed"
python3 - <<PY
import csv, glob
v = [float(r["Counter_Value"]) for f in glob.glob("$OUT/pmc_fetch0/**/*counter_collection.csv", recursive=True)
     for r in csv.DictReader(open(f)) if "syrk16" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
if v:
    open("$OUT/r06_syrk_traffic_ck0.txt", "w").write(f"GQ_OPTIONS=syrk_ck=0: {len(v)} SYRK launches, {sum(v) * 2048 / 1e9 / len(v):.2f} GB of L2-miss reads per launch\n")
    print(open("$OUT/r06_syrk_traffic_ck0.txt").read().strip())
PY
rm -rf $OUT/pmc_fetch0
cp $OUT/r06_syrk_traffic.json $R/profiles/r06_syrk_traffic.json  # bench.py reads it (and checks the source hash)
for W in $WLS; do
  EXTRA=""
  [ "$W" != "llama3-8b-block-q4k" ] && EXTRA="--no-whole-model"
  cd $R && timeout 900 python bench.py --workload $W --steps 5 --warmup 2 $EXTRA 2>$OUT/r06_bench_$W.err > $OUT/r06_bench_$W.out
  tail -2 $OUT/r06_bench_$W.out | head -1 > $OUT/r06_bench_$W.json          # the full dict
  tail -1 $OUT/r06_bench_$W.out > $OUT/r06_bench_${W}_compact.json            # the compact headline (the LAST stdout line)
  rm -f $OUT/r06_bench_$W.out
  (cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$W -o p -- \
     python $R/bench.py --workload $W --steps 5 --warmup 2 --no-cpu-baseline --no-whole-model --no-side-legs > $OUT/prof_$W.log 2>&1 || echo "rocprofv3 failed for $W")
  cp $(find $OUT/prof_$W -name "*kernel_stats.csv" | head -1) $OUT/r06_bench_${W}_kernel_stats.csv 2>/dev/null
  rm -rf $OUT/prof_$W
  python3 -c "
import json,sys
d=json.loads(open('$OUT/r06_bench_$W.json').read())
print('$W', d['ms_per_step'], 'ms/step', d['value'], 'Mparams/s  syrk frac', d['roofline']['frac'], ' trailing whole', (d.get('trailing_update') or {}).get('whole_alone',{}).get('frac'), ' cpu', (d.get('cpu_baseline') or {}).get('value'))
" 2>&1 | tail -1
done
