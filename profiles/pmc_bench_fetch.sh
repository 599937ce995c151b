#!/bin/bash
# roofline.traffic: L2-miss read bytes of the SYRK launches of the bench command itself (rocprofv3 --pmc FETCH_SIZE,
# its own run with --kernel-trace only).  FETCH_SIZE is in KiB-like units of 1 KB and under-reports by 2x on gfx950
# (MI355X_MICROARCH.md): bytes = FETCH_SIZE * 1024 * 2.   usage (GPU box): bash profiles/pmc_bench_fetch.sh [bench args]
# Writes gpurun_out/r02_syrk_traffic.json; copy it to profiles/r02_syrk_traffic.json (bench.py reads it there).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/pmc_fetch
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fetch/p -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-whole-model --no-side-legs "$@" > $R/gpurun_out/pmc_fetch/p.log 2>&1 || echo "pass failed"
python3 - <<PY
import csv, glob, hashlib, json
rows = []
for f in glob.glob("$R/gpurun_out/pmc_fetch/p/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "syrk16" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
            rows.append((int(r["Dispatch_Id"]), int(r["Grid_Size"]), float(r["Counter_Value"])))
rows.sort()
for d, g, v in rows:
    print(f"dispatch {d:6d} grid {g:8d} threads: FETCH_SIZE {v:.4g} -> {v * 1024 * 2 / 1e9:7.2f} GB")
if rows:
    per = sum(v for _, _, v in rows) * 2048 / 1e9 / len(rows)
    print(f"{len(rows)} launches, average {per:.2f} GB per launch")
    # algorithmic bytes of one launch of the default workload: the activations of 32 sequences (65536 tokens) of the
    # launch's inputs read once + H read and written once, averaged over the 8 launches of a step
    T, n4, n14 = 65536, 3, 1
    alg = (4 * (n4 * (T * 4096 * 2 + 2 * 4096 * 4096 * 4) ) + 4 * (T * 14336 * 2 + 2 * 14336 * 14336 * 4)) / 8 / 1e9
    sha = hashlib.sha256(open("$R/bench.py", "rb").read()).hexdigest()[:16]
    json.dump({"GB_per_launch": round(per, 2), "algorithmic_GB_per_launch": round(alg, 2), "launches": len(rows),
               "measured_on": f"rocprofv3 --pmc FETCH_SIZE x 2 KB (gfx950 correction), bench.py sha256 {sha}, "
                              "default workload, 3 steps"},
              open("$R/gpurun_out/r02_syrk_traffic.json", "w"))
PY
