#!/bin/bash
# far-update kernel A/B: durations of the large gemm32_chain_full launches with pieces of the loop removed (probe
# libraries built with -DGQ_FAR_NO*; results wrong, timing only)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for v in ${VARIANTS:-"" NOCOMMIT NOFETCH NOBARRIER NOCOMMITNOFETCHNOBARRIER MFMAONLY}; do
  d=$R/gpurun_out/r3/far_ab_$v; mkdir -p $d
  so=$R/gptq-gguf-toolkit_amd/csrc/libgptqgguf_hip.so; [ -n "$v" ] && so=$R/profiles/libgq_far_$v.so
  GQ_SO_PATH=$so timeout 300 rocprofv3 --kernel-trace --output-format csv -d $d/p -o p -- python $R/profiles/near_probe.py > $d/log.txt 2>&1 || echo "pass failed"
  python3 - $(find $d/p -name '*kernel_trace.csv' | head -1) "$v" <<'PY'
import csv, sys
t = n = 0
for r in csv.DictReader(open(sys.argv[1])):
    if "chain_full" in r["Kernel_Name"] and int(r.get("Grid_Size_X", r.get("Grid_Size", 0))) >= 4096:
        t += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); n += 1
print(f"{sys.argv[2] or 'as built':28s} far launches {n}  total {t / 1e6:.3f} ms")
PY
  rm -rf $d
done
