#!/bin/bash
# rocprofv3 kernel trace of profiles/chain_probe.py: per-launch duration and TFLOP/s of the chained far-update GEMM
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/trace_far
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_far/p -o p -- python $R/profiles/chain_probe.py > $R/gpurun_out/trace_far/p.log 2>&1 || echo "pass failed"
tail -12 $R/gpurun_out/trace_far/p.log
python3 - <<PY
import csv, glob
rows = []
for f in glob.glob("$R/gpurun_out/trace_far/p/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm32_kernel<false, 0, false, 0, 128" in r["Kernel_Name"] or "gemm32_chain_full_kernel" in r["Kernel_Name"]:
            rows.append(r)
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
print(len(rows), "far-update launches;", list(rows[0].keys()) if rows else "")
tot_f = tot_t = 0.0
for r in rows[-13:]:
    gx, gy = int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]) // int(r["Workgroup_Size_Y"])
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    fl = 2.0 * gx * 128 * gy * 128 * 1024
    tot_f += fl; tot_t += dur
    print(f"tiles {gx:4d} x {gy:3d}  {dur:8.1f} us  {fl / dur / 1e6:6.1f} TFLOP/s")
print(f"total {tot_t:.0f} us, {tot_f / tot_t / 1e6:.1f} TFLOP/s")
PY
