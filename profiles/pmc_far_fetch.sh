#!/bin/bash
# L2-miss read traffic (FETCH_SIZE x 2 KB on gfx950... per the guide: FETCH_SIZE counts 32-byte? see MI355X_MICROARCH.md) of the far launches
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
d=$R/gpurun_out/r3/pmc_far_fetch; mkdir -p $d
timeout 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $d/p -o p -- python $R/profiles/near_probe.py > $d/p.log 2>&1 || echo "pass failed"
python3 - $d <<'PY'
import csv, glob, collections, sys
d = sys.argv[1]
rows = collections.defaultdict(dict)
for f in glob.glob(d + "/p/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "chain_full_kernel" in r["Kernel_Name"]:
            rows[(int(r["Dispatch_Id"]), int(r["Grid_Size"]))][r["Counter_Name"]] = float(r["Counter_Value"])
for (disp, grid), c in sorted(rows.items())[-14:]:
    ntx = grid // 512 // 32  # N / 128 (M = 4096: 32 tile rows)
    N = ntx * 128
    alg = (4096 * 1024 + 1024 * N + 4096 * N) * 4 / 1e9  # A + B + C read once
    print(f"dispatch {disp} N={N:6d}: FETCH {c.get('FETCH_SIZE', 0) * 2048 / 1e9:7.3f} GB  WRITE {c.get('WRITE_SIZE', 0) * 2048 / 1e9:7.3f} GB   algorithmic reads {alg:6.3f} GB")
PY
rm -rf $d/p
