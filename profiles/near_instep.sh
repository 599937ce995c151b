R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
  d=$R/gpurun_out/r3/near_instep_$v; mkdir -p $d
  if [ $v = 1 ]; then export GQ_FAR_SYNC=1; fi
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $d/p -o p -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-whole-model --no-side-legs > $d/log.txt 2>&1
  echo "FAR_SYNC=$v: $(grep -o '"ms_per_step": [0-9.]*' $d/log.txt | tail -1)"
  python3 - $(find $d/p -name "*kernel_stats.csv" | head -1) <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if any(k in r["Name"] for k in ("near256", "chain_full", "segment", "diag_blk5")):
        print(f"   {r['Name'].split('(')[0][-48:]:48s} calls {r['Calls']:>6s} avg {float(r['AverageNs']) / 1e3:8.1f} us total {int(r['TotalDurationNs']) / 1e6:8.1f} ms")
PY
  rm -rf $d/p
done
