#!/bin/bash
# The judged artifacts of a round: the bench line and the rocprofv3 --kernel-trace --stats summary of the same
# command.  usage (GPU box): bash profiles/collect_bench_profile.sh r02   -> gpurun_out/<tag>_bench_line.json,
# gpurun_out/<tag>_bench_kernel_stats.csv (copy both into profiles/).
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-r02}
cd $R && python bench.py --steps 5 --warmup 2 2>$R/gpurun_out/${TAG}_bench.err | tail -1 > $R/gpurun_out/${TAG}_bench_line.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_prof -o p -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-whole-model --no-side-legs > $R/gpurun_out/${TAG}_prof.log 2>&1 || echo "rocprofv3 failed"
cp $(find $R/gpurun_out/${TAG}_prof -name "*kernel_stats.csv" | head -1) $R/gpurun_out/${TAG}_bench_kernel_stats.csv
tail -1 $R/gpurun_out/${TAG}_prof.log | cut -c1-300
head -12 $R/gpurun_out/${TAG}_bench_kernel_stats.csv | cut -c1-160
