#!/bin/bash
# r06: the default bench step under ENVIRONMENT variants ("VAR=value" strings; "-" = none), alternating on ONE box
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for rep in 1 2; do
for o in "$@"; do
  oo=$o; [ "$o" = "-" ] && oo="GQ_NOOP=1"
  env $oo timeout 300 python bench.py --steps ${STEPS:-10} --warmup 3 --no-cpu-baseline --no-whole-model --no-side-legs ${WL:+--workload $WL} 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('rep $rep [%-28s] %.2f ms/step  latency %.2f  syrk frac %.3f' % ('$o', d['ms_per_step'], d['step_latency_ms'], d['roofline']['frac']))
"
done
done
