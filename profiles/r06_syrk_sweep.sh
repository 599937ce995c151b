#!/bin/bash
# r06: the grouped SYRK of the bench's four-input fold (3 x 4096 + 14336, 65 536 tokens, random fp16) under option sets; best of 5
# usage: bash profiles/r06_syrk_sweep.sh "syrk_gw=4,syrk_ck=256" "syrk_gw=8" ...      (default: the gw x ck grid of r06)
R=${GRAFT_REPO_ROOT:-/root/repo}
if [ $# -eq 0 ]; then
  set --
  for gw in 4 8 2 16; do for ck in 256 0 128 512; do set -- "$@" "syrk_gw=$gw,syrk_ck=$ck"; done; done
fi
for rep in 1 2; do
for o in "$@"; do
  t=$(GQ_OPTIONS="$o" NSEQ=32 ITERS=6 timeout 200 python $R/profiles/syrk_probe.py | tail -5 | awk '{print $5}' | sort -n | tail -1)
  echo "rep $rep [$o]: $t TFLOP/s"
done
done
