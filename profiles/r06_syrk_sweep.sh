#!/bin/bash
# r06: the grouped SYRK of the bench's four-input fold (3 x 4096 + 14336, 65 536 tokens, random fp16) under option pairs; best of 5
R=${GRAFT_REPO_ROOT:-/root/repo}
for gw in 4 8 2 16; do for ck in 256 0 128 512; do
  t=$(GQ_OPTIONS="syrk_gw=$gw,syrk_ck=$ck" NSEQ=32 ITERS=6 timeout 200 python $R/profiles/syrk_probe.py | tail -5 | awk '{print $5}' | sort -n | tail -1)
  echo "syrk_gw=$gw syrk_ck=$ck: $t TFLOP/s"
done; done
