#!/bin/bash
# A/B of option syrk_ck (soft XCD rendezvous inside a tile every N half-stages; 0: none), alone on the GPU, profiles/syrk_probe.py
cd "$(dirname "$0")/.."
for rep in 1 2; do
 for data in zeros random; do
  for ck in 0 64 128 256 512 1024; do
   a=$(GQ_OPTIONS=syrk_ck=$ck DATA=$data CS=14336 ITERS=8 python profiles/syrk_probe.py 2>/dev/null | tail -3 | awk '{printf "%s ", $5}')
   b=$(GQ_OPTIONS=syrk_ck=$ck DATA=$data CS=4096,4096,4096 ITERS=8 python profiles/syrk_probe.py 2>/dev/null | tail -3 | awk '{printf "%s ", $5}')
   c=$(GQ_OPTIONS=syrk_ck=$ck DATA=$data CS=4096,4096,4096,14336 ITERS=8 python profiles/syrk_probe.py 2>/dev/null | tail -3 | awk '{printf "%s ", $5}')
   echo "[$data] [syrk_ck=$ck] 14336: $a | 3x4096: $b | four: $c"
  done
 done
done
