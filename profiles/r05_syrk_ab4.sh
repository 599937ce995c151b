#!/bin/bash
cd "$(dirname "$0")/.."
B=$PWD/profiles/micro/_build
for rep in 1 2; do
 for data in zeros random; do
  for v in shipped ck256; do
   so=""; [ $v != shipped ] && so=$B/libgq_$v.so
   a=$(GQ_SO_PATH=$so DATA=$data CS=14336 ITERS=8 python profiles/syrk_probe.py 2>/dev/null | tail -3 | awk '{printf "%s ", $5}')
   b=$(GQ_SO_PATH=$so DATA=$data CS=4096,4096,4096 ITERS=8 python profiles/syrk_probe.py 2>/dev/null | tail -3 | awk '{printf "%s ", $5}')
   c=$(GQ_SO_PATH=$so DATA=$data CS=4096,4096,4096,14336 ITERS=8 python profiles/syrk_probe.py 2>/dev/null | tail -3 | awk '{printf "%s ", $5}')
   echo "[$data] [$v] 14336: $a | 3x4096: $b | four: $c"
  done
 done
done
