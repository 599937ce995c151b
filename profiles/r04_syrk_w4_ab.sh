#!/bin/bash
# A/B of the forms of the in-place SYRK on the bench shapes, alone on the GPU (profiles/syrk_probe.py):
#   default    syrk16_256n_kernel: 8 waves (2 x 4), wave tile 128 x 64
#   syrk_w4=V  syrk16_256w_kernel<V>: 4 waves (2 x 2), wave tile 128 x 128, bit-identical results (V: stream variants)
# usage: bash profiles/r04_syrk_w4_ab.sh "" syrk_w4=1 syrk_w4=2 ...
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
[ $# -eq 0 ] && set -- "" "syrk_w4=1"
for rep in 1 2; do
  for v in "$@"; do
    a=$(GQ_OPTIONS="$v" CS=14336 ITERS=8 python profiles/syrk_probe.py 2>/dev/null | tail -3 | awk '{printf "%s ", $5}')
    b=$(GQ_OPTIONS="$v" CS=4096,4096,4096 ITERS=8 python profiles/syrk_probe.py 2>/dev/null | tail -3 | awk '{printf "%s ", $5}')
    echo "[${v:-default}] C=14336: $a TFLOP/s | 3 x C=4096: $b TFLOP/s"
  done
done
