#!/bin/bash
# A/B of the two forms of the in-place SYRK on the bench shapes, alone on the GPU (profiles/syrk_probe.py):
#   default  syrk16_256n_kernel: 8 waves (2 x 4), wave tile 128 x 64
#   w4       syrk16_256w_kernel: 4 waves (2 x 2), wave tile 128 x 128 (option syrk_w4 = 1), bit-identical results
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for rep in 1 2; do
  for v in "" "syrk_w4=1"; do
    echo "== GQ_OPTIONS='$v' CS=14336"
    GQ_OPTIONS="$v" CS=14336 ITERS=8 python profiles/syrk_probe.py | tail -3
    echo "== GQ_OPTIONS='$v' CS=4096,4096,4096"
    GQ_OPTIONS="$v" CS=4096,4096,4096 ITERS=8 python profiles/syrk_probe.py | tail -3
  done
done
