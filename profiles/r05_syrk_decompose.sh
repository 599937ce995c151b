#!/bin/bash
# the four-wave SYRK with one of its three streams removed (profiles/micro/build_syrk_r05_probes.sh), on
# all-zero and on random operands, alone on the GPU, C = 14336, 65 536 tokens.  TFLOP/s are "algorithmic flops / time" of the
# launch whatever the variant computes: for the variants without MFMAs the number says how fast the REST of the loop would let
# the matrix pipe run.  usage (GPU box): bash profiles/r05_syrk_decompose.sh > gpurun_out/r05_syrk_decompose.txt
cd "$(dirname "$0")/.."
B=profiles/micro/_build
# Both with the rendezvous inside a tile the round added (default, syrk_ck = 256) and without it (syrk_ck = 0: the r04 kernel).
for ck in 0 256; do
  for data in zeros random; do
    for v in shipped nomfma_nord nomfma nord nodma hotdma noepi norv; do
      so=""; [ $v != shipped ] && so=$PWD/$B/libgq_$v.so
      a=$(GQ_OPTIONS=syrk_ck=$ck GQ_SO_PATH=$so DATA=$data CS=14336 ITERS=8 python profiles/syrk_probe.py 2>/dev/null | tail -3 | awk '{printf "%s ", $5}')
      echo "[syrk_ck=$ck] [$data] [$v] C=14336: $a TFLOP/s"
    done
  done
done
