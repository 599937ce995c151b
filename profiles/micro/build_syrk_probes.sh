#!/bin/bash
# Timing-only probe builds of the SYRK main loop (results are WRONG in some of them): the shipped sources carry no probe
# switches, so the variants are made by editing a copy.  Builds profiles/micro/_build/libgq_<variant>.so (git-ignored; they
# travel to the GPU box); run with GQ_SO_PATH=... python profiles/syrk_probe.py
#   vm8    s_waitcnt vmcnt(8) instead of (4) before the step's barrier: never waits for the next half-stage's DMA (wrong data)
#   vm0    vmcnt(0): waits for every DMA in flight (correct, prefetch distance one step)
#   nobar  no s_barrier in the step (wrong data)
R=$(cd "$(dirname "$0")/../.." && pwd)
B=$R/profiles/micro/_build
mkdir -p $B/src/csrc $B/include
cp $R/gptq-gguf-toolkit_amd/csrc/*.hip $R/gptq-gguf-toolkit_amd/csrc/*.hpp $B/src/csrc/
cp $R/include/gptq_gguf.h $B/include/
sed -i 's#"../../include/gptq_gguf.h"#"../../include/gptq_gguf.h"#' $B/src/csrc/gq_common.hpp
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-function"
for f in gq_api gq_codec gq_scale_search gq_gptq gq_cholesky gq_forward; do
  [ -f $B/$f.o ] || /opt/rocm/bin/hipcc $FLAGS -c $B/src/csrc/$f.hip -o $B/$f.o &
done
wait
for V in vm8 vm0 nobar; do
  cp $R/gptq-gguf-toolkit_amd/csrc/gq_hessian.hip $B/src/csrc/gq_hessian_$V.hip
  case $V in
    vm8)   sed -i 's/#define GQ_NBAR() asm volatile("s_waitcnt vmcnt(4)\\n\\ts_barrier"/#define GQ_NBAR() asm volatile("s_waitcnt vmcnt(8)\\n\\ts_barrier"/' $B/src/csrc/gq_hessian_$V.hip ;;
    vm0)   sed -i 's/#define GQ_NBAR() asm volatile("s_waitcnt vmcnt(4)\\n\\ts_barrier"/#define GQ_NBAR() asm volatile("s_waitcnt vmcnt(0)\\n\\ts_barrier"/' $B/src/csrc/gq_hessian_$V.hip ;;
    nobar) sed -i 's/#define GQ_NBAR() asm volatile("s_waitcnt vmcnt(4)\\n\\ts_barrier"/#define GQ_NBAR() asm volatile("s_waitcnt vmcnt(4)"/' $B/src/csrc/gq_hessian_$V.hip ;;
  esac
  grep -c "define GQ_NBAR" $B/src/csrc/gq_hessian_$V.hip > /dev/null
  /opt/rocm/bin/hipcc $FLAGS -c $B/src/csrc/gq_hessian_$V.hip -o $B/gq_hessian_$V.o &
done
wait
for V in vm8 vm0 nobar; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $B/libgq_$V.so $B/gq_api.o $B/gq_codec.o $B/gq_scale_search.o $B/gq_gptq.o $B/gq_cholesky.o $B/gq_forward.o $B/gq_hessian_$V.o
  grep "define GQ_NBAR" $B/src/csrc/gq_hessian_$V.hip
done
ls -la $B/*.so
