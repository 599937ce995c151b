#!/bin/bash
# A/B builds of the column-loop kernel for timing probes: which part of gptq_segment_kernel's time is the dependent
# chain of wave 0, which the rank-1 tile updates.  usage (build container): bash profiles/seg_ab.sh ; then on the GPU box
#   GQ_SO_PATH=profiles/libgq_nochain.so python profiles/chain_probe.py   (results are wrong by construction)
R=$(cd "$(dirname "$0")/.." && pwd); C=$R/gptq-gguf-toolkit_amd/csrc
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-function"
for v in NOCHAIN NOUPDATE; do
  /opt/rocm/bin/hipcc $FL -DGQ_SEG_$v -c $C/gq_gptq.hip -o /tmp/gq_gptq_$v.o &&
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/profiles/libgq_$(echo $v | tr A-Z a-z).so /tmp/gq_gptq_$v.o \
      $C/gq_api.o $C/gq_codec.o $C/gq_scale_search.o $C/gq_hessian.o $C/gq_cholesky.o
done
