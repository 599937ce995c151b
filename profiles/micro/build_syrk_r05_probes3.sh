#!/bin/bash
# r05, third probe: "hotdma" -- the shipped loop whose DMA base never advances: every half-stage re-reads the first 32 token rows
# (L2-resident): the DMA instructions, their LDS writes and everything else stay, the memory side behind the L2 is taken out.
R=$(cd "$(dirname "$0")/../.." && pwd)
B=$R/profiles/micro/_build
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-function -Wno-unused-variable"
V=hotdma
S=$B/src/csrc/gq_hessian_$V.hip
cp $R/gptq-gguf-toolkit_amd/csrc/gq_hessian.hip $S
python3 - "$S" <<'PY'
import sys
p = sys.argv[1]
s = open(p).read()
a = s.index("void syrk16_256w_kernel(const SyrkGroup grp) {")
b = s.index("// K-split tiles: H tile = beta", a)
body = s[a:b]
old = "        if (hnext + 1 < nhs) {                                                                        \\"
assert body.count(old) == 1
body = body.replace(old, "        if (false && hnext + 1 < nhs) {                                                               \\")
s = s[:a] + body + s[b:]
open(p, "w").write(s)
PY
/opt/rocm/bin/hipcc $FLAGS -c $S -o $B/gq_hessian_$V.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $B/libgq_$V.so $B/gq_api.o $B/gq_codec.o $B/gq_scale_search.o $B/gq_gptq.o $B/gq_cholesky.o $B/gq_forward.o $B/gq_hessian_$V.o
ls -la $B/libgq_$V.so
