#!/bin/bash
# r05, second set of timing-only probe builds of syrk16_256w_kernel (see build_syrk_r05_probes.sh): what AROUND the main loop costs.
#   noepi   no epilogue (H is neither read nor written)          norv   persistent rounds without the XCD rendezvous
#   noepi_norv  both
R=$(cd "$(dirname "$0")/../.." && pwd)
B=$R/profiles/micro/_build
mkdir -p $B/src/csrc $B/include
cp $R/gptq-gguf-toolkit_amd/csrc/*.hip $R/gptq-gguf-toolkit_amd/csrc/*.hpp $B/src/csrc/
cp $R/include/gptq_gguf.h $B/include/
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-function -Wno-unused-variable"
for f in gq_api gq_codec gq_scale_search gq_gptq gq_cholesky gq_forward; do
  /opt/rocm/bin/hipcc $FLAGS -c $B/src/csrc/$f.hip -o $B/$f.o &
done
wait
VARS="noepi norv noepi_norv"
for V in $VARS; do
  S=$B/src/csrc/gq_hessian_$V.hip
  cp $R/gptq-gguf-toolkit_amd/csrc/gq_hessian.hip $S
  python3 - "$S" "$V" <<'PY'
import sys
p, v = sys.argv[1], sys.argv[2]
s = open(p).read()
a = s.index("void syrk16_256w_kernel(const SyrkGroup grp) {")
b = s.index("// K-split tiles: H tile = beta", a)
body = s[a:b]
if "noepi" in v:
    old = "    for (int i = 0; i < 8; ++i) {\n#pragma unroll\n        for (int j = 0; j < 8; ++j) {\n            const f32x4 v = w.c[i][j];"
    assert body.count(old) == 1
    body = body.replace(old, "    for (int i = 0; i < 8; ++i) {\n#pragma unroll\n        for (int j = 0; j < 8; ++j) {\n            const f32x4 v = w.c[i][j];\n            if (v[0] != 12345.678f) continue;")
if "norv" in v:
    old = "if (grp.bar && round > 0) {"
    assert body.count(old) == 1
    body = body.replace(old, "if (false && grp.bar && round > 0) {")
s = s[:a] + body + s[b:]
open(p, "w").write(s)
PY
  /opt/rocm/bin/hipcc $FLAGS -c $S -o $B/gq_hessian_$V.o &
done
wait
for V in $VARS; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $B/libgq_$V.so $B/gq_api.o $B/gq_codec.o $B/gq_scale_search.o $B/gq_gptq.o $B/gq_cholesky.o $B/gq_forward.o $B/gq_hessian_$V.o
done
ls $B/libgq_noepi*.so $B/libgq_norv.so
