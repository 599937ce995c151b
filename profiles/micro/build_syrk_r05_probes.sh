#!/bin/bash
# r05: where does the four-wave SYRK's structural ceiling come from?  Timing-only probe builds of
# syrk16_256w_kernel's main loop that REMOVE one of its three streams each (results are wrong in all of them); the shipped
# sources carry no probe switches, so the variants are made by editing a copy.  Builds profiles/micro/_build/libgq_<v>.so
# (git-ignored; they travel to the GPU box); run: GQ_SO_PATH=... DATA=zeros|random python profiles/syrk_probe.py
#   nomfma       no v_mfma: operand delivery (LDS-DMA ring + barrier) and the fragment reads only
#   nomfma_nord  no v_mfma, no ds_read: the LDS-DMA ring and its barrier alone = what the L2 -> LDS path delivers
#   nodma        MFMAs + fragment reads on a ring that is never refilled (prologue only): the compute side alone
#   nord         MFMAs + DMA, no fragment reads (stale registers): without the LDS read traffic
#   hotdma       the shipped loop whose DMA base never advances: every half-stage re-reads the first 32 token rows (L2-resident):
#                the DMA instructions, their LDS writes and everything else stay, the memory side behind the L2 is taken out
#   noepi        no epilogue (H neither read nor written)        norv   persistent rounds without the round-start XCD rendezvous
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
VARS="nomfma nomfma_nord nodma nord hotdma noepi norv"
for V in $VARS; do
  S=$B/src/csrc/gq_hessian_$V.hip
  cp $R/gptq-gguf-toolkit_amd/csrc/gq_hessian.hip $S
  python3 - "$S" "$V" <<'PY'
import re, sys
p, v = sys.argv[1], sys.argv[2]
s = open(p).read()
a = s.index("struct W4 {")
b = s.index("#undef GQ_WDL", a)
body = s[a:b]
if v in ("nomfma", "nomfma_nord"):
    n = body.count('asm volatile("v_mfma_f32_16x16x32')
    assert n == 2, n
    body = re.sub(r'if constexpr \(BF16\) asm volatile\("v_mfma_f32_16x16x32_bf16[^;]*;\n\s*else asm volatile\("v_mfma_f32_16x16x32_f16[^;]*;',
                  'asm volatile("" : "+a"(acc) : "v"(A_), "v"(B_));', body)
    assert "v_mfma" not in body
if v in ("nomfma_nord", "nord"):
    # the step's reads only (first_fragments keeps its own): read<SC, CODE>() becomes empty
    body = body.replace("if constexpr (CODE >= 0) {", "if constexpr (false && CODE >= 0) {", 1)
if v == "nodma":
    body = body.replace("if constexpr (S.dma[M] >= 0) {  // a DMA piece of half-stage n+3", "if constexpr (false && S.dma[M] >= 0) {", 1)
s = s[:a] + body + s[b:]
ka = s.index("void syrk16_256w_kernel(const SyrkGroup grp) {")
kb = s.index("// K-split tiles: H tile = beta", ka)
kern = s[ka:kb]
if v == "hotdma":
    old = "        if (hnext + 1 < nhs) {                                                                        \\"
    assert kern.count(old) == 1
    kern = kern.replace(old, "        if (false && hnext + 1 < nhs) {                                                               \\")
if v == "noepi":
    old = "    for (int i = 0; i < 8; ++i) {\n#pragma unroll\n        for (int j = 0; j < 8; ++j) {\n            const f32x4 v = w.c[i][j];"
    assert kern.count(old) == 1
    kern = kern.replace(old, old + "\n            if (v[0] != 12345.678f) continue;")
if v == "norv":
    old = "if (grp.bar && round > 0) {"
    assert kern.count(old) == 1
    kern = kern.replace(old, "if (false && grp.bar && round > 0) {")
s = s[:ka] + kern + s[kb:]
open(p, "w").write(s)
PY
  /opt/rocm/bin/hipcc $FLAGS -c $S -o $B/gq_hessian_$V.o &
done
wait
for V in $VARS; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $B/libgq_$V.so $B/gq_api.o $B/gq_codec.o $B/gq_scale_search.o $B/gq_gptq.o $B/gq_cholesky.o $B/gq_forward.o $B/gq_hessian_$V.o
done
ls -la $B/*.so
