#!/bin/bash
# r05, fourth probe: "ckN" -- a soft XCD rendezvous INSIDE the tile every N half-stages (the 32 workgroups of an XCD share 12 operand
# panels out of one 4 MB L2; between the round-start rendezvous they drift).  Run with GQ_OPTIONS=syrk_nosplit=1 (all units walk the
# same number of half-stages).  Timing probe; results stay correct.
R=$(cd "$(dirname "$0")/../.." && pwd)
B=$R/profiles/micro/_build
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-function -Wno-unused-variable"
for N in 64 256; do
V=ck$N
S=$B/src/csrc/gq_hessian_$V.hip
cp $R/gptq-gguf-toolkit_amd/csrc/gq_hessian.hip $S
python3 - "$S" "$N" <<'PY'
import sys
p, N = sys.argv[1], int(sys.argv[2])
s = open(p).read()
a = s.index("void syrk16_256w_kernel(const SyrkGroup grp) {")
b = s.index("// K-split tiles: H tile = beta", a)
body = s[a:b]
old = "    for (int n = 0; n < nhs; n += 4) {  // nhs % 4 == 0\n"
assert body.count(old) == 1
new = old + f'''        if (grp.bar && n && (n & {N - 1}) == 0) {{
            if (tid == 0) {{
                unsigned* b2 = grp.bar + 8 + (blockIdx.x & 7);
                __hip_atomic_fetch_add(b2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned want = ((unsigned)round * (unsigned)(nhs / {N} - 1) + (unsigned)(n / {N})) * (gridDim.x >> 3);
                for (int spin = 0; spin < 32 && __hip_atomic_load(b2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want; ++spin)
                    __builtin_amdgcn_s_sleep(4);
            }}
        }}
'''
body = body.replace(old, new)
s = s[:a] + body + s[b:]
# host: sixteen counters instead of eight
old = "for (int x = 0; x < 8; ++x) table.push_back(0u);"
assert s.count(old) == 1
s = s.replace(old, "for (int x = 0; x < 16; ++x) table.push_back(0u);")
open(p, "w").write(s)
PY
/opt/rocm/bin/hipcc $FLAGS -c $S -o $B/gq_hessian_$V.o &
done
wait
for N in 64 256; do V=ck$N
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $B/libgq_$V.so $B/gq_api.o $B/gq_codec.o $B/gq_scale_search.o $B/gq_gptq.o $B/gq_cholesky.o $B/gq_forward.o $B/gq_hessian_$V.o
done
ls -la $B/libgq_ck*.so
