#!/bin/bash
# r06: a timing-probe build of the resident Cholesky sub-problem kernel (csrc/gq_cholsub.hpp): every task records
# {claimed, dependencies met, work done, released} in 100 MHz wall-clock ticks + its workgroup into a buffer whose device
# address comes in through GQ_CSUB_TS.  Built from a patched COPY of csrc/ (the shipped sources carry no probe code) into
# profiles/micro/_build/stamps/libgptqgguf_hip.so; use with GQ_SO_PATH (profiles/r06_sub_stamps.py).
set -e
R=$(cd $(dirname $0)/../.. && pwd)
T=/tmp/csub_stamps; rm -rf $T; mkdir -p $T/gptq-gguf-toolkit_amd $T/include
cp -r $R/gptq-gguf-toolkit_amd/csrc $T/gptq-gguf-toolkit_amd/; cp $R/include/gptq_gguf.h $T/include/
cd $T/gptq-gguf-toolkit_amd/csrc
python3 - <<'PY'
p='gq_cholsub.hpp'; s=open(p).read()
def rep(a, b):
    global s
    assert a in s, a
    s = s.replace(a, b, 1)
rep("uint32_t* __restrict__ cnt) {", "uint32_t* __restrict__ cnt, unsigned long long* ts) {")
rep("        // [probe: claimed]\n", "        if (ts && tid == 0 && t < ntask) { ts[5 * t + 0] = wall_clock64(); ts[5 * t + 4] = blockIdx.x; }\n")
rep("        // [probe: dependencies met]\n", "        if (ts && tid == 0) ts[5 * t + 1] = wall_clock64();\n")
rep("        // [probe: work done]\n", "        if (ts && tid == 0) ts[5 * t + 2] = wall_clock64();\n")
rep("        // [probe: released]\n", "        if (ts && tid == 0) ts[5 * t + 3] = wall_clock64();\n")
open(p,'w').write(s)
p='gq_cholesky.hip'; s=open(p).read()
rep("A + o, X + o, Tmp + o, n, flag, plan, cnt);", 'A + o, X + o, Tmp + o, n, flag, plan, cnt, getenv("GQ_CSUB_TS") ? (unsigned long long*)strtoull(getenv("GQ_CSUB_TS"), nullptr, 16) : nullptr);')
open(p,'w').write(s)
PY
rm -f gq_cholesky.o libgptqgguf_hip.so
make -s -j8
mkdir -p $R/profiles/micro/_build/stamps && cp libgptqgguf_hip.so $R/profiles/micro/_build/stamps/
echo built $R/profiles/micro/_build/stamps/libgptqgguf_hip.so
