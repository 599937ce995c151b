#!/usr/bin/env python3
"""Per-task stamps of ONE resident sub-problem launch (probe build: profiles/micro/build_csub_stamps.sh).
usage: GQ_SO_PATH=profiles/micro/_build/stamps/libgptqgguf_hip.so [C=1792] [G=48] python profiles/r06_sub_stamps.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
C = int(os.environ.get("C", 1792)); G = int(os.environ.get("G", 48))
ts = torch.zeros(5 * 4096, dtype=torch.int64, device="cuda")
os.environ["GQ_CSUB_TS"] = hex(ts.data_ptr())
from gptq_gguf_toolkit_amd import ops, _cabi
torch.manual_seed(0)
X = (torch.randn(2 * C, C, device="cuda") * torch.exp(torch.randn(C, device="cuda") * 0.5)).half()
H0 = torch.zeros(C, C, device="cuda"); ops.h_accumulate(H0, X, 0.0, 0.5); del X
W0 = torch.randn(128, C, device="cuda")
L = _cabi.lib()
with ops.options(chol_sub=int(os.environ.get("SUB", 16)), chol_sub_wgs=G):
    nws = ops.workspace_bytes(_cabi.WS_H_PREPARE, 128, C)
    ws = torch.zeros(nws, dtype=torch.uint8, device="cuda")
    for it in range(3):
        ts.zero_()
        H = H0.clone(); W = W0.clone(); U = torch.empty_like(H); flag = torch.zeros(1, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize(); t0 = time.perf_counter()
        rc = L.gq_h_prepare(H.data_ptr(), W.data_ptr(), 128, C, 0.01, U.data_ptr(), flag.data_ptr(), None, ws.data_ptr(), ws.numel(),
                            torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    sub_ws = 64 * 1024 + (C // 128 + 1) * 128 * 4 + 1024
    off = ((ws.data_ptr() + nws - sub_ws + 255) & ~255) - ws.data_ptr()
    words = ws[off:off + 8192 * 4].view(torch.int32).cpu().tolist()
nph, ntask = words[0], words[1]
t = ts.cpu().view(-1, 5)[:ntask].tolist()
tasks = words[4 + 8 * nph: 4 + 8 * nph + ntask]
T0 = min(r[0] for r in t)
us = lambda x: (x - T0) / 100.0
names = "LEAF G1 G2 G3 G4".split()
print(f"C={C} G={G} whole call {dt*1e3:.2f} ms; kernel span {us(max(r[3] for r in t)):.1f} us, {nph} phases, {ntask} tasks")
agg = {}
for i, r in enumerate(t):
    p = (tasks[i] >> 16) & 0xffff
    ty = names[words[4 + 8 * p] & 15]
    a = agg.setdefault(ty, [0, 0.0, 0.0, 0.0])
    a[0] += 1; a[1] += us(r[1]) - us(r[0]); a[2] += us(r[2]) - us(r[1]); a[3] += us(r[3]) - us(r[2])
for ty, a in agg.items():
    print(f"  {ty:5s} n={a[0]:4d}  wait {a[1]/a[0]:7.2f} us  work {a[2]/a[0]:7.2f} us  release {a[3]/a[0]:6.2f} us   (avg per task)")
# the critical chain: leaf to leaf
prev = None
print("  leaf#  start   work   release | gap to next leaf start")
leaves = [(i, r) for i, r in enumerate(t) if names[words[4 + 8 * ((tasks[i] >> 16) & 0xffff)] & 15] == "LEAF"]
for k, (i, r) in enumerate(leaves):
    nxt = us(leaves[k + 1][1][1]) if k + 1 < len(leaves) else float("nan")
    print(f"  {k:4d} {us(r[1]):8.1f} {us(r[2]) - us(r[1]):6.1f} {us(r[3]) - us(r[2]):6.1f} | {nxt - us(r[3]):6.1f}")
if os.environ.get("DUMP"):
    for i, r in enumerate(t):
        p = (tasks[i] >> 16) & 0xffff
        print(i, names[words[4 + 8 * p] & 15], p, tasks[i] & 0xffff, "wg", r[4], " ".join(f"{us(x):8.1f}" for x in r[:4]))
