"""Coarse GPU timeline of a whole-model run from a rocprofv3 kernel trace: runs of kernels of one class
(forward = hipBLASLt / attention / torch / gq::fwd_*, hessian = syrk + staging copies, chain = every other gq:: kernel)
with their span, busy time (union) and the idle gap before them.   usage: python profiles/trace_phases.py <kernel_trace.csv>"""
import csv
import sys

rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))))


def cls(n):
    if "gq::fwd_" in n or "Cijk" in n or "attn_fwd" in n or "at::native" in n:
        return "forward"
    if "syrk" in n or "transpose16" in n or "copyBuffer" in n:
        return "hessian"
    if "gq::" in n:
        return "chain"
    return "other"


# merge: a phase = maximal run where 'chain' kernels are absent (forward/hessian) or present
phases = []
cur = None
for s, e, n in rows:
    c = "chain" if cls(n) == "chain" else "fwd+H"
    if cur is None or (c != cur[0] and s - cur[2] > 0 and (c == "chain" or s > cur[4])):
        cur = [c, s, e, 0, e, 0]  # class, start, end, busy, last_chain_or_fwd_end, n
        phases.append(cur)
    cur[2] = max(cur[2], e)
    cur[5] += 1
t0 = rows[0][0]
# coalesce tiny phases (< 200 kernels) into neighbours for readability
out = []
for p in phases:
    if out and (p[5] < 200 or out[-1][0] == p[0]):
        out[-1][2] = max(out[-1][2], p[2]); out[-1][5] += p[5]
    else:
        out.append(p)
prev_end = t0
for c, s, e, _, _, n in out:
    # busy = union of kernel intervals inside [s, e]
    busy, end = 0, s
    for ks, ke, _n in rows:
        if ke <= s or ks >= e:
            continue
        if ke > end:
            busy += ke - max(ks, end); end = ke
    print(f"{c:6s} start {(s - t0) / 1e6:9.1f} ms  span {(e - s) / 1e6:8.1f} ms  busy {busy / 1e6:8.1f} ms  gap before {(s - prev_end) / 1e6:7.1f} ms  kernels {n}")
    prev_end = e
