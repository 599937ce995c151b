"""GPU busy / idle of a whole-model run from a rocprofv3 --kernel-trace CSV: union of kernel intervals over all
queues, per-category kernel time.   usage: python profiles/trace_model.py <kernel_trace.csv>"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
t0, t1 = iv[0][0], max(e for _, e, _ in iv)
busy, end = 0, t0
gaps = []
for s, e, _ in iv:
    if s > end:
        gaps.append((s - end, end - t0))
    if e > end:
        busy += e - max(s, end)
        end = e
print(f"span {(t1 - t0) / 1e6:.1f} ms, busy (union) {busy / 1e6:.1f} ms, idle {(t1 - t0 - busy) / 1e6:.1f} ms, {len(iv)} kernels")
gaps.sort(reverse=True)
print("largest gaps (ms @ ms):", [(round(g / 1e6, 2), round(a / 1e6, 1)) for g, a in gaps[:12]])
print("gaps > 20 us:", sum(1 for g, _ in gaps if g > 20000), "total", round(sum(g for g, _ in gaps if g > 20000) / 1e6, 1), "ms;",
      "gaps <= 20 us total", round(sum(g for g, _ in gaps if g <= 20000) / 1e6, 1), "ms")
cat = collections.defaultdict(lambda: [0, 0])
for s, e, n in iv:
    if "Cijk" in n: k = "hipblaslt gemm"
    elif "gq::" in n: k = "gq:" + n.split("gq::")[1].split("(")[0].split("<")[0]
    elif "attn_fwd" in n: k = "attn_fwd"
    elif "copyBuffer" in n or "fillBuffer" in n: k = "rocclr copy/fill"
    elif "at::native" in n: k = "torch elementwise/other"
    else: k = n[:40]
    cat[k][0] += e - s
    cat[k][1] += 1
for k, (t, c) in sorted(cat.items(), key=lambda kv: -kv[1][0])[:14]:
    print(f"  {k:36s} {t / 1e6:9.2f} ms {c:7d}")
