"""Per-queue timeline of the LAST bench step from a rocprofv3 --kernel-trace CSV.
usage: python profiles/trace_step.py <kernel_trace.csv>"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
syrk = [i for i, r in enumerate(rows) if "syrk16_256" in r["Kernel_Name"]]
n_per = int(sys.argv[2]) if len(sys.argv) > 2 else 8
first = syrk[-n_per]
# the step starts with the staging copies before its first SYRK: walk back over stage_rows kernels
i0 = first
while i0 > 0 and ("stage_rows" in rows[i0 - 1]["Kernel_Name"] or "FillFunctor" in rows[i0 - 1]["Kernel_Name"]
                  or "syrk" in rows[i0 - 1]["Kernel_Name"] and i0 - 1 >= first):
    i0 -= 1
t0 = int(rows[i0]["Start_Timestamp"])
step = rows[i0:]
end = max(int(r["End_Timestamp"]) for r in step)
print(f"last step: {len(step)} kernels, span {(end - t0) / 1e6:.2f} ms")
byq = collections.OrderedDict()
for r in step:
    byq.setdefault(r["Queue_Id"], []).append(r)
for q, rs in byq.items():
    a, b = int(rs[0]["Start_Timestamp"]) - t0, max(int(r["End_Timestamp"]) for r in rs) - t0
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs)
    print(f"queue {q}: {a / 1e6:8.2f} .. {b / 1e6:8.2f} ms, {len(rs)} kernels, busy {busy / 1e6:7.2f} ms")
    agg = collections.OrderedDict()
    for r in rs:
        k = r["Kernel_Name"].split("(")[0][-48:]
        e = agg.setdefault(k, [int(r["Start_Timestamp"]) - t0, 0, 0, 0])
        e[1] = int(r["End_Timestamp"]) - t0
        e[2] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        e[3] += 1
    for k, (a, b, tot, n) in agg.items():
        if tot > 300000 or n > 20:
            print(f"      {k:48s} first {a / 1e6:8.2f} last {b / 1e6:8.2f} sum {tot / 1e6:7.2f} ms n={n}")
