"""Timeline of one bench step from a GQ_PROF_DUMP file (bench.py --breakdown).
usage: GQ_PROF_DUMP=gpurun_out/tl.txt python bench.py --breakdown ... ; python profiles/timeline.py gpurun_out/tl.txt
Prints, per stream, the first/last instant of each phase and, per 5-ms bucket, the busy fraction of each tag."""
import sys, collections
recs, cur = [], []
for line in open(sys.argv[1]):
    if line.startswith("#"):
        recs.append(cur); cur = []
        continue
    tag, name, st, a, b = line.split()
    cur.append((name, st, float(a), float(b)))
step = max(recs, key=len)  # the breakdown step records every tag
t0 = min(r[2] for r in step)
streams = sorted({r[1] for r in step}, key=lambda s: min(r[2] for r in step if r[1] == s))
print(f"{len(step)} intervals, {len(streams)} streams, span {max(r[3] for r in step) - t0:.2f} ms")
for i, s in enumerate(streams):
    rs = [r for r in step if r[1] == s]
    byname = collections.OrderedDict()
    for n, _, a, b in sorted(rs, key=lambda r: r[2]):
        e = byname.setdefault(n, [a, b, 0.0, 0])
        e[1] = max(e[1], b); e[2] += b - a; e[3] += 1
    print(f"stream {i}: {min(r[2] for r in rs) - t0:7.2f} .. {max(r[3] for r in rs) - t0:7.2f} ms")
    for n, (a, b, tot, cnt) in byname.items():
        print(f"    {n:22s} first {a - t0:7.2f}  last {b - t0:7.2f}  sum {tot:7.2f} ms  n={cnt}")
