"""Diagnostic: from a rocprofv3 kernel trace csv, busy time vs span of the last evaluations (gap = dispatch latency)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
k = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows]
# evaluations start at temb_kernel
starts = [i for i, r in enumerate(k) if "temb_kernel" in r[2]]
for a, b in zip(starts[-4:-1], starts[-3:]):
    seg = k[a:b]
    busy = sum(e - s for s, e, _ in seg)
    span = seg[-1][1] - seg[0][0]
    print(f"kernels {len(seg)}  busy {busy/1e6:.3f} ms  span {span/1e6:.3f} ms  mean gap {(span-busy)/len(seg)/1e3:.2f} us")
from collections import defaultdict
d = defaultdict(lambda: [0, 0])
for s, e, n in k[starts[-2]:starts[-1]]:
    n = n.split("(")[0][:70]; d[n][0] += 1; d[n][1] += e - s
for n, (c, t) in sorted(d.items(), key=lambda x: -x[1][1])[:14]:
    print(f"{t/1e3:9.1f} us {c:4d}  {n}")
