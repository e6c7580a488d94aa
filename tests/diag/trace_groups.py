"""Diagnostic: group a rocprofv3 kernel-trace csv by (kernel, grid, workgroup, LDS) -> count, mean, total; biggest first."""
import csv, sys
from collections import defaultdict
d = defaultdict(lambda: [0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"].replace("void ", "").replace("loco::", "").replace("(anonymous namespace)::", "").split("(")[0][:60]
    key = (n, r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", ""), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "")), r.get("LDS_Block_Size", ""))
    d[key][0] += 1; d[key][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
tot = sum(v[1] for v in d.values())
print(f"total {tot/1e6:.1f} ms, {len(d)} groups")
for k, (c, t) in sorted(d.items(), key=lambda x: -x[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 60]:
    print(f"{t/1e6:9.2f} ms {100*t/tot:5.1f}% {c:6d} x {t/c/1e3:9.1f} us  {k[0]:60s} grid {k[1]}x{k[2]}x{k[3]} wg {k[4]} lds {k[5]}")
