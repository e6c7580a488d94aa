"""Diagnostic (by hand): rank the layer shapes of a per-shape conv profile (bench.py with LOCO_BENCH_SHAPES=<file>, any workload; the
profile is the first engine context's) by their time above max(flops at 350 TFLOP/s, input + output bytes at 4 TB/s).
    LOCO_BENCH_SHAPES=gpurun_out/s.json python3 bench.py --workload tloco_sd15 --steps 1 --warmup 1 --no-extra --no-e2e --no-cpu-baseline
    python3 tests/diag/shape_excess.py gpurun_out/s.json [rows]"""
import json, re, sys
rep = json.load(open(sys.argv[1]))
N = int(sys.argv[2]) if len(sys.argv) > 2 else 40
tot = sum(v["ms"] for v in rep.values())
rows = []
for k, v in rep.items():
    m = re.search(r"t(\d)_m(\d)_ci(\d+)_co(\d+)_h(\d+)_b(\d+)_s(\d+)", k)
    bu = 0.0
    if m and v["flops"] > 0:
        taps, mode, ci, co, h, b, sp = map(int, m.groups())
        bu = max(v["flops"] / v["launches"] / 350e12, 4.0 * h * h * b * (ci + co) / 4e12) * 1e6
    rows.append((k, v, bu, v["ms"] * 1e3 - bu * v["launches"]))
print(f"conv total {tot:.2f} ms over {sum(v['launches'] for v in rep.values())} launches")
for k, v, bu, ex in sorted(rows, key=lambda r: -r[3])[:N]:
    print(f"{k:78s} n={v['launches']:4d} ms={v['ms']:8.3f} ({100 * v['ms'] / tot:4.1f}%) {v['flops'] / max(v['ms'], 1e-9) / 1e9:7.1f} TF/s  "
          f"{v['ms'] * 1e3 / v['launches']:7.1f} us/launch, bound {bu:6.1f}  excess {ex / 1e3:7.2f} ms")
