"""Aggregate two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs as the MI355X guide
prescribes) into profiles/traffic.json: HBM-side bytes per launch for each conv kernel variant.
Units: FETCH_SIZE / WRITE_SIZE are KiB.  gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE reports half of the bytes of
a WIDE COALESCED streaming read (16 B per lane, 1 KiB per wave); other patterns are to be calibrated on a known byte count.
The conv kernels read 64-byte runs of halo rows (four lanes x 16 B), and the calibration on their own pattern
(profiles/r03_fetch_size_calibration.md: the tangent form's extra fetch at one probe is 69.0 MB raw against 67.1 MB of unique /
89.5 MB of requested primal-cache bytes) shows the raw counter is exact for them: NO doubling (rounds 1-2 doubled the
vector-staged variants and over-stated their traffic).  WRITE_SIZE is exact for these stores.

The result carries the sha256 of the libloco_hip.so it was measured on (`_lib_sha256`): bench.py only reports
`roofline.traffic` from a traffic.json whose hash matches the library it is running (a stale file yields null).

    python profiles/make_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write r03 [output dir, default profiles/]
"""
import collections, csv, glob, hashlib, json, os, re, sys

def load(d, cn):
    f = (glob.glob(os.path.join(d, "*counter_collection.csv")) + glob.glob(os.path.join(d, "*", "*counter_collection.csv")))[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == cn:
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return agg

def norm(name):
    m = re.match(r"void loco::(conv_pair_bf16x3)<(\d+)>\(", name)      # round 6: the 16x16x32 tap-pair kernel (same 64-byte-run halo loads)
    if m:
        return f"{m.group(1)}<{m.group(2)}>"
    m = re.match(r"void loco::(conv_mfma_\w+)<([\d, ]+?)(?:, (?:true|false))?>\(", name)
    if not m:
        return None
    p = m.group(2).replace(' ', '').split(',')
    # the no-doubling calibration was made on the vector-staged path (16-byte runs of halo rows, 7th parameter 0 / 3); the
    # per-pixel staging variants (1, 2: 4-byte loads) are NOT calibrated -- a conv cannot fetch less than it stores, and the raw
    # counter says it does for them -- so they get no traffic.json entry (their rows stay in the per-kernel csv, marked)
    if len(p) > 6 and p[6] not in ("0", "3"):
        return None
    # bench.py names a variant by <TAPS,WM,WN,TM,TN,MODE>
    return f"{m.group(1)}<{','.join(p[:6])}>"

fd, wd, tag = sys.argv[1], sys.argv[2], sys.argv[3]
outdir = sys.argv[4] if len(sys.argv) > 4 else os.path.dirname(os.path.abspath(__file__))
F, W = load(fd, "FETCH_SIZE"), load(wd, "WRITE_SIZE")
out, rows = {}, []
def fetch_corr(name):
    return 1.0      # calibrated on the kernel's own access pattern, see the header

for k in F:
    n = norm(k)
    f = sum(F[k]) / len(F[k]) * 1024 * fetch_corr(k)
    w = sum(W.get(k, [0])) / max(1, len(W.get(k, [0]))) * 1024
    rows.append((k, len(F[k]), f, w))
    if n:
        if n in out:   # staging flavours share a variant name: launch-weighted mean
            o = out[n]
            tot = o["launches"] + len(F[k])
            o["bytes"] = (o["bytes"] * o["launches"] + (f + w) * len(F[k])) / tot
            o["launches"] = tot
        else:
            out[n] = {"bytes": f + w, "launches": len(F[k])}
here = outdir
lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "loco-edit_amd", "libloco_hip.so")
res = {k: round(v["bytes"]) for k, v in out.items()}
res["_lib_sha256"] = hashlib.sha256(open(lib, "rb").read()).hexdigest() if os.path.exists(lib) else None
res["_source"] = (f"profiles/{tag}: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE passes (separate runs) of `bench.py --steps 1 "
                  "--warmup 0`, mean bytes per launch of each conv variant; raw counters x 1024 (calibrated on the kernel's own "
                  "64-byte-run access pattern: profiles/r03_fetch_size_calibration.md), WRITE exact for 16-byte streaming stores")
json.dump(res, open(os.path.join(here, "traffic.json"), "w"), indent=1)
with open(os.path.join(here, f"{tag}_pmc_traffic_per_kernel.csv"), "w") as fh:
    fh.write("kernel,launches,fetch_bytes_per_launch_raw,write_bytes_per_launch,calibrated\n")
    for k, n, f, w in sorted(rows, key=lambda r: -r[1] * (r[2] + r[3])):
        fh.write(f"\"{k}\",{n},{f:.0f},{w:.0f},{'yes' if norm(k) else 'no'}\n")
print(json.dumps({k: round(v['bytes'] / 1e6, 1) for k, v in out.items()}, indent=1))
