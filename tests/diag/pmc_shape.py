"""Diagnostic: mean FETCH_SIZE / WRITE_SIZE (KiB -> MB, fetch doubled for the 16-byte-load variants: gfx950 correction)
of the conv kernel in a rocprofv3 --pmc run of tests/conv_pmc.py.  python tests/diag/pmc_shape.py <dir> <counter>"""
import csv, glob, os, sys
f = (glob.glob(os.path.join(sys.argv[1], "*counter_collection.csv")) + glob.glob(os.path.join(sys.argv[1], "*", "*counter_collection.csv")))[0]
v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if r["Counter_Name"] == sys.argv[2] and "conv_mfma" in r["Kernel_Name"]]
corr = 2.0 if sys.argv[2] == "FETCH_SIZE" else 1.0
print(sys.argv[2], len(v), "launches, mean MB per launch", round(sum(v) / len(v) * 1024 * corr / 1e6, 1), "(min", round(min(v) * 1024 * corr / 1e6, 1), "max", round(max(v) * 1024 * corr / 1e6, 1), ")")
