"""Diagnostic: per-counter mean over the launches of one kernel in a rocprofv3 --pmc run (CSV out: counter, mean per
launch, launches).  python tests/diag/pmc_summarise.py <dir> <kernel substring> > profiles/rNN_..._pmc_sq_X.csv"""
import collections, csv, glob, os, sys
files = glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True)
acc = collections.defaultdict(list)
for f in files:
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("counter,mean_per_launch,launches")
for k in sorted(acc):
    print(f"{k},{sum(acc[k]) / len(acc[k]):.6g},{len(acc[k])}")
