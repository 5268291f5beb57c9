"""Summarise rocprofv3 --pmc CSVs written by tools/pmc.sh: per-launch averages of
every counter for the kernels whose name contains a pattern (default k_step)."""
import csv, glob, sys, collections
d = sys.argv[1]; pat = sys.argv[2] if len(sys.argv) > 2 else "k_step"
acc = collections.defaultdict(list)
for f in sorted(glob.glob(f"{d}/pass*/*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(f"{k:32s} launches={len(v):4d} mean={sum(v)/len(v):16.1f}")
