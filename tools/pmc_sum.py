"""Sums rocprofv3 --pmc counter_collection.csv files per counter and kernel: python tools/pmc_sum.py DIR [DIR...]"""
import csv, glob, sys, collections
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(float); n = collections.defaultdict(int)
        for row in csv.DictReader(open(f)):
            key = (row["Kernel_Name"][:40], row["Counter_Name"])
            agg[key] += float(row["Counter_Value"]); n[key] += 1
        for k in sorted(agg):
            print(f"{k[0]:40s} {k[1]:28s} total {agg[k]:.4g} over {n[k]} dispatch rows")
