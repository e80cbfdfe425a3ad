#!/usr/bin/env python
"""Summarise rocprofv3 --pmc counter_collection.csv: per kernel name, mean of each counter and duration."""
import csv, glob, sys, collections
for d in sys.argv[1:]:
    for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(path)):
            k = r["Kernel_Name"][:60]
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if "Start_Timestamp" in r and r.get("End_Timestamp"):
                acc[k]["_dur_us"].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
        for k, cs in acc.items():
            if "conv" not in k and "bn_" not in k:
                continue
            print(d, "|", k)
            for c, v in sorted(cs.items()):
                print(f"    {c:28s} {sum(v)/len(v):16.1f}  (n={len(v)})")
