#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output under a directory: per-kernel mean duration and mean counter values per launch."""
import collections, csv, glob, json, sys
root = sys.argv[1]
out = {"kernels": {}, "counters": {}}
for f in glob.glob(root + "/**/*_kernel_trace.csv", recursive=True):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        d[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for k, v in d.items():
        out["kernels"][k] = {"calls": len(v), "mean_us": sum(v) / len(v) / 1e3, "min_us": min(v) / 1e3, "max_us": max(v) / 1e3}
for f in glob.glob(root + "/**/*_counter_collection.csv", recursive=True):
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        d[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in d.items():
        if "rt_" not in k:
            continue
        for c, v in cs.items():
            out["counters"].setdefault(k, {})[c] = {"launches": len(v), "mean": sum(v) / len(v)}
json.dump(out, sys.stdout, indent=1)
