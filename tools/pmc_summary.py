#!/usr/bin/env python3
"""Mean of every collected counter per solver kernel from a rocprofv3 --pmc output directory
(development tool): python tools/pmc_summary.py <dir> [out.json]"""
import collections
import csv
import glob
import json
import os
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(os.path.join(sys.argv[1], "**", "*_counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].split("(")[0].split("<")[0].replace("void fs::", "")
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}
for k, cs in out.items():
    cs["launches"] = len(next(iter(acc[k].values())))
print(json.dumps(out, indent=1))
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
