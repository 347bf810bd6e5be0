"""Raw per-kernel averages of one rocprofv3 --pmc counter (calibration helper): python tools/pmc_raw.py <counter_collection.csv> <COUNTER>"""
import csv
import re
import sys
from collections import defaultdict

acc = defaultdict(lambda: [0, 0.0])
with open(sys.argv[1], newline="") as f:
    for r in csv.DictReader(f):
        if r["Counter_Name"] == sys.argv[2]:
            k = re.sub(r"\(.*$", "", re.sub(r"^void ", "", re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])))
            acc[k][0] += 1
            acc[k][1] += float(r["Counter_Value"])
for k, (n, v) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:6]:
    print(f"{k[:70]:70s} launches {n:4d}  {sys.argv[2]} per launch {v / n:14.1f}")
