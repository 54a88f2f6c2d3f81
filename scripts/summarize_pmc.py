"""Summarise rocprofv3 counter_collection CSVs: per kernel, per counter: mean value per dispatch."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
for path in sorted(glob.glob(os.path.join(out, "pmc_*.csv"))):
    acc = defaultdict(lambda: [0.0, 0])
    with open(path) as f:
        for row in csv.DictReader(f):
            k = (row.get("Kernel_Name", "?")[:60], row.get("Counter_Name", "?"))
            acc[k][0] += float(row.get("Counter_Value", 0) or 0)
            acc[k][1] += 1
    print("==", os.path.basename(path))
    for (kern, ctr), (tot, n) in sorted(acc.items()):
        print("%-60s %-24s mean/dispatch %.6g  (n=%d)" % (kern, ctr, tot / max(n, 1), n))
