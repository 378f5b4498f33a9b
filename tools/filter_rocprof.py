"""Shrink a rocprofv3 output directory in place (run on the GPU box after each pass): keep the per-kernel statistics
whole, keep only the library's kernels' rows (namespace ff::) of the per-dispatch CSVs (the Euler-Maruyama workload alone
launches thousands of torch random-number kernels), drop everything else.  usage: filter_rocprof.py DIR"""
import csv
import sys
from pathlib import Path

root = Path(sys.argv[1])
for f in list(root.rglob("*")):
    if not f.is_file():
        continue
    if f.name.endswith("kernel_stats.csv"):
        continue
    if f.name.endswith("kernel_trace.csv") or f.name.endswith("counter_collection.csv"):
        with open(f, newline="") as g:
            r = csv.reader(g)
            head = next(r)
            k = head.index("Kernel_Name")
            rows = [row for row in r if "ff::" in row[k]]
        with open(f, "w", newline="") as g:
            w = csv.writer(g)
            w.writerow(head)
            w.writerows(rows)
        continue
    f.unlink()
