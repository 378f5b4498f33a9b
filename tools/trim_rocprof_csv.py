"""Trim the kernel names in a rocprofv3 CSV (torch's template names run to kilobytes) so the
summary is readable; keeps every numeric column untouched.  usage: trim_rocprof_csv.py in.csv out.csv"""
import csv
import sys

with open(sys.argv[1], newline="") as f, open(sys.argv[2], "w", newline="") as g:
    r, w = csv.reader(f), csv.writer(g)
    for row in r:
        w.writerow([c if len(c) <= 100 else c[:97] + "..." for c in row])
