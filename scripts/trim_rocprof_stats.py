#!/usr/bin/env python3
"""Trim a rocprofv3 *_kernel_stats.csv for committing under profiles/: shorten the
(kilobyte-long Tensile) kernel names, keep every numeric column.
usage: trim_rocprof_stats.py in.csv out.csv [max_rows]"""
import csv
import sys

src, dst = sys.argv[1], sys.argv[2]
max_rows = int(sys.argv[3]) if len(sys.argv) > 3 else 40
rows = list(csv.reader(open(src)))
with open(dst, 'w', newline='') as f:
    w = csv.writer(f)
    w.writerow(rows[0])
    for r in rows[1:1 + max_rows]:
        r[0] = r[0] if len(r[0]) <= 110 else r[0][:107] + '...'
        w.writerow(r)
