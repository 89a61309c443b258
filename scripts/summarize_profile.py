#!/usr/bin/env python3
"""Condense a scripts/gpu_profile.sh output directory into small files for profiles/:
  <tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary (names trimmed)
  <tag>_hbm_traffic.csv    per-kernel FETCH_SIZE / WRITE_SIZE per launch (separate --pmc passes)
HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE is in KB and on gfx950
reads exactly half of a wide coalesced stream (MI355X_MICROARCH.md, HBM section), WRITE_SIZE is in KB.
usage: summarize_profile.py gpurun_out/prof_<tag> <tag> profiles/"""
import collections
import csv
import os
import sys

src, tag, dst = sys.argv[1], sys.argv[2], sys.argv[3]


def short(name):
    name = name.split('(')[0].replace('void ', '')
    return name if len(name) <= 100 else name[:97] + '...'


rows = list(csv.reader(open(os.path.join(src, 'trace', tag + '_kernel_stats.csv'))))
with open(os.path.join(dst, tag + '_kernel_stats.csv'), 'w', newline='') as f:
    w = csv.writer(f)
    w.writerow(rows[0])
    for r in rows[1:41]:
        r[0] = short(r[0])
        w.writerow(r)

agg = collections.defaultdict(lambda: dict(n=0, FETCH_SIZE=0.0, WRITE_SIZE=0.0))
for which in ('fetch', 'write'):
    path = os.path.join(src, 'pmc_' + which, tag + '_counter_collection.csv')
    for r in csv.DictReader(open(path)):
        a = agg[short(r['Kernel_Name'])]
        a[r['Counter_Name']] += float(r['Counter_Value'])
        if which == 'fetch':
            a['n'] += 1
with open(os.path.join(dst, tag + '_hbm_traffic.csv'), 'w', newline='') as f:
    w = csv.writer(f)
    w.writerow(['kernel', 'launches', 'FETCH_SIZE_KB_per_launch', 'WRITE_SIZE_KB_per_launch',
                'hbm_bytes_per_launch=(2*FETCH+WRITE)*1024'])
    for k, a in sorted(agg.items(), key=lambda kv: -(2 * kv[1]['FETCH_SIZE'] + kv[1]['WRITE_SIZE'])):
        n = max(1, a['n'])
        w.writerow([k, a['n'], '%.1f' % (a['FETCH_SIZE'] / n), '%.1f' % (a['WRITE_SIZE'] / n),
                    int((2 * a['FETCH_SIZE'] + a['WRITE_SIZE']) / n * 1024)])
print(open(os.path.join(dst, tag + '_hbm_traffic.csv')).read())
