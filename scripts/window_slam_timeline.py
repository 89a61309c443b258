"""The device-side time line of window_slam.run from a rocprofv3 kernel trace (kernel_trace.csv of
`rocprofv3 --kernel-trace --output-format csv -- python scripts/window_slam_profile.py`): for the LAST 91 launches of the resident
loop (the timed run; the warm-up run before it takes three steps a window) - kernel time, the other device work of a window,
and the idle time between a window's kernel and the next window's.  usage: python scripts/window_slam_timeline.py kernel_trace.csv"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows))
res = [i for i, e in enumerate(ev) if 'k_resident_lm' in e[2]][-91:]
first = res[0]
# the window of launch i: everything after the previous resident kernel up to and including this one
tot_res = tot_other = tot_idle = 0
names = {}
for a, i in zip([res[0] - 8] + res[:-1], res):
    s, e, _ = ev[i]
    tot_res += e - s
    prev_end = ev[a][1]
    busy = 0
    for k in range(a + 1, i):
        busy += ev[k][1] - ev[k][0]
        names[ev[k][2][:60]] = names.get(ev[k][2][:60], 0) + 1
    tot_other += busy
    if a >= first: tot_idle += s - prev_end - busy
span = ev[res[-1]][1] - ev[res[0]][0]
print('last 91 resident launches: span %.1f ms; resident kernels %.1f ms (%.0f us each); other device work %.2f ms; device idle between windows %.1f ms (%.0f us per window)'
      % (span / 1e6, tot_res / 1e6, tot_res / 91e3, tot_other / 1e6, tot_idle / 1e6, tot_idle / 90e3))
print('other kernels per run:', names)
