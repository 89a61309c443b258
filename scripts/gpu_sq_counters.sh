#!/bin/bash
# SQ counters of the trial's kernels (own rocprofv3 pass, counters only): matrix-core busy cycles, wave cycles, parked / stalled / issuing
# usage (via gpurun): bash scripts/gpu_sq_counters.sh <tag> [bench.py flags]
TAG=${1:-r03}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/sq_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-lm --windows 0 --no-other-configs --no-live-pmc $*"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc -o $TAG -- $CMD > $OUT/pmc.log 2>&1
cd $GRAFT_REPO_ROOT
python - $OUT $TAG <<'PY'
import collections, csv, glob, os, sys
out, tag = sys.argv[1], sys.argv[2]
f = glob.glob(os.path.join(out, 'pmc', '**', '*counter_collection.csv'), recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f[0])):
    k = r['Kernel_Name'].split('(')[0].replace('void ', '')[:60]
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    if r['Counter_Name'] == 'SQ_WAVE_CYCLES': n[k] += 1
cols = ['SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_BUSY_CYCLES', 'SQ_WAVE_CYCLES', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_WAIT_INST_LDS']
with open(os.path.join(out, tag + '_sq_counters.csv'), 'w', newline='') as g:
    w = csv.writer(g); w.writerow(['kernel', 'launches'] + [c + '_per_launch' for c in cols] + ['wait_any/wave', 'wait_inst/wave', 'active/wave'])
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1]['SQ_WAVE_CYCLES']):
        m = max(1, n[k]); wc = max(1., v['SQ_WAVE_CYCLES'])
        w.writerow([k, n[k]] + ['%.0f' % (v[c] / m) for c in cols] + ['%.3f' % (v['SQ_WAIT_ANY'] / wc), '%.3f' % (v['SQ_WAIT_INST_ANY'] / wc), '%.3f' % (v['SQ_ACTIVE_INST_ANY'] / wc)])
print(open(os.path.join(out, tag + '_sq_counters.csv')).read())
PY
