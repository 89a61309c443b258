#!/bin/bash
# SQ counters of the trial's kernels (own rocprofv3 passes, counters only - no trace domains): what `limited_by` in the bench
# line rests on.  Pass A: cycles (busy, wave, waiting, issuing, matrix-core busy).  Pass B: instruction mix (fp64 MFMA ops,
# fp64 FMAs, VALU, LDS), LDS bank conflicts, waves launched.  Pass C: occupancy (SQ_LEVEL_WAVES accumulates, own pass).
# usage (via gpurun): bash scripts/gpu_sq_counters.sh <tag> [bench.py flags]
TAG=${1:-r05}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/sq_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-lm --windows 0 --no-other-configs --no-live-pmc $*"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmcA -o $TAG -- $CMD > $OUT/pmcA.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES --output-format csv -d $OUT/pmcB -o $TAG -- $CMD > $OUT/pmcB.log 2>&1
rocprofv3 --pmc SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmcC -o $TAG -- $CMD > $OUT/pmcC.log 2>&1
cd $GRAFT_REPO_ROOT
python - $OUT $TAG <<'PY'
import collections, csv, glob, os, sys
out, tag = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for p in ('pmcA', 'pmcB', 'pmcC'):
    f = glob.glob(os.path.join(out, p, '**', '*counter_collection.csv'), recursive=True)
    if not f:
        print('pass', p, 'left no counter file:', open(os.path.join(out, p + '.log')).read()[-400:])
        continue
    seen = set()
    for r in csv.DictReader(open(f[0])):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')[:60]
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        if p == 'pmcA' and r['Counter_Name'] == 'SQ_WAVE_CYCLES': n[k] += 1
cols = ['SQ_BUSY_CYCLES', 'SQ_WAVE_CYCLES', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_WAIT_INST_LDS', 'SQ_VALU_MFMA_BUSY_CYCLES',
        'SQ_INSTS_VALU_MFMA_MOPS_F64', 'SQ_INSTS_VALU_FMA_F64', 'SQ_INSTS_VALU', 'SQ_INSTS_LDS', 'SQ_LDS_BANK_CONFLICT', 'SQ_LDS_IDX_ACTIVE', 'SQ_WAVES',
        'SQ_LEVEL_WAVES', 'SQ_BUSY_CU_CYCLES', 'GRBM_GUI_ACTIVE']
with open(os.path.join(out, tag + '_sq_counters.csv'), 'w', newline='') as g:
    w = csv.writer(g)
    w.writerow(['kernel', 'launches'] + [c + '_per_launch' for c in cols] +
               ['wait_any/wave_cycles', 'wait_inst/wave_cycles', 'active/wave_cycles', 'mfma_busy/busy_cycles(4 SIMDs per SQ_BUSY cycle)',
                'lds_conflict/lds_active', 'mean_waves_per_busy_CU', 'fp64_flops_per_launch=(512*MOPS_F64+128*FMA_F64)'])
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1]['SQ_WAVE_CYCLES']):
        m = max(1, n[k]); wc = max(1., v['SQ_WAVE_CYCLES'])
        w.writerow([k, n[k]] + ['%.0f' % (v[c] / m) for c in cols] +
                   ['%.3f' % (v['SQ_WAIT_ANY'] / wc), '%.3f' % (v['SQ_WAIT_INST_ANY'] / wc), '%.3f' % (v['SQ_ACTIVE_INST_ANY'] / wc),
                    '%.3f' % (v['SQ_VALU_MFMA_BUSY_CYCLES'] / max(1., 4 * v['SQ_BUSY_CYCLES'])),
                    '%.3f' % (v['SQ_LDS_BANK_CONFLICT'] / max(1., v['SQ_LDS_IDX_ACTIVE'])),
                    '%.2f' % (v['SQ_LEVEL_WAVES'] / max(1., v['SQ_BUSY_CU_CYCLES'])),
                    '%.0f' % ((512 * v['SQ_INSTS_VALU_MFMA_MOPS_F64'] + 128 * v['SQ_INSTS_VALU_FMA_F64']) / m)])
print(open(os.path.join(out, tag + '_sq_counters.csv')).read())
PY
