#!/usr/bin/env python3
"""Condense the three counter-only rocprofv3 passes of scripts/gpu_sq_counters.sh into one row per kernel.
usage: summarize_sq_counters.py gpurun_out/sq_<tag> <tag>
Units on MI355X (checked against kernel durations): SQ_BUSY_CYCLES is summed over the 32 shader engines, SQ_WAVE_CYCLES /
SQ_WAIT_* / SQ_ACTIVE_* over all wavefronts, SQ_VALU_MFMA_BUSY_CYCLES over the 1024 SIMDs (a v_mfma_f64_16x16x4_f64 = 64 cycles = 4 of
SQ_INSTS_VALU_MFMA_MOPS_F64's 512-flop units), GRBM_GUI_ACTIVE over the 8 XCDs.  SQ_LEVEL_WAVES (occupancy) reads 0 in this
counter-only mode on this pool."""
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
               ['wait_any/wave_cycles', 'wait_inst/wave_cycles', 'active/wave_cycles', 'mfma_busy_fraction_of_a_SIMD = MFMA_BUSY / (32 * SQ_BUSY_CYCLES)  [1024 SIMDs, SQ_BUSY summed over 32 shader engines]',
                'lds_conflict/lds_active', 'mean_waves_per_busy_CU', 'fp64_flops_per_launch=(512*MOPS_F64+128*FMA_F64)'])
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1]['SQ_WAVE_CYCLES']):
        m = max(1, n[k]); wc = max(1., v['SQ_WAVE_CYCLES'])
        w.writerow([k, n[k]] + ['%.0f' % (v[c] / m) for c in cols] +
                   ['%.3f' % (v['SQ_WAIT_ANY'] / wc), '%.3f' % (v['SQ_WAIT_INST_ANY'] / wc), '%.3f' % (v['SQ_ACTIVE_INST_ANY'] / wc),
                    '%.3f' % (v['SQ_VALU_MFMA_BUSY_CYCLES'] / max(1., 32 * v['SQ_BUSY_CYCLES'])),
                    '%.3f' % (v['SQ_LDS_BANK_CONFLICT'] / max(1., v['SQ_LDS_IDX_ACTIVE'])),
                    '%.2f' % (v['SQ_LEVEL_WAVES'] / max(1., v['SQ_BUSY_CU_CYCLES'])),
                    '%.0f' % ((512 * v['SQ_INSTS_VALU_MFMA_MOPS_F64'] + 128 * v['SQ_INSTS_VALU_FMA_F64']) / m)])
print(open(os.path.join(out, tag + '_sq_counters.csv')).read())
