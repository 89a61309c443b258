#!/bin/bash
# kernel stats of LM trials on the dense-visibility scene (scripts/oleg_timing.py): ba_dense.h kernels
OUT=$GRAFT_REPO_ROOT/gpurun_out/dense_prof; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o d -- python $GRAFT_REPO_ROOT/scripts/oleg_timing.py > $OUT/log.txt 2>&1
python3 - <<PY
import csv
for r in list(csv.reader(open('$OUT/d_kernel_stats.csv')))[:14]: print(r[0].split('(')[0][-44:], r[1], r[3][:8], r[4][:5])
PY
rm -f $OUT/*kernel_trace.csv
