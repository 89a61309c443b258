#!/bin/bash
# kernel stats of the reduced solve at a wide band (scripts/band_cliff.py restricted to one track length)
OUT=$GRAFT_REPO_ROOT/gpurun_out/wide_prof; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
BAND_L=${1:-${BAND_L:-22}} timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o w -- python $GRAFT_REPO_ROOT/scripts/band_cliff.py > $OUT/log.txt 2>&1
python3 - <<PY
import csv
for r in list(csv.reader(open('$OUT/w_kernel_stats.csv')))[:9]: print(r[0].split('(')[0][-40:], r[1], r[3][:8], r[4][:5])
PY
rm -f $OUT/*kernel_trace.csv
