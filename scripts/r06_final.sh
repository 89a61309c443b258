#!/bin/bash
# The last GPU act of the round: smoke(), `python -m pytest tests -m gpu -x -q` twice at HEAD, the solve-accuracy table.
#   usage (GPU box): bash scripts/r06_final.sh <commit>        -> gpurun_out/r06_final/
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_final
mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
L=$O/r06_gpu_tests_at_head.log
echo "# python -m pytest tests -m gpu -x -q, twice, at HEAD (commit $1) $(date -u +%Y-%m-%dT%H:%MZ)" > $L
rc=0
for i in 1 2; do
  python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/run$i.log 2>&1 || rc=1
  grep -E "^[.sFEx]+ +\[|passed|failed|^FAILED|^ERROR" $O/run$i.log >> $L
done
tail -3 $L
python scripts/solve_accuracy.py 20 > $O/r06_solve_accuracy.txt 2>&1; tail -6 $O/r06_solve_accuracy.txt
exit $rc
