#!/bin/bash
# the whole GPU suite N times, no -x: which tests ever fail (run-to-run differences come from fp64 atomics and from timing)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/flake
N=${1:-3}
for i in $(seq 1 $N); do
  python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/flake/run$i.log 2>&1
  grep -E "passed|failed" gpurun_out/flake/run$i.log | tail -1
  grep -E "^FAILED|^ERROR" gpurun_out/flake/run$i.log
done
