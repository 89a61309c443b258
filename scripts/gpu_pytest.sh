#!/bin/bash
# pytest on the GPU box with the summary where the tail of the output can see it (RCCL prints its banner last)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest -m gpu -q -x -p no:cacheprovider "$@" > gpurun_out/pytest_last.log 2>&1
rc=$?
grep -E "passed|failed|error|FAILED|ERROR|^E  " gpurun_out/pytest_last.log | tail -25
exit $rc
