#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out
timeout 2400 python -m pytest tests -m gpu -q > $O/r02b_pytest.log 2>&1; echo "pytest rc=$?" >> $O/r02b_pytest.log
tail -15 $O/r02b_pytest.log
