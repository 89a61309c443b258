#!/bin/bash
# time line of the fused elimination (PROFILE build on the box)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03c
make -C pysfm_amd/csrc PROFILE=1 > gpurun_out/r03c/make.log 2>&1
python scripts/bcr_phase_trace.py 1000 100000 2> gpurun_out/r03c/trace.log | tail -3
tail -40 gpurun_out/r03c/trace.log
