#!/bin/bash
# Round 6: the time lines of the solve and of the reduction at HEAD (PROFILE build on the box; the product library is not touched:
# the box's copy of the tree is scratch).  Output: gpurun_out/r06p/
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06p
mkdir -p $O
make -C pysfm_amd/csrc -j8 PROFILE=1 > $O/make.log 2>&1 || { tail -20 $O/make.log; exit 1; }
python scripts/bcr_phase_trace.py 1000 100000 > $O/bcr_phase_trace.out 2> $O/bcr_phase_trace.txt
python scripts/schur_phase_trace.py > $O/schur_phase_trace.out 2> $O/schur_phase_trace.txt
tail -5 $O/bcr_phase_trace.out
grep -c . $O/bcr_phase_trace.txt $O/schur_phase_trace.txt
