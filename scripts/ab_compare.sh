cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for V in old new; do cp pysfm_amd/libpysfm_ba_$V.so pysfm_amd/libpysfm_ba.so; python bench.py --windows 5 --no-cpu-baseline --no-lm 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$V', round(d['ms_per_step_windows']['min'],4), round(d['ms_per_step_windows']['median'],4), round(d['kernel_ms_per_step']['bcr_eliminate']*1000,1))"; done; done
