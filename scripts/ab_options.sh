#!/bin/bash
# A/B of library options on the GPU box: `gpurun -- bash scripts/ab_options.sh "" "fast_paths=0" ...`: three alternating runs
# of bench.py per option set; prints min / median of five 20-step windows, the kernel sum and the per-kernel microseconds.
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for V in "$@"; do
  OPTS=""; for kv in $V; do OPTS="$OPTS --option $kv"; done
  python bench.py --full-line --windows 5 --no-cpu-baseline --no-lm --no-other-configs --no-live-pmc $OPTS $BENCH_ARGS 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('%-24s' % '[$V]', 'min %.4f median %.4f ms' % (d['ms_per_step_windows']['min'], d['ms_per_step_windows']['median']), 'kernel sum %.1f us' % (1e3 * sum(k.values())), {n: round(v*1000,1) for n, v in k.items()})"
done; done
