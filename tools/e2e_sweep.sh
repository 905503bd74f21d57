#!/bin/bash
# e2e_stage2 against the number of stream groups and host threads (GPU box): bash tools/e2e_sweep.sh [repeats]
ARGS="--steps 3 --warmup 2 --cpu-sample 0 --cpu-procs 0 --pcie-steps 0 --unbatched-frames 0 --lm-depth5 0 --e2e-depth5-streams 0"
R=${1:-1}
for cfg in "2 0" "3 12" "4 12" "4 8" "3 16"; do
  set -- $cfg
  if [ "$2" != "0" ]; then export SSK_RBA_BATCH_THREADS=$2 SSK_LM_BUILD_THREADS=$2; else unset SSK_RBA_BATCH_THREADS SSK_LM_BUILD_THREADS; fi
  for r in $(seq $R); do
  python3 bench.py $ARGS --e2e-groups $1 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d['e2e_stage2']
print('groups $1 threads $2: value %.0f ms_per_step %.2f (serial %.0f)' % (e['value'], e['ms_per_step'], e.get('serial',{}).get('value',0)))"
  done
done
