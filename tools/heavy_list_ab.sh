#!/bin/bash
# A/B of the work-item length on the data-file-order shape (and the Zipf shape): VFM_HEAVY_LIST forces vfm_heavy_list_for.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$1; shift
: > $OUT
for e in "$@"; do
  envs=$e; [ "$e" = "-" ] && envs=""
  for pt in "--user-order --zipf 1.1 --n-batches 160 --steps 320 --warmup 20" "--zipf 1.1 --steps 200 --warmup 20"; do
    env $envs python3 $R/bench.py $pt --no-regions --no-cpu-baseline --sustained-steps 0 --streamed-steps 0 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.readline())
print('[$e] [$pt]: ms/step', j['ms_per_step'], {k: v['avg_us'] for k, v in j['kernels'].items()})" >> $OUT
  done
done
cat $OUT
