#!/bin/bash
# A/B on the GPU box: how the plan build on the side stream shares the chip with the step's kernels.
# usage: tools/overlap_ab.sh OUTFILE "ENV1" "ENV2" ...      (each ENV: space-separated assignments; "-" = none)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=${1:-$R/gpurun_out/overlap_ab.txt}; shift
: > $OUT
run() {
  echo "== $1" >> $OUT
  local envs=$1; [ "$envs" = "-" ] && envs=""
  env $envs python3 $R/bench.py --plans stream --steps 300 --warmup 20 --no-cpu-baseline --sustained-steps 300 --streamed-steps 0 --no-regions --settle-steps 0 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.readline())
print('streamed ms/step', j['ms_per_step'], 'host', j['host_enqueue_ms_per_step'], '| resident sustained', j['sustained']['ms_per_step'], '| kernels', {k: v['avg_us'] for k, v in j['kernels'].items()})" >> $OUT
}
for e in "$@"; do run "$e"; done
cat $OUT
