#!/bin/bash
# A/B on the GPU box: does the plan build on the side stream overlap the step when the main kernels leave block slots free?
# usage: tools/overlap_ab.sh OUTFILE
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=${1:-$R/gpurun_out/overlap_ab.txt}
: > $OUT
run() {
  echo "== $1" >> $OUT
  env $1 python3 $R/bench.py --plans stream --steps 300 --warmup 20 --no-cpu-baseline --sustained-steps 300 --streamed-steps 0 --no-regions --settle-steps 0 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.readline())
print('streamed ms/step', j['ms_per_step'], 'host', j['host_enqueue_ms_per_step'], '| resident sustained', j['sustained']['ms_per_step'], '| kernels', {k: v['avg_us'] for k, v in j['kernels'].items()})" >> $OUT
}
run "VFM_PLAN_PRIORITY=0"
run "VFM_PLAN_PRIORITY=-1"
run "VFM_PLAN_PRIORITY=-1 VFM_BWD_GRID=1024"
run "VFM_PLAN_PRIORITY=-1 VFM_BWD_GRID=960 VFM_FWD2_GRID=960"
run "VFM_PLAN_PRIORITY=-1 VFM_BWD_GRID=896 VFM_FWD2_GRID=896"
run "VFM_PLAN_PRIORITY=-1 VFM_BWD_GRID=768 VFM_FWD2_GRID=768"
run "VFM_PLAN_PRIORITY=-1 VFM_BWD_GRID=768"
cat $OUT
