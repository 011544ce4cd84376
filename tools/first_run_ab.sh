#!/bin/bash
# Is the FIRST bench.py process of a gpurun call slower than the ones after it, and does it depend on the HIP events inside the
# timed region?  usage: tools/first_run_ab.sh OUT "ARGS of the first run" ["ARGS of later runs" ...]
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$1; shift
: > $OUT
i=0
for a in "$@"; do
  i=$((i + 1))
  python3 $R/bench.py --no-cpu-baseline --sustained-steps 500 --streamed-steps 0 --no-regions $a 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.readline())
print('run $i [$a]: ms/step', j['ms_per_step'], 'host', j['host_enqueue_ms_per_step'], 'cold', j['settle']['cold_ms_per_step'] if j.get('settle') else None, 'sustained', j['sustained']['ms_per_step'], 'kernels', {k: v['avg_us'] for k, v in (j.get('kernels') or {}).items()})" >> $OUT
done
cat $OUT
