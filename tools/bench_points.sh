#!/bin/bash
# Run on the GPU box (via gpurun): the measurement points SURVEY 8(d) asks for besides the default bench line, one
# JSON line each, into gpurun_out/points_$1.jsonl  (copy the file to profiles/ to have it judged).
#   cfg2 (ML-100K shape) | cfg3 at B = 1,048,576 rows | cfg3 with Zipf(1.1) item popularity | the same with the rows in the
#   data files' order (sorted by user: a batch = ~900 users' consecutive ratings, 8 % of the table) | cfg5 shape on one GPU
#   (lazy exact Adam: auto) and the same with the dense step | the software-pipelined step on / off where "auto"
#   decides either way | cfg3 forward A/B: k_fwd (every occurrence sampled), k_fwd2 with table eps, k_fwd2 with the
#   RNG compiled out, rows not sorted | cfg3 without the look-ahead lazy Adam form (every row every step) | cfg2 with the
#   three-launch backward (VFM_BWD_SMALL=0) | cfg3 with every plan built inside the timed region (--plans stream)
# usage: tools/bench_points.sh r04
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/points_$TAG.jsonl
mkdir -p $R/gpurun_out
: > $OUT
run() {  # label, env assignments (may be empty), bench args...
  local label=$1; shift
  local envs=$1; shift
  echo "[points] $label" >&2
  env $envs python3 $R/bench.py "$@" --no-cpu-baseline 2>> $R/gpurun_out/points_$TAG.err | python3 -c "
import json,sys
j=json.loads(sys.stdin.readline()); j['point']='$label'; print(json.dumps(j))" >> $OUT || echo "{\"point\": \"$label\", \"error\": true}" >> $OUT
}
COMMON="--sustained-steps 1000"
run cfg2_ml100k_d20 "" --workload ml100k_d20 --steps 300 --warmup 30 $COMMON
run cfg2_ml100k_d20_three_launch_step "VFM_BWD_SMALL=0" --workload ml100k_d20 --steps 300 --warmup 30 $COMMON
run cfg3_plans_streamed "" --plans stream --steps 400 --warmup 20 --no-regions --sustained-steps 0 --streamed-steps 0
run cfg3_no_packed_records "" --no-wrec --steps 200 --warmup 20 --no-regions $COMMON
run cfg5_criteo_d256_k_fwd "VFM_FWD_KERNEL=1" --workload criteo_d256 --steps 200 --warmup 20 --sustained-steps 0
run cfg3_B1048576 "" --batch 1048576 --n-batches 4 --steps 60 --warmup 6 --sustained-steps 0
run cfg3_B1048576_plain_step "" --batch 1048576 --n-batches 4 --steps 60 --warmup 6 --pipeline off --sustained-steps 0
run cfg3_B100000_pipelined_step "" --pipeline on --steps 200 --warmup 20 --no-regions --sustained-steps 0
run cfg2_ml100k_d20_pipelined_step "" --workload ml100k_d20 --steps 300 --warmup 30 --pipeline on --sustained-steps 0
run cfg3_zipf1.1 "" --zipf 1.1 --steps 200 --warmup 20 --sustained-steps 0
run cfg3_data_file_order_auto "" --user-order --zipf 1.1 --n-batches 160 --steps 320 --warmup 20 --no-regions --sustained-steps 1600
run cfg3_data_file_order_pipelined_step "" --user-order --zipf 1.1 --n-batches 160 --steps 320 --warmup 20 --no-regions --pipeline on --sustained-steps 0
run cfg5_criteo_d256_auto "" --workload criteo_d256 --steps 200 --warmup 20 $COMMON
run cfg5_criteo_d256_row_list_form "" --workload criteo_d256 --lazy-adam on --steps 200 --warmup 20 --sustained-steps 0
run cfg3_B5000_auto "" --batch 5000 --steps 300 --warmup 20 --no-regions $COMMON
run cfg5_criteo_d256_dense "" --workload criteo_d256 --lazy-adam off --lookahead off --steps 40 --warmup 4 --sustained-steps 0
run cfg3_lookahead_off "" --lookahead off --steps 200 --warmup 20 --no-regions --sustained-steps 0
run cfg3_fwd_ab_k_fwd "VFM_FWD_KERNEL=1" --steps 200 --warmup 20 --no-regions --sustained-steps 0
run cfg3_fwd_ab_k_fwd2 "" --steps 200 --warmup 20 --no-regions --sustained-steps 0
run cfg3_fwd_ab_k_fwd2_table_eps "" --fwd-eps table --steps 200 --warmup 20 --no-regions --sustained-steps 0
run cfg3_fwd_ab_k_fwd2_no_rng "VFM_FWD_AB_NORNG=1" --steps 200 --warmup 20 --no-regions --sustained-steps 0
run cfg3_unsorted_rows "" --no-sort --steps 200 --warmup 20 --no-regions --sustained-steps 0
python3 - <<PY
import json
for l in open("$OUT"):
    j = json.loads(l)
    if j.get("error"): print(j); continue
    print(j["point"], "ms/step", j["ms_per_step"], "value", j["value"], {k: v["avg_us"] for k, v in j["kernels"].items()},
          "sustained", (j.get("sustained") or {}).get("ms_per_step"), "streamed", j.get("ms_per_step_streamed"))
PY
