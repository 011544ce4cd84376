set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/fin
bash tools/profile_round.sh r03 > gpurun_out/fin/prof_r03.log 2>&1
EXTRA_ARGS="--workload criteo_d256" bash tools/profile_round.sh r03cfg5 > gpurun_out/fin/prof_r03cfg5.log 2>&1
cd $R
python bench.py > gpurun_out/fin/bench_default.json 2> gpurun_out/fin/bench_default.err
bash tools/bench_points.sh r03 > gpurun_out/fin/points.log 2>&1
tools/rehearse_ranks.sh 2 gpurun_out/fin/rehearse_2.json
tools/rehearse_ranks.sh 4 gpurun_out/fin/rehearse_4.json
tools/rehearse_ranks.sh 5 gpurun_out/fin/rehearse_5.json --steps 5 --warmup 2 --mode-budget-s 40
EXCHANGE=stats python tools/shard_host_probe.py > gpurun_out/fin/probe_stats.txt 2>&1
EXCHANGE=grads python tools/shard_host_probe.py > gpurun_out/fin/probe_grads.txt 2>&1
python tools/call_overhead_probe.py > gpurun_out/fin/call_overhead.txt 2>&1
tail -3 gpurun_out/fin/points.log
