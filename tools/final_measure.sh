set -o pipefail
# Run on the GPU box (via gpurun, <= 1200 s per call: pass the stage): the evidence of a round, into gpurun_out/fin/.
#   tools/final_measure.sh r04 profiles1  kernel trace + PMC passes: cfg3 default, cfg2, cfg5
#   tools/final_measure.sh r04 profiles2  ... B = 1,048,576, data-file order, cfg5 row-list form
#   tools/final_measure.sh r04 points     bench_points.sh + the default bench line
#   tools/final_measure.sh r04 ranks      gloo rehearsals of `bench.py --gpus N` (2 and 5 ranks on the one card) + host probes
TAG=${1:-r04}; STAGE=${2:-profiles1}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/fin
case $STAGE in
profiles1)
  bash tools/profile_round.sh $TAG > gpurun_out/fin/prof_$TAG.log 2>&1
  EXTRA_ARGS="--workload ml100k_d20" bash tools/profile_round.sh ${TAG}cfg2 > gpurun_out/fin/prof_${TAG}cfg2.log 2>&1
  EXTRA_ARGS="--workload criteo_d256" bash tools/profile_round.sh ${TAG}cfg5 > gpurun_out/fin/prof_${TAG}cfg5.log 2>&1
  ;;
profiles2)
  EXTRA_ARGS="--batch 1048576 --n-batches 4 --steps 40 --warmup 6" bash tools/profile_round.sh ${TAG}b1m > gpurun_out/fin/prof_${TAG}b1m.log 2>&1
  EXTRA_ARGS="--user-order --zipf 1.1 --n-batches 160 --steps 160 --warmup 20 --no-regions" bash tools/profile_round.sh ${TAG}fileorder > gpurun_out/fin/prof_${TAG}fileorder.log 2>&1
  EXTRA_ARGS="--workload criteo_d256 --lazy-adam on" bash tools/profile_round.sh ${TAG}cfg5list > gpurun_out/fin/prof_${TAG}cfg5list.log 2>&1
  ;;
points)
  python bench.py > gpurun_out/fin/bench_default.json 2> gpurun_out/fin/bench_default.err
  bash tools/bench_points.sh $TAG > gpurun_out/fin/points.log 2>&1
  tail -3 gpurun_out/fin/points.log
  ;;
ranks)
  tools/rehearse_ranks.sh 2 gpurun_out/fin/rehearse_2.json
  tools/rehearse_ranks.sh 5 gpurun_out/fin/rehearse_5.json
  python tools/call_overhead_probe.py > gpurun_out/fin/call_overhead.txt 2>&1
  ;;
esac
