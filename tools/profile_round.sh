#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel trace + stats and the HBM PMC passes of the bench
# command; leaves raw output under gpurun_out/prof_$1/ and a compact summary for profiles/.
# usage: tools/profile_round.sh r01
set -o pipefail
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
(cd $R && python3 -c "from vae_amd.build import sources_digest; print(sources_digest())") > $OUT/csrc_sha1.txt     # what is being profiled
cd /tmp
ARGS="--steps 100 --warmup 10 --no-cpu-baseline --sustained-steps 0 --streamed-steps 0 --event-every 1 ${EXTRA_ARGS:-}"     # e.g. EXTRA_ARGS="--workload ml20m_d16 --n-batches 4"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py $ARGS > $OUT/trace.json 2> $OUT/trace.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py $ARGS --no-events > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py $ARGS --no-events > $OUT/pmc_write.json 2> $OUT/pmc_write.err
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/pmc_sq -- python3 $R/bench.py $ARGS --no-events > $OUT/pmc_sq.json 2> $OUT/pmc_sq.err
python3 $R/tools/pmc_summary.py --json $OUT/summary.json $OUT/trace $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq > $OUT/summary.txt
cp $(find $OUT/trace -name '*kernel_stats.csv' | tail -1) $OUT/kernel_stats.csv
# the per-dispatch CSVs of four passes are 10-40 MB per point (gpurun merges at most 64 MiB per call): summarised above,
# dropped here unless KEEP_RAW=1
[ "${KEEP_RAW:-0}" = "1" ] || rm -rf $OUT/trace $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq
cat $OUT/summary.txt | head -60
