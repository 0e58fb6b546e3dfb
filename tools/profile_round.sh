#!/bin/bash
# Profiling recipe of a round (run on the GPU box through gpurun):  tools/profile_round.sh <tag>
# 1. bench.py with its defaults, as the driver runs it (full: detect + BA + CPU baseline + the traffic passes) -> gpurun_out/bench_<tag>.json
# 2. rocprofv3 --kernel-trace --stats of the same command -> gpurun_out/prof_<tag>/
# 3. two PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, kernel-trace only) -> gpurun_out/pmc_<tag>_{fetch,write}/   (SKIP_PMC=1 leaves them out)
tag=${1:-rX}
R=$(pwd)
export TMPDIR=/tmp
mkdir -p gpurun_out
python bench.py > gpurun_out/bench_${tag}.json 2> gpurun_out/bench_${tag}.err
tail -c 3000 gpurun_out/bench_${tag}.json
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${tag} -o prof -- python $R/bench.py --no-measure-traffic --steps 10 --warmup 2 --no-cpu-baseline --rp-frames 0 --latency-calls 0 --lines-images 0 > $R/gpurun_out/bench_prof_${tag}.log 2>&1
if [ -z "$SKIP_PMC" ]; then
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_${tag}_fetch -o pmc -- python $R/bench.py --no-measure-traffic --steps 3 --warmup 1 --no-cpu-baseline --ba-iters 3 --inflight 1 --rp-frames 0 --latency-calls 0 --lines-images 0 > $R/gpurun_out/pmc_${tag}_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_${tag}_write -o pmc -- python $R/bench.py --no-measure-traffic --steps 3 --warmup 1 --no-cpu-baseline --ba-iters 3 --inflight 1 --rp-frames 0 --latency-calls 0 --lines-images 0 > $R/gpurun_out/pmc_${tag}_write.log 2>&1
cd $R
python tools/pmc_summary.py gpurun_out/pmc_${tag}_fetch gpurun_out/pmc_${tag}_write gpurun_out/pmc_${tag}_traffic.json > gpurun_out/pmc_${tag}_summary.csv
cat gpurun_out/pmc_${tag}_summary.csv | head -40
# the raw counter dumps are large; keep the summaries
find gpurun_out/pmc_${tag}_fetch gpurun_out/pmc_${tag}_write -name '*counter_collection.csv' -size +20M -delete
fi
cd $R
find gpurun_out/prof_${tag} -name '*kernel_trace.csv' -size +20M -delete
