#!/bin/bash
# kernel trace of the detect path with N batches in flight + a summary of how well the kernels overlap:  INFLIGHT=4 tools/trace_overlap.sh <tag>
tag=${1:-x}; R=$(pwd); export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/ov_${tag} -o ov -- python $R/bench.py --no-measure-traffic --steps 24 --warmup 2 --steady-steps 0 --no-cpu-baseline --ba none --no-edge --rp-frames 0 --latency-calls 0 --lines-images 0 --image-in-steps 0 --inflight ${INFLIGHT:-3} > $R/gpurun_out/ov_${tag}.log 2>&1
cd $R
python tools/trace_overlap.py $(find gpurun_out/ov_${tag} -name '*kernel_trace.csv' | head -1)
find gpurun_out/ov_${tag} -name '*kernel_trace.csv' -size +30M -delete
