#!/bin/bash
# Serial (one stream, eager) kernel trace of the step, reduced to WARM steps (tools/warm_stats.py), 64 clips and the 8-clip shard:
#   tools/prof_step.sh <tag> [bench args]   -> gpurun_out/prof_<tag>/{step,step8}_warm_kernel_stats.csv
set -e
tag=${1:-r04}; shift || true
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
SER="--steps 3 --warmup 2 --no-cpu-baseline --no-f32-mode --no-f16x2-mode --no-bf16-mode --no-kernel-timing --no-inference --no-graph"
rocprofv3 --kernel-trace --output-format csv -d $out -o step_serial -- python3 bench.py $SER "$@" > $out/step_serial.log 2>&1
python3 tools/warm_stats.py $(find $out -name "step_serial_kernel_trace.csv") 3 > $out/step_warm_kernel_stats.csv
rocprofv3 --kernel-trace --output-format csv -d $out -o step8_serial -- python3 bench.py $SER --batch 8 "$@" > $out/step8_serial.log 2>&1
python3 tools/warm_stats.py $(find $out -name "step8_serial_kernel_trace.csv") 3 > $out/step8_warm_kernel_stats.csv
find $out -name "*_kernel_trace.csv" -delete
tail -1 $out/step_warm_kernel_stats.csv; tail -1 $out/step8_warm_kernel_stats.csv
