#!/bin/bash
# HBM-side traffic of every kernel of the step in one math mode (two rocprofv3 counter passes; run on the GPU box from the repo root):
#   tools/traffic_step.sh <tag> <math mode> [more bench flags]   -> gpurun_out/traffic_<tag>/<mode>_step_traffic_by_kernel.txt
set -e
tag=${1:-r06}; m=${2:-bf16x3}; shift 2 || true
out=gpurun_out/traffic_$tag
mkdir -p $out
export TMPDIR=/tmp
SER="--steps 2 --warmup 1 --no-cpu-baseline --no-f32-mode --no-f16x2-mode --no-bf16-mode --no-kernel-timing --no-inference --no-live-traffic --no-graph"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_$m -o step_fetch -- python3 bench.py $SER --math $m "$@" > $out/step_fetch_$m.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_$m -o step_write -- python3 bench.py $SER --math $m "$@" > $out/step_write_$m.log 2>&1
python3 tools/step_traffic.py $(find $out/pmc_$m -name "step_fetch_counter_collection.csv") $(find $out/pmc_$m -name "step_write_counter_collection.csv") 3 \
    "HBM-side traffic per launch of every kernel of the 64-clip step ($tag, --math $m $*): rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (two runs) -- python3 bench.py --steps 2 --warmup 1 --no-graph" > $out/${m}_step_traffic_by_kernel.txt
rm -rf $out/pmc_$m
tail -1 $out/${m}_step_traffic_by_kernel.txt
