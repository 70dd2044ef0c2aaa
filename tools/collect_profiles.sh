#!/bin/bash
# The rocprofv3 records the DESIGN / bench numbers are checked against (run on the GPU box from the repo root):
#   tools/collect_profiles.sh <tag> [bench args, e.g. --math f16x2]   -> gpurun_out/prof_<tag>/...
# copy the summaries you want judged into profiles/.  Counters (--pmc) run in passes of their own, never combined with trace
# domains other than --kernel-trace.
set -e
tag=${1:-r03}; shift || true
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
SER="--steps 3 --warmup 1 --no-cpu-baseline --no-f32-mode --no-f16x2-mode --no-bf16-mode --no-kernel-timing --no-inference --no-graph"
# 1. whole step, every kernel on one stream (true per-kernel durations), 64 clips and the 8-clip shard
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o step_serial -- python3 bench.py $SER "$@" > $out/step_serial.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o step8_serial -- python3 bench.py $SER --batch 8 "$@" > $out/step8_serial.log 2>&1
# 2. the dominant kernel alone (22 launches at its dominant shape)
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o dominant -- python3 bench.py --kernel-only "$@" > $out/dominant.log 2>&1
# 3. HBM-side traffic of the dominant kernel: FETCH_SIZE and WRITE_SIZE in separate passes
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out -o pmc_fetch -- python3 bench.py --kernel-only "$@" > $out/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out -o pmc_write -- python3 bench.py --kernel-only "$@" > $out/pmc_write.log 2>&1
# 4. matrix-pipe occupancy
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out -o pmc_mfma -- python3 bench.py --kernel-only "$@" > $out/pmc_mfma.log 2>&1
# 5. HBM-side traffic of the WHOLE step, per kernel (tools/step_traffic.py)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out -o step_fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-f32-mode --no-f16x2-mode --no-bf16-mode --no-kernel-timing --no-inference --no-graph "$@" > $out/step_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out -o step_write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-f32-mode --no-f16x2-mode --no-bf16-mode --no-kernel-timing --no-inference --no-graph "$@" > $out/step_write.log 2>&1
python3 tools/step_traffic.py $(find $out -name "step_fetch_counter_collection.csv") $(find $out -name "step_write_counter_collection.csv") 3 \
    "HBM-side traffic per launch of every kernel of the 64-clip step ($tag $*): rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (two runs) -- python3 bench.py --steps 2 --warmup 1 --no-graph" > $out/step_traffic_by_kernel.txt
find $out -name "*.csv" | head -40
