#!/bin/bash
# The rocprofv3 records the DESIGN / bench numbers are checked against (run on the GPU box from the repo root):
#   tools/collect_profiles.sh <tag>        -> gpurun_out/prof_<tag>/...; copy the summaries you want judged into profiles/
# Counters (--pmc) run in passes of their own, never combined with trace domains other than --kernel-trace.
set -e
tag=${1:-r02}
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
# 1. whole step, every kernel on one stream (true per-kernel durations)
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o step_serial -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline \
    --no-f32-mode --no-f16x2-mode --no-kernel-timing --no-graph --wgrad-stream main > $out/step_serial.log 2>&1
# 2. the dominant kernel alone (22 launches at its dominant shape)
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o dominant -- python3 bench.py --kernel-only > $out/dominant.log 2>&1
# 3. HBM-side traffic of the dominant kernel: FETCH_SIZE and WRITE_SIZE in separate passes
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out -o pmc_fetch -- python3 bench.py --kernel-only > $out/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out -o pmc_write -- python3 bench.py --kernel-only > $out/pmc_write.log 2>&1
# 4. matrix-pipe occupancy
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out -o pmc_mfma -- python3 bench.py --kernel-only > $out/pmc_mfma.log 2>&1
ls -la $out
