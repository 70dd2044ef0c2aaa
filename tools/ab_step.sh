#!/bin/bash
# Same-box A/B of the whole step between the baseline build (tools/probes/libfgcn_alt.so via FGCN_LIB) and the tree's build:
#   tools/ab_step.sh "<bench args>" ...
mkdir -p gpurun_out/ab; rm -f gpurun_out/ab/abl.log
B="python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-f32-mode --no-f16x2-mode --no-bf16-mode --no-kernel-timing --no-inference"
for rep in 1 2; do
for t in "$@"; do
  for which in base new; do
    echo "== $which $t" >> gpurun_out/ab/abl.log
    if [ $which = base ]; then export FGCN_LIB=$PWD/tools/probes/libfgcn_alt.so; else unset FGCN_LIB; fi
    $B $t 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['loss'])" >> gpurun_out/ab/abl.log 2>&1 || exit 1
  done
done
done
unset FGCN_LIB
cat gpurun_out/ab/abl.log
