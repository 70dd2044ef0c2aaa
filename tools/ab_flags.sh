#!/bin/bash
# Same-box A/B of bench.py under different FLAGS (run on the GPU box from the repo root):
#     tools/ab_flags.sh [-f "<common bench flags>"] "--tune 23=1" "" ...        ("" = no extra flags)
# Boxes of the pool differ by +-2-3 %, so only numbers from ONE call compare.  Output: gpurun_out/ab/ab_flags.log
FLAGS=""
if [ "$1" = "-f" ]; then FLAGS="$2"; shift 2; fi
mkdir -p gpurun_out/ab; rm -f gpurun_out/ab/ab_flags.log
B="python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-f32-mode --no-f16x2-mode --no-bf16-mode --no-kernel-timing --no-inference --no-live-traffic $FLAGS"
for rep in 1 2; do
for t in "$@"; do
    echo "== [$t]" >> gpurun_out/ab/ab_flags.log
    $B $t 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['loss'])" >> gpurun_out/ab/ab_flags.log 2>&1 || exit 1
done
done
cat gpurun_out/ab/ab_flags.log
