#!/bin/bash
# Same-box A/B of the whole step between builds of libfgcn: tools/ab_libs.sh <out.log> <reps> base=<path|-> name=<path> ...
# ("-" = the tree's own build).  Every build runs the 64-clip and the 8-clip step, interleaved, <reps> times.
out=$1; reps=$2; shift 2
mkdir -p $(dirname $out); : > $out
B="python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-f32-mode --no-f16x2-mode --no-bf16-mode --no-kernel-timing --no-inference"
for rep in $(seq $reps); do
  for spec in "$@"; do
    name=${spec%%=*}; lib=${spec#*=}
    if [ "$lib" = "-" ]; then unset FGCN_LIB; else export FGCN_LIB=$PWD/$lib; fi
    for clips in 64 8; do
      steps=10; [ $clips = 8 ] && steps=30
      r=$($B --batch $clips --steps $steps 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['loss'])" 2>&1)
      echo "$name $clips : $r" >> $out
    done
  done
done
unset FGCN_LIB
cat $out
