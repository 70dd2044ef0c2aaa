#!/bin/bash
# Step time of the strong-scaling shards on ONE GPU (bench.py --batch n, HIP-graph replay), both float32-class arithmetics:
#   tools/shards.sh > gpurun_out/shards.log
for m in bf16x3 f16x2; do
for n in 64 32 16 8; do
    python3 bench.py --steps 20 --warmup 5 --batch $n --math $m --no-cpu-baseline --no-f32-mode --no-f16x2-mode --no-bf16-mode --no-kernel-timing --no-inference 2>/dev/null \
        | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$m', $n, d['ms_per_step'], d['value'])"
done
done
