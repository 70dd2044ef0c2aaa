#!/bin/bash
# The end-of-round records in two gpurun calls (run on the GPU box from the repo root; copy what is to be judged into profiles/):
#   tools/final_records.sh <tag> a   -> gpurun_out/final_<tag>/: the driver's bench line, other configs, shards, warm-step kernel tables (64 / 8 clips)
#   tools/final_records.sh <tag> b   -> per-kernel HBM traffic of the step, dominant-kernel stats, the bf16 step's tables
set -e
tag=${1:-r06}; part=${2:-a}
out=gpurun_out/final_$tag
mkdir -p $out
export TMPDIR=/tmp
Q="--no-cpu-baseline --no-f32-mode --no-f16x2-mode --no-bf16-mode --no-kernel-timing --no-inference --no-live-traffic"
if [ "$part" = "a" ]; then
    python3 bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err
    echo "bench: $(python3 -c "import json; d=json.load(open('$out/bench.json')); print(d['value'], d['ms_per_step'])")"
    : > $out/bench_other_configs.jsonl
    python3 bench.py --steps 20 --warmup 5 --joints 27 $Q >> $out/bench_other_configs.jsonl 2>> $out/bench.err
    python3 bench.py --steps 20 --warmup 5 --joints 22 $Q >> $out/bench_other_configs.jsonl 2>> $out/bench.err
    python3 bench.py --steps 20 --warmup 5 --math bf16 $Q >> $out/bench_other_configs.jsonl 2>> $out/bench.err
    tools/shards.sh > $out/shards.log 2>> $out/bench.err
    cat $out/shards.log
    : > $out/inference.jsonl
    python3 tools/infer_bench.py >> $out/inference.jsonl 2>> $out/bench.err
    python3 tools/infer_bench.py --math bf16 >> $out/inference.jsonl 2>> $out/bench.err
    python3 tools/infer_bench.py --batch 8 >> $out/inference.jsonl 2>> $out/bench.err
    cat $out/inference.jsonl
    tools/prof_step.sh ${tag}_tmp > $out/prof_step.log 2>&1
    cp gpurun_out/prof_${tag}_tmp/step_warm_kernel_stats.csv $out/bf16x3_step_warm_kernel_stats.csv
    cp gpurun_out/prof_${tag}_tmp/step8_warm_kernel_stats.csv $out/bf16x3_8clips_step_warm_kernel_stats.csv
    tail -2 $out/prof_step.log
else
    SER="--steps 2 --warmup 1 $Q --no-graph"
    for m in bf16x3 bf16; do
        rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_$m -o step_fetch -- python3 bench.py $SER --math $m > $out/step_fetch_$m.log 2>&1
        rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_$m -o step_write -- python3 bench.py $SER --math $m > $out/step_write_$m.log 2>&1
        python3 tools/step_traffic.py $(find $out/pmc_$m -name "step_fetch_counter_collection.csv") $(find $out/pmc_$m -name "step_write_counter_collection.csv") 3 \
            "HBM-side traffic per launch of every kernel of the 64-clip step ($tag, --math $m): rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (two runs) -- python3 bench.py --steps 2 --warmup 1 --no-graph" > $out/${m}_step_traffic_by_kernel.txt
        tail -3 $out/${m}_step_traffic_by_kernel.txt
        rm -rf $out/pmc_$m
    done
    rocprofv3 --kernel-trace --stats --output-format csv -d $out/dom -o dominant -- python3 bench.py --kernel-only > $out/dominant_kernel_live.json 2> $out/dominant.err
    cp $(find $out/dom -name "dominant_kernel_stats.csv") $out/bf16x3_dominant_kernel_stats.csv
    rm -rf $out/dom
    tools/prof_step.sh ${tag}_bf16_tmp --math bf16 > $out/prof_step_bf16.log 2>&1
    cp gpurun_out/prof_${tag}_bf16_tmp/step_warm_kernel_stats.csv $out/bf16_step_warm_kernel_stats.csv
    tail -2 $out/prof_step_bf16.log
fi
ls $out
