set -x
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_f16x2_edge_gpu.py -q -s > gpurun_out/t10_edge.log 2>&1; echo "edge tests rc=$?"
grep -E "^\[|passed|failed|Error" gpurun_out/t10_edge.log | tail -60
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "f16x2 and (wgrad or weight_gradient or halo or pointwise)" > gpurun_out/t10_k.log 2>&1; echo "kernel tests rc=$?"
tail -4 gpurun_out/t10_k.log
timeout -k 10 900 python -m pytest tests/test_block_model_gpu.py tests/test_grad_parity_gpu.py -x -q -k "f16x2" > gpurun_out/t10_m.log 2>&1; echo "model tests f16x2 rc=$?"
tail -4 gpurun_out/t10_m.log
B="python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-f32-mode --no-kernel-timing"
for rep in 1 2; do
  for m in bf16x3 f16x2; do
    for clips in 64 8; do
      steps=10; [ $clips = 8 ] && steps=30
      r=$($B --math $m --batch $clips --steps $steps 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['loss'])" 2>&1)
      echo "$m clips=$clips : $r" >> gpurun_out/t10_ab.log
    done
  done
done
cat gpurun_out/t10_ab.log
