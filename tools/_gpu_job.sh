set -x
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_f16x2_edge_gpu.py -q -s > gpurun_out/t14_edge.log 2>&1; echo "edge tests rc=$?"
grep -E "spatial|passed|failed|^FAILED|^E  " gpurun_out/t14_edge.log | tail -20
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_packing_gpu.py -x -q -k "f16x2" > gpurun_out/t14_k.log 2>&1; echo "kernel tests f16x2 rc=$?"
tail -4 gpurun_out/t14_k.log
timeout -k 10 900 python -m pytest tests/test_block_model_gpu.py tests/test_grad_parity_gpu.py -x -q -k "f16x2" > gpurun_out/t14_m.log 2>&1; echo "model tests f16x2 rc=$?"
tail -4 gpurun_out/t14_m.log
timeout -k 10 300 python tools/kbench.py --only spatial --math f16x2 > gpurun_out/t14_kb_f16x2.log 2>&1; grep -i "spatial" gpurun_out/t14_kb_f16x2.log | head
timeout -k 10 300 python tools/kbench.py --only spatial --math bf16x3 > gpurun_out/t14_kb_bf16x3.log 2>&1; grep -i "spatial" gpurun_out/t14_kb_bf16x3.log | head
B="python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-f32-mode --no-f16x2-mode --no-kernel-timing"
for rep in 1 2; do
  for m in bf16x3 f16x2; do
    for clips in 64 8; do
      steps=10; [ $clips = 8 ] && steps=30
      r=$($B --math $m --batch $clips --steps $steps 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['loss'])" 2>&1)
      echo "$m clips=$clips : $r" >> gpurun_out/t14_ab.log
    done
  done
done
cat gpurun_out/t14_ab.log
