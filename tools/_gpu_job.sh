set -x
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "pointwise or halo" > gpurun_out/t6_pw.log 2>&1; echo "pw tests rc=$?"
tail -5 gpurun_out/t6_pw.log
timeout -k 10 600 python tools/kbench.py --only pw --math bf16x3 > gpurun_out/t6_kbench_pw.log 2>&1; echo "kbench rc=$?"
cat gpurun_out/t6_kbench_pw.log | tail -80
timeout -k 10 900 python -m pytest tests/test_block_model_gpu.py tests/test_packing_gpu.py tests/test_grad_parity_gpu.py -x -q > gpurun_out/t6_model.log 2>&1; echo "model tests rc=$?"
tail -5 gpurun_out/t6_model.log
B="python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-f32-mode --no-kernel-timing"
for rep in 1 2; do
  for mk in 100000 64 128 192; do
    for clips in 64 8; do
      steps=10; [ $clips = 8 ] && steps=30
      r=$(FGCN_PW_MIN_K=$mk $B --batch $clips --steps $steps 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['loss'])" 2>&1)
      echo "pw_min_k=$mk clips=$clips : $r" >> gpurun_out/t6_ab_pw.log
    done
  done
done
cat gpurun_out/t6_ab_pw.log
