cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
python -m pytest tests -x -q -m gpu > gpurun_out/r4/test.log 2>&1
tail -5 gpurun_out/r4/test.log
python tools/kbench.py --only gemm,tconv,spatial,joint > gpurun_out/r4/kbench.log 2>&1
python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r4/bench.log 2>&1
tail -1 gpurun_out/r4/bench.log | cut -c1-400
