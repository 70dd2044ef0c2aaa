cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r8
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/r8/test.log 2>&1
tail -3 gpurun_out/r8/test.log
timeout 600 python tools/kbench.py --only wgrad > gpurun_out/r8/kbench.log 2>&1
grep tconv gpurun_out/r8/kbench.log
python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r8/bench.log 2>&1
tail -1 gpurun_out/r8/bench.log | cut -c1-200
