cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r10
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/r10/test.log 2>&1
tail -3 gpurun_out/r10/test.log
timeout 600 python tools/kbench.py --only wgrad,joint > gpurun_out/r10/kbench.log 2>&1
grep "tconv_wgrad\|gram" gpurun_out/r10/kbench.log
python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r10/bench.log 2>&1
tail -1 gpurun_out/r10/bench.log | cut -c1-200
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --batch 8 --no-kernel-timing > gpurun_out/r10/bench8.log 2>&1
tail -1 gpurun_out/r10/bench8.log | cut -c1-200
