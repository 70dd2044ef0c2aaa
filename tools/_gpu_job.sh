set -x
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_f16x2_edge_gpu.py -q -s > gpurun_out/t11_edge.log 2>&1; echo "edge tests rc=$?"
grep -E "passed|failed|^FAILED" gpurun_out/t11_edge.log | tail -8
timeout -k 10 1100 python -m pytest tests -m gpu -q > gpurun_out/t11_all.log 2>&1; echo "gpu tests rc=$?"
tail -8 gpurun_out/t11_all.log
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > gpurun_out/t11_bench.json 2> gpurun_out/t11_bench.err; echo "bench rc=$?"
tail -2 gpurun_out/t11_bench.err; cut -c1-300 gpurun_out/t11_bench.json
