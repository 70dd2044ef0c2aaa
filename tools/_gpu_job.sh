set -x
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/t2_all.log 2>&1; echo "gpu tests rc=$?"
tail -4 gpurun_out/t2_all.log
timeout -k 10 300 python tools/probes/copy_sources_probe.py 8 > gpurun_out/t2_copies.log 2>&1; echo "probe rc=$?"
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > gpurun_out/t2_bench.json 2> gpurun_out/t2_bench.err; echo "bench rc=$?"
tail -3 gpurun_out/t2_bench.err; cut -c1-400 gpurun_out/t2_bench.json
timeout -k 10 300 python bench.py --batch 8 --steps 30 --warmup 5 --no-cpu-baseline --no-f32-mode --no-kernel-timing > gpurun_out/t2_bench8.json 2>> gpurun_out/t2_bench.err; echo "bench8 rc=$?"
cut -c1-300 gpurun_out/t2_bench8.json
