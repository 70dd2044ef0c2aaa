cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r13
python bench.py --steps 10 --warmup 3 > gpurun_out/r13/bench.log 2>&1
tail -1 gpurun_out/r13/bench.log > gpurun_out/r13/bench.json
python tools/kbench.py > gpurun_out/r13/kbench.log 2>&1
rocprofv3 --kernel-trace --stats -d gpurun_out/r13/prof -o p --output-format csv -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-graph > gpurun_out/r13/prof.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d gpurun_out/r13/pmc -o c --output-format csv -- python3 tools/kbench.py --only tconv,wgrad,gemm,spatial --reps 2 > gpurun_out/r13/pmc.log 2>&1
ls gpurun_out/r13 gpurun_out/r13/prof gpurun_out/r13/pmc
cut -c1-400 gpurun_out/r13/bench.json
