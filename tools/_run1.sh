cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r12
timeout 600 python tools/kbench.py --only gemm > gpurun_out/r12/a.log 2>&1
timeout 600 python tools/kbench.py --only gemm --tune 5=8 > gpurun_out/r12/b.log 2>&1
paste <(grep " ms " gpurun_out/r12/a.log | cut -c1-75) <(grep " ms " gpurun_out/r12/b.log | cut -c59-75)
