set -x
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/t3_all.log 2>&1; echo "gpu tests rc=$?"
tail -6 gpurun_out/t3_all.log
