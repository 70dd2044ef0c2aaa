#!/bin/bash
# Same-box A/B of two library builds per kernel family (tools/kbench.py): tools/ab_kernels.sh "<kbench args>"   (baseline = tools/probes/libfgcn_alt.so via FGCN_LIB)
mkdir -p gpurun_out/ab
for rep in 1 2; do
  echo "== baseline (rep $rep)" >> gpurun_out/ab/kab.log
  FGCN_LIB=$PWD/tools/probes/libfgcn_alt.so python3 tools/kbench.py $1 2>/dev/null | grep -v "^--\|amdgpu" >> gpurun_out/ab/kab.log || exit 1
  echo "== new (rep $rep)" >> gpurun_out/ab/kab.log
  python3 tools/kbench.py $1 2>/dev/null | grep -v "^--\|amdgpu" >> gpurun_out/ab/kab.log || exit 1
done
