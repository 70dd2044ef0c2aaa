#!/bin/bash
# Timing probes of the fused spatial backward (FGCN_PROBE_SB bits, fgcn_spatial_bwd_tile.hip): tools/build_probe.py sb<bits> ... first
out=gpurun_out/probe_sbwd.txt; : > $out
for b in 0 4 7 16 24 64 127 128 144; do
    if [ $b = 0 ]; then lib=""; else lib="FGCN_LIB=$PWD/tools/probes/libfgcn_sb$b.so"; fi
    echo "== FGCN_PROBE_SB=$b" >> $out
    env $lib python3 tools/kbench.py --math bf16x3 --only spatial_bwd 2>/dev/null | grep fused >> $out
done
cat $out
