#!/bin/bash
# Timing probes of the tile-form conv_d weight gradient (FGCN_PROBE_SW bits, fgcn_spatial_wgrad_tile.hip): tools/build_probe.py sw<bits> ... first
out=gpurun_out/probe_swt.txt; : > $out
for b in 0 "$@"; do
    if [ $b = 0 ]; then lib=""; else lib="FGCN_LIB=$PWD/tools/probes/libfgcn_sw$b.so"; fi
    echo "== FGCN_PROBE_SW=$b" >> $out
    env $lib python3 tools/kbench.py --math bf16x3 --only spatial_wgrad --b 128 2>/dev/null | grep -A1 "spatial_wgrad_tile" | grep -v "^--" >> $out
done
cat $out
