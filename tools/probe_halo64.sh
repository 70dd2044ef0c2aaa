#!/bin/bash
# Timing probes of the 9-tap halo conv at 64 channels (FGCN_PROBE_HALO bits, fgcn_tconv.hip): tools/build_probe.py halo<bits> ... first
out=gpurun_out/probe_halo64.txt; : > $out
for b in 0 "$@"; do
    if [ $b = 0 ]; then lib=""; else lib="FGCN_LIB=$PWD/tools/probes/libfgcn_halo$b.so"; fi
    echo "== FGCN_PROBE_HALO=$b" >> $out
    env $lib python3 tools/kbench.py --math bf16x3 --only tconv --b 128 2>/dev/null | grep "s1 C64\|s1 C128" >> $out
done
cat $out
