#!/bin/bash
# Timing probes of the embedding-backward tile kernels (FGCN_PROBE_EMB bits, fgcn_emb_tile.hip): tools/build_probe.py emb<bits> ... first
#   tools/probe_emb.sh [-t "18=1"] <bits> ...
tune=""
if [ "$1" = "-t" ]; then tune="--tune $2"; shift 2; fi
out=gpurun_out/probe_emb.txt; : > $out
for b in 0 "$@"; do
    if [ $b = 0 ]; then lib=""; else lib="FGCN_LIB=$PWD/tools/probes/libfgcn_emb$b.so"; fi
    echo "== FGCN_PROBE_EMB=$b" >> $out
    env $lib python3 tools/kbench.py --math bf16x3 --only emb_bwd --b 128 $tune 2>/dev/null | grep "^emb_" >> $out
done
cat $out
