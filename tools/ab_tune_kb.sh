#!/bin/bash
# Same-call A/B of one kbench family under different fgcn_set_tuning settings:
#     tools/ab_tune_kb.sh "<kbench args>" "<grep pattern>" "<tune A or ->" "<tune B>" ...
args="$1"; pat="$2"; shift 2
for rep in 1 2; do
for t in "$@"; do
    echo "== $t"
    if [ "$t" = - ]; then tt=""; else tt="--tune $t"; fi
    python3 tools/kbench.py $args $tt 2>/dev/null | grep "$pat"
done
done
