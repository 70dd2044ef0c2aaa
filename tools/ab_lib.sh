#!/bin/bash
# Same-call A/B of one kbench family between the in-tree library and probe libraries (tools/build_probe.py):
#     tools/ab_lib.sh "<kbench args>" "<grep pattern>" <probe name> ...
args="$1"; pat="$2"; shift 2
for rep in 1 2; do
for n in - "$@"; do
    if [ $n = - ]; then lib=""; else lib="FGCN_LIB=$PWD/tools/probes/libfgcn_$n.so"; fi
    echo "== $n"
    env $lib python3 tools/kbench.py $args 2>/dev/null | grep "$pat"
done
done
