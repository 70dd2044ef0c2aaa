#!/bin/bash
# SQ counters + time of the fused spatial backward with its HBM-born operands pre-split (stand-in probes, FGCN_PROBE_SB bits 8 / 9):
#   tools/pmc_sbwd.sh   (after: for b in 256 512 768; do python tools/build_probe.py sb$b fgcn_spatial_bwd_tile.hip -DFGCN_PROBE_SB=$b; done)
out=gpurun_out/pmc_sbwd; mkdir -p $out; export TMPDIR=/tmp
: > $out/summary.txt
for b in 0 256 512 768; do
  if [ $b = 0 ]; then unset FGCN_LIB; else export FGCN_LIB=$PWD/tools/probes/libfgcn_sb$b.so; fi
  for shape in "300 64 64" "75 256 256"; do
    d=$out/b${b}_$(echo $shape | tr ' ' '_'); mkdir -p $d
    rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $d -o a -- python3 tools/_dbg_one.py sb $shape > $d/a.log 2>&1
    rocprofv3 --kernel-trace --stats --output-format csv -d $d -o t -- python3 tools/_dbg_one.py sb $shape > $d/t.log 2>&1
    python3 - "$d" "$b" "$shape" >> $out/summary.txt <<PY
import csv, glob, collections, sys
d, b, shape = sys.argv[1], sys.argv[2], sys.argv[3]
acc = collections.defaultdict(list)
for f in glob.glob(d + "/**/a_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "spatial_bwd_tile" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
avg = {k: sum(v) / len(v) for k, v in acc.items()}
ms = None
for f in glob.glob(d + "/**/t_kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "spatial_bwd_tile" in r["Name"]:
            ms = float(r["AverageNs"]) / 1e6
print(f"probe bits {b:>3s}  (T, Cin, Cout) = ({shape})  {ms if ms is None else round(ms, 4)} ms  VALU {avg.get('SQ_INSTS_VALU', 0):.3e}  MFMA {avg.get('SQ_INSTS_MFMA', 0):.3e}  "
      f"VALU:MFMA {avg.get('SQ_INSTS_VALU', 0) / max(avg.get('SQ_INSTS_MFMA', 1), 1):.2f}  SALU {avg.get('SQ_INSTS_SALU', 0):.3e}  LDS {avg.get('SQ_INSTS_LDS', 0):.3e}  VMEM {avg.get('SQ_INSTS_VMEM', 0):.3e}  "
      f"MFMA-busy {avg.get('SQ_VALU_MFMA_BUSY_CYCLES', 0):.3e}  wave-cycles {avg.get('SQ_WAVE_CYCLES', 0):.3e}")
PY
  done
done
cat $out/summary.txt
