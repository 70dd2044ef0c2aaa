set -x
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
B="python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-f32-mode --no-f16x2-mode --no-kernel-timing --math f16x2"
for rep in 1 2; do
  for mk in 128 64; do
    for clips in 64 8; do
      steps=10; [ $clips = 8 ] && steps=30
      r=$(FGCN_PW_MIN_K=$mk $B --batch $clips --steps $steps 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['loss'])" 2>&1)
      echo "f16x2 pw_min_k=$mk clips=$clips : $r" >> gpurun_out/t13_ab.log
    done
  done
done
cat gpurun_out/t13_ab.log
timeout -k 10 300 python tools/rccl_world1_check.py --steps 12 > gpurun_out/t13_rccl.json 2> gpurun_out/t13_rccl.err; echo "rccl rc=$?"; cat gpurun_out/t13_rccl.json
