#!/bin/bash
# Whole-step A/B of experiment builds: tools/bench_quick.sh <out file under gpurun_out/> <lib[:kind]> ...   ("" or base = liborbfe.so)
# Each entry runs bench.py (150 steps, no side measurements) through tools/bench_ab.py and appends one line: value, ms per step, stage times.
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$1; shift; mkdir -p $(dirname $OUT)
for ent in "$@"; do
  lib=${ent%%:*}; kind=0; [[ "$ent" == *:* ]] && kind=${ent##*:}
  [ "$lib" = base ] && lib=""
  ORBFE_AB_LIB=$lib python3 tools/bench_ab.py --steps 150 --cpu-sample 0 --e2e-steps 0 --per-frame 0 --content-steps 0 --blur-kind $kind 2>>$OUT.err | \
    python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['value'], d['ms_per_step'], d['roofline']['stage_ms_per_batch'])" "$ent" >> $OUT
done
cat $OUT
