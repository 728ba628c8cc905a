#!/bin/bash
# per-kernel time table of a short bench run: tools/kstats.sh <outdir> [bench args]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$1; shift; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o run -- python3 bench.py --steps 10 --warmup 2 --cpu-sample 0 --e2e-steps 0 --per-frame 0 --content-steps 0 "$@" > $OUT/bench.log 2>&1 || { tail -5 $OUT/bench.log; exit 1; }
tail -1 $OUT/bench.log | cut -c1-400
python3 - $OUT <<'PY'
import csv, sys, glob
f = glob.glob(sys.argv[1] + '/**/run_kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print("| kernel | calls | avg us | total ms | % |\n|---|---|---|---|---|")
for r in rows[:22]:
    print(f"| {r['Name'][:60]} | {r['Calls']} | {float(r['AverageNs'])/1e3:.1f} | {float(r['TotalDurationNs'])/1e6:.2f} | {100*float(r['TotalDurationNs'])/tot:.1f} |")
PY
