#!/bin/bash
# HBM-side traffic counters of a short bench run (all stages): tools/pmc_bench_mem.sh <outdir>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$1; mkdir -p $OUT
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT -o m$i -- python3 bench.py --steps 4 --warmup 1 --cpu-sample 0 --e2e-steps 0 > $OUT/m$i.log 2>&1 || echo "failed pass $i"
done
python3 tools/pmc_summary.py $(find $OUT -name "m*counter_collection.csv")
