#!/bin/bash
# HBM traffic counter passes of the extractor-only run (tools/stage_times.py): tools/pmc_mem.sh <outdir>
# FETCH_SIZE and WRITE_SIZE do not fit one pass (TCC slots): one rocprofv3 run each, kernel trace only.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$1; mkdir -p $OUT
export R=3
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT -o m$i -- python3 tools/stage_times.py > $OUT/m$i.log 2>&1 || echo "failed pass $i"
done
python3 tools/pmc_summary.py $(find $OUT -name "m*counter_collection.csv")
