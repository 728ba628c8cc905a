#!/bin/bash
# SQ counter passes of a short bench run (all stages): tools/pmc_bench.sh <outdir> <kernel-name-substring>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$1; mkdir -p $OUT
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT -o p$i -- python3 bench.py --steps 4 --warmup 1 --cpu-sample 0 --e2e-steps 0 > $OUT/p$i.log 2>&1 || echo "failed pass $i"
done
python3 tools/pmc_summary.py $(find $OUT -name "p*counter_collection.csv") | grep -i "$2"
