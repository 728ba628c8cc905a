#!/bin/bash
# SQ counter passes of the extractor-only run (tools/stage_times.py): tools/pmc_stage.sh <outdir> <kernel-name-substring>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$1; mkdir -p $OUT
export R=3
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT -o p$i -- python3 tools/stage_times.py > $OUT/p$i.log 2>&1 || echo "failed pass $i"
done
python3 tools/pmc_summary.py $(find $OUT -name "*counter_collection.csv") | grep -i "$2"
