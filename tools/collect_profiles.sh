#!/bin/bash
# Regenerates the raw data behind profiles/ on a GPU box (run through gpurun from the repo root):
#   three separate rocprofv3 --pmc passes (FETCH_SIZE | WRITE_SIZE | SQ_*), one --kernel-trace --stats pass of bench.py and one of
#   tools/bench_local_points.py, then a plain bench.py run.  Outputs land under gpurun_out/; tools/prof_summary.py turns a
#   kernel_stats.csv into the markdown tables committed under profiles/.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc gpurun_out/prof gpurun_out/prof_lp
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY"; do
  name=$(echo $set | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmc -o $name -- python3 bench.py --steps 2 --warmup 1 --cpu-sample 0 > gpurun_out/pmc/$name.log 2>&1 || echo "failed $name"
done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o run -- python3 bench.py --steps 10 --warmup 2 --cpu-sample 0 > gpurun_out/prof/bench.log 2>&1
tail -1 gpurun_out/prof/bench.log | cut -c1-300
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_lp -o lp -- python3 tools/bench_local_points.py > gpurun_out/prof_lp/bench.log 2>&1
tail -1 gpurun_out/prof_lp/bench.log | cut -c1-300
python bench.py | cut -c1-2500
