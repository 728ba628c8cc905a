cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof4
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof4 -o r4 -- python3 bench.py --steps 10 --warmup 2 --cpu-sample 0 > gpurun_out/prof4/bench.log 2>&1
tail -1 gpurun_out/prof4/bench.log | cut -c1-400
ls gpurun_out/prof4
