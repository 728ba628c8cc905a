set -x
cd $GRAFT_REPO_ROOT
make -C oracle >/dev/null 2>&1
timeout -k 10 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 || exit 1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o r2 -- python3 bench.py --steps 10 --warmup 2 --cpu-sample 0 > gpurun_out/prof/bench2.log 2>&1
tail -1 gpurun_out/prof/bench2.log | cut -c1-1500
timeout -k 10 300 python bench.py --steps 20 --warmup 3 --cpu-sample 0 | cut -c1-1600
