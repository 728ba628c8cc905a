cd $GRAFT_REPO_ROOT
make -C oracle >/dev/null 2>&1
timeout -k 10 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -2 || exit 1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc3
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc3 -o FETCH_SIZE -- python3 bench.py --steps 2 --warmup 1 --cpu-sample 0 > gpurun_out/pmc3/f.log 2>&1
python bench.py --cpu-sample 0 | cut -c1-300
