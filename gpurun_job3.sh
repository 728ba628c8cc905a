cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_extractor_gpu.py -x -q -m gpu 2>&1 | tail -2
for sh in -1 3 4 5; do echo "shift=$sh"; ORBFE_XCD_RUN_SHIFT=$sh python bench.py --cpu-sample 0 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['stage_ms_per_batch'])"; done
