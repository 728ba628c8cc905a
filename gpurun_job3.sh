cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
for ov in 0 1; do echo "overlap_blur=$ov"; ORBFE_OVERLAP_BLUR=$ov python bench.py --cpu-sample 0 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['stage_ms_per_batch'])"; done
