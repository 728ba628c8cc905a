cd $GRAFT_REPO_ROOT
for f in 32 64 128 256; do echo "frames=$f"; timeout -k 10 300 python bench.py --steps 20 --warmup 3 --cpu-sample 0 --frames $f | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['stage_ms_per_batch'])"; done
