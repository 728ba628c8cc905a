mkdir -p gpurun_out/r06
for cfg in "128 3" "64 3" "128 4" "64 4" "192 3" "256 3"; do set -- $cfg; F=$1; S=$2
python bench.py --frames $F --sets $S --steps $((38400/F)) --per-frame 0 --content-steps 0 --cpu-sample 0 --e2e-steps 0 --cabi-steps 0 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('F', d['config']['frames_per_gpu_per_step'], 'sets', $S, d['value'], d['ms_per_step'], d['roofline']['stage_ms_per_batch'])"
done
