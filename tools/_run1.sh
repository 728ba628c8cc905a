mkdir -p gpurun_out/r06
(for i in $(seq 1 40); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo; sleep 0.25; done) > gpurun_out/r06/clocks.txt &
sleep 1
python bench.py --steps 2500 --per-frame 0 --content-steps 0 --cpu-sample 0 --e2e-steps 0 --cabi-steps 0 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
wait
cat gpurun_out/r06/clocks.txt | sort | uniq -c | sort -rn | head -12
