"""Print the headline and the per-content stage times of a bench.py JSON line: tools/bench_summary.py <file>"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["stage_ms_per_batch"])
for k, v in d.get("content", {}).items():
    print(k, v["frames_per_s"], v.get("stage_ms_per_batch"))
