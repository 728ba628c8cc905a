#!/bin/bash
# Raw data behind profiles/rNN_*: tools/profile_round.sh <tag> [bench args, e.g. --config tum_bow]   (run through gpurun from the repo root)
#   one rocprofv3 --kernel-trace --stats pass of bench.py as it runs by default (extractors on two streams, two handle sets in turn: launches
#   overlap), one with --lr-streams 1 --sets 1 (every kernel alone on the chip), then separate --pmc passes (rocprofv3 serialises kernels there) (never combined with other trace
#   domains): two SQ sets, FETCH_SIZE, WRITE_SIZE, TCC hit/miss, the fabric requests by size (TCC_EA0_RDREQ / _32B / _128B, TCC_EA0_WRREQ / _64B).  tools/profile_report.py turns gpurun_out/<tag>/ into
#   the markdown / json files committed under profiles/.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$1; shift; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o stats -- python3 bench.py --steps 10 --warmup 2 --cpu-sample 0 --e2e-steps 0 --per-frame 0 --content-steps 0 --cabi-steps 0 "$@" > $OUT/stats.log 2>&1 || { echo "stats pass failed"; tail -3 $OUT/stats.log; exit 1; }
tail -1 $OUT/stats.log | cut -c1-200
# the same with every kernel alone on the chip (one stream, one set of handles): the per-kernel durations of the PMC tables
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o stats1 -- python3 bench.py --steps 10 --warmup 2 --cpu-sample 0 --e2e-steps 0 --per-frame 0 --content-steps 0 --cabi-steps 0 --lr-streams 1 --sets 1 "$@" > $OUT/stats1.log 2>&1 || { echo "one-stream stats pass failed"; tail -3 $OUT/stats1.log; exit 1; }
tail -1 $OUT/stats1.log | cut -c1-200
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT -o pmc$i -- python3 bench.py --steps 3 --warmup 1 --cpu-sample 0 --e2e-steps 0 --per-frame 0 --content-steps 0 --cabi-steps 0 "$@" > $OUT/pmc$i.log 2>&1 || { echo "pmc pass $i failed"; exit 1; }
  echo "pass $i done"
done
# ---- what the committed tables are derived from, small enough to keep: the two kernel-stats tables and the per-kernel averages of
#      every counter pass (tools/profile_report.py copies gpurun_out/<tag>/final/ to profiles/<round>_raw/)
mkdir -p $OUT/final
cp $(find $OUT -name "stats_kernel_stats.csv" | head -1) $OUT/final/stats_kernel_stats.csv
cp $(find $OUT -name "stats1_kernel_stats.csv" | head -1) $OUT/final/stats1_kernel_stats.csv
grep "^{" $OUT/stats.log | tail -1 > $OUT/final/bench_line_stats_pass.json
grep "^{" $OUT/stats1.log | tail -1 > $OUT/final/bench_line_one_stream_pass.json
python3 - $OUT <<'PY'
import collections, csv, glob, json, os, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sorted(glob.glob(os.path.join(sys.argv[1], "**", "pmc*counter_collection.csv"), recursive=True)):
    for r in csv.DictReader(open(path)):
        acc[r["Kernel_Name"].split("(")[0].replace("void ", "").strip()][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: {"mean_of_later_launches": sum(v[len(v) // 3:]) / max(len(v[len(v) // 3:]), 1), "launches": len(v)} for c, v in cs.items()}
       for k, cs in acc.items() if not k.startswith("at::") and "rocclr" not in k}
json.dump(out, open(os.path.join(sys.argv[1], "final", "pmc_averages.json"), "w"), indent=1)
print("final/: kernel stats x 2, bench lines x 2, pmc_averages.json for", len(out), "kernels")
PY
# ---- the reports, written on the GPU box (gpurun merges at most 64 MiB back: the per-dispatch CSVs stay there): $OUT/report/<name>_*.md / .json /
#      _raw/; copy them into profiles/ afterwards.  REPORT_NAME (default r06), REPORT_CONFIG (default kitti_stereo)
mkdir -p $OUT/report
python3 tools/profile_report.py $OUT $OUT/report/${REPORT_NAME:-r06} ${REPORT_CONFIG:-kitti_stereo}
find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
du -sh $OUT
