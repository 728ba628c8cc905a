#!/usr/bin/env python3
"""gpurun_out/<tag>/ (tools/profile_round.sh) -> profiles/<round>_kernel_stats.md, <round>_pmc_counters.md, <round>_pmc_dominant.json

usage: tools/profile_report.py gpurun_out/r02 profiles/r02 [config]
FETCH_SIZE is reported in KB and, on gfx950, tallies the 128-byte requests of 16-byte-per-lane coalesced loads at 64 bytes
(MI355X_MICROARCH.md, HBM section): kernels whose bulk reads are such loads are listed in WIDE and get the x2 correction.
"""
import collections
import csv
import glob
import json
import os
import sys

# 16-byte-per-lane STREAMING reads (a wave's load = whole 128-byte lines).  orient_describe also loads 16 bytes per lane, but as a
# gather of 48- / 64-byte row pieces: its requests are 64 bytes and FETCH_SIZE counts them as such (TCC_MISS x 128 B = 0.91 GB per
# launch agrees with the uncorrected 0.78 GB, not with twice that)
WIDE = ("copy_level0", "pyr_resize", "fast_cells", "fast_groups")
ALG_KB = {  # algorithmic KB per 256-image launch at KITTI geometry
    "copy_level0": 2 * 466616 * 256 / 1024,
    "pyr_resize": ((1444097 - 36330) + (1444097 - 466616)) * 256 / 1024 / 7,
    "fast_cells": (1444097 + 8 * 4600) * 256 / 1024,
    "gauss_blur7": 2 * 1444097 * 256 / 1024,
    # the fused level chain: seven launches read a level, write its blurred plane and the next level (averaged per launch), the last
    # level's launch only blurs
    "blur_level_kernel<true>": ((1444097 - 36330) + (1444097 - 36330) + (1444097 - 466616)) * 256 / 1024 / 7,
    "blur_level_kernel<false>": 2 * 36330 * 256 / 1024,
    "orient_describe": 2000 * (749 + 512 + 60) * 256 / 1024,
    # stereo association per pair: both eyes' keypoint records + descriptors (2 x 2000 x 60 B), per matched left keypoint (~1 850) the 11 x 11
    # left SAD window and the 11 x 21 right strip it slides over, 8 B of result per left keypoint (an estimate: windows overlap)
    "stereo_match": (2 * 2000 * 60 + 1850 * (121 + 231) + 2000 * 8) * 256 / 1024,
}


def fabric_kb(c):
    """Bytes the L2 moved to / from the fabric, from the request counters by size: read requests are 32, 64 or 128 bytes
    (TCC_EA0_RDREQ counts all, _32B and _128B the two named sizes), write requests 32 or 64 (TCC_EA0_WRREQ / _64B).  None without
    the passes.  FETCH_SIZE = RDREQ x 64 B tallies a 128-byte request at half its bytes: the guide's x2 for wide streaming loads."""
    if "TCC_EA0_RDREQ_sum" not in c or "TCC_EA0_WRREQ_sum" not in c:
        return None, None
    r, r32, r128 = c["TCC_EA0_RDREQ_sum"], c.get("TCC_EA0_RDREQ_32B_sum", 0), c.get("TCC_EA0_RDREQ_128B_sum", 0)
    w, w64 = c["TCC_EA0_WRREQ_sum"], c.get("TCC_EA0_WRREQ_64B_sum", 0)
    return (r128 * 128 + r32 * 32 + max(r - r128 - r32, 0) * 64) / 1024, (w64 * 64 + max(w - w64, 0) * 32) / 1024


def short(name):
    return name.split("(")[0].replace("void ", "").strip()


def counters(tag_dir):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in sorted(glob.glob(os.path.join(tag_dir, "**", "pmc*counter_collection.csv"), recursive=True)):
        for r in csv.DictReader(open(path)):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v[len(v) // 3:]) / max(len(v[len(v) // 3:]), 1) for c, v in cs.items()} for k, cs in acc.items()}


def main(tag_dir, out_prefix, config="kitti_stereo"):
    stats = glob.glob(os.path.join(tag_dir, "**", "stats_kernel_stats.csv"), recursive=True)[0]
    rows = list(csv.DictReader(open(stats)))
    lines = [l for l in open(os.path.join(tag_dir, "stats.log")).read().split("\n") if l.startswith("{")]
    bench_line = lines[-1] if lines else ""
    try:
        b = json.loads(bench_line)
        head = f"{b['value']:.0f} {b['unit']} under the profiler ({b['ms_per_step']:.3f} ms per step of {b['config']['frames_per_gpu_per_step']} stereo frames)"
    except Exception:
        head = "bench line not parsed"
    def table(rws):
        out = "| kernel | calls | total us | avg us | min us | max us | % |\n|---|---:|---:|---:|---:|---:|---:|\n"
        for r in rws:
            if float(r["Percentage"]) < 0.05:
                continue
            out += (f"| {short(r['Name'])[:60]} | {r['Calls']} | {float(r['TotalDurationNs'])/1e3:.1f} | {float(r['AverageNs'])/1e3:.1f} | "
                    f"{float(r['MinNs'])/1e3:.1f} | {float(r['MaxNs'])/1e3:.1f} | {float(r['Percentage']):.2f} |\n")
        return out

    foot = ("\n`__amd_rocclr_copyBuffer` / `fillBufferAligned` and the `at::native` kernels belong to bench.py's set-up (uploading the synthetic "
            "frames, one copy per image, before the first step) and result read-back; the kernel trace of a steady-state step holds none of "
            "them (two `at::native` fills of the match buffers excepted).  Percentages are of the whole profiled process.\n")
    cmd = "`rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 10 --warmup 2 --cpu-sample 0 --e2e-steps 0 --per-frame 0 --content-steps 0"
    avg_us = {short(r["Name"]): float(r["AverageNs"]) / 1e3 for r in rows}
    one = glob.glob(os.path.join(tag_dir, "**", "stats1_kernel_stats.csv"), recursive=True)
    with open(out_prefix + "_kernel_stats.md", "w") as f:
        f.write(f"# {os.path.basename(out_prefix)}: rocprofv3 kernel stats of the bench step\n\n")
        if one:   # the one-stream pass: every kernel alone on the chip -- the durations the PMC tables and `roofline.avg_launch_ms` go with
            rows1 = list(csv.DictReader(open(one[0])))
            avg_us = {short(r["Name"]): float(r["AverageNs"]) / 1e3 for r in rows1}
            l1 = [l for l in open(os.path.join(tag_dir, "stats1.log")).read().split("\n") if l.startswith("{")]
            try:
                b1 = json.loads(l1[-1])
                head1 = f"{b1['value']:.0f} {b1['unit']} under the profiler ({b1['ms_per_step']:.3f} ms per step)"
            except Exception:
                head1 = "bench line not parsed"
            f.write("Two passes of the same build (tools/profile_round.sh), times in microseconds per launch of 256 images: first every kernel alone on "
                    "the chip (what `roofline.avg_launch_ms`, the PMC tables and earlier rounds' tables show), then the default layout, in which launches "
                    "overlap by design.\n\n## Every kernel alone on the chip (one stream, one set of handles): the per-kernel figures\n\n"
                    f"{cmd} --lr-streams 1 --sets 1`: a step starts when the one before it has ended, left and right extractor on the same stream; {head1}.\n\n")
            f.write(table(rows1))
            f.write("\n## Default layout: overlapping launches (left | right extractor on two streams, handle sets in turn)\n\n"
                    f"{cmd}`; {head}.  The left and right extractor's launches overlap and the matching half of the step before runs beside them, so a "
                    "launch's own duration here contains the other launches' share of the chip, and under the profiler the launches interleave "
                    "differently than in a plain run (`roofline.timed_region` in bench.py's line is measured there).\n\n")
            f.write(table(rows))
        else:
            f.write(f"{cmd}` (tools/profile_round.sh), times in microseconds per launch of 256 images; {head}.\n\n")
            f.write(table(rows))
        f.write(foot)
    cs = counters(tag_dir)
    with open(out_prefix + "_pmc_counters.md", "w") as f:
        f.write(f"# {os.path.basename(out_prefix)}: PMC counters per kernel launch (256 KITTI images / stereo frames per launch)\n\n"
                "Separate `rocprofv3 --kernel-trace --pmc <set>` passes of `python3 bench.py --steps 3 --warmup 1 --cpu-sample 0 --e2e-steps 0 --per-frame 0 --content-steps 0` "
                "(tools/profile_round.sh: two SQ sets, FETCH_SIZE, WRITE_SIZE, TCC hit/miss, GRBM_GUI_ACTIVE), averaged over the later launches of "
                "each kernel (rocprofv3 serialises kernels in counter passes); durations from the one-stream `--stats` pass of the same build.  FETCH / WRITE are KB as rocprofv3 reports them; FETCH is doubled "
                "(x2) for kernels whose bulk reads are coalesced 16-byte-per-lane loads, which gfx950 tallies at half their bytes "
                "(MI355X_MICROARCH.md, HBM section); Infinity-Cache hits are counted, so read them as fabric-side traffic.  SQ_WAVE_CYCLES, "
                "SQ_WAIT_* and SQ_ACTIVE_* count quad-cycles summed over waves.  `VALU busy` = SQ_ACTIVE_INST_VALU x 4 clocks / (1024 SIMDs x kernel "
                "clocks) -- the time the SIMDs' vector ALUs were executing; an average above 2 clocks per instruction means slow-class instructions "
                "(tools/valu_bench.hip: min/max, shifts, v_perm, packed-16, 3-operand integer at ~4.1 clocks; add/sub/logic/mov/FP32/16-bit VOP2 at ~2.1-2.8) --, kernel clocks = GRBM_GUI_ACTIVE / 8; `LDS busy` = SQ_LDS_IDX_ACTIVE / (256 CUs x kernel clocks).\n\n")
        f.write("| kernel | us | FETCH KB | WRITE KB | alg. KB | L2 hit | waves | VALU / wave | SALU / wave | LDS / wave | VMEM / wave | wait | issue stall | VALU busy | LDS busy | LDS conflict |\n")
        f.write("|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|\n")
        for k, c in cs.items():
            if k.startswith("at::") or "rocclr" in k or k not in avg_us:
                continue
            wide = any(k.startswith(w) for w in WIDE)
            fetch = c.get("FETCH_SIZE", 0) * (2 if wide else 1)
            alg = next((v for a, v in ALG_KB.items() if k.startswith(a)), None)
            w = max(c.get("SQ_WAVES", 0), 1)
            clocks = c.get("GRBM_GUI_ACTIVE", 0) / 8
            hit = c.get("TCC_HIT_sum", 0) / max(c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0), 1)
            wc = max(c.get("SQ_WAVE_CYCLES", 0), 1)
            f.write(f"| {k[:40]} | {avg_us[k]:.1f} | {fetch:.3g}{' (x2)' if wide else ''} | {c.get('WRITE_SIZE', 0):.3g} | "
                    f"{'' if alg is None else format(alg, '.3g')} | {hit:.2f} | {w:.3g} | {c.get('SQ_INSTS_VALU', 0)/w:.0f} | {c.get('SQ_INSTS_SALU', 0)/w:.0f} | "
                    f"{c.get('SQ_INSTS_LDS', 0)/w:.0f} | {(c.get('SQ_INSTS_VMEM_RD', 0)+c.get('SQ_INSTS_VMEM_WR', 0))/w:.1f} | "
                    f"{c.get('SQ_WAIT_ANY', 0)/wc:.2f} | {c.get('SQ_WAIT_INST_ANY', 0)/wc:.2f} | "
                    f"{(c.get('SQ_ACTIVE_INST_VALU', 0)*4/(1024*clocks) if clocks else 0):.2f} | {(c.get('SQ_LDS_IDX_ACTIVE', 0)/(256*clocks) if clocks else 0):.2f} | "
                    f"{c.get('SQ_LDS_BANK_CONFLICT', 0)/max(c.get('SQ_LDS_IDX_ACTIVE', 0), 1):.2f} |\n")
        f.write("\n(`blur_level_kernel<true>` / `pyr_resize_dot_kernel`: average of the seven level launches.)\n")
        # ---- fabric traffic from the request-size split, against the algorithmic bytes, for every kernel above 0.2 ms per step
        f.write("\n## Fabric traffic by request size\n\n`TCC_EA0_RDREQ` (all read requests), `_32B`, `_128B` and `TCC_EA0_WRREQ`, `_64B` in passes of their own: read bytes = "
                "128 x RDREQ_128B + 32 x RDREQ_32B + 64 x the rest, write bytes = 64 x WRREQ_64B + 32 x the rest.  This settles which kernels need the "
                "guide's x 2 on FETCH_SIZE (= RDREQ x 64 B): all whose read requests are 128-byte ones -- on this build every kernel of the table, the "
                "4-byte-per-lane window loads of the level kernel and the patch gathers of the describe kernel included.  Per launch; `x calls` = launches of "
                "that kernel per extractor (or per step for the matcher kernels).\n\n")
        f.write("| kernel | us | read req | 128 B | 32 B | read MB | write req | 64 B | write MB | traffic MB | alg. MB | traffic / alg. |\n|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|\n")
        for k, c in cs.items():
            if k.startswith("at::") or "rocclr" in k or k not in avg_us:
                continue
            rk, wk = fabric_kb(c)
            if rk is None or avg_us[k] < 20:
                continue
            alg = next((v for a, v in ALG_KB.items() if k.startswith(a)), None)
            f.write(f"| {k[:40]} | {avg_us[k]:.1f} | {c['TCC_EA0_RDREQ_sum']:.3g} | {c.get('TCC_EA0_RDREQ_128B_sum', 0):.3g} | {c.get('TCC_EA0_RDREQ_32B_sum', 0):.3g} | {rk / 1024:.1f} | "
                    f"{c['TCC_EA0_WRREQ_sum']:.3g} | {c.get('TCC_EA0_WRREQ_64B_sum', 0):.3g} | {wk / 1024:.1f} | {(rk + wk) / 1024:.1f} | "
                    f"{'' if alg is None else format(alg / 1024, '.1f')} | {'' if alg is None else format((rk + wk) / alg, '.2f')} |\n")
    dom = max((k for k in cs if k in avg_us and not k.startswith("at::") and "rocclr" not in k), key=lambda k: avg_us[k])
    stage = {"fast_cells": "fast", "fast_groups": "fast", "orient_describe": "describe", "gauss_blur7": "blur", "blur_level": "pyramid"}
    st = next((v for a, v in stage.items() if dom.startswith(a)), dom)
    wide = any(dom.startswith(w) for w in WIDE)
    rk, wk = fabric_kb(cs[dom])
    cd = cs[dom]
    wv, wcyc = max(cd.get("SQ_WAVES", 0), 1), max(cd.get("SQ_WAVE_CYCLES", 0), 1)
    clk = cd.get("GRBM_GUI_ACTIVE", 0) / 8
    issue = {"waves_per_launch": wv, "valu_per_wave": round(cd.get("SQ_INSTS_VALU", 0) / wv, 1), "salu_per_wave": round(cd.get("SQ_INSTS_SALU", 0) / wv, 1),
             "lds_per_wave": round(cd.get("SQ_INSTS_LDS", 0) / wv, 1),
             "wave_cycles": {"active": round(cd.get("SQ_ACTIVE_INST_ANY", 0) / wcyc, 3), "wait_issue": round(cd.get("SQ_WAIT_INST_ANY", 0) / wcyc, 3),
                             "wait_memory": round(cd.get("SQ_WAIT_ANY", 0) / wcyc, 3)},
             "valu_busy": round(cd.get("SQ_ACTIVE_INST_VALU", 0) * 4 / (1024 * clk), 3) if clk else None,
             "lds_busy": round(cd.get("SQ_LDS_IDX_ACTIVE", 0) / (256 * clk), 3) if clk else None,
             "lds_bank_conflict_share": round(cd.get("SQ_LDS_BANK_CONFLICT", 0) / max(cd.get("SQ_LDS_IDX_ACTIVE", 0), 1), 3)}
    # the raw material of every table above, a few KB: profiles/<round>_raw/
    fin = os.path.join(tag_dir, "final")
    if os.path.isdir(fin):
        import shutil
        raw = out_prefix + "_raw"
        os.makedirs(raw, exist_ok=True)
        for fn in os.listdir(fin):
            shutil.copy(os.path.join(fin, fn), os.path.join(raw, fn))
    json.dump({"stage": st, "config": config, "kernel": dom, "images_per_launch": 256, "issue": issue, "fetch_kb": cs[dom].get("FETCH_SIZE", 0),
               "fetch_correction": 2.0 if wide else 1.0, "write_kb": cs[dom].get("WRITE_SIZE", 0), "avg_us_stats_pass": avg_us[dom],
               "read_kb_by_request_size": rk, "write_kb_by_request_size": wk,
               "source": "tools/profile_round.sh: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) -- python3 bench.py --steps 3 "
                         "--warmup 1 --cpu-sample 0 --e2e-steps 0 --per-frame 0 --content-steps 0; FETCH_SIZE x 2 for 16-byte-per-lane coalesced loads on gfx950 (MI355X_MICROARCH.md, HBM section); "
                         "read / write_kb_by_request_size: TCC_EA0_RDREQ / WRREQ split by request size (passes of their own)"},
              open(out_prefix + "_pmc_dominant.json", "w"), indent=1)
    print("dominant:", dom, avg_us[dom])


if __name__ == "__main__":
    main(*sys.argv[1:])
