#!/usr/bin/env python3
"""Turns the raw rocprofv3 output of one round (gpurun_out/<dir>/) into the small, committed summaries under profiles/.

    python profiles/summarize.py gpurun_out/r2p r02

Inputs (all produced by the commands listed in profiles/README.md, program directly after `--`):
  trace2s/ , trace1s/ : rocprofv3 --kernel-trace --stats  -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline
                        (default two-stream engine / VSRLAB_AMD_SINGLE_STREAM=1)
  pmc_fetch/, pmc_write/, pmc_mfma/ : rocprofv3 --kernel-trace --pmc <counters> -- python3 bench.py --roofline-only
Outputs: <tag>_bench_kernel_stats.csv (two-stream), <tag>_bench_kernel_stats_single_stream.csv, <tag>_roofline.json.
Counter handling follows /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE is doubled for wide coalesced
reads on gfx950, WRITE_SIZE is taken as is, both are in KiB; PMC passes are separate runs and are never compared in time
with un-profiled ones."""
import csv
import json
import os
import shutil
import sys
from collections import defaultdict

KERN = "conv3x3_c64_persist_kernel"
P540 = 540 * 960


def rows(path):
    with open(path, newline="") as f:
        yield from csv.DictReader(f)


def pmc_per_launch(d, counter, last=48):
    """mean counter value over the LAST `last` persistent-conv dispatches (= the timed launches of the roofline leg)."""
    vals = defaultdict(float)
    order = []
    for r in rows(os.path.join(d, "rl_counter_collection.csv")):
        if KERN in r["Kernel_Name"] and r["Counter_Name"] == counter:
            k = int(r["Dispatch_Id"])
            if k not in vals:
                order.append(k)
            vals[k] += float(r["Counter_Value"])
    sel = order[-last:]
    return sum(vals[k] for k in sel) / len(sel), len(sel)


def main():
    src, tag = sys.argv[1], sys.argv[2]
    here = os.path.dirname(os.path.abspath(__file__))
    shutil.copy(os.path.join(src, "trace2s", "bench_kernel_stats.csv"), os.path.join(here, f"{tag}_bench_kernel_stats.csv"))
    shutil.copy(os.path.join(src, "trace1s", "bench_kernel_stats.csv"), os.path.join(here, f"{tag}_bench_kernel_stats_single_stream.csv"))
    out = {"_how": __doc__.split("Outputs:")[0].strip()}
    # ---- which code was profiled: bench.py --roofline-only prints its tree id (sha1 of the kernel sources + the git HEAD build()
    # recorded); bench.py replays the counters below only when the running tree has the same sha1 ----
    for log in ("pmc_fetch.log", "pmc_write.log", "pmc_mfma.log"):
        try:
            for line in open(os.path.join(src, log)):
                if line.startswith("{") and '"tree"' in line:
                    tree = json.loads(line)["tree"]
                    out["csrc_sha1"], out["git_head"] = tree.get("csrc_sha1"), tree.get("git_head")
        except OSError:
            pass
    # ---- in-kernel clock of the dominant kernel: tools/ab_conv.py with the ABL=0 diagnostic build (s_memtime / s_memrealtime around
    # the kernel, median over workgroups; MI355X_MICROARCH "DVFS give-back" item 6), same box and call as the traces ----
    try:
        import re
        txt = open(os.path.join(src, "clock.log")).read()
        m = {k: re.findall(k + r": clock ([0-9.]+) GHz", txt) for k in ("relu", "skip")}
        if m["relu"] and m["skip"]:
            out["in_kernel_clock_GHz_by_variant"] = {k: float(v[-1]) for k, v in m.items()}
            out["in_kernel_clock_GHz"] = round(0.5 * (float(m["relu"][-1]) + float(m["skip"][-1])), 3)      # the leg alternates the two
    except OSError:
        pass
    # ---- roofline leg: its 48 timed launches are the last 48 persistent-conv rows of any trace of bench.py ----
    tr = [r for r in rows(os.path.join(src, "trace1s", "bench_kernel_trace.csv")) if KERN in r["Kernel_Name"]]
    tail = tr[-48:]
    dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in tail]
    out["roofline_leg_avg_us_rocprof"] = round(sum(dur) / len(dur), 2)
    fetch, n1 = pmc_per_launch(os.path.join(src, "pmc_fetch"), "FETCH_SIZE")
    write, n2 = pmc_per_launch(os.path.join(src, "pmc_write"), "WRITE_SIZE")
    out["fetch_size_kib_per_launch"] = round(fetch, 1)
    out["write_size_kib_per_launch"] = round(write, 1)
    out["hbm_bytes_per_launch"] = round((2.0 * fetch + write) * 1024.0)
    out["algorithmic_bytes_per_launch"] = (64 + 64) * P540 * 2 + 0.5 * 64 * P540 * 2
    mf = {}
    for c in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_VALU_MFMA_MOPS_BF16", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "GRBM_GUI_ACTIVE"):
        try:
            mf[c], _ = pmc_per_launch(os.path.join(src, "pmc_mfma"), c)
        except ZeroDivisionError:
            pass
    if mf:
        mf = {k: round(v) for k, v in mf.items()}
        # SQ_INSTS_VALU_MFMA_MOPS_BF16 x 512 = FLOP (MI355X_MICROARCH cycle constants); the kernel's work is 2*P*64*576
        mf["flop_from_mops"] = mf.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0) * 512
        mf["flop_algorithmic"] = 2 * P540 * 64 * 576
        if "SQ_VALU_MFMA_BUSY_CYCLES" in mf:
            # busy cycles are summed over the chip's 1024 SIMDs.  GRBM_GUI_ACTIVE / 8 over-reads the clock on dispatches this
            # short (MI355X_MICROARCH, DVFS give-back), so the fraction is quoted against the in-kernel clock measured with
            # s_memtime / s_memrealtime stamps (clock.log above; 1.6 GHz if that log is missing) and the trace's duration.
            per_simd = mf["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0
            mf["mfma_busy_cycles_per_simd"] = round(per_simd)
            ghz = out.get("in_kernel_clock_GHz", 1.6)
            mf["mfma_busy_frac_at_in_kernel_clock"] = round(per_simd / (out["roofline_leg_avg_us_rocprof"] * 1000.0 * ghz), 4)
            mf["clock_GHz_used"] = ghz
        out["mfma_busy"] = mf
    # ---- the same kernel inside ONE timed step (single-stream trace: no overlap inflating durations) ----
    # bench.py --steps 1 --warmup 1: the trace holds 2 steps, then the roofline leg (8 warm + 48 timed launches + 4 set-up)
    body = tr[:-(48 + 8 + 4)]
    step = body[len(body) // 2:]
    by = defaultdict(list)
    for r in step:
        gx = int(r["Grid_Size_X"])
        name = r["Kernel_Name"].split("kernel")[1].split("(")[0]             # "<ACT, RES, MASK[, PIPE]>"
        name = "<" + ", ".join(x.strip() for x in name.strip("<>").split(",")[:3]) + ">"
        by[(name, gx)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    # trunk-shape launches only (grid = 256 CUs x 512 threads over a 540x960 image is indistinguishable by grid size from
    # the HR launches, so they are separated by duration: a 540p launch moves 133-199 MB, the HR ones 4-16x that)
    alg = {"<1, false, 0>": 2 * 64 * P540 * 2, "<0, true, 0>": 3 * 64 * P540 * 2, "<0, false, 3>": 2 * 64 * P540 * 2 + 4.2e6}
    tot_b = tot_t = 0.0
    per = {}
    for (name, gx), d in by.items():
        if name in alg:
            small = [x for x in d if x < 80.0]
            per[name] = {"launches": len(small), "avg_us": round(sum(small) / max(len(small), 1), 2)}
            tot_b += alg[name] * len(small)
            tot_t += sum(small)
    out["in_step"] = {"what": "540p trunk launches of conv3x3_c64_persist inside one timed step (single-stream trace), call-weighted",
                      "per_variant": per, "algorithmic_GBs": round(tot_b / tot_t / 1e3, 1), "frac_of_8TBs": round(tot_b / tot_t / 1e3 / 8000.0, 4)}
    json.dump(out, open(os.path.join(here, f"{tag}_roofline.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
