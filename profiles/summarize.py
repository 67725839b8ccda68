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

KERN = "conv3x3_c64_chain_kernel"          # the dominant kernel (round 3: the trunk chains); its leg: bench.py dominant_kernel_roofline
KERN_LAYER = "conv3x3_c64_persist_kernel"   # one launch per layer (bench.py per_layer_kernel_roofline)
P540 = 540 * 960
CHAIN_LAYERS, CHAIN_ITERS = 16, 12          # bench.py


def rows(path):
    with open(path, newline="") as f:
        yield from csv.DictReader(f)


def pmc_per_launch(d, counter, last=CHAIN_ITERS):
    """mean counter value over the LAST `last` chain dispatches (= the timed launches of the roofline leg)."""
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
    # ---- in-kernel clock of the dominant kernel: tools/ab_chain.py with the ABL=0 diagnostic build (s_memtime / s_memrealtime around
    # the kernel, median over workgroups; MI355X_MICROARCH "DVFS give-back" item 6), same box and call as the traces ----
    try:
        import re
        m = re.findall(r"chain: clock ([0-9.]+) GHz", open(os.path.join(src, "clock_chain.log")).read())
        if m:
            out["in_kernel_clock_GHz"] = float(m[-1])
    except OSError:
        pass
    # ---- roofline leg: its 12 timed launches are the last 12 chain rows of any trace of bench.py; the one-launch-per-layer leg
    # behind it is the last 48 persistent-conv rows ----
    allrows = list(rows(os.path.join(src, "trace1s", "bench_kernel_trace.csv")))
    us = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tr = [r for r in allrows if KERN in r["Kernel_Name"]]
    dur = [us(r) for r in tr[-CHAIN_ITERS:]]
    out["roofline_leg_avg_us_rocprof"] = round(sum(dur) / len(dur), 2)
    out["roofline_leg_avg_us_per_layer_rocprof"] = round(sum(dur) / len(dur) / CHAIN_LAYERS, 2)
    trl = [r for r in allrows if KERN_LAYER in r["Kernel_Name"]]
    durl = [us(r) for r in trl[-48:]]
    out["per_layer_launch_leg_avg_us_rocprof"] = round(sum(durl) / len(durl), 2)
    fetch, n1 = pmc_per_launch(os.path.join(src, "pmc_fetch"), "FETCH_SIZE")
    write, n2 = pmc_per_launch(os.path.join(src, "pmc_write"), "WRITE_SIZE")
    out["fetch_size_kib_per_launch"] = round(fetch, 1)
    out["write_size_kib_per_launch"] = round(write, 1)
    out["hbm_bytes_per_launch"] = round((2.0 * fetch + write) * 1024.0)
    out["algorithmic_bytes_per_launch"] = sum((3 if l & 1 else 2) * 64 * P540 * 2 for l in range(CHAIN_LAYERS))
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
        mf["flop_algorithmic"] = 2 * P540 * 64 * 576 * CHAIN_LAYERS
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
    # bench.py --steps 1 --warmup 1: the trace holds 2 steps (per step 14 forward chains <0> of 60 layers and 14 backward chains <2>
    # of 59), then the roofline leg (2 warm + 12 timed launches of <0>)
    fw = [r for r in tr if "<0>" in r["Kernel_Name"]][:-(CHAIN_ITERS + 2)]
    bw = [r for r in tr if "<2>" in r["Kernel_Name"]]
    fw, bw = fw[len(fw) // 2:], bw[len(bw) // 2:]
    bits = 4.2e6                                   # sign bits of one 540p activation
    alg_fw = 30 * ((2 + 3) * 64 * P540 * 2 + bits)
    alg_bw = 30 * (2 * 64 * P540 * 2 + bits) + 29 * 3 * 64 * P540 * 2
    tf, tb = sum(us(r) for r in fw), sum(us(r) for r in bw)
    tot_b, tot_t = alg_fw * len(fw) + alg_bw * len(bw), tf + tb
    per = {"forward chain <CHAIN_RELU>, 60 layers": {"launches": len(fw), "avg_us": round(tf / max(len(fw), 1), 1), "avg_us_per_layer": round(tf / max(len(fw), 1) / 60, 2)},
           "backward chain <CHAIN_MASK>, 59 layers": {"launches": len(bw), "avg_us": round(tb / max(len(bw), 1), 1), "avg_us_per_layer": round(tb / max(len(bw), 1) / 59, 2)}}
    out["in_step"] = {"what": "the trunk chains of conv3x3_c64_chain_kernel inside one timed step (single-stream trace), call-weighted",
                      "per_variant": per, "algorithmic_GBs": round(tot_b / tot_t / 1e3, 1), "frac_of_8TBs": round(tot_b / tot_t / 1e3 / 8000.0, 4)}
    # ---- HBM bytes of the dominant kernels INSIDE a step (r04): FETCH_SIZE / WRITE_SIZE passes over `bench.py --clips 1 --steps 1 --warmup 1`
    # (pmc_step_fetch/, pmc_step_write/), per launch, against the algorithmic bytes of the same launch.  In the step a layer's input was
    # written one layer earlier: what the counters show below the algorithmic read bytes is served by the 256 MiB Infinity Cache. ----
    try:
        def per_kernel(d, counter):
            vals, names = defaultdict(float), {}
            for r in rows(os.path.join(src, d, "st_counter_collection.csv")):
                if r["Counter_Name"] == counter:
                    k = int(r["Dispatch_Id"]); vals[k] += float(r["Counter_Value"]); names[k] = r["Kernel_Name"]
            return vals, names
        fv, fn = per_kernel("pmc_step_fetch", "FETCH_SIZE")
        wv, wn = per_kernel("pmc_step_write", "WRITE_SIZE")
        def mean_of(vals, names, pred, drop_last=0):
            ks = [k for k in sorted(vals) if pred(names[k])]
            ks = ks[:len(ks) - drop_last] if drop_last else ks
            return (sum(vals[k] for k in ks) / len(ks), len(ks)) if ks else (0.0, 0)
        st = {}
        for tag2, pred, drop, alg in (("forward chain, 60 layers", lambda n: KERN in n and "<0>" in n, CHAIN_ITERS + 2, alg_fw),
                                      ("backward chain, 59 layers", lambda n: KERN in n and "<2>" in n, 0, alg_bw),
                                      ("wgrad3x3_c64_pc (all launches of the step)", lambda n: "wgrad3x3_c64_pc" in n, 0, None)):
            f, nf = mean_of(fv, fn, pred, drop)
            w, nw = mean_of(wv, wn, pred, drop)
            st[tag2] = {"launches": nf, "fetch_x2_GB_per_launch": round(2.0 * f * 1024 / 1e9, 3), "write_GB_per_launch": round(w * 1024 / 1e9, 3)}
            if alg:
                st[tag2]["algorithmic_GB_per_launch"] = round(alg / 1e9, 3)
        out["in_step_traffic"] = st
    except (OSError, KeyError, ZeroDivisionError):
        pass
    json.dump(out, open(os.path.join(here, f"{tag}_roofline.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))
    lds_conflicts(src, tag, here, out.get("csrc_sha1"), out.get("git_head"))


def lds_conflicts(src, tag, here, sha1, head):
    """<tag>_lds_conflicts.md from pmc_lds/ (rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -- python3 bench.py
    --train-flow --clips 1 --steps 1 --warmup 1 --no-cpu-baseline): conflict cycles / active cycles per kernel, summed over its launches.
    Stamped with the profiled tree's csrc_sha1 (round-3 VERDICT #11: a stale file under a "final tree" heading)."""
    path = os.path.join(src, "pmc_lds", "lds_counter_collection.csv")
    if not os.path.exists(path):
        return
    tree = {"csrc_sha1": sha1, "git_head": head}
    try:
        for line in open(os.path.join(src, "pmc_lds.log")):
            if line.startswith("{") and '"tree"' in line:
                tree = json.loads(line)["tree"]
    except OSError:
        pass
    acc = defaultdict(lambda: defaultdict(float))
    launches = defaultdict(set)
    for r in rows(path):
        k = r["Kernel_Name"]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        launches[k].add(r["Dispatch_Id"])
    lines = [f"# LDS bank conflicts per kernel over `bench.py --train-flow --clips 1 --steps 1 --warmup 1` under rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE",
             f"# profiled tree: csrc_sha1 {tree.get('csrc_sha1')}, git {tree.get('git_head')} (written by profiles/summarize.py {tag}; regenerate with profiles/{tag}_commands.sh)",
             "", "| kernel | launches | SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE |", "|---|---|---|"]
    for k in sorted(acc):
        a = acc[k]
        if a.get("SQ_LDS_IDX_ACTIVE", 0) <= 0 or "rocclr" in k or "at::" in k:
            continue
        lines.append(f"| `{k[:70]}` | {len(launches[k])} | {a.get('SQ_LDS_BANK_CONFLICT', 0.0) / a['SQ_LDS_IDX_ACTIVE']:.4f} |")
    open(os.path.join(here, f"{tag}_lds_conflicts.md"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
