#!/usr/bin/env python3
"""Turn what tools/collect_profiles.sh left under gpurun_out/prof/ into the committed profiles/<round>_* files.

    python tools/summarize_profiles.py [round_tag]          (default r02)
"""
import collections, csv, glob, json, os, re, sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(REPO, "gpurun_out", "prof")
DST = os.path.join(REPO, "profiles")
TAG = sys.argv[1] if len(sys.argv) > 1 else "r03"


def short(name):
    """oiva kernel name without namespace and argument list; None for kernels that are not ours"""
    m = re.search(r"oiva::\(anonymous namespace\)::([A-Za-z0-9_]+(?:<[^(]*>)?)\(", name)
    return m.group(1) if m else None


def stats(sub, out):
    rows = []
    for f in glob.glob(os.path.join(SRC, sub, "**", "*_kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Name"])
            if k:
                rows.append([k, r["Calls"], r["TotalDurationNs"], f'{float(r["AverageNs"]):.0f}', r["Percentage"], r["MinNs"], r["MaxNs"]])
    rows.sort(key=lambda r: -float(r[2]))
    with open(os.path.join(DST, out), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kernel", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        w.writerows(rows)
    return rows


def counters(sub):
    """{kernel: {counter: average per launch}} plus launches and average duration under the counters"""
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(SRC, sub, "**", "*_counter_collection.csv"), recursive=True):
        per_dispatch = collections.defaultdict(float)
        names = {}
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if not k:
                continue
            key = (r["Dispatch_Id"], r["Counter_Name"])
            per_dispatch[key] += float(r["Counter_Value"])      # one row per dimension instance
            names[r["Dispatch_Id"]] = k
        for (d, c), v in per_dispatch.items():
            acc[names[d]][c].append(v)
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} | {"launches_per_pass": max(len(v) for v in cs.values())}
            for k, cs in acc.items()}


def main():
    os.makedirs(DST, exist_ok=True)
    for name, out in (("bench_n1.json", "bench_n1.json"), ("bench_cfg5.json", "bench_cfg5.json"),
                      ("bench_sharded_1rank.json", "bench_sharded_1rank.json"), ("bench_m16k2.json", "bench_m16k2.json")):
        path = os.path.join(SRC, name)
        if os.path.exists(path):
            line = [l for l in open(path).read().splitlines() if l.startswith("{")][-1]
            json.dump(json.loads(line), open(os.path.join(DST, f"{TAG}_{out}"), "w"), indent=1)
    stats("stats_headline", f"{TAG}_kernel_stats.csv")                 # the headline workload (all three arithmetic modes)
    stats("stats_configs", f"{TAG}_configs_kernel_stats.csv")          # the same command with its secondary configs
    stats("stats_cfg5", f"{TAG}_cfg5_kernel_stats.csv")
    stats("stats_m16k2", f"{TAG}_m16k2_kernel_stats.csv")
    stats("stats_resident", f"{TAG}_resident_shard8_kernel_stats.csv")
    res = {}
    for sub, what in (("pmc_resident_shard8", "256 x 4000 x 8 / 2 (65.5 MB of X), 50 iterations per launch"),
                      ("pmc_resident_cfg2", "513 x 1000 x 4 / 2 (16.4 MB of X), 50 iterations per launch")):
        for k, cs in counters(sub).items():
            if k.startswith("resident_kernel") and "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
                res[f"{k} on {what}"] = {"FETCH_SIZE_KiB_avg": cs["FETCH_SIZE"], "WRITE_SIZE_KiB_avg": cs["WRITE_SIZE"],
                                         "fetch_bytes_corrected_x2": cs["FETCH_SIZE"] * 2048, "fetch_bytes_uncorrected": cs["FETCH_SIZE"] * 1024,
                                         "write_bytes": cs["WRITE_SIZE"] * 1024, "iterations_per_launch": 50}
    json.dump({"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) around tools/resident_case.py: HBM-side bytes of ONE launch of the "
                       "X-resident kernel = 50 iterations.  X is read once per launch; the rest is the exchange words (8-byte atomic loads "
                       "and stores, for which the x2 FETCH correction of 16 B/lane streaming reads need not hold: both figures are given)",
               "kernels": res}, open(os.path.join(DST, f"{TAG}_pmc_resident_traffic.json"), "w"), indent=1)

    head = counters("pmc_headline")
    traffic = {}
    sq = {}
    for k, cs in head.items():
        if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
            fetch = cs["FETCH_SIZE"] * 1024 * 2          # gfx950: FETCH_SIZE reports half of 16 B/lane streaming reads
            write = cs["WRITE_SIZE"] * 1024
            traffic[k] = {"FETCH_SIZE_KiB_avg": cs["FETCH_SIZE"], "WRITE_SIZE_KiB_avg": cs["WRITE_SIZE"],
                          "fetch_bytes_corrected": fetch, "write_bytes": write, "hbm_bytes_per_launch": fetch + write}
        sq[k] = {c: v for c, v in cs.items() if c not in ("FETCH_SIZE", "WRITE_SIZE")}
    for k, cs in counters("pmc_precise").items():
        if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs and k not in traffic:
            fetch, write = cs["FETCH_SIZE"] * 2048, cs["WRITE_SIZE"] * 1024
            traffic[k + " (precise mode)"] = {"FETCH_SIZE_KiB_avg": cs["FETCH_SIZE"], "WRITE_SIZE_KiB_avg": cs["WRITE_SIZE"],
                                              "fetch_bytes_corrected": fetch, "write_bytes": write,
                                              "hbm_bytes_per_launch": fetch + write}
    json.dump({"note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (each with --kernel-trace only), "
                       "python3 bench.py --steps 6 --warmup 2 --no-cpu --no-other-mode --graph 0 (tools/collect_profiles.sh), "
                       "MI355X gfx950; averages per launch",
               "units": "FETCH_SIZE and WRITE_SIZE are reported in KiB",
               "calibration": "gfx950 FETCH_SIZE reports half of the bytes of 16 B/lane streaming reads (MI355X_MICROARCH.md, "
                              "HBM section; calibrated in round 1 on a known 524 288 000-byte read: factor 1.99998)",
               "kernels": traffic}, open(os.path.join(DST, f"{TAG}_pmc_hbm_traffic.json"), "w"), indent=1)
    json.dump({"note": "rocprofv3 --pmc <4 counters per pass> --kernel-trace, python3 bench.py --steps 6 --warmup 2 --no-cpu "
                       "--no-other-mode --graph 0 (fast mode, headline shape 2048 x 4000 x 8 / 2; tools/collect_profiles.sh), "
                       "MI355X gfx950.  Averages per launch, summed over the chip; SQ *_CYCLES / ACTIVE / WAIT counters are "
                       "in units of 4 shader cycles.",
               "kernels": sq}, open(os.path.join(DST, f"{TAG}_pmc_sq_counters.json"), "w"), indent=1)
    cfg5 = counters("pmc_cfg5")
    for k, cs in cfg5.items():
        if "FETCH_SIZE" in cs:
            cs["fetch_bytes_corrected"] = cs["FETCH_SIZE"] * 2048
        if "WRITE_SIZE" in cs:
            cs["write_bytes"] = cs["WRITE_SIZE"] * 1024
    json.dump({"note": "rocprofv3 --pmc (4 counters per pass) --kernel-trace, python3 bench.py --config cfg5 --steps 6 --warmup 2 "
                       "--no-cpu --no-other-mode --no-configs --graph 0 (default arithmetic of that shape: mixed; "
                       "tools/collect_profiles.sh), MI355X; averages per launch, summed over the chip.  SQ *_CYCLES / ACTIVE / WAIT "
                       "counters are in units of 4 shader cycles; SQ_ACTIVE_INST_VALU x 4 / 1024 SIMDs against GRBM_GUI_ACTIVE / 8 "
                       "XCDs is the fraction of the launch the vector ALU is busy.",
               "kernels": cfg5}, open(os.path.join(DST, f"{TAG}_cfg5_pmc.json"), "w"), indent=1)
    m16 = counters("pmc_m16k2")
    for k, cs in m16.items():
        if "FETCH_SIZE" in cs:
            cs["fetch_bytes_corrected"] = cs["FETCH_SIZE"] * 2048
        if "WRITE_SIZE" in cs:
            cs["write_bytes"] = cs["WRITE_SIZE"] * 1024
    if m16:
        json.dump({"note": "rocprofv3 --pmc (one pass per counter group) --kernel-trace, python3 bench.py --config m16k2 --steps 6 "
                           "--warmup 2 --no-cpu --no-other-mode --no-configs --graph 0 (2048 x 4000 x 16 / 2, mixed mode, 8 frame "
                           "splits; tools/collect_profiles.sh), MI355X; averages per launch, summed over the chip.  SQ *_CYCLES / "
                           "ACTIVE / WAIT counters are in units of 4 shader cycles; GRBM_GUI_ACTIVE / 8 XCDs = shader cycles of the "
                           "launch.", "kernels": m16}, open(os.path.join(DST, f"{TAG}_m16k2_pmc.json"), "w"), indent=1)
    for k, cs in cfg5.items():
        if "SQ_VALU_MFMA_BUSY_CYCLES" in cs and cs.get("GRBM_GUI_ACTIVE"):
            print(f"{k}: matrix pipes busy {cs['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / (cs['GRBM_GUI_ACTIVE'] / 8):.2f}")
    for k, t in traffic.items():
        print(f"{k}: HBM bytes per launch {t['hbm_bytes_per_launch'] / 1e6:.1f} MB")
    # the tools' own logs, as they are (minus the loader's complaint about a file the image lacks)
    for name in ("shard_step.log", "launch_cost.log", "resident_trace_shard8.log", "resident_trace_shard8_loopback8.log",
                 "resident_configs.log", "cfg5_cov_sweep.log", "e2e_phases.log", "e2e_host.log", "stage_times_reference_shapes.log",
                 "shard_fused_ab.log", "sweep_shape_splits.log"):
        path = os.path.join(SRC, name)
        if os.path.exists(path):
            lines = [l for l in open(path).read().splitlines() if "amdgpu.ids" not in l]
            open(os.path.join(DST, f"{TAG}_{name}"), "w").write("\n".join(lines) + "\n")
    # the achieved parity errors of every end-to-end row (tests/test_gpu_parity.py with $OIVA_PARITY_LOG)
    pj = os.path.join(SRC, "parity.jsonl")
    if os.path.exists(pj):
        import subprocess
        md = subprocess.run([sys.executable, os.path.join(REPO, "tools", "parity_table.py"), pj], capture_output=True, text=True, check=True).stdout
        open(os.path.join(DST, f"{TAG}_parity_errors.md"), "w").write(md)


if __name__ == "__main__":
    main()
