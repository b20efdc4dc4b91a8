#!/usr/bin/env python3
"""Turn gpurun_out/prof_<tag>/ (written by tools/profile_round.sh on the GPU box) into the committed
evidence under profiles/:

  profiles/<tag>_<cfg>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary of `python bench.py <cfg args>`
  profiles/<tag>_<cfg>_bench.json         the plain (un-profiled) bench line of the same command
  profiles/<tag>_summary.md               one table: bench avg launch vs rocprof avg, PMC traffic vs algorithmic bytes
  profiles/traffic.json                   HBM bytes per launch per workload (read back by bench.py's roofline.traffic)

HBM bytes follow MI355X_MICROARCH.md §HBM: FETCH_SIZE and WRITE_SIZE are collected in separate passes, are in KiB,
and on gfx950 FETCH_SIZE reports exactly half of a wide coalesced streaming read -> traffic = (2*FETCH + WRITE) KiB.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))



def newest(pattern):
    files = sorted(glob.glob(pattern), key=os.path.getmtime)
    return files[-1] if files else None


def rows(pattern):
    f = newest(pattern)      # gpurun merges new runs into the same directory: take the latest
    return list(csv.DictReader(open(f))) if f else []


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    traffic_path = os.path.join(dst, "traffic.json")
    traffic = json.load(open(traffic_path)) if os.path.exists(traffic_path) else {}
    lines = [f"# rocprofv3 evidence, {tag}", "",
             "| config (bench args) | bench value | bench avg launch µs (dispatch timestamps) | rocprof avg µs (calls) | "
             "alg. bytes/launch | PMC HBM bytes/launch (2·FETCH+WRITE) | PMC/alg | achieved GB/s | frac of 8 TB/s |",
             "|---|---|---|---|---|---|---|---|---|"]
    workload_of = {"a1": "a1", "a1_batched": "a1_batched", "a1_serial": "a1_serial", "a1_one_launch": "a1_one_launch", "hd5": "hd5",
                   "a2_1gpu": "a2"}
    for cfg in ("a1", "a1_batched", "a1_serial", "a1_one_launch", "hd5", "a2_1gpu"):
        bj = os.path.join(src, f"bench_{cfg}.json")
        if not os.path.exists(bj):
            continue
        text = [l for l in open(bj).read().splitlines() if l.startswith("{")]
        if not text:
            continue
        bench = json.loads(text[-1])
        KERNEL = bench["roofline"].get("kernel", "blur_tiled_kernel")       # blur_fused_kernel for the fused stream
        shutil.copy(bj, os.path.join(dst, f"{tag}_{cfg}_bench.json"))
        stats = rows(os.path.join(src, f"trace_{cfg}", "*", "*_kernel_stats.csv"))
        f = newest(os.path.join(src, f"trace_{cfg}", "*", "*_kernel_stats.csv"))
        if f:
            shutil.copy(f, os.path.join(dst, f"{tag}_{cfg}_kernel_stats.csv"))
        srow = [r for r in stats if KERNEL in r["Name"]]
        srow.sort(key=lambda r: -float(r["TotalDurationNs"]))
        roc = f"{float(srow[0]['AverageNs']) / 1e3:.2f} ({srow[0]['Calls']})" if srow else "n/a"
        # PMC: mean per dispatch of the dominant grid size
        pm = {}
        for cnt, sub in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
            rr = [r for r in rows(os.path.join(src, f"{sub}_{cfg}", "*", "*_counter_collection.csv"))
                  if r["Counter_Name"] == cnt and KERNEL in r["Kernel_Name"]]
            by_grid = collections.defaultdict(list)
            for r in rr:
                by_grid[r["Grid_Size"]].append(float(r["Counter_Value"]))
            if by_grid:
                grid = max(by_grid, key=lambda g: len(by_grid[g]))
                pm[cnt] = sum(by_grid[grid]) / len(by_grid[grid])
        rf = bench["roofline"]
        alg = rf["algorithmic_bytes_per_launch"]
        if len(pm) == 2:
            hbm = (2 * pm["FETCH_SIZE"] + pm["WRITE_SIZE"]) * 1024
            traffic[workload_of[cfg]] = {"hbm_bytes_per_launch": round(hbm), "fetch_kib": pm["FETCH_SIZE"], "write_kib": pm["WRITE_SIZE"],
                                         "note": "2*FETCH_SIZE + WRITE_SIZE KiB per dispatch (gfx950 FETCH_SIZE = 1/2 of streamed bytes)", "tag": tag}
            hb, ratio = f"{hbm:,.0f}", f"{hbm / alg:.3f}"
        else:
            hb, ratio = "n/a", "n/a"
        lines.append(f"| `{cfg}` | {bench['value']:,.0f} {bench['unit']} | {rf['avg_launch_us']} | {roc} | {alg:,} | {hb} | {ratio} | "
                     f"{rf['achieved']} | {rf['frac']} |")
    ex = os.path.join(src, "bench_a1_extra.json")
    if os.path.exists(ex):
        shutil.copy(ex, os.path.join(dst, f"{tag}_a1_extra_bench.json"))
    # SQ / LDS / TCC counter passes (pmc_<workload>_<group>): mean per dispatch, per kernel and grid size
    out = [f"rocprofv3 --pmc passes of `python bench.py [--workload hd5] --no-extra --steps 3 --warmup 1`, {tag}: mean per dispatch",
           "(SQ_* cycle counters are in quad-cycles summed over the chip; FETCH/TCC per MI355X_MICROARCH.md)", ""]
    for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
        if not os.path.isdir(d):
            continue
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in rows(os.path.join(d, "*", "*_counter_collection.csv")):
            if "blur_" not in r["Kernel_Name"]:
                continue
            key = (r["Kernel_Name"].split("(")[0].replace("void mi_blur::", "")[:48], r["Grid_Size"])
            acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for (k, g), cs in sorted(acc.items()):
            n = max(len(v) for v in cs.values())
            out.append(f"{os.path.basename(d):12s} {k:48s} grid {g:>9s} x{n:<3d} " + "  ".join(f"{c}={sum(v) / len(v):.0f}" for c, v in sorted(cs.items())))
    if len(out) > 3:
        open(os.path.join(dst, f"{tag}_pmc_counters.txt"), "w").write("\n".join(out) + "\n")
    open(os.path.join(dst, f"{tag}_summary.md"), "w").write("\n".join(lines) + "\n")
    json.dump(traffic, open(traffic_path, "w"), indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
