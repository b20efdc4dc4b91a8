#!/usr/bin/env python3
"""Fold a `rocprofv3 --hip-trace --kernel-trace` run of tools/e2e_timeline.py into a per-submit timeline summary:

    python tools/e2e_timeline_report.py gpurun_out/r03/e2e_trace35/runc [n_submits]

Looks at the LAST n_submits zero-copy dispatches (the timed loop of the probe): how many are executing at any instant,
per-stream gaps between one dispatch's end and the next one's start, dispatch durations, and the host time inside every
HIP API called in that window (per submit)."""
import collections
import csv
import glob
import sys


def main():
    d = sys.argv[1]
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    api = list(csv.DictReader(open(glob.glob(d + "/*_hip_api_trace.csv")[0])))
    ker = [r for r in csv.DictReader(open(glob.glob(d + "/*_kernel_trace.csv")[0])) if "blur" in r["Kernel_Name"]]
    ker.sort(key=lambda r: int(r["Start_Timestamp"]))
    k = ker[-n:]
    name = k[0]["Kernel_Name"].split("(")[0]
    t0 = min(int(r["Start_Timestamp"]) for r in k)
    t1 = max(int(r["End_Timestamp"]) for r in k)
    span = t1 - t0
    print(f"kernel {name}, grid {k[0]['Grid_Size_X']} / workgroup {k[0]['Workgroup_Size_X']}; last {n} dispatches span {span / 1e3:.1f} us "
          f"= {span / n / 1e3:.1f} us per submit")
    dur = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in k)
    print(f"dispatch duration us: min {dur[0]:.0f}  median {dur[len(dur) // 2]:.0f}  max {dur[-1]:.0f}")
    # concurrency profile
    ev = []
    for r in k:
        ev.append((int(r["Start_Timestamp"]), 1))
        ev.append((int(r["End_Timestamp"]), -1))
    ev.sort()
    hist = collections.Counter()
    cur, last = 0, t0
    for t, dlt in ev:
        hist[cur] += t - last
        last = t
        cur += dlt
    print("dispatches executing at once (fraction of the window): " + "  ".join(f"{c}: {hist[c] / span * 100:.1f} %" for c in sorted(hist)))
    # per-stream gaps
    by = collections.defaultdict(list)
    for r in k:
        by[r["Stream_Id"]].append(r)
    gaps = []
    for s, rs in by.items():
        for a, b in zip(rs, rs[1:]):
            gaps.append((int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3)
    gaps.sort()
    if gaps:
        print(f"same-stream gap (end of one dispatch -> start of the next) us: min {gaps[0]:.1f}  median {gaps[len(gaps) // 2]:.1f}  max {gaps[-1]:.1f}")
    # start-to-start spacing over all streams
    st = sorted(int(r["Start_Timestamp"]) for r in k)
    sp = sorted((b - a) / 1e3 for a, b in zip(st, st[1:]))
    print(f"start-to-start spacing us: min {sp[0]:.1f}  p10 {sp[len(sp) // 10]:.1f}  median {sp[len(sp) // 2]:.1f}  p90 {sp[9 * len(sp) // 10]:.1f}  max {sp[-1]:.1f}")
    # host API time in the window
    tot = collections.Counter()
    cnt = collections.Counter()
    for r in api:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if s >= t0 - 200000 and e <= t1:
            tot[r["Function"]] += e - s
            cnt[r["Function"]] += 1
    print("host time inside HIP calls, per submit (calls per submit x mean us):")
    for f, t in tot.most_common(12):
        print(f"  {f:32s} {cnt[f] / n:5.2f} x {t / cnt[f] / 1e3:7.2f} us = {t / n / 1e3:7.2f} us")
    print(f"  total {sum(tot.values()) / n / 1e3:.1f} us of HIP calls per submit")


if __name__ == "__main__":
    main()
