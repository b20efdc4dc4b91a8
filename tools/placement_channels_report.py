#!/usr/bin/env python3
"""Fold a `rocprofv3 --kernel-trace --pmc <TCC counters> --output-format json` run of tools/placement_probe.py into
per-allocation figures: kernel duration and, per counter, the distribution over the TCC instances (channels x XCCs).
    python3 tools/placement_channels_report.py <results.json> <launches per allocation>"""
import json
import sys
from collections import defaultdict

import numpy as np


def main():
    d = json.load(open(sys.argv[1]))["rocprofiler-sdk-tool"][0]
    per = int(sys.argv[2])
    names = {}
    for c in d["counters"]:
        names[c["id"]["handle"]] = c["name"]
    kern = {k["kernel_id"]: k.get("formatted_kernel_name", k.get("kernel_name", "?")) for k in d["kernel_symbols"]}
    rows = []
    for rec in d["callback_records"]["counter_collection"]:
        dd = rec["dispatch_data"]
        name = kern.get(dd["dispatch_info"]["kernel_id"], "?")
        if "blur_direct_kernel" not in name and "blur_tiled_kernel" not in name:
            continue
        vals = defaultdict(list)
        for r in rec["records"]:
            vals[names.get(r["counter_id"]["handle"], str(r["counter_id"]["handle"]))].append(r["value"])
        rows.append((dd["dispatch_info"]["dispatch_id"], (dd["end_timestamp"] - dd["start_timestamp"]) / 1e3, vals))
    rows.sort()
    n_alloc = len(rows) // per
    print(f"{len(rows)} blur dispatches = {n_alloc} allocations x {per} launches; counters: {sorted(rows[0][2])}; instances per counter: "
          f"{len(next(iter(rows[0][2].values())))}")
    summary = []
    for a in range(n_alloc):
        grp = rows[a * per + per // 4:(a + 1) * per]           # drop each allocation's first quarter (clock ramp after the fill)
        us = float(np.median([g[1] for g in grp]))
        line = f"allocation {a}: kernel {us:7.1f} us (median of {len(grp)})"
        stats = {}
        for cname in sorted(grp[0][2]):
            m = np.array([g[2][cname] for g in grp], dtype=np.float64).mean(axis=0)       # per instance, mean over launches
            stats[cname] = m
            line += f" | {cname}: sum {m.sum():.3e} max/mean {m.max() / max(m.mean(), 1e-9):.3f} min/mean {m.min() / max(m.mean(), 1e-9):.3f}"
        print(line)
        summary.append((us, stats))
    # fast vs slow: per-instance ratio
    order = sorted(range(n_alloc), key=lambda i: summary[i][0])
    fast, slow = order[0], order[-1]
    print(f"\nfastest allocation {fast} ({summary[fast][0]:.1f} us) vs slowest {slow} ({summary[slow][0]:.1f} us):")
    for cname in sorted(summary[fast][1]):
        f, s = summary[fast][1][cname], summary[slow][1][cname]
        with np.errstate(divide="ignore", invalid="ignore"):
            ratio = np.where(f > 0, s / f, np.nan)
        print(f"  {cname}: total slow/fast {s.sum() / max(f.sum(), 1e-9):.3f}; per-instance slow/fast min {np.nanmin(ratio):.3f} median {np.nanmedian(ratio):.3f} "
              f"max {np.nanmax(ratio):.3f}; coefficient of variation over instances fast {f.std() / max(f.mean(), 1e-9):.4f} slow {s.std() / max(s.mean(), 1e-9):.4f}")
        if len(f) % 16 == 0:                                    # fold to 16 channels (sum over XCCs) and to XCCs (sum over channels)
            nx = len(f) // 16
            for label, fa, sa in (("per XCC", f.reshape(nx, 16).sum(axis=1), s.reshape(nx, 16).sum(axis=1)),
                                  ("per channel", f.reshape(nx, 16).sum(axis=0), s.reshape(nx, 16).sum(axis=0))):
                print(f"      {label}: fast " + " ".join(f"{v / max(fa.mean(), 1e-9):.3f}" for v in fa))
                print(f"      {label}: slow " + " ".join(f"{v / max(sa.mean(), 1e-9):.3f}" for v in sa))


if __name__ == "__main__":
    main()
