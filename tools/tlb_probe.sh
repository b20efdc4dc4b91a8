#!/bin/bash
# Developer: do the fast / slow allocations differ in address-translation misses?  ab_alloc (1080p 5x5, direct kernel, 6 fresh
# allocations) under rocprofv3 --pmc; the per-dispatch counters are then folded per allocation (60 dispatches each... see below).
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/tlb_probe; rm -rf $OUT; mkdir -p $OUT
python3 tools/ab_alloc.py --shape hd5 --opts "prefer_direct=2" --allocs 6 --burst 20 > $OUT/plain.txt 2>&1
rocprofv3 --kernel-trace --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum --output-format csv -d $OUT/pmc -- python3 tools/ab_alloc.py --shape hd5 --opts "prefer_direct=2" --allocs 6 --burst 20 > $OUT/pmc.txt 2>&1
rocprofv3 --kernel-trace --pmc GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc2 -- python3 tools/ab_alloc.py --shape hd5 --opts "prefer_direct=2" --allocs 6 --burst 20 > $OUT/pmc2.txt 2>&1
grep -h "hd5 " $OUT/plain.txt $OUT/pmc.txt $OUT/pmc2.txt
python3 - <<'PY'
import csv, glob, collections
for d in ("pmc", "pmc2"):
    f = sorted(glob.glob(f"gpurun_out/tlb_probe/{d}/*/*_counter_collection.csv"))
    if not f: print(d, "no csv"); continue
    rows = [r for r in csv.DictReader(open(f[-1])) if "blur_direct" in r["Kernel_Name"]]
    by = collections.defaultdict(list)
    for r in rows: by[r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    for c, v in by.items():
        v.sort()
        n = len(v); per = n // 6
        print(d, c, "dispatches", n, "mean per allocation:", " ".join(f"{sum(x for _, x in v[i*per:(i+1)*per]) / max(per,1):12.0f}" for i in range(6)))
PY
