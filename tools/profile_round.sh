#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box (run through gpurun):
#   tools/profile_round.sh <tag>   ->  gpurun_out/prof_<tag>/...
# then, back in the build container:  python tools/summarize_profiles.py <tag>   (copies summaries to profiles/)
#
# For each bench configuration: (0) the plain bench line, (1) kernel trace + stats of the same command,
# (2)+(3) separate PMC passes (FETCH_SIZE, WRITE_SIZE) with kernel-trace only, as MI355X_MICROARCH.md prescribes.
set -u
TAG=${1:-r03}
PART=${2:-all}          # all | a (the a1 forms) | b (hd5, a2_1gpu, counter passes): two gpurun calls when one would be too long
OUT=/root/repo/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /root/repo
export TMPDIR=/tmp

run_cfg() {   # name, bench args...
    local name=$1; shift
    python3 bench.py "$@" > $OUT/bench_$name.json 2> $OUT/bench_$name.err
    echo "$name plain rc=$?"
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$name -- python3 bench.py "$@" --no-cpu-baseline > $OUT/bench_${name}_traced.json 2> $OUT/trace_$name.err
    echo "$name trace rc=$?"
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_$name -- python3 bench.py "$@" --no-cpu-baseline --no-extra --ramp-seconds 0.05 --steps 2 --warmup 1 > /dev/null 2> $OUT/fetch_$name.err
    echo "$name fetch rc=$?"
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write_$name -- python3 bench.py "$@" --no-cpu-baseline --no-extra --ramp-seconds 0.05 --steps 2 --warmup 1 > /dev/null 2> $OUT/write_$name.err
    echo "$name write rc=$?"
}

if [ $PART != b ]; then
run_cfg a1                                                                  # the judged command: python bench.py (fused stream + every secondary point)
run_cfg a1_batched --dispatch batched --no-cpu-baseline --no-extra           # one launch per batch, 4 streams
run_cfg a1_serial --dispatch batched --streams 1 --time-every 4 --no-cpu-baseline --no-extra   # same launches, one stream: dispatches do not overlap
run_cfg a1_one_launch --batch 5000 --streams 1 --time-every 1 --no-cpu-baseline --no-extra   # whole stream in one launch (HBM-bound point)
fi
if [ $PART != a ]; then
run_cfg hd5 --workload hd5 --streams 1 --no-cpu-baseline                    # BASELINE configs[2]: 1920x1080 5x5
run_cfg a2_1gpu --workload a2 --no-cpu-baseline                             # BASELINE configs[4] at N=1: 8192x8192 3x3

# SQ / LDS / TCC counters of the fused stream kernel (the headline's dominant kernel) and of the 1080p 5x5 launch:
# separate --pmc passes, kernel-trace only
pmc() {   # name, counters..., then "--", bench args
    local name=$1; shift
    local ctrs=()
    while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done
    shift
    rocprofv3 --kernel-trace --pmc "${ctrs[@]}" --output-format csv -d $OUT/pmc_$name -- python3 bench.py "$@" --no-cpu-baseline --no-extra --steps 3 --warmup 1 --ramp-seconds 0.05 > /dev/null 2> $OUT/pmc_$name.err
    echo "pmc $name rc=$?"
}
for w in a1 hd5; do
    extra=""; [ $w = hd5 ] && extra="--workload hd5"
    pmc ${w}_sq SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU -- $extra
    pmc ${w}_lds SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS -- $extra
    pmc ${w}_tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum -- $extra
done
fi
# the per-dispatch traces are large (the whole round would not fit gpurun's 64 MiB return) and nothing downstream reads them:
# summarize_profiles.py works from the *_kernel_stats.csv and *_counter_collection.csv files
find $OUT -name "*_kernel_trace.csv" -delete
echo "done"; du -sh $OUT
