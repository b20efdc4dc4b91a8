#!/bin/bash
# Collect the rocprofv3 evidence for bench.py on the GPU box (run through gpurun).
#   tools/profile_round.sh <tag>      -> gpurun_out/prof_<tag>/{trace,fetch,write,sq}/...
set -u
TAG=${1:-r01}
OUT=/root/repo/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /root/repo
export TMPDIR=/tmp
# 1. kernel trace + stats of the judged command
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --no-cpu-baseline > $OUT/bench_under_trace.json 2> $OUT/trace.err
echo "trace rc=$?"
# 2. PMC passes (separate runs, counters only + kernel-trace): HBM traffic of the headline workload
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra > $OUT/bench_fetch.json 2> $OUT/fetch.err
echo "fetch rc=$?"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra > $OUT/bench_write.json 2> $OUT/write.err
echo "write rc=$?"
# 3. same two counters for the HBM-bound points (one launch over 5000 images; 1080p 5x5; 8192^2)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_big -- python3 tools/kbench.py --shape a1one,hd5,big --reps 2 --opts "stage_dma=1;rows_per_thread=8;xcd_remap=1" > $OUT/kbench_fetch.log 2> $OUT/fetch_big.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write_big -- python3 tools/kbench.py --shape a1one,hd5,big --reps 2 --opts "stage_dma=1;rows_per_thread=8;xcd_remap=1" > $OUT/kbench_write.log 2> $OUT/write_big.err
# 4. SQ counters for diagnosis
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $OUT/sq -- python3 tools/kbench.py --shape a1,a1one,hd5,big --reps 2 --opts "stage_dma=1;rows_per_thread=8;xcd_remap=1" > $OUT/kbench_sq.log 2> $OUT/sq.err
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM --output-format csv -d $OUT/sq2 -- python3 tools/kbench.py --shape a1,a1one,hd5,big --reps 2 --opts "stage_dma=1;rows_per_thread=8;xcd_remap=1" > $OUT/kbench_sq2.log 2> $OUT/sq2.err
echo "done"; find $OUT -name "*.csv" | head -40; du -sh $OUT
