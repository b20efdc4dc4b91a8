#!/bin/bash
# Evidence for profiles/<tag>_e2e_timeline.md: the batch-35 end-to-end stream (pinned host buffers in -> out, 256x256x3, 3x3)
# as one launch per batch (traced timeline) and through the batch server (fixed share vs tickets, workers, batch sizes).
TAG=${1:-r03}
OUT=/root/repo/gpurun_out/$TAG
mkdir -p $OUT
cd /root/repo
export TMPDIR=/tmp
T="timeout -k 10 120"
{
echo "## A. one launch per batch (zero_copy_server=0): rocprofv3 --hip-trace --kernel-trace of 60 submits, 4 and 8 buffer sets"
for s in 4 8; do
  rocprofv3 --hip-trace --kernel-trace --output-format csv -d $OUT/trace_classic_s$s -- python3 tools/e2e_timeline.py 35 $s 60 zero_copy_server=0 2>/dev/null | grep "^batch"
  python3 tools/e2e_timeline_report.py $OUT/trace_classic_s$s/* 60
done
echo
echo "## B. one launch per batch: what does NOT move it (400 submits each)"
for cfg in "4 zero_copy_streams=4 zero_copy_blocks=24" "8 zero_copy_streams=4 zero_copy_blocks=24" "16 zero_copy_streams=8 zero_copy_blocks=12" "16 zero_copy_streams=8 zero_copy_blocks=8" \
           "4 zero_copy_streams=4 zero_copy_blocks=28" "4 zero_copy_events=0" "64" "4 arena=1024"; do
  set -- $cfg; s=$1; shift
  $T python3 tools/e2e_timeline.py 35 $s 400 zero_copy_server=0 "$@" 2>/dev/null | grep "^batch" | cut -d';' -f1
done
echo
echo "## C. batch server, fixed share per worker (zero_copy_tickets=0) vs ticket counter: per-worker trace"
for tk in 0 1; do for b in 35 500; do n=400; [ $b = 500 ] && n=40; $T python3 tools/zc_trace_probe.py $b 4 $n zero_copy_tickets=$tk zero_copy_workers=96 2>/dev/null | grep -v amdgpu; done; done
$T python3 tools/zc_trace_probe.py 35 4 400 2>/dev/null | grep -v amdgpu
echo
echo "## D. batch server: workers (batch 35 / 500), batch sizes, buffer sets"
for wk in 8 16 24 32 48 64 96 128 192 256; do $T python3 tools/e2e_timeline.py 35 4 600 zero_copy_workers=$wk 2>/dev/null | grep "^batch" | cut -d';' -f1; done
for wk in 24 48 64 128; do $T python3 tools/e2e_timeline.py 500 4 40 zero_copy_workers=$wk 2>/dev/null | grep "^batch" | cut -d';' -f1; done
for b in 4 8 16 35 70 140 500; do n=$((14000 / b)); $T python3 tools/e2e_timeline.py $b 4 $n 2>/dev/null | grep "^batch" | cut -d';' -f1; done
for s in 2 3 4 8; do $T python3 tools/e2e_timeline.py 35 $s 600 2>/dev/null | grep "^batch" | cut -d';' -f1; done
echo
echo "## E. same stream, server off vs on, back to back (3 runs each)"
for rep in 1 2 3; do for sv in 0 1; do $T python3 tools/e2e_timeline.py 35 4 600 zero_copy_server=$sv 2>/dev/null | grep "^batch" | cut -d';' -f1; done; done
} > $OUT/e2e_round.txt 2>&1
rm -rf $OUT/trace_classic_s4 $OUT/trace_classic_s8
wc -l $OUT/e2e_round.txt
