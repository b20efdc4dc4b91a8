#!/bin/bash
# round-2 A/B batch 1: correctness of the new variants, then sustained (burst) timings
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "experiment or updown or stream_kernel_bit_exact" > gpurun_out/ab1_pytest.log 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/ab1_pytest.log
tail -3 gpurun_out/ab1_pytest.log
{
echo "## experiment (tiled row pass variants), sustained bursts of 100 launches"
timeout -k 10 300 python tools/kbench.py --shape hd5,hd3,a1one,big1 --reps 5 --burst 60 --opts "experiment=0,1,2,3"
echo "## rows per thread, sustained"
timeout -k 10 300 python tools/kbench.py --shape hd5,hd3,a1one,big1 --reps 3 --burst 60 --opts "rows_per_thread=4,8;experiment=0,1"
echo "## stream variant: updown x band rows, sustained"
timeout -k 10 300 python tools/kbench.py --shape hd5,hd3,big1 --reps 3 --burst 60 --opts "prefer_stream=1;stream_updown=0,1;stream_band_rows=0,64,100,128"
echo "## isolated launches for reference"
timeout -k 10 300 python tools/kbench.py --shape hd5,a1one --reps 7 --burst 1 --opts "experiment=0,1"
} > gpurun_out/ab1_kbench.txt 2>&1
tail -60 gpurun_out/ab1_kbench.txt
