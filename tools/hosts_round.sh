#!/bin/bash
# Run the C++ hosts in their reference command-line modes on the GPU box and keep the report tails (profiles/<tag>_hosts.txt).
OUT=/root/repo/gpurun_out/hosts_$1.txt
A=/root/repo/heterogeneous-opencl-image-processing-engine_amd/apps
cd /tmp
{
echo "# C++ hosts on the MI355X box ($(nproc) host threads visible), synthetic 256x256x3 / 320x240x3 streams of 5000 images"
for cmd in "heterogeneous_blur cpu 0.5 35 --size 256x256" "heterogeneous_blur cpu 0.5 35 --size 256x256 --threads 1" \
           "heterogeneous_blur gpu 1.0 35 --size 256x256" "heterogeneous_blur gpu 1.0 500 --size 256x256" \
           "heterogeneous_blur both auto 35 --size 256x256" "heterogeneous_blur both 0.728 35 --size 320x240" \
           "heterogeneous_blur gpu 1.0 35 --size 256x256 --resident" "heterogeneous_blur gpu 1.0 35 --size 256x256 --images 50000 --resident --fused" "heterogeneous_blur gpu 1.0 35 --size 1920x1080 --ksize 5 --images 500" \
           "split_image_blur 0.837 35 --size 320x240" "split_image_blur --resident --gpus 1 --size 8192x8192 --iters 50"; do
  echo; echo "\$ $cmd"
  $A/$cmd 2>&1 | grep -E "Mode:|CPU device|Auto-cal|wall-clock|processed|Transfer|Kernel exec|imbalance|Images per second|Megapixels|Recommended|Launches|Kernel-only|Per-dispatch|Host link|EQUALS|per image|Algorithmic bandwidth"
done
} > $OUT 2>&1
tail -5 $OUT
