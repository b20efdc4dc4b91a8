#!/bin/bash
# The hosts over the reference's own batch sizes (data/approach{1,2}: 35 50 100 200 500 800 1200) and the extremes, 320x240x3, 5000 images.
A=/root/repo/heterogeneous-opencl-image-processing-engine_amd/apps
cd /tmp
for b in 1 8 35 50 100 200 500 800 1200 5000; do
  echo -n "batch $b | gpu: "; $A/heterogeneous_blur gpu 1.0 $b --size 320x240 2>&1 | grep "Images per second" | tr -d '\n' | sed 's/   Images per second: //'
  echo -n " | both 0.728: "; $A/heterogeneous_blur both 0.728 $b --size 320x240 2>&1 | grep "Images per second" | tr -d '\n' | sed 's/   Images per second: //'
  echo -n " | cpu: "; $A/heterogeneous_blur cpu 0.5 $b --size 320x240 2>&1 | grep "Images per second" | tr -d '\n' | sed 's/   Images per second: //'
  echo -n " | split 0.837: "; $A/split_image_blur 0.837 $b --size 320x240 2>&1 | grep "Images per second" | tr -d '\n' | sed 's/   Images per second: //'
  echo
done
