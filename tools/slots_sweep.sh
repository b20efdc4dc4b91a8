#!/bin/bash
# Buffer sets in flight (--slots) by host mode, with the batch server: heterogeneous_blur gpu / both auto, split_image_blur.
A=/root/repo/heterogeneous-opencl-image-processing-engine_amd/apps
for rep in 1 2 3 4; do
  for s in 2 3 4; do
    echo -n "rep $rep slots $s | gpu 35: ";  $A/heterogeneous_blur gpu 1.0 35 --size 256x256 --images 20000 --slots $s 2>&1 | grep "Images per second" | tr -d '\n'
    echo -n " | gpu 500: "; $A/heterogeneous_blur gpu 1.0 500 --size 256x256 --images 20000 --slots $s 2>&1 | grep "Images per second" | tr -d '\n'
    echo -n " | both auto 35: "; $A/heterogeneous_blur both auto 35 --size 256x256 --images 20000 --slots $s 2>&1 | grep "Images per second" | tr -d '\n'
    echo -n " | both 0.728 35: "; $A/heterogeneous_blur both 0.728 35 --size 256x256 --images 20000 --slots $s 2>&1 | grep "Images per second" | tr -d '\n'
    echo -n " | split 0.837 35: "; $A/split_image_blur 0.837 35 --size 320x240 --images 20000 --slots $s 2>&1 | grep "Images per second" | tr -d '\n'
    echo
  done
done
