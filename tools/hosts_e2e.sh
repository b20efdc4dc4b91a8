#!/bin/bash
# Developer: the C++ host in gpu / both mode, by number of rotating batch-buffer sets (--slots), 256x256 frames
cd /root/repo/heterogeneous-opencl-image-processing-engine_amd/apps
for b in 35 500; do for s in 2 3 4 6; do
  for rep in 1 2; do ./heterogeneous_blur gpu 1.0 $b --size 256x256 --images 20000 --slots $s | grep "Images per second" | sed "s/^/gpu batch $b slots $s: /"; done
done; done
for s in 2 4; do ./heterogeneous_blur both auto 35 --size 256x256 --images 20000 --slots $s | grep "Images per second" | sed "s/^/both auto batch 35 slots $s: /"; done
for s in 2 4; do ./split_image_blur 0.837 35 --size 320x240 --images 5000 --slots $s | grep "Images per second" | sed "s/^/split 0.837 batch 35 slots $s: /"; done
