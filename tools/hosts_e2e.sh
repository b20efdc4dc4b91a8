#!/bin/bash
# The host app in gpu mode by number of buffer sets / helper threads / images (profiles/<tag>_e2e*.txt):  tools/hosts_e2e.sh
A=/root/repo/heterogeneous-opencl-image-processing-engine_amd/apps
cd /tmp
for b in 35 500; do
  for cfg in "4 4" "4 8" "4 12" "6 8" "8 8"; do
    set -- $cfg
    for rep in 1 2; do
      echo -n "gpu batch $b slots $1 host-threads $2: "
      $A/heterogeneous_blur gpu 1.0 $b --size 256x256 --images 20000 --slots $1 --host-threads $2 2>&1 | grep "Images per second"
    done
  done
done
echo -n "both auto batch 35: "; $A/heterogeneous_blur both auto 35 --size 256x256 --images 20000 2>&1 | grep "Images per second"
echo -n "split 0.837 batch 35: "; $A/split_image_blur 0.837 35 --size 320x240 --images 20000 2>&1 | grep "Images per second"
MI_BLUR_NO_AFFINITY=1 $A/heterogeneous_blur gpu 1.0 35 --size 256x256 --images 20000 --host-threads 8 2>&1 | grep "Images per second\|placement" | tr '\n' ' '; echo "(MI_BLUR_NO_AFFINITY=1)"
