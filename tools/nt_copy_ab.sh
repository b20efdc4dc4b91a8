#!/bin/bash
# heterogeneous_blur gpu: the batch build with non-temporal stores (default) against glibc memcpy (MI_BLUR_HOST_NT_COPY=0),
# and by helper threads.  -> profiles/r03_hosts_e2e.txt (appendix)
A=/root/repo/heterogeneous-opencl-image-processing-engine_amd/apps
cd /tmp
for rep in 1 2 3; do for nt in 0 1; do for b in 35 500; do
  echo -n "NT=$nt gpu batch $b: "
  MI_BLUR_HOST_NT_COPY=$nt $A/heterogeneous_blur gpu 1.0 $b --size 256x256 --images 20000 2>&1 | grep "Images per second"
done; done; done
for ht in 2 4 6 8 12; do for b in 35 500; do for rep in 1 2; do
  echo -n "NT=1 host-threads $ht gpu batch $b: "
  $A/heterogeneous_blur gpu 1.0 $b --size 256x256 --images 20000 --host-threads $ht 2>&1 | grep "Images per second"
done; done; done
