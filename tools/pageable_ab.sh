#!/bin/bash
# Submits of PAGEABLE caller memory (the reference's malloc'd batch buffers, heterogeneous_blur.c:431-432, kept as they are), 256x256x3:
# the batch server on the slot's pinned staging ("staged_server" 1, default) against a DMA copy each way around a launch (0), by
# buffer sets and by gather/scatter threads (MI_BLUR_STAGING_THREADS; default 8); pinned and registered buffers beside them.
cd /root/repo
run() { timeout -k 10 120 python3 tools/e2e_timeline.py "$@" 2>/dev/null | grep "^batch" | cut -d";" -f1; }
for rep in 1 2; do
  for b in 35 500; do n=600; [ $b = 500 ] && n=40
    for sl in 2 4; do
      echo -n "DMA staging      : "; run $b $sl $n pageable=1 staged_server=0
      echo -n "server on staging: "; run $b $sl $n pageable=1
    done
    echo -n "registered       : "; run $b 4 $n pageable=2
    echo -n "pinned           : "; run $b 4 $n
  done
done
for th in 2 4 8 12; do for b in 35 500; do n=600; [ $b = 500 ] && n=40
  echo -n "server on staging, $th threads: "; MI_BLUR_STAGING_THREADS=$th run $b 4 $n pageable=1
done; done
