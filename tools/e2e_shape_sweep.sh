#!/bin/bash
# The pinned host -> host stream (batch server) over frame sizes and batch sizes: GB/s each way.  Cliff hunting off the BASELINE shape.
cd /root/repo
run() { timeout -k 10 120 python3 tools/e2e_timeline.py "$@" 2>/dev/null | grep "^batch" | cut -d";" -f1 | sed 's/ submits in .* per submit,/:/'; }
for shape in 64x64 250x250 256x256 320x240 640x480 1366x768 1920x1080 3840x2160; do
  px=$(( ${shape%x*} * ${shape#*x} ))
  for b in 1 8 35 500; do
    [ $(( px * 3 * b )) -gt 400000000 ] && continue
    n=$(( 60000000 / (px * 3 * b) + 8 )); [ $n -gt 2000 ] && n=2000
    for r in 1 2; do run $b 4 $n shape=$shape radius=$r; done
  done
done
