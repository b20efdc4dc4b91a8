#!/bin/bash
# heterogeneous_blur --frames on the GPU box: N distinct 256x256 PPM frames in tmpfs, by helper threads / batch / output form.
# (profiles/<tag>_frames.txt)   tools/frames_round.sh [n_frames]
N=${1:-4000}
A=/root/repo/heterogeneous-opencl-image-processing-engine_amd/apps
D=/dev/shm/mi_blur_frames_$$
mkdir -p $D/in
python3 - <<PY
import numpy as np, os
rng = np.random.default_rng(1)
for i in range($N):
    with open("$D/in/f_%05d.ppm" % i, "wb") as f:
        f.write(b"P6\n256 256\n255\n"); f.write(rng.integers(0, 256, 256 * 256 * 3, dtype=np.uint8).tobytes())
PY
echo "# $N distinct 256x256x3 PPM frames in tmpfs ($(du -sh $D/in | cut -f1)); heterogeneous_blur gpu 1.0 B --frames DIR [--planar-out] [--save-dir DIR]"
cd /tmp
for b in 35 140; do
  for t in 4 8 16; do
    for extra in "" "--planar-out" "--native-layout"; do
      echo "## batch $b, $t helper threads $extra"
      $A/heterogeneous_blur gpu 1.0 $b --frames $D/in --host-threads $t $extra 2>&1 | grep -E "Decode:|GPU repack-in|GPU blur|GPU copy-out|blocked|Ingest-inclusive"
    done
  done
done
echo "## batch 35, 8 helper threads, saving every frame"
$A/heterogeneous_blur gpu 1.0 35 --frames $D/in --host-threads 8 --save-dir $D/out 2>&1 | grep -E "Decode:|Save:|blocked|Ingest-inclusive"
echo "## cpu device, batch 35, 8 helper threads"
$A/heterogeneous_blur cpu 0.5 35 --frames $D/in --host-threads 8 2>&1 | grep -E "Decode:|CPU repack|blocked|Ingest-inclusive"
rm -rf $D
