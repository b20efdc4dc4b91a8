export TMPDIR=/tmp
mkdir -p gpurun_out/r03
out=gpurun_out/r03/e2e_matrix.txt
: > $out
for cfg in "4 4 24" "8 4 24" "8 8 12" "16 8 12" "12 6 16" "16 8 16" "16 8 8" "8 8 24" "16 8 6"; do
  set -- $cfg
  python3 tools/e2e_timeline.py 35 $1 400 zero_copy_streams=$2 zero_copy_blocks=$3 2>&1 | grep "^batch" >> $out
done
for cfg in "4 4 28" "4 4 20" "8 4 28" "4 4 35" "4 4 40" "8 8 14" "8 8 10"; do
  set -- $cfg
  python3 tools/e2e_timeline.py 35 $1 400 zero_copy_streams=$2 zero_copy_blocks=$3 2>&1 | grep "^batch" >> $out
done
cat $out | cut -c1-140
