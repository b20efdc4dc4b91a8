mkdir -p gpurun_out/r03
for rep in 1 2 3; do for t in 0 30; do
  MI_BLUR_FUSED_TAIL=$t timeout -k 10 300 python bench.py --no-extra --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('tail=$t', d['value'], d['roofline']['avg_launch_us'], d['roofline']['frac'], d['roofline']['kernel'], d['parity']['status'])"
done; done
