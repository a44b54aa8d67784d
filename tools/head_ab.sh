#!/bin/bash
# Development aid: split-path step time with and without the fused descriptor head, alternating in one GPU session
for i in 1 2; do
  for v in "" 1; do   # (fused | unfused)
    BUF_NO_FUSED_HEAD=$v python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('unfused' if '$v' else 'fused  ', 'f32 step', round(d['ms_per_step'],2), 'cyl', round(d['roofline']['avg_us']/1e3,2), '| split step', round(d['ms_per_step_split'],2), 'cyl', round(d['roofline_split']['avg_us']/1e3,2))"
  done
done
