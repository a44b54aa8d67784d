#!/bin/bash
# tools/final_round.sh without the -m gpu suite (when that has just run on the same tree)
set -x
tools/profile_round.sh r04 > gpurun_out/r04_profile.log 2>&1
tail -3 gpurun_out/r04_profile.log
python tests/eval_recall.py --backend gpu --out gpurun_out/r04/recall_gpu.json > gpurun_out/r04/recall_gpu.log 2>&1
python tests/eval_ransac_rr.py --out gpurun_out/r04/ransac_rr.json > gpurun_out/r04/ransac_rr.log 2>&1
python tests/eval_ransac_rr.py --overlaps 0.35,0.3,0.25,0.2 --out gpurun_out/r04/ransac_rr_low_overlap.json > gpurun_out/r04/ransac_rr_low.log 2>&1
