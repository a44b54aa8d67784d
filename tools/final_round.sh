#!/bin/bash
# The evidence set of a round in one GPU session: the -m gpu suite, tools/profile_round.sh, the recall / RANSAC-RR scripts.
set -x
python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > gpurun_out/r05_tests.log
tools/profile_round.sh r05 > gpurun_out/r05_profile.log 2>&1
tail -3 gpurun_out/r05_profile.log
python tests/eval_recall.py --backend gpu --out gpurun_out/r05/recall_gpu.json > gpurun_out/r05/recall_gpu.log 2>&1
python tests/eval_ransac_rr.py --out gpurun_out/r05/ransac_rr.json > gpurun_out/r05/ransac_rr.log 2>&1
python tests/eval_ransac_rr.py --overlaps 0.35,0.3,0.25,0.2 --out gpurun_out/r05/ransac_rr_low_overlap.json > gpurun_out/r05/ransac_rr_low.log 2>&1
tail -2 gpurun_out/r05/recall_gpu.log gpurun_out/r05/ransac_rr.log gpurun_out/r05/ransac_rr_low.log
cat gpurun_out/r05_tests.log
