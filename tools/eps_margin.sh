#!/bin/bash
# Development aid: the same parity checks with the error bounds of the two matrix-pipe filters scaled down (builds with
# -DNNF_EPS_SCALE / -DVOX_EPS_SCALE): how much margin do the shipped bounds have?
cd "$(dirname "$0")/.."
for sc in 0.0625 0.00390625 0; do
  echo "== eps x $sc"
  BUF_LIB_PATH=$PWD/build/libbuffer_eps_${sc}.so python3 -m pytest tests/test_ops_gpu.py tests/test_edge_cases_gpu.py -q -k "knn1 or voxelize" 2>&1 | tail -4
done
