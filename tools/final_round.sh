#!/bin/bash
# The evidence set of a round in ONE GPU session: tools/profile_round.sh (rewrites profiles/traffic.json + instr.json first, so that the
# -m gpu suite behind it sees instruction counts of THIS build), the suite, the drop-in surface line, the recall / RANSAC-RR scripts.
set -x
tag=${1:-r06}
mkdir -p gpurun_out/$tag
tools/profile_round.sh $tag > gpurun_out/${tag}_profile.log 2>&1
tail -3 gpurun_out/${tag}_profile.log
cp profiles/traffic.json profiles/instr.json gpurun_out/$tag/            # (the box's tree is scratch: bring the regenerated files home)
python -m pytest tests -m gpu -q 2>&1 | tail -6 > gpurun_out/${tag}_tests.log
python bench.py --workload surface --steps 5 --detail-json gpurun_out/$tag/surface_detail.json > gpurun_out/$tag/surface.json 2> gpurun_out/$tag/surface.err
python tools/split_safe_probe.py > gpurun_out/$tag/split_safe_probe.txt 2>&1
python tools/a1_time.py 32 > gpurun_out/$tag/a1_time.txt 2>&1
python tests/eval_recall.py --backend gpu --out gpurun_out/$tag/recall_gpu.json > gpurun_out/$tag/recall_gpu.log 2>&1
python tests/eval_ransac_rr.py --out gpurun_out/$tag/ransac_rr.json > gpurun_out/$tag/ransac_rr.log 2>&1
python tests/eval_ransac_rr.py --overlaps 0.35,0.3,0.25,0.2 --out gpurun_out/$tag/ransac_rr_low_overlap.json > gpurun_out/$tag/ransac_rr_low.log 2>&1
for f in recall_gpu ransac_rr ransac_rr_low; do tail -n 2 gpurun_out/$tag/$f.log; done
cat gpurun_out/${tag}_tests.log
