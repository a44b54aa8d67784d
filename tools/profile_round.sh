#!/bin/bash
# The measurement set of a round, on the GPU box: bench lines (headline, stream, kitti), rocprofv3 kernel stats of the
# headline command, the two HBM-traffic PMC passes, SQ counters.  Everything lands in gpurun_out/<tag>/; copy what is to be
# judged into profiles/.
set -ex
tag=${1:-r05}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/$tag
mkdir -p $O
export BUF_NO_TRAFFIC=1                                 # no stale bytes in the lines of the profiled runs (traffic: null there); the tracked file stays
tools/prof.sh ${tag}_stats stats -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline
cp $(find $R/gpurun_out/${tag}_stats -name "*kernel_stats.csv") $O/kernel_stats.csv
tools/prof.sh ${tag}_fetch pmc "FETCH_SIZE" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --detail-json $O/bench_pmc.json
tools/prof.sh ${tag}_write pmc "WRITE_SIZE" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline
python tools/make_traffic.py $(find $R/gpurun_out/${tag}_fetch -name "*counter_collection.csv") $(find $R/gpurun_out/${tag}_write -name "*counter_collection.csv") $O/bench_pmc.json $O/traffic.json
cp $O/traffic.json $R/profiles/traffic.json     # the bench lines below replay THIS build's measured bytes per unit
tools/prof.sh ${tag}_instr pmc "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVES" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline
python tools/make_instr.py $(find $R/gpurun_out/${tag}_instr -name "*counter_collection.csv") $O/bench_pmc.json $O/instr.json
cp $O/instr.json $R/profiles/instr.json         # ... and price the vector-issue roofline with THIS build's instruction counts
unset BUF_NO_TRAFFIC
python bench.py --steps 20 --warmup 5 --detail-json $O/bench_detail.json > $O/bench.json 2> $O/bench.err
python bench.py --workload stream > $O/stream.json 2>> $O/bench.err
python bench.py --workload stream --stream-pairs 1781 --stream-overlaps 0.3,0.25,0.2,0.15 > $O/stream_lomatch.json 2>> $O/bench.err
python bench.py --workload kitti --steps 6 --warmup 2 --no-cpu-baseline > $O/kitti.json 2>> $O/bench.err
python bench.py --pairs-per-step 64 --steps 4 --warmup 2 --no-cpu-baseline --detail-json $O/bench_64pairs_detail.json > $O/bench_64pairs.json 2>> $O/bench.err
python bench.py --workload stream --arith split > $O/stream_split.json 2>> $O/bench.err
python bench.py --workload kitti --steps 6 --warmup 2 --no-cpu-baseline --arith split > $O/kitti_split.json 2>> $O/bench.err
tools/prof.sh ${tag}_sq pmc "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline
python tools/pmc_summary.py $(find $R/gpurun_out/${tag}_sq -name "*counter_collection.csv") > $O/pmc_sq.txt
python tools/pmc_summary.py $(find $R/gpurun_out/${tag}_fetch -name "*counter_collection.csv") > $O/pmc_FETCH_SIZE.txt
python tools/pmc_summary.py $(find $R/gpurun_out/${tag}_write -name "*counter_collection.csv") > $O/pmc_WRITE_SIZE.txt
rm -rf $R/gpurun_out/${tag}_instr $R/gpurun_out/${tag}_stats $R/gpurun_out/${tag}_fetch $R/gpurun_out/${tag}_write $R/gpurun_out/${tag}_sq
ls -la $O
