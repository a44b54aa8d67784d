#!/bin/bash
# Development aid (round 4): the cell-centric A2 self query against the query-centric kernel in ONE GPU session -- wall time
# (tools/bench_ops.py radius <pairs>) and SQ counters of both (separate PMC runs, kernel trace only).
#   tools/a2_profile.sh [pairs]      -> gpurun_out/a2_<kernel>.txt, gpurun_out/a2pmc_<kernel>/
cd "$(dirname "$0")/.."
pairs=${1:-64}
C="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAVES SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES"
python3 tools/bench_ops.py radius $pairs > gpurun_out/a2_cell.txt 2>&1
BUF_A2_QUERY_CENTRIC=1 python3 tools/bench_ops.py radius $pairs > gpurun_out/a2_wave.txt 2>&1
tools/prof.sh a2pmc_cell pmc "$C" -- python3 tools/bench_ops.py radius $pairs 3 > /dev/null 2>&1
export BUF_A2_QUERY_CENTRIC=1
tools/prof.sh a2pmc_wave pmc "$C" -- python3 tools/bench_ops.py radius $pairs 3 > /dev/null 2>&1
unset BUF_A2_QUERY_CENTRIC
python3 tools/pmc_sum.py gpurun_out/a2pmc_cell k_grid_query > gpurun_out/a2pmc_cell.txt 2>&1
python3 tools/pmc_sum.py gpurun_out/a2pmc_wave k_grid_query > gpurun_out/a2pmc_wave.txt 2>&1
tail -3 gpurun_out/a2_cell.txt gpurun_out/a2_wave.txt; cat gpurun_out/a2pmc_cell.txt gpurun_out/a2pmc_wave.txt
