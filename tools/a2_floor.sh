#!/bin/bash
# Round 6: the cell-centric A2 self query before / after the per-cell neighbourhood records, ONE GPU session:
# wall time (tools/bench_ops.py radius 32), SQ counters and the two HBM-traffic counters (separate PMC passes, kernel trace only) of
# k_grid_query_cell for the library in build/pk/cur.so (dense-table lookups at every cell change: rounds 4-5) and the shipped one.
#   tools/a2_floor.sh [pairs]   (on the GPU box)   -> gpurun_out/r06/a2_floor_raw.txt
# build/pk/cur.so is not kept in the tree: build it from the round-5 sources first (git stash; git checkout 28b95c4 -- buffer_amd/csrc;
# python3 -c "from buffer_amd import build; build.build(force=True, out='build/pk/cur.so')"; git checkout HEAD -- buffer_amd/csrc; git stash pop).
cd "$(dirname "$0")/.."
pairs=${1:-32}
mkdir -p gpurun_out/r06
out=gpurun_out/r06/a2_floor_raw.txt
: > $out
C="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAVES SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES"
for v in old new; do
    if [ $v = old ]; then export BUF_LIB_PATH=$PWD/build/pk/cur.so; else unset BUF_LIB_PATH; fi
    echo "== library $v (${BUF_LIB_PATH:-buffer_amd/libbuffer_hip.so})" >> $out
    timeout 200 python3 tools/bench_ops.py radius $pairs 2>&1 | tail -3 >> $out
    for pass in sq fetch write; do
        case $pass in sq) cnt="$C";; fetch) cnt="FETCH_SIZE";; write) cnt="WRITE_SIZE";; esac
        timeout 300 tools/prof.sh a2f_${v}_$pass pmc "$cnt" -- python3 tools/bench_ops.py radius $pairs 3 > /dev/null 2>&1
        echo "-- $v $pass" >> $out
        python3 tools/pmc_sum.py gpurun_out/a2f_${v}_$pass k_grid_query_ >> $out 2>&1
        rm -rf gpurun_out/a2f_${v}_$pass
    done
done
unset BUF_LIB_PATH
cat $out
