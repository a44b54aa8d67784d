#!/bin/bash
# Development aid: SQ counters of k_cyl_net_wg for library variants build/ab/<name>.so (one rocprofv3 --pmc pass each, kernel trace only).
#   tools/wg_pmc.sh old new   (run through gpurun)
R=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp WG_ONLY=1
for v in "$@"; do
    out=$R/gpurun_out/wgpmc_$v; rm -rf $out; mkdir -p $out
    export BUF_LIB_PATH=$R/build/ab/$v.so
    (cd /tmp && rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_MFMA --output-format csv -d $out -o p -- python3 $R/tools/wg_probe.py 20000 > $out/run.log 2>&1)
    echo "== $v"; grep "winograd:" $out/run.log
    python3 $R/tools/pmc_summary.py $(find $out -name "*counter_collection.csv") k_cyl_net_wg | awk '{print $1, $NF}' 
    rm -rf $out
done
