#!/bin/bash
# Development aid: build ablation variants of csrc/convnet_wg.hip (WG_EXP bit mask) as small stand-alone libraries under
# build/wgv/ for tools/wg_probe.py --lib.   tools/wg_variants.sh 0 1 2 4 ...
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/build/wgv
cat > $R/build/wgv/unit.hip <<EOF2
#include <stdlib.h>
#include "$R/buffer_amd/csrc/core.hip"
#include "$R/buffer_amd/csrc/convnet_wg.hip"
EOF2
for e in "$@"; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -Wno-unused-function \
        -DWG_EXP=$e ${WG_FLAGS} -o $R/build/wgv/libwg_$e.so $R/build/wgv/unit.hip &
done
wait
ls -la $R/build/wgv/*.so
