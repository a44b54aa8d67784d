#!/bin/bash
# Development aid: time variants of the library in ONE GPU session (boxes differ by several per cent, so an A/B across two
# gpurun calls says little).  Builds build/ab/<i>.so with each flag set, then runs the probe once per variant on the box.
#   tools/ab.sh [-n patches] "<hipcc flags of variant 0>" "<flags of variant 1>" ...
set -e
cd "$(dirname "$0")/.."
n=20000
if [ "$1" = "-n" ]; then n=$2; shift 2; fi
mkdir -p build/ab
rm -f build/ab/*.so
i=0
for flags in "$@"; do
    BUF_EXTRA_HIPCC_FLAGS="$flags" python3 -c "import sys; sys.path.insert(0, '.'); from buffer_amd import build; build.build(force=True, out='build/ab/$i.so')" 2>&1 | grep -v "warning\|^ \|^$" || true &
    i=$((i + 1))
    if [ $((i % 4)) = 0 ]; then wait; fi
done
wait
cmd=""
for ((k = 0; k < i; k++)); do cmd="$cmd echo variant $k; WG_ONLY=1 BUF_LIB_PATH=\$PWD/build/ab/$k.so python3 tools/wg_probe.py $n 2>&1 | grep -E 'winograd|WG_STAMP|layer|rror' | head -12 ;"; done
cmd="$cmd echo variant 0 again; WG_ONLY=1 BUF_LIB_PATH=\$PWD/build/ab/0.so python3 tools/wg_probe.py $n 2>&1 | grep 'winograd:.*patches'"
timeout 1700 /usr/local/graft/bin/gpurun --timeout 900 -- "$cmd"
