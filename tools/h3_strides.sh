#!/bin/bash
# Development aid: time the split kernels for several builds of the library in ONE GPU session.  tools/h3_strides.sh lib1.so lib2.so ...
cd "$(dirname "$0")/.."
for lib in "$@"; do
  echo "== $lib"
  H3_ONLY=1 BUF_LIB_PATH=$PWD/$lib python3 tools/h3_probe.py 20000 2>&1 | grep "patches\|relu"
  BUF_LIB_PATH=$PWD/$lib python3 tools/cost_h3_probe.py 25600 2>&1 | grep "split f16x3"
done
