#!/bin/bash
# Development aid: bitwise A/B of the voxelisation kernel, in-tree library against build/libbuffer_old.so
cd "$(dirname "$0")/.."
BUF_LIB_PATH=$PWD/build/libbuffer_old.so python3 tools/vox_ab.py gen gpurun_out/vox_old.npz 2>&1 | tail -8
python3 tools/vox_ab.py gen gpurun_out/vox_new.npz 2>&1 | tail -8
python3 tools/vox_ab.py cmp gpurun_out/vox_old.npz gpurun_out/vox_new.npz
rm -f gpurun_out/vox_old.npz gpurun_out/vox_new.npz
