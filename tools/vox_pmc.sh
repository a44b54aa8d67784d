#!/bin/bash
# Development aid: SQ counters of k_patch_voxelize (separate PMC runs, kernel trace only) -> gpurun_out/voxpmc{1,2}.txt
cd "$(dirname "$0")/.."
python3 tools/vox_probe.py 2>&1 | grep patches
C1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA"
C2="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_WAVES"
C3="SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_VMEM"
tools/prof.sh voxpmc1 pmc "$C1" -- python3 tools/vox_probe.py > /dev/null 2>&1
tools/prof.sh voxpmc2 pmc "$C2" -- python3 tools/vox_probe.py > /dev/null 2>&1
tools/prof.sh voxpmc3 pmc "$C3" -- python3 tools/vox_probe.py > /dev/null 2>&1
for n in 1 2 3; do python3 tools/pmc_sum.py gpurun_out/voxpmc$n k_patch_voxelize > gpurun_out/voxpmc$n.txt 2>&1; rm -rf gpurun_out/voxpmc$n; done
cat gpurun_out/voxpmc1.txt gpurun_out/voxpmc2.txt gpurun_out/voxpmc3.txt
