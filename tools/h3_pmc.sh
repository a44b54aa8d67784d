#!/bin/bash
# Development aid (round 4): SQ counters of the two split-f16 CNN kernels (separate PMC runs, kernel trace only).
#   tools/h3_pmc.sh  -> gpurun_out/h3pmc_{cyl,cost}{1,2}.txt
cd "$(dirname "$0")/.."
export H3_ONLY=1
C1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA"
C2="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_WAVES"
tools/prof.sh h3pmc_cyl1 pmc "$C1" -- python3 tools/h3_probe.py 20000 > /dev/null 2>&1
tools/prof.sh h3pmc_cyl2 pmc "$C2" -- python3 tools/h3_probe.py 20000 > /dev/null 2>&1
tools/prof.sh h3pmc_cost1 pmc "$C1" -- python3 tools/cost_h3_probe.py 25600 > /dev/null 2>&1
tools/prof.sh h3pmc_cost2 pmc "$C2" -- python3 tools/cost_h3_probe.py 25600 > /dev/null 2>&1
for n in cyl1 cyl2; do python3 tools/pmc_sum.py gpurun_out/h3pmc_$n k_cyl_net_h3 > gpurun_out/h3pmc_$n.txt 2>&1; done
for n in cost1 cost2; do python3 tools/pmc_sum.py gpurun_out/h3pmc_$n k_cost_net_h3 > gpurun_out/h3pmc_$n.txt 2>&1; done
cat gpurun_out/h3pmc_cyl1.txt gpurun_out/h3pmc_cyl2.txt gpurun_out/h3pmc_cost1.txt gpurun_out/h3pmc_cost2.txt
