cd "$GRAFT_REPO_ROOT"
export H3_ONLY=1
C2="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_WAVES"
tools/prof.sh h3cmp_old pmc "$C2" -- python3 tools/h3_probe.py 20000 > /dev/null 2>&1
python3 tools/pmc_sum.py gpurun_out/h3cmp_old k_cyl_net_h3 > gpurun_out/h3cmp_old.txt 2>&1
export BUF_LIB_PATH=$PWD/build/ab/h3new.so
tools/prof.sh h3cmp_new pmc "$C2" -- python3 tools/h3_probe.py 20000 > /dev/null 2>&1
python3 tools/pmc_sum.py gpurun_out/h3cmp_new k_cyl_net_h3 > gpurun_out/h3cmp_new.txt 2>&1
rm -rf gpurun_out/h3cmp_old gpurun_out/h3cmp_new
echo OLD; cat gpurun_out/h3cmp_old.txt; echo NEW; cat gpurun_out/h3cmp_new.txt
