#!/bin/bash
# Round 6: the packed-fp32 finding, stand-alone and bisected, in ONE GPU session.
#   1. tools/micro/pk_hazard: register-only / LDS-fed / register-fed packed victims beside f16-MFMA, fp32-MFMA and nothing
#   2. the library's own victim (k_vn_gather6_lds, tools/race_probe3.py) on builds of the library:
#        cur   = the shipped flags (no packed-fp32 instructions)          pk      = packed instructions allowed (the round-5 state)
#        pknop = packed allowed + s_waitcnt lgkmcnt(0); s_nop 7 between the LDS reads of a slot and its arithmetic
#      each beside k_nn1f_sweep (f16 MFMA), k_cyl_net_wg (fp32 MFMA) and nothing; pk also with the no-LDS kernel (BUF_VN_GATHER_DIRECT=1)
# Builds happen HERE (no GPU needed); the run goes through gpurun.
set -e
cd "$(dirname "$0")/.."
runs=${1:-300}
mkdir -p build/pk
PK_BUILD_ONLY=1 tools/micro/pk_hazard.sh
bld() { BUF_EXTRA_HIPCC_FLAGS="$2" python3 -c "import sys; sys.path.insert(0, '.'); from buffer_amd import build; build.build(force=True, out='build/pk/$1.so')" 2>&1 | grep -v "warning\|^ \|^$" || true; }
bld cur "" &
bld pk "-Xclang -target-feature -Xclang +packed-fp32-ops" &
bld pknop "-Xclang -target-feature -Xclang +packed-fp32-ops -DVG6_PROBE_NOP=7" &
wait
ls -la build/pk
cmd="mkdir -p gpurun_out/r06; (tools/micro/pk_hazard 10000; "
for v in cur pk pknop; do for a in nn1 wg none; do
  cmd="$cmd echo \"== library $v, aggressor $a\"; BUF_LIB_PATH=\$PWD/build/pk/$v.so timeout 600 python3 tools/race_probe3.py $a $runs 2>&1 | tail -4;"
done; done
cmd="$cmd echo '== library pk, no-LDS kernel (BUF_VN_GATHER_DIRECT=1), aggressor nn1'; BUF_VN_GATHER_DIRECT=1 BUF_LIB_PATH=\$PWD/build/pk/pk.so timeout 600 python3 tools/race_probe3.py nn1 $runs 2>&1 | tail -4) > gpurun_out/r06/pk_bisect.log 2>&1; cat gpurun_out/r06/pk_bisect.log"
/usr/local/graft/bin/gpurun --timeout 1700 -- "$cmd"
