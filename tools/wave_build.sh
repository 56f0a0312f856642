#!/bin/bash
# build the product library and the -DF16_EXP_STAMPW diagnostic variant; non-zero exit if either fails
set -e
python -c 'import __graft_entry__ as g; g.build()' > /tmp/build.log 2>&1 || { grep -E "error" -A4 /tmp/build.log | head -30; exit 1; }
F16HIP_SO=build/libf16hip_stampw.so F16_HIPCC_EXTRA=-DF16_EXP_STAMPW python -c "from f16_mpc_oop_py_amd import lib; lib.build(force=True)" > /tmp/build2.log 2>&1 || { grep -E "error" -A4 /tmp/build2.log | head -30; exit 1; }
F16HIP_SO=build/libf16hip_stampg.so F16_HIPCC_EXTRA=-DF16_EXP_STAMPG python -c "from f16_mpc_oop_py_amd import lib; lib.build(force=True)" > /tmp/build3.log 2>&1 || { grep -E "error" -A4 /tmp/build3.log | head -30; exit 1; }
echo build ok
