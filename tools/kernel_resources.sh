#!/bin/bash
# tools/kernel_resources.sh [file.hip ...] -- registers / scratch / LDS / occupancy of every kernel, from the compiler's
# -Rpass-analysis=kernel-resource-usage remarks (same flags as the product build).
cd "$(dirname "$0")/../f16_mpc_oop_py_amd/csrc"
files=${@:-f16_dynamics.hip f16_control.hip f16_mpc_solve.hip f16_trim.hip}
for f in $files; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -DF16_FAST_TAN -DF16_FAST_POW -DF16_FAST_TRIG \
    -DF16_FAST_DIV $F16_HIPCC_EXTRA -c -o /tmp/kres.o $f -Rpass-analysis=kernel-resource-usage 2>&1 |
    awk '/Function Name/ {name=$(NF-1)} /VGPRs:/ {v=$(NF-1)} /AGPRs:/ {a=$(NF-1)} /ScratchSize/ {s=$(NF-1)} /Occupancy/ {o=$(NF-1)} /LDS Size/ {print name, "VGPR", v, "AGPR", a, "scratch", s, "occ", o, "LDS", $(NF-1)}' |
    sed -e 's/\[-Rpass-analysis=kernel-resource-usage\]//g' | while read n rest; do echo "$(echo $n | c++filt) $rest"; done
done
