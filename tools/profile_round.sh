#!/bin/bash
# Round profile on the GPU box (run through gpurun from the repo root):  bash tools/profile_round.sh
# Separate rocprofv3 passes, as MI355X_MICROARCH.md prescribes: --kernel-trace --stats alone; each --pmc set alone.
# Output under gpurun_out/prof_*; tools/*_summary.py (run in the build container) turn it into profiles/<tag>_*.
# The library must be built BEFORE the first rocprofv3 line (load() never compiles: the profiler's preload would make the
# compiler chain an exec hop of a GPU-initialised process); here only its presence is checked.
set -e
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out
mkdir -p $O
cd $R
python3 -c 'from f16_mpc_oop_py_amd import lib; lib.load(); print("library ok:", lib.SO_PATH)'
step() { echo "== $1 ($(date +%T))"; }
# bash tools/profile_round.sh [a|b]: the passes in two halves (a gpurun call is limited to 20 minutes); no argument = everything
HALF=${1:-ab}
if [[ $HALF == *a* ]]; then
step "kernel-trace + stats of the default bench command"
rocprofv3 --output-format csv --kernel-trace --stats -d $O/prof_stats -o st -- python3 bench.py --steps 30 --warmup 5 > $O/bench_under_rocprof.json 2> $O/prof_stats.err
step "kernel-trace + stats of the MPC leg ALONE, headline settings only (B = 4096, N = 30, osqp defaults)"
rocprofv3 --output-format csv --kernel-trace --stats -d $O/prof_stats_mpc -o stm -- python3 tools/gpu_mpc_only.py > $O/prof_stats_mpc.log 2> $O/prof_stats_mpc.err
step "kernel-trace + stats of the config-5 closed loop ALONE as one launch (B = 8192, T = 100, N = 30: f16_rollout_mpc), then its issue counters"
rocprofv3 --output-format csv --kernel-trace --stats -d $O/prof_stats_c5 -o c5 -- python3 tools/gpu_config5_only.py > $O/prof_stats_c5.log 2> $O/prof_stats_c5.err
rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE \
  -d $O/prof_c5_sq -o c5 -- python3 tools/gpu_config5_only.py > $O/prof_c5_sq.log 2> $O/prof_c5_sq.err
rocprofv3 --output-format csv --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
  -d $O/prof_c5_fp -o c5f -- python3 tools/gpu_config5_only.py > $O/prof_c5_fp.log 2> $O/prof_c5_fp.err
step "HBM traffic: FETCH_SIZE, WRITE_SIZE (separate passes), B = 4096 and B = 262,144 rollouts"
rocprofv3 --output-format csv --pmc FETCH_SIZE -d $O/prof_fetch -o f -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-mpc --no-config5 > /dev/null 2> $O/prof_fetch.err
rocprofv3 --output-format csv --pmc WRITE_SIZE -d $O/prof_write -o w -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-mpc --no-config5 > /dev/null 2> $O/prof_write.err
step "MPC kernels (headline settings only): matrix-core and issue counters, then LDS pipe + fp64 instruction mix"
rocprofv3 --output-format csv --pmc SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE \
  -d $O/prof_mfma -o m -- python3 tools/gpu_mpc_only.py > $O/prof_mfma.log 2> $O/prof_mfma.err
rocprofv3 --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_LDS \
  -d $O/prof_lds -o l -- python3 tools/gpu_mpc_only.py > $O/prof_lds.log 2> $O/prof_lds.err
fi
if [[ $HALF == *b* ]]; then
step "dynamics kernels: issue counters, then LDS pipe + fp64 instruction mix"
rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT \
  -d $O/prof_dyn -o d -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-mpc --no-config5 > /dev/null 2> $O/prof_dyn.err
rocprofv3 --output-format csv --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_SALU \
  -d $O/prof_dyn2 -o d2 -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-mpc --no-config5 > /dev/null 2> $O/prof_dyn2.err
step "the horizon sweep (env.py:426-436, B = 64, N = 1..150): kernel trace, then HBM-side traffic and issue counters of its solve launch"
rocprofv3 --output-format csv --kernel-trace --stats -d $O/prof_stats_sweep -o sw -- python3 tools/gpu_sweep_only.py 64 150 > $O/prof_stats_sweep.log 2> $O/prof_stats_sweep.err
rocprofv3 --output-format csv --pmc FETCH_SIZE -d $O/prof_sweep_fetch -o f -- python3 tools/gpu_sweep_only.py 64 150 > /dev/null 2> $O/prof_sweep_fetch.err
rocprofv3 --output-format csv --pmc WRITE_SIZE -d $O/prof_sweep_write -o w -- python3 tools/gpu_sweep_only.py 64 150 > /dev/null 2> $O/prof_sweep_write.err
rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE \
  -d $O/prof_sweep_sq -o s -- python3 tools/gpu_sweep_only.py 64 150 > /dev/null 2> $O/prof_sweep_sq.err
step "plain bench (the record the profile is compared with)"
python3 bench.py > $O/bench.json 2> $O/bench.err
fi
step "done"
