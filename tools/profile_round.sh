#!/bin/bash
# Round profile on the GPU box (run through gpurun from the repo root):  bash tools/profile_round.sh
# Separate rocprofv3 passes, as MI355X_MICROARCH.md prescribes: --kernel-trace --stats alone; each --pmc set alone.
set -e
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out
mkdir -p $O
cd $R
rocprofv3 --output-format csv --kernel-trace --stats -d $O/prof_stats -o st -- python3 bench.py --steps 10 --warmup 2 > $O/bench_under_rocprof.json 2> $O/prof_stats.err
echo "stats pass done"
rocprofv3 --output-format csv --pmc FETCH_SIZE -d $O/prof_fetch -o f -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-mpc --no-large > /dev/null 2> $O/prof_fetch.err
rocprofv3 --output-format csv --pmc WRITE_SIZE -d $O/prof_write -o w -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-mpc --no-large > /dev/null 2> $O/prof_write.err
echo "hbm passes done"
rocprofv3 --output-format csv --pmc SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU \
  -d $O/prof_mfma -o m -- python3 tools/gpu_mpc_only.py > $O/prof_mfma.log 2> $O/prof_mfma.err
rocprofv3 --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 \
  -d $O/prof_lds -o l -- python3 tools/gpu_mpc_only.py > $O/prof_lds.log 2> $O/prof_lds.err
echo "mpc passes done"
python3 bench.py > $O/bench.json 2> $O/bench.err
echo "plain bench done"
# SQ counters of the dynamics kernels (issue utilisation behind the "VALU-bound when the chip is full" reading)
rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT \
  -d $O/prof_dyn -o d -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-mpc > /dev/null 2> $O/prof_dyn.err
echo "dynamics SQ pass done"
