set -e
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
rocprofv3 --output-format csv --kernel-trace --stats -d $O/pw_stats -o st -- python3 tools/gpu_mpc_only.py > $O/pw_stats.log 2>&1
rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS -d $O/pw_pmc -o p -- python3 tools/gpu_mpc_only.py > $O/pw_pmc.log 2>&1
rocprofv3 --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU -d $O/pw_pmc2 -o p -- python3 tools/gpu_mpc_only.py > $O/pw_pmc2.log 2>&1
echo done
