set -e
export TMPDIR=/tmp
O=$PWD/gpurun_out
python3 -c 'from f16_mpc_oop_py_amd import lib; lib.load()'
rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE -d $O/prof_q1 -o m -- python3 tools/gpu_mpc_only.py > $O/prof_q1.log 2> $O/prof_q1.err
rocprofv3 --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_SALU -d $O/prof_q2 -o l -- python3 tools/gpu_mpc_only.py > $O/prof_q2.log 2> $O/prof_q2.err
