#!/bin/bash
export TMPDIR=/tmp
ROOT=$(pwd)
cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $ROOT/gpurun_out/cpmc1 -- python3 $ROOT/tools/pipeline_bench.py --reps 2 > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $ROOT/gpurun_out/cpmc2 -- python3 $ROOT/tools/pipeline_bench.py --reps 2 > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 --output-format csv -d $ROOT/gpurun_out/cpmc3 -- python3 $ROOT/tools/pipeline_bench.py --reps 2 > /dev/null 2>&1
find $ROOT/gpurun_out/cpmc* -name "*.db" -delete
ls -R $ROOT/gpurun_out/cpmc1 | head
