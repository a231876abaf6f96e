#!/bin/bash
# tools/call_pmc.sh TAG — stall / issue counters of the fused call-path kernel (tools/io_kernel_rates.py), separate passes
set -u
TAG=$1
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/$TAG
mkdir -p $O
python3 $R/tools/io_kernel_rates.py 1000 > $O/warm.log 2>&1   # kernels into the cache before the profiler is attached
cd /tmp
i=0
for c in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" \
  "SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM" \
  "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS" \
  "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout -k 10 170 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$i -- python3 $R/tools/io_kernel_rates.py > $O/pmc_$i.log 2>&1
  echo "pass $i rc=$?"
done
cd $R
python3 tools/pmc_summary.py $O
rm -rf $O/pmc_*/
