#!/bin/bash
# tools/pmc_explore.sh TAG [bench args...] — stall/occupancy counters of the generated kernels
# (separate --pmc passes, kernel-trace only).  Output: gpurun_out/TAG/pmc_summary.json + stdout.
set -u
TAG=$1; shift
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/$TAG
mkdir -p $O
# the library refuses to spawn hipcc under a profiler: make sure this run's kernels are in the cache first
python3 $R/bench.py --warm-only "$@" > $O/warm.log 2>&1 || { echo "warm-up failed"; cat $O/warm.log; exit 1; }
cd /tmp
i=0
for c in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LEVEL_WAVES" \
  "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM" \
  "SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_TRANS_F64" \
  "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_IFETCH" \
  "TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_64B_sum" \
  "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout -k 10 170 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$i -- python3 $R/bench.py --no-cpu-baseline --no-side-configs --sites 1000000 --steps 2 --warmup 1 "$@" > $O/pmc_$i.log 2>&1
  echo "pass $i rc=$?"
done
cd $R
python3 tools/pmc_summary.py $O
rm -rf $O/pmc_*/
