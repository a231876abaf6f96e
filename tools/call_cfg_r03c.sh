# tools/call_cfg_r03c.sh — call-path form of the ten-member sum-product kernel under rocprofv3: one-wave workgroups at two waves
# per SIMD (no barrier coupling between waves), against the shipped 256-lane form (run on the GPU box)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in "256 2 0 0" "64 2 0 0" "64 2 4 1" "128 2 0 0" "64 2 1 0" "64 2 2 0"; do
  set -- $cfg
  export FAMSEQ_ELIM_BT=$1 FAMSEQ_ELIM_MINWAVES=$2 FAMSEQ_VARIANT_MIN=$3 FAMSEQ_KERNEL_CACHE=/tmp/kc_cfg_$1_$2_$3_$4
  if [ $4 = 1 ]; then export FAMSEQ_ELIM_CALL_REGS=1; else unset FAMSEQ_ELIM_CALL_REGS; fi
  python3 $R/tools/io_kernel_rates.py 1000000 elim ped10 > /dev/null 2>&1
  rm -rf /tmp/cp_cfg; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cp_cfg -- python3 $R/tools/io_kernel_rates.py 1000000 elim ped10 > /tmp/cp_cfg.log 2>&1
  echo "bt=$1 min_waves=$2 first_variant=$3 call_regs=$4: $(cat /tmp/cp_cfg/*/*kernel_stats.csv | grep famseq_elim | cut -d, -f4) ns  ($(ls /tmp/kc_cfg_$1_$2_$3_$4/*.res | wc -l) variants compiled: $(cat /tmp/kc_cfg_$1_$2_$3_$4/*.res | tr '\n' ' '))"
done
