# tools/call_group_ab.sh — round 3 experiment (profiles/r03c/exp_call_phred_group_x_ct_out.txt): call-path kernels by logarithms per basic
# block (FAMSEQ_CALL_PHRED_GROUP) x constant-width stage-out (then an environment switch, FAMSEQ_CALL_CT_OUT; since folded into the
# kernels' variant contest: kElimCallVariants) x lane kernels' waves per SIMD, small pedigrees (GPU box)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in "2 1 0" "1 1 0" "2 0 0" "1 0 0" "2 1 2" "1 1 2"; do
  set -- $cfg
  export FAMSEQ_CALL_PHRED_GROUP=$1 FAMSEQ_CALL_CT_OUT=$2 FAMSEQ_KERNEL_CACHE=/tmp/kc_g_$1_$2_$3
  if [ $3 != 0 ]; then export FAMSEQ_LANE_MINWAVES=$3; else unset FAMSEQ_LANE_MINWAVES; fi
  for run in "ped5 enum" "ped5 elim" "trio enum" "trio elim" "quad enum" "ped10 elim"; do
    set -- $run
    if [ "$FAMSEQ_LANE_MINWAVES" != "" ] && [ $2 = elim ]; then continue; fi
    python3 $R/tools/io_kernel_rates.py 1000000 $2 $1 > /dev/null 2>&1
    rm -rf /tmp/cp_cfg; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cp_cfg -- python3 $R/tools/io_kernel_rates.py 1000000 $2 $1 > /tmp/cp_cfg.log 2>&1
    echo "group=$FAMSEQ_CALL_PHRED_GROUP ct_out=$FAMSEQ_CALL_CT_OUT lane_minwaves=${FAMSEQ_LANE_MINWAVES:-default} $1 $2: $(cat /tmp/cp_cfg/*/*kernel_stats.csv | grep 'famseq_e' | cut -d, -f4 | tr '\n' ' ')"
  done
done
