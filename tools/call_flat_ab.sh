# tools/call_flat_ab.sh — the sum-product kernel's call-path form with and without flat staging of the packed PLs, under rocprofv3
# (run on the GPU box)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in "0 1024" "1 1024" "1 256" "1 1"; do
  set -- $cfg
  export FAMSEQ_CALL_FLAT=$1 FAMSEQ_CALL_LUT_LDS=$2 FAMSEQ_KERNEL_CACHE=/tmp/kc_flat_$1_$2
  for ped in ped10 ped5 ped15; do
    python3 $R/tools/io_kernel_rates.py 1000000 elim $ped > /dev/null 2>&1
    rm -rf /tmp/cp_cfg; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cp_cfg -- python3 $R/tools/io_kernel_rates.py 1000000 elim $ped > /tmp/cp_cfg.log 2>&1
    echo "flat=$1 lut_lds=$2 $ped: $(cat /tmp/cp_cfg/*/*kernel_stats.csv | grep famseq_elim | cut -d, -f4) ns  (scratch of the variants compiled: $(cat $FAMSEQ_KERNEL_CACHE/*.res | tr '\n' ' '))"
  done
done
