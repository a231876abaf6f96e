# tools/call_now.sh [ROOT ...] — call-path kernel times (sum-product and enumeration forms) under rocprofv3, for each source tree
# given (default: this one), side by side on the same box: ns per 1 M sites (GPU box)
cd /tmp && export TMPDIR=/tmp
ROOTS="${@:-$GRAFT_REPO_ROOT}"
for ped in ped10 ped5 ped15 trio quad; do
  for eng in elim enum; do
    if [ $eng = enum ] && [ $ped = ped15 ]; then continue; fi
    line="$ped $eng:"
    for R in $ROOTS; do
      export FAMSEQ_KERNEL_CACHE=/tmp/kc_now_$(echo $R | md5sum | cut -c1-8)
      python3 $R/tools/io_kernel_rates.py 1000000 $eng $ped > /dev/null 2>&1
      rm -rf /tmp/cp_cfg; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cp_cfg -- python3 $R/tools/io_kernel_rates.py 1000000 $eng $ped > /tmp/cp_cfg.log 2>&1
      line="$line $(cat /tmp/cp_cfg/*/*kernel_stats.csv | grep 'famseq_e' | cut -d, -f4 | cut -d. -f1)"
    done
    echo "$line"
  done
done
