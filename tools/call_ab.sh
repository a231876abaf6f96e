# tools/call_ab.sh — the fused call-path kernels (packed PLs in, Phred + call out) under rocprofv3, default block
# configuration against the 256-lane / two-waves one, per pedigree and engine (run on the GPU box).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { tag=$1; python3 $R/tools/io_kernel_rates.py $SITES $ENG $PED > /dev/null 2>&1; rm -rf /tmp/cp_$tag; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cp_$tag -- python3 $R/tools/io_kernel_rates.py $SITES $ENG $PED > /tmp/cp_$tag.log 2>&1; echo "== $tag $PED $ENG"; cat /tmp/cp_$tag/*/*kernel_stats.csv | grep "famseq" | cut -d, -f1-4 | cut -c1-60; }
for PED in ${PEDS:-trio quad ped5 ped10}; do
  SITES=4000000; [ $PED = ped10 ] && SITES=1000000
  unset FAMSEQ_ELIM_BT FAMSEQ_ELIM_MINWAVES FAMSEQ_VARIANT_MIN FAMSEQ_LANE_BT FAMSEQ_LANE_MINWAVES
  export FAMSEQ_KERNEL_CACHE=/tmp/kc_new; ENG=elim run new; ENG=enum run new
  export FAMSEQ_KERNEL_CACHE=/tmp/kc_old FAMSEQ_ELIM_BT=256 FAMSEQ_ELIM_MINWAVES=2 FAMSEQ_VARIANT_MIN=0 FAMSEQ_LANE_BT=256 FAMSEQ_LANE_MINWAVES=2
  ENG=elim run old; ENG=enum run old
done
