#!/bin/bash
# times every tools/exp_variants/*.hsaco (name: {el|ln}{N}_tag.hsaco) with tools/kernel_bench
for f in tools/exp_variants/*.hsaco; do
  b=$(basename $f .hsaco)
  kind=${b:0:2}; n=$(echo $b | sed -E 's/^[a-z]+([0-9]+)_.*/\1/')
  entry=famseq_elim; [ $kind = ln ] && entry=famseq_enum_lane
  bt=$(grep -m1 "define BT" tools/exp_variants/$b.hip | awk '{print $3}')
  sites=${SITES:-4000000}
  [ $kind = ln ] && [ $n -ge 13 ] && sites=262144   # deep enumerations: 0.14 s per launch at this size
  timeout -k 10 120 ./tools/kernel_bench $f $entry $n $bt 0 $sites ${LC:-1.0} || exit 1
done
