#!/bin/bash
# Registers / scratch / LDS of every generated kernel whose source was kept (FAMSEQ_KEEP_SRC=1).
# usage: tools/kernel_resources.sh [kernel cache dir]
dir=${1:-famseq_amd/lib/kernels}
for f in "$dir"/*.hip; do
  printf '%s\n  ' "$(head -1 "$f" | cut -c36-130)"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off --genco \
    -Rpass-analysis=kernel-resource-usage -o /dev/null "$f" 2>&1 |
    grep -E " VGPRs:|VGPRs Spill|ScratchSize|Occupancy|LDS Size" | sed -e 's/.*remark: *//' -e 's/ \[-Rpass.*//' | tr '\n' ';'
  echo
done
