#!/bin/bash
# A/B timing of split-kernel build variants on the GPU box: scratch/split_ab.sh "<flags A>" "<flags B>" ...   (results must still be right:
# the last line of each run prints the kernel times; add --check to compare the final states of all variants)
set -e
i=0
for V in "$@"; do
  bash scratch/build_split_variant.sh /tmp/ab$i.so "$V" > /dev/null
  echo "[$V] $(FLOWFUSION_AMD_LIB=/tmp/ab$i.so python scratch/split_prof.py 2>&1 | tail -1)"
  i=$((i + 1))
done
