#!/bin/bash
# timing ablations of the split kernel on the GPU box
set -e
mkdir -p gpurun_out/abl
for V in "" "-DFF_SPLIT_NOACT" "-DFF_SPLIT_NODMA" "-DFF_SPLIT_NODMA -DFF_SPLIT_NOBARRIER" "-DFF_SPLIT_NODMA -DFF_SPLIT_NOBARRIER -DFF_SPLIT_NOACT"; do
  bash scratch/build_split_variant.sh /tmp/v.so "$V" > /dev/null
  echo "variant [$V]: $(FLOWFUSION_AMD_LIB=/tmp/v.so python scratch/split_prof.py 2>&1 | tail -1)"
done
