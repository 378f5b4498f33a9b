#!/bin/bash
# cycle stamps of the split kernel (scratch/kbench_split.hip) for a list of build variants (timing experiments; results wrong)
set -e
for V in "$@"; do
  hipcc -O3 -std=c++17 -Wno-inline-asm --offload-arch=gfx950 -DFF_SPLIT_STAMPS $V -Iflowfusion_amd/csrc -Iinclude scratch/kbench_split.hip -o /tmp/kbs
  echo "== [$V]"
  timeout -k 10 120 /tmp/kbs | head -2
done
