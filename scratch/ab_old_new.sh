#!/bin/bash
# A/B on one box: the committed (HEAD) split kernel vs the working tree's
set -e
R0=${GRAFT_REPO_ROOT:-/root/repo}
GRAFT_REPO_ROOT=$R0/scratch/old_tree bash scratch/build_split_variant.sh /tmp/old.so "" > /dev/null
bash scratch/build_split_variant.sh /tmp/new.so "" > /dev/null
for i in 1 2; do
  echo "old: $(FLOWFUSION_AMD_LIB=/tmp/old.so python scratch/split_prof.py | tail -1)"
  echo "new: $(FLOWFUSION_AMD_LIB=/tmp/new.so python scratch/split_prof.py | tail -1)"
done
