#!/bin/bash
# cost of the activation micro-ops of the split kernel, one class at a time (timing only: results are wrong):
# bits of FF_SPLIT_SKIP_OPS = MicroOp codes (ff_mlp_ode_split.hpp): 0 LOAD 1 SCALE 2,3 EXP 4 ADD1 5,6 RCP 7 VALUE 8-13 tangent ops
# 14 TOPH 15 RESM 16 PACKH 17 PACKM 18 RESL 19 PACKL
set -e
run() { bash scratch/build_split_variant.sh /tmp/ops.so "$2" > /dev/null; echo "[$1] $(FLOWFUSION_AMD_LIB=/tmp/ops.so python scratch/split_prof.py 2>&1 | tail -1)"; }
run "the kernel" ""
run "no micro-ops at all" "-DFF_SPLIT_SKIP_OPS=0xFFFFF"
run "only LOAD (accvgpr reads)" "-DFF_SPLIT_SKIP_OPS=0xFFFFE"
run "no transcendentals" "-DFF_SPLIT_SKIP_OPS=0x6C"
run "no LOAD" "-DFF_SPLIT_SKIP_OPS=0x1"
run "no split (TOPH..PACKL)" "-DFF_SPLIT_SKIP_OPS=0xFC000"
run "no packs (v_perm)" "-DFF_SPLIT_SKIP_OPS=0xB0000"
run "only transcendentals" "-DFF_SPLIT_SKIP_OPS=0xFFF93"
