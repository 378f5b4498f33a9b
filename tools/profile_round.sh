#!/bin/bash
# Collect the profiles the bench line refers to (run ON the GPU box, from the repo root):
#   tools/profile_round.sh OUTDIR [trace|pmc|all]      (two gpurun calls of <= 20 minutes: `trace`, then `pmc`)
# 1. the bench line itself (headline + split-precision record + extra_configs: BASELINE configs 2..5);
# 2. rocprofv3 --kernel-trace --stats of the same command (every workload's kernel in one trace);
# 3. PMC passes (one counter set per run; never combined with tracing) for HBM traffic, clock, MFMA busy, LDS.
# Every step writes under OUTDIR (normally gpurun_out/prof_rNN); tools/summarize_profiles.py turns it into the files
# committed under profiles/.  A progress line per step keeps the box's watchdog informed.
set -e
OUT=${1:-gpurun_out/prof}
STAGE=${2:-all}
mkdir -p "$OUT"
export TMPDIR=/tmp
if [ "$STAGE" != "pmc" ]; then
python bench.py --steps 3 --warmup 1 > "$OUT/bench.json" 2> "$OUT/bench.err"
echo "bench done: $(cut -c1-120 "$OUT/bench.json")"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python bench.py --steps 2 --warmup 1 --cpu-batch 0 \
    > "$OUT/bench_under_rocprof.json" 2> "$OUT/trace.err"
python tools/filter_rocprof.py "$OUT/trace"
echo "kernel trace done"
fi
if [ "$STAGE" = "trace" ]; then exit 0; fi
i=0
for SET in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"; do
    rocprofv3 --pmc $SET --output-format csv -d "$OUT/pmc$i" -- python bench.py --steps 1 --warmup 0 --cpu-batch 0 \
        > "$OUT/pmc$i.json" 2> "$OUT/pmc$i.err"
    python tools/filter_rocprof.py "$OUT/pmc$i"
    echo "pmc pass $i ($SET) done"
    i=$((i + 1))
done
