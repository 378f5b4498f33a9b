#!/bin/bash
# rocprofv3 passes over scratch/split_prof.py (run on the GPU box from the repo root): OUT [args...]
set -e
OUT=${1:-gpurun_out/split_prof}; shift || true
mkdir -p "$OUT"
export TMPDIR=/tmp
python scratch/split_prof.py "$@" > "$OUT/plain.txt" 2>&1; cat "$OUT/plain.txt" | tail -1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python scratch/split_prof.py "$@" > "$OUT/trace.txt" 2>&1
i=0
for SET in "GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS" "FETCH_SIZE" "WRITE_SIZE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC"; do
    rocprofv3 --pmc $SET --output-format csv -d "$OUT/pmc$i" -- python scratch/split_prof.py "$@" > "$OUT/pmc$i.txt" 2>&1 || echo "pmc pass $i ($SET) failed"
    i=$((i + 1))
done
python - "$OUT" <<'PY'
import csv, sys, glob
out = sys.argv[1]
for f in sorted(glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True)):
    for row in list(csv.reader(open(f)))[:6]:
        print(row[0][:70], row[1:6])
vals = {}
for f in sorted(glob.glob(out + "/pmc*/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        if "mlp_ode" in row["Kernel_Name"]:
            vals.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
for k, v in vals.items():
    print(k, max(v))
PY
