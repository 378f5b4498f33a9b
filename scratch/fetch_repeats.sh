#!/bin/bash
# run ON the GPU box from the repo root: FETCH_SIZE of the headline launch in N single-process runs (default 6) -- is the
# read side of its memory traffic one number or two (38,07x KiB raw in ten of ten launches of round 3; 118,310 KiB in one
# profile of round 2 and one of round 4)?
set -e
export TMPDIR=/tmp
N=${1:-6}
OUT=gpurun_out/fetch_repeats
mkdir -p $OUT
: > $OUT/summary.txt
for i in $(seq 1 $N); do
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/run$i -- python bench.py --steps 1 --warmup 0 --cpu-batch 0 --no-extras > $OUT/run$i.json 2> $OUT/run$i.err
    python - $OUT/run$i $i <<'PY' >> $OUT/summary.txt
import csv, sys
from pathlib import Path
root, i = Path(sys.argv[1]), sys.argv[2]
for f in root.rglob("*counter_collection.csv"):
    with open(f, newline="") as g:
        for row in csv.DictReader(g):
            if "mlp_ode_kernel<16, 256, 4, 0, false, 2, 8" in row["Kernel_Name"] and row["Counter_Name"] == "FETCH_SIZE" and row.get("Grid_Size", row.get("Grid_Size_X", "")) in ("4194304", ""):
                v = float(row["Counter_Value"])
                if v > 1000:
                    print(f"run {i}  dispatch {row.get('Dispatch_Id', '?'):>4}  FETCH_SIZE {v:.3f} KiB  (x2: {v * 2048 / 1e6:.1f} MB read)")
PY
    rm -rf $OUT/run$i
    echo "run $i done"
done
cat $OUT/summary.txt
