#!/bin/bash
# does the 16x16x32 MFMA shape buy clock in THIS kernel?  (timing only; wrong results)
set -e
export TMPDIR=/tmp
mkdir -p gpurun_out/clk
i=0
for V in "" "-DFF_SPLIT_MFMA16" "-DFF_SPLIT_NODMA -DFF_SPLIT_NOBARRIER -DFF_SPLIT_NOACT -DFF_SPLIT_NOSPLIT -DFF_SPLIT_NOWREAD" \
         "-DFF_SPLIT_MFMA16 -DFF_SPLIT_NODMA -DFF_SPLIT_NOBARRIER -DFF_SPLIT_NOACT -DFF_SPLIT_NOSPLIT -DFF_SPLIT_NOWREAD"; do
  bash scratch/build_split_variant.sh /tmp/v$i.so "$V" > /dev/null
  T=$(FLOWFUSION_AMD_LIB=/tmp/v$i.so python scratch/split_prof.py 2>&1 | tail -1)
  FLOWFUSION_AMD_LIB=/tmp/v$i.so rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d gpurun_out/clk/p$i -- python scratch/split_prof.py > /dev/null 2>&1
  python - "$i" "$V" "$T" <<'PY'
import csv, glob, sys
i, v, t = sys.argv[1:4]
vals = {}
for f in glob.glob(f"gpurun_out/clk/p{i}/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "mlp_ode" in row["Kernel_Name"]:
            vals.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
g = max(vals["GRBM_GUI_ACTIVE"]) / 8
ms = float(t.split("'")[1])
print(f"[{v}] {ms:.1f} ms  cycles {g:.4g} ({g / 1.0027e9:.3f} x ideal)  clock {g / (ms * 1e-3) / 1e9:.3f} GHz", flush=True)
PY
  rm -rf gpurun_out/clk/p$i
  i=$((i + 1))
done
