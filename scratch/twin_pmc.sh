#!/bin/bash
# MFMA-pipe busy fraction and clock of the 128-wide two-part kernel with and without its four-slot twin (run on the GPU box)
set -e
export TMPDIR=/tmp
for MODE in twin plain; do
  if [ $MODE = plain ]; then export FF_SPLIT_NO_TWIN=1; else unset FF_SPLIT_NO_TWIN; fi
  rm -rf gpurun_out/twin_$MODE
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d gpurun_out/twin_$MODE -- python scratch/h128_two_waves.py > gpurun_out/twin_$MODE.txt 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/twin_${MODE}_t -- python scratch/h128_two_waves.py > /dev/null 2>&1
  python - "$MODE" <<'PY'
import csv, glob, sys
mode = sys.argv[1]
vals = {}
for f in glob.glob(f"gpurun_out/twin_{mode}/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "mlp_ode_split" in row["Kernel_Name"]:
            vals.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
dur = []
for f in glob.glob(f"gpurun_out/twin_{mode}_t/**/*kernel_trace.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "mlp_ode_split" in row["Kernel_Name"]:
            dur.append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-6)
g = max(vals["GRBM_GUI_ACTIVE"]) / 8
busy = max(vals["SQ_VALU_MFMA_BUSY_CYCLES"]) / (g * 1024)
print(f"{mode}: longest launch {max(dur):.1f} ms, cycles {g:.3e}, clock {g / (max(dur) * 1e-3) / 1e9:.2f} GHz, MFMA pipe busy {busy:.3f}")
PY
done
