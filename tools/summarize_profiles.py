"""Turn the output of tools/profile_round.sh into the files committed under profiles/rNN.

    python tools/summarize_profiles.py gpurun_out/prof_r01 profiles/r01

* bench_c2_1gpu.json, bench_c2_under_rocprof.json       the bench lines
* bench_c2_kernel_stats.csv, bench_c2_kernel_trace_mlp_ode.csv   rocprofv3 --kernel-trace --stats (names trimmed)
* bench_c2_pmc_summary.json  and  ../hbm_traffic.json   PMC counters of the mlp_ode kernel, per launch, with the
  gfx950 corrections of MI355X_MICROARCH.md (FETCH_SIZE counts 64 B per 128-B request -> doubled; KiB units)
"""
import csv
import json
import sys
from pathlib import Path


def _head_commit():
    """Commit the profiled tree was snapshotted from (the summariser runs in the authoring container, where git is)."""
    import subprocess
    try:
        return subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True,
                              cwd=Path(__file__).resolve().parents[1]).stdout.strip() or None
    except OSError:
        return None


def find(root: Path, suffix: str):
    hits = sorted(root.rglob(f"*{suffix}"))
    return hits[0] if hits else None


def trim(name: str, n: int = 100) -> str:
    return name if len(name) <= n else name[: n - 3] + "..."


def main(src: Path, dst: Path):
    dst.mkdir(parents=True, exist_ok=True)
    (dst / "bench_c2_1gpu.json").write_text((src / "bench.json").read_text())
    (dst / "bench_c2_under_rocprof.json").write_text((src / "bench_under_rocprof.json").read_text())
    bench = json.loads((src / "bench.json").read_text())

    stats = find(src / "trace", "kernel_stats.csv")
    with open(stats) as f, open(dst / "bench_c2_kernel_stats.csv", "w", newline="") as g:
        r, w = csv.reader(f), csv.writer(g)
        for row in r:
            row[0] = trim(row[0])
            w.writerow(row)
    trace = find(src / "trace", "kernel_trace.csv")
    kernel_full = None
    with open(trace) as f, open(dst / "bench_c2_kernel_trace_mlp_ode.csv", "w", newline="") as g:
        r = csv.DictReader(f)
        keep = ["Kernel_Name", "Start_Timestamp", "End_Timestamp", "Workgroup_Size", "Grid_Size", "LDS_Block_Size",
                "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count"]
        keep = [k for k in keep if k in r.fieldnames]
        w = csv.DictWriter(g, fieldnames=keep + ["Duration_ns"])
        w.writeheader()
        for row in r:
            if "mlp_ode_kernel" in row["Kernel_Name"]:
                kernel_full = row["Kernel_Name"]
                out = {k: row[k] for k in keep}
                out["Duration_ns"] = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
                out["Kernel_Name"] = trim(out["Kernel_Name"])
                w.writerow(out)

    counters = {}
    i = 0
    while (src / f"pmc{i}").exists():
        cc = find(src / f"pmc{i}", "counter_collection.csv")
        with open(cc) as f:
            for row in csv.DictReader(f):
                if "mlp_ode_kernel" in row["Kernel_Name"]:
                    counters.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
        i += 1
    # one dispatch per pass (steps 1, warmup 0) plus the parity launches at other batch sizes: keep the
    # largest value per counter = the timed 2^20 launch
    per_launch = {k: max(v) for k, v in counters.items()}
    B = bench["config"]["batch_per_gpu"]
    read_b = per_launch["FETCH_SIZE"] * 1024 * 2          # KiB; 64-B tally of 128-B requests on gfx950
    write_b = per_launch["WRITE_SIZE"] * 1024
    cycles = per_launch["GRBM_GUI_ACTIVE"] / 8            # the counter is summed over the 8 XCDs
    ms = bench["roofline"]["kernel_ms_avg"]
    summary = {
        "command": "rocprofv3 --pmc <counter set> --output-format csv -- python bench.py --steps 1 --warmup 0 --cpu-batch 0"
                   "   (one pass per counter set; tools/profile_round.sh)",
        "kernel": kernel_full,
        "batch": B,
        "FETCH_SIZE_KB_raw": per_launch["FETCH_SIZE"],
        "WRITE_SIZE_KB_raw": per_launch["WRITE_SIZE"],
        "correction": "MI355X_MICROARCH.md HBM section: on gfx950 FETCH_SIZE reports 1/2 of the bytes of a coalesced "
                      "stream (64-B tally of 128-B requests) -> doubled; WRITE_SIZE taken as is; units are KiB",
        "read_bytes_corrected": read_b,
        "write_bytes": write_b,
        "bytes_per_launch": read_b + write_b,
        "algorithmic_bytes_per_launch": bench["roofline"]["algorithmic_hbm_bytes_per_launch"],
        "effective_clock_GHz": cycles / (ms * 1e-3) / 1e9,
        # SQ_VALU_MFMA_BUSY_CYCLES sums over the 4 SIMDs x 256 CUs
        "mfma_pipe_busy_fraction": per_launch["SQ_VALU_MFMA_BUSY_CYCLES"] / (cycles * 256 * 4)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in per_launch else None,
        "counters_per_launch": per_launch,
    }
    (dst / "bench_c2_pmc_summary.json").write_text(json.dumps(summary, indent=1))
    (dst.parent / "hbm_traffic.json").write_text(json.dumps(
        {"bytes_per_launch": read_b + write_b, "source": f"{dst}/bench_c2_pmc_summary.json", "kernel": kernel_full,
         "batch": B, "commit": _head_commit()}, indent=1))
    print(json.dumps({k: summary[k] for k in ("kernel", "bytes_per_launch", "effective_clock_GHz",
                                                "mfma_pipe_busy_fraction")}, indent=1))


if __name__ == "__main__":
    main(Path(sys.argv[1]), Path(sys.argv[2]))
