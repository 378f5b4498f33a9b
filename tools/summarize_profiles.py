"""Turn the output of tools/profile_round.sh into the files committed under profiles/rNN.

    python tools/summarize_profiles.py gpurun_out/prof_r02 profiles/r02

* bench_1gpu.json, bench_under_rocprof.json     the bench lines (headline, split-precision record, extra_configs)
* kernel_stats.csv, kernel_trace_mlp_ode.csv    rocprofv3 --kernel-trace --stats of the same command (names trimmed);
                                                the trace lists every launch of the library's kernels with its duration,
                                                each dispatch ONCE (rocprofv3 repeats rows when a run writes several
                                                per-process files)
* kernel_durations_by_grid.csv                  launches grouped by (kernel, grid size): calls, mean / min / max / sigma
                                                in ms -- kernel_stats.csv averages over every batch size a kernel ran
                                                at; the dominant launch's average is one row of THIS table
* pmc_summary.json  and  ../hbm_traffic.json    PMC counters per fused kernel, per launch (the LARGEST launch of each
                                                kernel = the full-size timed one), with the gfx950 corrections of
                                                MI355X_MICROARCH.md (FETCH_SIZE counts 64 B per 128-B request -> doubled;
                                                KiB units), effective clock and MFMA-pipe busy fraction
"""
import csv
import json
import re
import subprocess
import sys
from pathlib import Path


def _head_commit():
    """Commit the profiled tree was snapshotted from (the summariser runs in the authoring container, where git is)."""
    try:
        return subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True,
                              cwd=Path(__file__).resolve().parents[1]).stdout.strip() or None
    except OSError:
        return None


def find(root: Path, suffix: str):
    hits = sorted(root.rglob(f"*{suffix}"))
    return hits[0] if hits else None


def trim(name: str, n: int = 110) -> str:
    return name if len(name) <= n else name[: n - 3] + "..."


def main(src: Path, dst: Path):
    dst.mkdir(parents=True, exist_ok=True)
    (dst / "bench_1gpu.json").write_text((src / "bench.json").read_text())
    (dst / "bench_under_rocprof.json").write_text((src / "bench_under_rocprof.json").read_text())
    bench = json.loads((src / "bench.json").read_text())

    stats = find(src / "trace", "kernel_stats.csv")
    with open(stats) as f, open(dst / "kernel_stats.csv", "w", newline="") as g:
        r, w = csv.reader(f), csv.writer(g)
        for row in r:
            row[0] = trim(row[0])
            w.writerow(row)
    trace = find(src / "trace", "kernel_trace.csv")
    durations = {}
    by_grid = {}
    seen = set()
    with open(trace) as f, open(dst / "kernel_trace_mlp_ode.csv", "w", newline="") as g:
        r = csv.DictReader(f)
        keep = ["Kernel_Name", "Start_Timestamp", "End_Timestamp", "Workgroup_Size", "Workgroup_Size_X", "Grid_Size", "Grid_Size_X",
                "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count"]
        keep = [k for k in keep if k in r.fieldnames]
        grid_col = "Grid_Size_X" if "Grid_Size_X" in r.fieldnames else ("Grid_Size" if "Grid_Size" in r.fieldnames else None)
        w = csv.DictWriter(g, fieldnames=keep + ["Duration_ns"])
        w.writeheader()
        for row in r:
            if "ff::" not in row["Kernel_Name"]:
                continue
            key = (row["Kernel_Name"], row["Start_Timestamp"], row["End_Timestamp"])
            if key in seen:                 # the same dispatch listed twice
                continue
            seen.add(key)
            out = {k: row[k] for k in keep}
            out["Duration_ns"] = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
            out["Kernel_Name"] = trim(out["Kernel_Name"])
            durations.setdefault(row["Kernel_Name"], []).append(out["Duration_ns"])
            by_grid.setdefault((trim(row["Kernel_Name"]), row[grid_col] if grid_col else "?"), []).append(out["Duration_ns"])
            if "mlp_ode" in row["Kernel_Name"]:
                w.writerow(out)
    with open(dst / "kernel_durations_by_grid.csv", "w", newline="") as g:
        w = csv.writer(g)
        w.writerow(["Kernel_Name", "Grid_Size_X", "Calls", "Mean_ms", "Min_ms", "Max_ms", "Stddev_ms", "Total_ms"])
        for (name, grid), ds in sorted(by_grid.items(), key=lambda kv: -sum(kv[1])):
            n = len(ds)
            mean = sum(ds) / n
            sd = (sum((d - mean) ** 2 for d in ds) / n) ** 0.5
            w.writerow([name, grid, n] + [f"{v / 1e6:.4f}" for v in (mean, min(ds), max(ds), sd, sum(ds))])

    counters = {}          # kernel -> counter -> [values over dispatches]
    i = 0
    while (src / f"pmc{i}").exists():
        cc = find(src / f"pmc{i}", "counter_collection.csv")
        if cc is not None:
            with open(cc) as f:
                for row in csv.DictReader(f):
                    if "mlp_ode" in row["Kernel_Name"]:
                        counters.setdefault(row["Kernel_Name"], {}).setdefault(row["Counter_Name"], []).append(
                            float(row["Counter_Value"]))
        i += 1
    kernels = {}
    for name, cs in counters.items():
        per_launch = {k: max(v) for k, v in cs.items()}     # the largest launch = the full-size timed one
        ms = max(durations.get(name, [0])) / 1e6             # its duration in the kernel trace
        entry = {"longest_launch_ms_in_trace": ms, "counters_of_largest_launch": per_launch}
        if "FETCH_SIZE" in per_launch and "WRITE_SIZE" in per_launch:
            entry["hbm_read_bytes_corrected"] = per_launch["FETCH_SIZE"] * 1024 * 2
            entry["hbm_write_bytes"] = per_launch["WRITE_SIZE"] * 1024
            entry["hbm_bytes_per_launch"] = entry["hbm_read_bytes_corrected"] + entry["hbm_write_bytes"]
        if "GRBM_GUI_ACTIVE" in per_launch and ms > 0:
            cycles = per_launch["GRBM_GUI_ACTIVE"] / 8      # the counter is summed over the 8 XCDs
            entry["effective_clock_GHz"] = cycles / (ms * 1e-3) / 1e9
            if "SQ_VALU_MFMA_BUSY_CYCLES" in per_launch:     # summed over the 4 SIMDs x 256 CUs
                entry["mfma_pipe_busy_fraction"] = per_launch["SQ_VALU_MFMA_BUSY_CYCLES"] / (cycles * 256 * 4)
        kernels[trim(name)] = entry
    summary = {
        "command": "rocprofv3 --pmc <counter set> --output-format csv -- python bench.py --steps 1 --warmup 0 --cpu-batch 0"
                   "   (one pass per counter set; tools/profile_round.sh); clock = GRBM_GUI_ACTIVE / 8 / duration of the "
                   "same kernel's longest launch in the kernel-trace pass (profiled passes clock slightly lower)",
        "correction": "MI355X_MICROARCH.md HBM section: on gfx950 FETCH_SIZE reports 1/2 of the bytes of a coalesced "
                      "stream (64-B tally of 128-B requests) -> doubled; WRITE_SIZE taken as is; units are KiB",
        "commit": _head_commit(),
        "kernels": kernels,
    }
    (dst / "pmc_summary.json").write_text(json.dumps(summary, indent=1))
    head = bench["roofline"]["kernel"]                      # e.g. mlp_ode_m16_h256_d4_c0_t0_w2
    # the headline instance: 16-column tile, width 256, state only, two wavefronts per SIMD, ring 8, SiLU, not cooperative
    tag = re.compile(r"mlp_ode_kernel<16, 256, 4, 0, false, 2, 8, (0, )?false(, false)?>")
    for name, e in kernels.items():
        if tag.search(name) and "hbm_bytes_per_launch" in e:
            (dst.parent / "hbm_traffic.json").write_text(json.dumps(
                {"bytes_per_launch": e["hbm_bytes_per_launch"], "source": f"{dst}/pmc_summary.json", "kernel": name,
                 "batch": bench["config"]["batch_per_gpu"], "commit": _head_commit(), "bench_kernel": head}, indent=1))
    for name, e in kernels.items():
        print(name[:90], {k: (round(v, 4) if isinstance(v, float) else v) for k, v in e.items() if k != "counters_of_largest_launch"})


if __name__ == "__main__":
    main(Path(sys.argv[1]), Path(sys.argv[2]))
