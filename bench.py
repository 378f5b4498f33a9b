#!/usr/bin/env python
"""Headline benchmark: samples/s of the fused probability-flow ODE sampler on MI355X.

Workload (BASELINE.json configs[1], the configuration the metric is quoted on): 16-dim VP-SDE score
model, MLP 4x256 (embedding 8, no conditionals), 100-step RK4 (torchdiffeq's "rk4" = 3/8 rule, 400
network evaluations per sample), batch 2^20 per GPU, random-init weights (seed 0), synthetic
standard-normal base samples resident in HBM.  One "step" = one full solve of the batch.

    python bench.py --gpus 1 --steps 5 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU, every rank solves its own 2^20-sample shard (weak scaling) and the
shards meet in one RCCL all-gather per step, inside the timed region.  Rank 0 prints ONE JSON line.
After the timed loop the N > 1 run also solves the two BASELINE configurations that are WORDED for a
node -- configs[3] (64-dim flow, 2^22 rows sharded N ways) and configs[4] (conditional 32-dim VE,
1000-step Euler-Maruyama, 2^20 rows sharded N ways, noise keyed by the global row) -- one all-gather
each, under `extra_configs`, and checks that a rank's rows equal a single-launch solve of those rows
(`rank_invariant`).  Host threads are capped at usable_cores() // N per rank.

`roofline`: the path is a dense fp32 contraction (1.3e6 FLOP per algorithmic HBM byte), so the
bounding roofline is the fp32 MFMA peak; `achieved` = algorithmic FLOPs of one launch (2 x MACs of
the Linear layers x 400 evaluations x batch) / the kernel's launch duration measured with HIP
events on the launch stream.  `cpu_baseline`: the CPU oracle (unfused torch ops + Python stepping
loop, i.e. the reference's CPU algorithm restated) timed on the host cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

DIM, UNITS, EMB = 16, [256, 256, 256, 256], 8
N_STEPS = 100
PEAK_FP32_MFMA_TFLOPS = 157.3        # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_BF16_MFMA_TFLOPS = 2500.0       # MI355X_MICROARCH.md: dense bf16 MFMA peak (v_mfma_f32_32x32x16_bf16: 32 cycles)
PEAK_HBM_GBS = 8000.0


def mac_per_eval(dim, units, cond=0):
    sizes = [dim + cond + EMB] + list(units) + [dim]
    return sum(a * b for a, b in zip(sizes[:-1], sizes[1:]))


def build_model(device):
    from flowfusion_amd.diffusion import MLP, VPSDE, ScoreModel
    torch.manual_seed(0)
    mlp = MLP(n_dimensions=DIM, n_conditionals=0, embedding_dimensions=EMB, units=UNITS)
    return ScoreModel(mlp, VPSDE(), no_sigma=True).eval().to(device)


def usable_cores():
    """Host cores this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = Path("/sys/fs/cgroup/cpu.max").read_text().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


def cpu_baseline(sm, batch_cpu, opts, budget_s=20.0, threads=None):
    """Oracle ("port") timed on the host: same workload on a bounded sample (chunks of 2048 samples
    until `batch_cpu` samples or `budget_s` seconds, whichever comes first), on `threads` torch threads
    (default: every core this process may use)."""
    from oracle import flowfusion_oracle as O
    sd = {k: v.detach().cpu() for k, v in sm.state_dict().items()}
    so = O.ScoreOracle(O.mlp_params_from_state_dict(sd, "model."), O.VP(), no_sigma=True)
    cores = threads or usable_cores()
    torch.set_num_threads(cores)
    torch.manual_seed(1234)
    z = torch.randn(batch_cpu, DIM)
    so.sample_ode_from_base(z[:256], None, "rk4", {"step_size": opts["step_size"] * 10})   # warm-up
    outs, done = [], 0
    t0 = time.perf_counter()
    chunk = 2048 if cores > 1 else 256
    while done < batch_cpu and (time.perf_counter() - t0 < budget_s or done == 0):
        outs.append(so.sample_ode_from_base(z[done:done + chunk], None, "rk4", opts))
        done += outs[-1].shape[0]
    dt = time.perf_counter() - t0
    return torch.cat(outs), z[:done], {
        "value": done / dt, "unit": "samples/s", "cores": cores, "kind": "port",
        "sample": f"{done} samples x 100 RK4 steps (same model and grid, torch fp32 CPU oracle, {dt:.1f} s)"}


def _timed(fn, device):
    """(result, wall seconds, HIP-event milliseconds) of one call, synchronised on both sides."""
    torch.cuda.synchronize(device)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    r = fn()
    e1.record()
    torch.cuda.synchronize(device)
    return r, time.perf_counter() - t0, e0.elapsed_time(e1)


def _record(name, units, unit, wall_s, kernel, kernel_ms, flop, note):
    ach = flop / (kernel_ms * 1e-3) / 1e12
    return {"workload": name, "value": units / wall_s, "unit": unit, "wall_ms": 1e3 * wall_s, "kernel": kernel,
            "kernel_ms": kernel_ms, "launches": note.get("launches", 1), "flop_algorithmic": flop,
            "roofline": {"bound": "mfma", "achieved": ach, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": ach / PEAK_FP32_MFMA_TFLOPS}, "dtype": "f32", **{k: v for k, v in note.items() if k != "launches"}}


def split_precision_record(device, z, opts, steps, warmup, f32_x, precision="bf16x3"):
    """The SAME workload as the headline loop on the opt-in split-precision kernels (precision="bf16x3": every fp32
    operand cut into three bf16 parts, six bf16 MFMAs per product term, fp32 accumulate -- fp32-class accuracy).  A
    second record beside the f32 line, never instead of it.  Two rooflines: the bf16 MFMA peak against the MFMA work
    actually executed (6 x the algorithmic MACs), and the algorithmic (fp32-equivalent) rate for comparison with
    the f32 line."""
    from flowfusion_amd import _native
    sm = build_model(device)
    sm.precision = precision
    parts, products = (3, 6) if precision == "bf16x3" else (2, 3)
    B = z.shape[0]
    for _ in range(warmup):
        x, _ = sm.sample_ode_from_base(z, method="rk4", options=opts)
    torch.cuda.synchronize(device)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    t0 = time.perf_counter()
    for a, b in ev:
        a.record()
        x, _ = sm.sample_ode_from_base(z, method="rk4", options=opts)
        b.record()
    torch.cuda.synchronize(device)
    elapsed = time.perf_counter() - t0
    kms = sum(a.elapsed_time(b) for a, b in ev) / steps
    n_evals = 4 * N_STEPS
    flop_alg = 2.0 * mac_per_eval(DIM, UNITS) * n_evals * B
    # executed: 12 bf16 MFMAs (six products x two column blocks of 16 samples) per (16-row tile, 32-wide k-step) -- first
    # layer 16 tiles x 1 k-step, hidden 16 x 8 each, output 1 x 8 -- of 2 x 16 x 16 x 32 FLOP each, per 32 samples and
    # evaluation
    mfma_per_eval = 2 * products * (16 * 1 + 3 * 16 * 8 + 8)
    flop_exec = mfma_per_eval * 16384.0 * n_evals * (B / 32)
    err = float((x - f32_x).abs().max() / f32_x.abs().max())
    return {
        "metric": "samples/sec (whole node), 16-dim VP-SDE 100-step RK4", "value": B * steps / elapsed, "unit": "samples/s",
        "n_gpus": 1, "steps": steps, "warmup": warmup, "ms_per_step": 1e3 * elapsed / steps,
        "dtype": f"{precision}-split, f32 accumulate", "precision_option": f"{precision} (opt-in; the default and the headline are f32)",
        "arithmetic": ("three bf16 parts per operand by truncation (exact), six products per term" if parts == 3 else
                       "two bf16 parts per operand by round-to-nearest (16 significand bits, unbiased), three products per term"),
        "kernel": _native.kernel_name(sm._net().plan(0)), "kernel_ms_avg": kms,
        "roofline": {"bound": "mfma", "achieved": flop_exec / (kms * 1e-3) / 1e12, "peak": PEAK_BF16_MFMA_TFLOPS,
                     "unit": "TFLOP/s", "frac": flop_exec / (kms * 1e-3) / 1e12 / PEAK_BF16_MFMA_TFLOPS,
                     "flop_executed_per_launch": flop_exec, "note": f"executed bf16 MFMA FLOPs ({products} products per term) vs the dense bf16 peak"},
        "fp32_equivalent": {"achieved": flop_alg / (kms * 1e-3) / 1e12, "unit": "TFLOP/s", "flop_per_launch": flop_alg,
                            "vs_fp32_mfma_peak": flop_alg / (kms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS},
        "max_abs_diff_vs_f32_kernel_over_max_abs": err,
    }


def extra_configs(device):
    """BASELINE configs 3, 4 and 5 (and the reference's default exact-trace log_prob), ONE full-size call each
    after a small warm-up call: `value` = units / wall clock of the public method (what a user sees, host work
    included), `kernel_ms` = HIP events around the fused launch alone where the public method does host work
    first (configs 3 / exact: the probe is drawn on the CPU like the reference, diffusion.py:701), else around
    the call.  FLOPs are algorithmic: 2 x MACs of the Linear layers x evaluations x MFMA columns a sample needs
    (1 state, +1 Hutchinson tangent, +D unit tangents)."""
    from flowfusion_amd import _native
    from flowfusion_amd import flow as Fm
    from flowfusion_amd.diffusion import MLP, VESDE, ScoreModel
    name_of = lambda net, mode: _native.lib().ff_kernel_name(net.plan(mode).kernel_id).decode()
    out = []
    g = torch.Generator(device=device).manual_seed(4321)
    # --- config 3: same 16-dim model, log_prob with the Hutchinson divergence, 2^20 ------------------------------
    sm = build_model(device)
    sm.hutch = True
    eps = float(sm.sde.epsilon)
    opts = {"step_size": (1.0 - eps) / N_STEPS}
    B = 1 << 20
    x0 = torch.randn(B, DIM, device=device, generator=g) * 0.9
    sm.log_prob(x0[:256], method="rk4", options=opts)
    _, wall, _ = _timed(lambda: sm.log_prob(x0, method="rk4", options=opts), device)
    net = sm._net()
    tab = sm._ode_table(torch.tensor([eps, 1.0]), "rk4", opts, 1).to(device)
    _, _, kms = _timed(lambda: net.integrate(x0, tab, 1, probe=sm.e), device)
    out.append(_record("BASELINE configs[2]: 16-dim VP-SDE 4x256, log_prob, Hutchinson divergence, 100-step RK4, batch 2^20",
                       B, "log-probs/s", wall, name_of(net, 1), kms, 2 * 2.0 * mac_per_eval(DIM, UNITS) * tab.shape[0] * B,
                       {"note": "wall includes the reference's CPU draw of the probe (diffusion.py:701) and its upload"}))
    # the same call with the probe taken from the counter-based stream on the device (probe="philox": no host draw)
    sm.log_prob(x0[:256], method="rk4", options=opts, probe="philox", seed=5)
    _, wall_p, _ = _timed(lambda: sm.log_prob(x0, method="rk4", options=opts, probe="philox", seed=5), device)
    out[-1]["wall_ms_probe_philox"] = 1e3 * wall_p
    out[-1]["value_probe_philox"] = B / wall_p
    # the same Hutchinson call on the opt-in 16-bit-operand kernels
    sm.precision = "bf16x2"
    sm.log_prob(x0[:256], method="rk4", options=opts)
    _, wall_h2, _ = _timed(lambda: sm.log_prob(x0, method="rk4", options=opts), device)
    net2 = sm._net()
    _, _, kms_h2 = _timed(lambda: net2.integrate(x0, tab, 1, probe=sm.e, stage_slots=4), device)
    out.append({"workload": "BASELINE configs[2] on precision='bf16x2' (opt-in): the Hutchinson log_prob above", "value": B / wall_h2,
                "unit": "log-probs/s", "wall_ms": 1e3 * wall_h2, "kernel": _native.kernel_name(net2.plan(1)), "kernel_ms": kms_h2,
                "dtype": "bf16x2-split, f32 accumulate", "speedup_vs_f32_entry": out[-1]["wall_ms"] / (1e3 * wall_h2)})
    sm.precision = "f32"
    net = sm._net()
    # --- the reference's default divergence: exact trace (D unit tangents), 2^16 ---------------------------------
    sm.hutch = False
    Be = 1 << 16
    sm.log_prob(x0[:64], method="rk4", options=opts)
    _, wall, _ = _timed(lambda: sm.log_prob(x0[:Be].contiguous(), method="rk4", options=opts), device)
    tab2 = sm._ode_table(torch.tensor([eps, 1.0]), "rk4", opts, 2).to(device)
    xe = x0[:Be].contiguous()
    _, _, kms = _timed(lambda: net.integrate(xe, tab2, 2), device)
    from flowfusion_amd.fused import exact_trace_passes
    passes = exact_trace_passes(DIM, net.plan(2).tile)
    out.append(_record("16-dim VP-SDE 4x256, log_prob with the exact trace (reference default divergence), 100-step RK4, batch 2^16",
                       Be, "log-probs/s", wall, name_of(net, 2), kms, (DIM + 1) * 2.0 * mac_per_eval(DIM, UNITS) * tab2.shape[0] * Be,
                       {"launches": len(passes), "note": "kernel_ms sums the launches; columns carried: "
                        + " + ".join(f"(1+{c})" for _, c in passes) + f" for {DIM}+1 needed"}))
    # the same exact-trace call on the opt-in 16-bit-operand kernels (a value column + unit tangents per column block of 16)
    sm.precision = "bf16x2"
    sm.log_prob(xe[:64].contiguous(), method="rk4", options=opts)
    _, wall2, kms2 = _timed(lambda: sm.log_prob(xe, method="rk4", options=opts), device)
    out.append({"workload": "the exact-trace log_prob above on precision='bf16x2' (opt-in)", "value": Be / wall2, "unit": "log-probs/s",
                "wall_ms": 1e3 * wall2, "kernel": _native.kernel_name(sm._net().plan(2)), "kernel_ms": kms2,
                "dtype": "bf16x2-split, f32 accumulate", "speedup_vs_f32_entry": out[-1]["wall_ms"] / (1e3 * wall2)})
    sm.precision = "f32"
    # --- small batches: one solve of config 2 at 4096 samples (latency-bound: 256 tiles for 1024 SIMDs) -----------------
    # default dispatch = the cooperative twin (a tile per workgroup); FF_COOP=0 pins the one-wavefront kernel
    sm.hutch = False
    zs = x0[:4096].contiguous()
    fwd = {"step_size": (1.0 - eps) / N_STEPS}
    lat = {}
    for pin in (None, "0"):
        if pin is None:
            os.environ.pop("FF_COOP", None)
        else:
            os.environ["FF_COOP"] = pin
        sm.sample_ode_from_base(zs, method="rk4", options=fwd)
        lat[pin] = min(_timed(lambda: sm.sample_ode_from_base(zs, method="rk4", options=fwd), device)[1] for _ in range(3))
    os.environ.pop("FF_COOP", None)
    out.append({"workload": "small batch: BASELINE configs[1] model, 100-step RK4, batch 4096 (latency of ONE solve)",
                "value": 4096 / lat[None], "unit": "samples/s", "wall_ms": 1e3 * lat[None],
                "wall_ms_one_wavefront_kernel": 1e3 * lat["0"], "speedup_of_cooperative_twin": lat["0"] / lat[None],
                "dtype": "f32", "note": "below ~3/4 of a chip's worth of tiles the launcher gives each tile to a workgroup "
                "(rows of a layer split over its 4 wavefronts, LDS exchange per layer); bitwise the same results"})
    # --- the reference's DEFAULT log_prob arguments at scale: adaptive dopri5 (atol = rtol = 1e-4, min_step 1e-6) with the
    # Hutchinson probe and with the exact trace (two unit-tangent passes per attempted step), 2^18 points of config 2's model.
    # The whole loop runs on the device (csrc/ff_adaptive.hip); FF_HOST_CONTROLLER=1 is round 2's host loop, for the record.
    Bd = 1 << 18
    xd = x0[:Bd].contiguous()
    for hutch in (True, False):
        sm.hutch = hutch
        entry = {"workload": "default-argument log_prob (adaptive dopri5, atol = rtol = 1e-4, min_step 1e-6), 16-dim VP-SDE 4x256, "
                             + ("Hutchinson probe" if hutch else "exact trace (reference default divergence)") + ", batch 2^18",
                 "unit": "log-probs/s", "dtype": "f32", "kernel": name_of(sm._net(), 1 if hutch else 2)}
        try:
            sm.log_prob(xd[:4096].contiguous())
            for tag, env in (("device_controller", None), ("host_controller", "1")):
                if env is None:
                    os.environ.pop("FF_HOST_CONTROLLER", None)
                else:
                    os.environ["FF_HOST_CONTROLLER"] = env
                _, wall, kms = _timed(lambda: sm.log_prob(xd), device)
                st = dict(sm.last_solver_stats)
                cols = 2 if hutch else DIM + 1
                flop = cols * 2.0 * mac_per_eval(DIM, UNITS) * (6 * st["attempts"] + 2) * Bd
                entry[tag] = {"wall_ms": 1e3 * wall, "hip_event_ms": kms, "value": Bd / wall, **st,
                              "achieved_TFLOPs_algorithmic": flop / wall / 1e12,
                              "frac_of_fp32_mfma_peak": flop / wall / 1e12 / PEAK_FP32_MFMA_TFLOPS}
            entry["value"] = entry["device_controller"]["value"]
            entry["speedup_vs_host_controller"] = entry["host_controller"]["wall_ms"] / entry["device_controller"]["wall_ms"]
        except RuntimeError as e:
            entry["error"] = str(e)
        finally:
            os.environ.pop("FF_HOST_CONTROLLER", None)
        out.append(entry)
    sm.hutch = False
    del sm, net, x0, xe, zs, xd
    # --- the reference's notebook shape (BASELINE configs[0] as demo_diffusion.ipynb has it): 2-D VE, 3x128, 50,000 points,
    # DEFAULT arguments -- adaptive dopri5 for sampling (cell 388), dopri5 + exact trace for log_prob (cell 467)
    torch.manual_seed(0)
    nb = ScoreModel(MLP(2, 0, EMB, [128] * 3), VESDE()).eval().to(device)
    zb = torch.randn(50000, 2, device=device, generator=g)
    xb = torch.randn(50000, 2, device=device, generator=g) * 0.5
    entry = {"workload": "BASELINE configs[0] as the notebook runs it: 2-D VE-SDE 3x128, 50,000 points, default arguments "
                         "(sample_ode_from_base: adaptive dopri5; log_prob: dopri5 + exact trace), wall ms of ONE call", "unit": "ms"}
    for prec in ("f32", "bf16x2"):
        nb.precision = prec
        nb.sample_ode_from_base(zb[:512].contiguous())
        nb.log_prob(xb[:512].contiguous())
        entry[f"sample_ms_{prec}"] = 1e3 * min(_timed(lambda: nb.sample_ode_from_base(zb), device)[1] for _ in range(3))
        entry[f"sample_attempts_{prec}"] = dict(nb.last_solver_stats)
        entry[f"log_prob_ms_{prec}"] = 1e3 * min(_timed(lambda: nb.log_prob(xb), device)[1] for _ in range(3))
        entry[f"log_prob_attempts_{prec}"] = dict(nb.last_solver_stats)
    # round 2's host-side step controller on the same calls (FF_HOST_CONTROLLER=1), for the record
    nb.precision = "f32"
    os.environ["FF_HOST_CONTROLLER"] = "1"
    try:
        nb.sample_ode_from_base(zb[:512].contiguous())
        nb.log_prob(xb[:512].contiguous())
        entry["sample_ms_f32_host_controller"] = 1e3 * min(_timed(lambda: nb.sample_ode_from_base(zb), device)[1] for _ in range(3))
        entry["sample_attempts_f32_host_controller"] = dict(nb.last_solver_stats)
        entry["log_prob_ms_f32_host_controller"] = 1e3 * min(_timed(lambda: nb.log_prob(xb), device)[1] for _ in range(3))
    finally:
        os.environ.pop("FF_HOST_CONTROLLER", None)
    out.append(entry)
    # --- the only timings the reference prints (CPU tensors, hardware not stated): the notebooks' own
    # log_prob calls.  Random-init weights here (a trained network's flow is smoother or stiffer, so attempt counts differ):
    # same shapes, same arguments, reference seconds beside ours for scale -- not a like-for-like speed-up.
    nbk = {"workload": "the reference notebooks' timed calls, same shapes and arguments (random-init weights): wall ms of ONE call",
           "unit": "ms", "dtype": "f32"}
    nb.precision = "f32"
    for name, hutch, ref_s, cell in (("diffusion_log_prob_exact_50000", False, 30.88, "demo_diffusion.ipynb:453/467"),
                                     ("diffusion_log_prob_hutchinson_50000", True, 15.79, "demo_diffusion.ipynb:454/472")):
        nb.hutch = hutch
        nb.log_prob(xb[:512].contiguous())
        ms = 1e3 * min(_timed(lambda: nb.log_prob(xb), device)[1] for _ in range(3))
        nbk[name] = {"ms": ms, **dict(nb.last_solver_stats), "reference_notebook_s": ref_s, "reference_cell": cell}
    nb.hutch = False
    # Hutch++ / XTrace with the notebook's arguments (one probe each: hpp_rank = hpp_vecs = xt_vecs = 1; everything else at
    # its default): the fused attempt records every row's Jacobian, one launch estimates them, the controller stays on the
    # device.  FF_HOST_CONTROLLER=1 + FF_TORCH_ESTIMATOR=1 is rounds 1-3's route (a launch per right-hand side, torch estimator, host controller).
    for name, attr, ref_s, cell in (("diffusion_log_prob_hutchpp_50000", "hutchpp", 46.28, "demo_diffusion.ipynb:455"),
                                    ("diffusion_log_prob_xtrace_50000", "xtrace", 34.35, "demo_diffusion.ipynb:456")):
        setattr(nb, attr, True)
        try:
            nb.log_prob(xb[:512].contiguous())
            ms = 1e3 * min(_timed(lambda: nb.log_prob(xb), device)[1] for _ in range(3))
            nbk[name] = {"ms": ms, **dict(nb.last_solver_stats), "reference_notebook_s": ref_s, "reference_cell": cell}
            os.environ["FF_HOST_CONTROLLER"] = os.environ["FF_TORCH_ESTIMATOR"] = "1"
            nb.log_prob(xb[:512].contiguous())
            nbk[name]["ms_host_route_of_rounds_1_to_3"] = 1e3 * _timed(lambda: nb.log_prob(xb), device)[1]
        except RuntimeError as e:
            nbk[name] = {"error": str(e)}
        finally:
            os.environ.pop("FF_HOST_CONTROLLER", None)
            os.environ.pop("FF_TORCH_ESTIMATOR", None)
            setattr(nb, attr, False)
    torch.manual_seed(0)
    fl = Fm.ODEFlow(target_dimension=2, hidden_units=[128, 128, 128]).eval().to(device)      # demo_flow.ipynb cell 7
    xf = torch.randn(25000, 2, device=device, generator=g) * 2.0
    fl.log_prob(xf[:512].contiguous(), atol=1e-4, rtol=1e-4)
    ms = 1e3 * min(_timed(lambda: fl.log_prob(xf, atol=1e-4, rtol=1e-4), device)[1] for _ in range(3))
    nbk["flow_log_prob_exact_25000"] = {"ms": ms, **dict(fl.last_solver_stats), "reference_notebook_s": 10.79,
                                        "reference_cell": "demo_flow.ipynb:409/419"}
    zf = torch.randn(50000, 2, device=device, generator=g)
    fl.sample(zf[:512].contiguous())
    ms = 1e3 * min(_timed(lambda: fl.sample(zf), device)[1] for _ in range(3))
    nbk["flow_sample_50000_default_tolerances"] = {"ms": ms, **dict(fl.last_solver_stats), "reference_notebook_s": None,
                                                   "reference_cell": "demo_flow.ipynb:377 (not timed by the reference)"}
    out.append(nbk)
    del nb, zb, xb, fl, xf, zf
    # --- config 4: 64-dim flow matching, 5x512, 200 fixed Dormand-Prince steps, 2^22 / 8 GPUs = 2^19 per GPU ------
    torch.manual_seed(0)
    f = Fm.ODEFlow(64, [512] * 5).to(device).eval()
    B4 = 1 << 19
    xT = torch.randn(B4, 64, device=device, generator=g)
    o4 = {"step_size": 1.0 / 200}
    f.sample(xT[:64].contiguous(), method="dopri5_fixed", options=o4)
    _, wall, kms = _timed(lambda: f.sample(xT, method="dopri5_fixed", options=o4), device)
    mac4 = 65 * 512 + 4 * 512 * 512 + 512 * 64
    out.append(_record("BASELINE configs[3]: 64-dim flow matching 5x512, 200-step fixed Dormand-Prince (1200 evals), "
                       "per-GPU share 2^19 of 2^22", B4, "samples/s", wall, name_of(f._net(), 0), kms,
                       2.0 * mac4 * 1200 * B4, {}))
    del f, xT
    # --- config 5: conditional 32-dim VE, 4x256, C = 8, 1000-step Euler-Maruyama, 2^20 ---------------------------
    torch.manual_seed(0)
    sm5 = ScoreModel(MLP(32, 8, EMB, UNITS), VESDE()).eval().to(device)
    B5 = 1 << 20
    cond = torch.randn(B5, 8, device=device, generator=g)
    mac5 = mac_per_eval(32, UNITS, 8)
    sm5.sample_sde((256, 32), conditional=cond[:256].contiguous(), steps=1000)
    for kind, kw, note in (("torch noise (the reference's random stream; 1 GiB noise buffers filled on a side stream)", {}, "torch"),
                           ("in-kernel counter-based noise, one launch", {"noise": "philox", "seed": 1}, "philox")):
        sm5.sample_sde((256, 32), conditional=cond[:256].contiguous(), steps=8, **kw)
        _, wall, kms = _timed(lambda: sm5.sample_sde((B5, 32), conditional=cond, steps=1000, **kw), device)
        out.append(_record(f"BASELINE configs[4]: conditional 32-dim VE-SDE 4x256 (8 conditionals), 1000-step Euler-Maruyama, "
                           f"batch 2^20, {kind}", B5, "samples/s", wall, name_of(sm5._net(), 0), kms,
                           2.0 * mac5 * 1000 * B5,
                           {"launches": 1 if note == "philox" else -(-1000 // max(1, (1 << 28) // (B5 * 32))),
                            "note": "wall and kernel_ms bracket the whole public call (prior draw on the host like the "
                                    "reference, noise kernels, launches)"}))
    # the same workload on the opt-in 16-bit-operand kernels (precision="bf16x2": dim <= 32, Euler-Maruyama rows on the split
    # kernels); extra records, f32 stays the arithmetic of the entries above
    f32_wall = {r["workload"].split("batch 2^20, ")[1][:5]: r["wall_ms"] for r in out[-2:]}
    sm5.precision = "bf16x2"
    for kind, kw, tag in (("torch noise", {}, "torch"), ("in-kernel counter-based noise, one launch", {"noise": "philox", "seed": 1}, "in-ke")):
        sm5.sample_sde((256, 32), conditional=cond[:256].contiguous(), steps=8, **kw)
        _, wall, kms = _timed(lambda: sm5.sample_sde((B5, 32), conditional=cond, steps=1000, **kw), device)
        out.append({"workload": f"BASELINE configs[4] on precision='bf16x2' (opt-in): conditional 32-dim VE-SDE 4x256, 1000-step "
                                f"Euler-Maruyama, batch 2^20, {kind}", "value": B5 / wall, "unit": "samples/s", "wall_ms": 1e3 * wall,
                    "kernel": _native.kernel_name(sm5._net().plan(0)), "kernel_ms": kms, "dtype": "bf16x2-split, f32 accumulate",
                    "speedup_vs_f32_entry": f32_wall[tag] / (1e3 * wall),
                    "roofline": {"bound": "mfma", "achieved": 3 * 2.0 * mac5 * 1000 * B5 / (kms * 1e-3) / 1e12, "peak": PEAK_BF16_MFMA_TFLOPS,
                                 "unit": "TFLOP/s", "frac": 3 * 2.0 * mac5 * 1000 * B5 / (kms * 1e-3) / 1e12 / PEAK_BF16_MFMA_TFLOPS,
                                 "note": "executed bf16 MFMA work (three products per term, padding of the 40-feature first layer not "
                                         "counted) vs the dense bf16 peak"}})
    return out


def sharded_extras(device, world, rank, dist, backend, rows_c4, rows_c5, flow_steps=200, em_steps=1000, rows_c3=1 << 20,
                   logp_steps=N_STEPS):
    """N > 1: the BASELINE configurations beside the headline, sharded `world` ways, one all-gather each.
    configs[2]: the headline model's log_prob with the Hutchinson divergence, `logp_steps`-step RK4, `rows_c3` rows over the
    node; evaluation points and the +-1 probe keyed by the GLOBAL row (distributed.log_prob_sharded, probe="philox").
    configs[3]: 64-dim flow 5x512, `flow_steps`-step fixed Dormand-Prince, `rows_c4` rows over the node; base samples from
    the library's counter-based stream keyed by the GLOBAL row, so every world size transports the same points.
    configs[4]: conditional 32-dim VE 4x256 (8 conditionals), `em_steps`-step Euler-Maruyama, `rows_c5` rows; prior,
    per-step noise and the conditionals keyed by the global row (distributed.sample_sde_sharded).
    Every rank reports its kernel ms (HIP events around its solve) and gather ms; rank-invariance: the rank's first 256
    rows equal a single-launch solve of those rows, bit for bit."""
    from flowfusion_amd import _native
    from flowfusion_amd import flow as Fm
    from flowfusion_amd.diffusion import MLP, VESDE, ScoreModel
    from flowfusion_amd.distributed import flow_sample_sharded, gather_rows, log_prob_sharded, sample_sde_sharded, shard_bounds
    gather_dev = device if backend == "nccl" else torch.device("cpu")
    recs, invariant = [], True

    def timed_gather(local, n_total):
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        full = gather_rows(local if backend == "nccl" else local.cpu(), n_total)
        if backend == "nccl":
            torch.cuda.synchronize(device)
        return full, 1e3 * (time.perf_counter() - t0)

    def per_rank(values):
        mine = torch.tensor(values, device=gather_dev, dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        return [[float(t[i]) for t in allr] for i in range(len(values))]

    # --- configs[2] ---------------------------------------------------------------------------------------------------
    sm3 = build_model(device)
    sm3.hutch = True
    lo, hi = shard_bounds(rows_c3, world, rank)
    x3 = _native.normal_fill(hi - lo, DIM, 4321, lo, device, scale=0.8)
    o3 = {"step_size": (1.0 - float(sm3.sde.epsilon)) / logp_steps}
    kw3 = {"local_x": x3, "n_total": rows_c3, "seed": 5, "gather": False, "method": "rk4", "options": o3}
    sm3.log_prob(x3[:64].contiguous(), method="rk4", options=o3, probe="philox", seed=5, sample_offset=lo)
    dist.barrier()
    (lp, _), wall, kms = _timed(lambda: log_prob_sharded(sm3, **kw3), device)
    full, gms = timed_gather(lp, rows_c3)
    head = sm3.log_prob(x3[:256].contiguous(), method="rk4", options=o3, probe="philox", seed=5, sample_offset=lo)
    ok3 = bool(torch.equal(head, lp[:256])) and bool(torch.equal(full[lo:lo + 256].to(device), lp[:256]))
    invariant &= ok3
    walls, kmss, gmss = per_rank([1e3 * wall, kms, gms])
    n_evals3 = 4 * logp_steps
    slow = max(w + g for w, g in zip(walls, gmss)) * 1e-3
    recs.append({"workload": f"BASELINE configs[2]: 16-dim VP-SDE 4x256 log_prob, Hutchinson divergence, {logp_steps}-step RK4, "
                             f"{rows_c3} rows sharded over {world} GPUs, probe keyed by the global row, one all-gather",
                 "value": rows_c3 / slow, "unit": "log-probs/s", "rows_per_rank": hi - lo,
                 "kernel": _native.kernel_name(sm3._net().plan(1)),
                 "per_rank": {"wall_ms": walls, "kernel_ms": kmss, "allgather_ms": gmss}, "dtype": "f32",
                 "roofline": {"bound": "mfma", "achieved": 4.0 * mac_per_eval(DIM, UNITS) * n_evals3 * (hi - lo) / (kms * 1e-3) / 1e12,
                              "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                              "frac": 4.0 * mac_per_eval(DIM, UNITS) * n_evals3 * (hi - lo) / (kms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                              "note": "rank 0's launch (probe draw on the device included in kernel_ms); value + tangent column "
                                      "= 2x the state-only work"}, "rank_invariant": ok3})
    del sm3, x3, lp, full, head
    # --- configs[3] ---------------------------------------------------------------------------------------------------
    torch.manual_seed(0)
    f = Fm.ODEFlow(64, [512] * 5).to(device).eval()
    lo, hi = shard_bounds(rows_c4, world, rank)
    o4 = {"step_size": 1.0 / flow_steps}
    kw4 = {"seed": 2024, "gather": False, "method": "dopri5_fixed", "options": o4}
    f.sample(_native.normal_fill(64, 64, 2024, lo, device), method="dopri5_fixed", options=o4)
    dist.barrier()
    # distributed.flow_sample_sharded: base samples drawn on the device keyed by the global row, this rank's rows integrated
    (y, span), wall, kms = _timed(lambda: flow_sample_sharded(f, rows_c4, **kw4), device)
    assert span == (lo, hi)
    full, gms = timed_gather(y, rows_c4)
    head = f.sample(_native.normal_fill(min(256, hi - lo), 64, 2024, lo, device), method="dopri5_fixed", options=o4)
    ok4 = bool(torch.equal(head, y[:256])) and bool(torch.equal(full[lo:lo + 256].to(device), y[:256]))
    invariant &= ok4
    walls, kmss, gmss = per_rank([1e3 * wall, kms, gms])
    mac4 = 65 * 512 + 4 * 512 * 512 + 512 * 64
    n_evals4 = 6 * flow_steps
    slow = max(w + g for w, g in zip(walls, gmss)) * 1e-3
    recs.append({"workload": f"BASELINE configs[3]: 64-dim flow matching 5x512, {flow_steps}-step fixed Dormand-Prince "
                             f"({n_evals4} evals), {rows_c4} rows sharded over {world} GPUs, one all-gather",
                 "value": rows_c4 / slow, "unit": "samples/s", "rows_per_rank": hi - lo, "kernel": _native.kernel_name(f._net().plan(0)),
                 "per_rank": {"wall_ms": walls, "kernel_ms": kmss, "allgather_ms": gmss}, "dtype": "f32",
                 "roofline": {"bound": "mfma", "achieved": 2.0 * mac4 * n_evals4 * (hi - lo) / (kms * 1e-3) / 1e12,
                              "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                              "frac": 2.0 * mac4 * n_evals4 * (hi - lo) / (kms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                              "note": "rank 0's launch (base-sample draw on the device included in kernel_ms)"}, "rank_invariant": ok4})
    del f, y, full, head
    # --- configs[4] ---------------------------------------------------------------------------------------------------
    torch.manual_seed(0)
    sm5 = ScoreModel(MLP(32, 8, EMB, UNITS), VESDE()).eval().to(device)
    lo, hi = shard_bounds(rows_c5, world, rank)
    cond = _native.normal_fill(hi - lo, 8, 77, lo, device)            # conditionals keyed by the global row as well
    sample_sde_sharded(sm5, (rows_c5, 32), steps=8, seed=1, gather=False, local_conditional=cond)
    dist.barrier()
    (y, _), wall, kms = _timed(lambda: sample_sde_sharded(sm5, (rows_c5, 32), steps=em_steps, seed=1, gather=False,
                                                           local_conditional=cond), device)
    full, gms = timed_gather(y, rows_c5)
    # the same rows as a launch of their own: prior, noise and conditionals are functions of the global row
    scale = float(sm5.sde.sigma_max)
    x256 = _native.normal_fill(min(256, hi - lo), 32, 1, lo, device, scale=scale)
    head = sm5._sample_sde_from(x256, None, cond[:256].contiguous(), em_steps, rng=(1, lo))
    ok5 = bool(torch.equal(head, y[:256])) and bool(torch.equal(full[lo:lo + 256].to(device), y[:256]))
    invariant &= ok5
    walls, kmss, gmss = per_rank([1e3 * wall, kms, gms])
    mac5 = mac_per_eval(32, UNITS, 8)
    slow = max(w + g for w, g in zip(walls, gmss)) * 1e-3
    recs.append({"workload": f"BASELINE configs[4]: conditional 32-dim VE-SDE 4x256 (8 conditionals), {em_steps}-step "
                             f"Euler-Maruyama, {rows_c5} rows sharded over {world} GPUs, in-kernel noise keyed by the global row, "
                             "one all-gather", "value": rows_c5 / slow, "unit": "samples/s", "rows_per_rank": hi - lo,
                 "kernel": _native.kernel_name(sm5._net().plan(0)),
                 "per_rank": {"wall_ms": walls, "kernel_ms": kmss, "allgather_ms": gmss}, "dtype": "f32",
                 "roofline": {"bound": "mfma", "achieved": 2.0 * mac5 * em_steps * (hi - lo) / (kms * 1e-3) / 1e12,
                              "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                              "frac": 2.0 * mac5 * em_steps * (hi - lo) / (kms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                              "note": "rank 0's launch (prior draw included in kernel_ms)"}, "rank_invariant": ok5})
    flag = torch.tensor([1.0 if invariant else 0.0], device=gather_dev, dtype=torch.float64)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return recs, bool(flag.item() == 1.0)


def adaptive_exchange_extra(device, world, rank, dist, backend, rows):
    """N > 1: the one exchange step of the path.  The headline model's default-argument log_prob (adaptive dopri5,
    Hutchinson probe keyed by the global row), `rows` points cut over the ranks, step size from the error norm of the WHOLE
    batch as torchdiffeq takes it (distributed.global_step_control: the sums of squares behind every norm are all-reduced
    between the reduction and the controller kernel).  Every rank also solves the whole batch alone (it fits one GPU) and
    compares: same attempt / accept counts, its rows within rounding of the whole-batch solve."""
    from flowfusion_amd import _native
    from flowfusion_amd.distributed import log_prob_sharded, shard_bounds
    gather_dev = device if backend == "nccl" else torch.device("cpu")
    sm = build_model(device)
    sm.hutch = True
    lo, hi = shard_bounds(rows, world, rank)
    x = _native.normal_fill(rows, DIM, 4321, 0, device, scale=0.8)
    mine = x[lo:hi].contiguous()
    log_prob_sharded(sm, local_x=mine, n_total=rows, seed=5, gather=False)                       # warm-up (same collectives on all ranks)
    dist.barrier()
    (lp, _), wall, _ = _timed(lambda: log_prob_sharded(sm, local_x=mine, n_total=rows, seed=5, gather=False), device)
    st = dict(sm.last_solver_stats)
    (alone, _), wall_alone, _ = _timed(lambda: log_prob_sharded(sm, local_x=mine, n_total=rows, seed=5, gather=False,
                                                                global_control=False), device)
    st_alone = dict(sm.last_solver_stats)
    whole, wall_whole, _ = _timed(lambda: sm.log_prob(x, probe="philox", seed=5), device)
    st_whole = dict(sm.last_solver_stats)
    rel = lambda a, b: float(((a - b).abs() / b.abs().clamp_min(1.0)).max())
    vals = torch.tensor([1e3 * wall, 1e3 * wall_alone, st["attempts"], st["accepted"], st_alone["attempts"],
                         rel(lp, whole[lo:hi]), rel(alone, whole[lo:hi])], device=gather_dev, dtype=torch.float64)
    allr = [torch.zeros_like(vals) for _ in range(world)]
    dist.all_gather(allr, vals)
    col = lambda i: [float(t[i]) for t in allr]
    same = all(int(a) == st_whole["attempts"] for a in col(2)) and all(int(a) == st_whole["accepted"] for a in col(3))
    return {"workload": f"default-argument log_prob (adaptive dopri5, Hutchinson), 16-dim VP-SDE 4x256, {rows} points sharded over "
                        f"{world} GPUs with whole-batch step control (one all-reduce of 8 doubles per error norm)",
            "value": rows / (max(col(0)) * 1e-3), "unit": "log-probs/s", "dtype": "f32",
            "whole_batch_on_one_gpu": {"wall_ms": 1e3 * wall_whole, **st_whole},
            "per_rank": {"wall_ms": col(0), "wall_ms_own_norm_only": col(1), "attempts": col(2), "accepted": col(3),
                         "attempts_own_norm_only": col(4), "max_rel_diff_vs_whole_batch_solve": col(5),
                         "max_rel_diff_vs_whole_batch_solve_own_norm_only": col(6)},
            "steps_equal_whole_batch_solve_on_all_ranks": bool(same)}


EXIT_EXTRAS_FAILED = 3     # the JSON line is printed, but a watchdog fired or an extras section carries `error`


def run_guarded(fn, seconds, device=None):
    """fn() in a daemon thread: (result, None), or (None, reason) after an exception or `seconds` without an answer -- a
    collective that never completes must not take the line this run has already measured with it."""
    import threading
    box = {}

    def body():
        try:
            if device is not None:
                torch.cuda.set_device(device)          # (the current device is per thread)
            box["r"] = fn()
        except Exception as exc:          # noqa: BLE001
            box["e"] = repr(exc)
    t = threading.Thread(target=body, daemon=True)
    t.start()
    t.join(seconds)
    if t.is_alive():
        return None, f"no answer within {seconds} s"
    return box.get("r"), box.get("e")


def streaming_helpers(device):
    """The memory-bound kernels beside the fused integrator (csrc/ff_aux.hip: Runge-Kutta stage algebra of the module path,
    error norms of an adaptive step, the counter-based prior draw) at solver-sized arrays (2^22 x 16 fp32): algorithmic
    bytes / HIP-event time against the 8 TB/s HBM roofline, next to torch's own copy kernel on the same box."""
    from flowfusion_amd import _native
    B, D = 1 << 22, 16
    n = B * D
    x = torch.randn(B, D, device=device)
    ks = [torch.randn(B, D, device=device) for _ in range(7)]
    out = torch.empty_like(x)

    def ms_of(fn, reps=10):
        for _ in range(2):
            fn()
        torch.cuda.synchronize(device)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize(device)
        return a.elapsed_time(b) / reps

    rows = []

    def rec(kernel, nbytes, ms):
        gbs = nbytes / (ms * 1e-3) / 1e9
        rows.append({"kernel": kernel, "ms": ms, "roofline": {"bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                                                "frac": gbs / PEAK_HBM_GBS, "algorithmic_bytes": nbytes}})

    rec("torch copy_ (yardstick: 8 B per element)", 8 * n, ms_of(lambda: out.copy_(x)))
    rec("ff_stage_combine, 4 terms + x (24 B per element)", 24 * n,
        ms_of(lambda: _native.stage_combine(out, x, ks[:4], [0.1, 0.2, 0.3, 0.4], 1.0)))
    rec("ff_stage_combine, 7 terms + x (36 B per element)", 36 * n,
        ms_of(lambda: _native.stage_combine(out, x, ks, [0.1 * (i + 1) for i in range(7)], 1.0)))
    rec("ff_scaled_rms, error norm of an adaptive step + finiteness check of y1 (err, y0, y1: 12 B per element, one pass; "
        "the kernel alone, as the device-side controller runs it: no read-back)", 12 * n,
        ms_of(lambda: _native.scaled_rms([(ks[1], None, x, ks[0])], 1e-5, 1e-5, check=ks[0], read=False)))
    rec("ff_normal_fill (4 B per element written; Philox4x32-10 + Box-Muller)", 4 * n,
        ms_of(lambda: _native.normal_fill(B, D, 1234, 0, device)))
    # the Hutch++ / XTrace estimator launch of an adaptive attempt at solver size: 6 evaluation rows x 2^16 samples of 16 x 16
    # Jacobians (one read of them is the algorithmic traffic), one sketch probe and one residual probe as in the reference's
    # defaults; the LDS-tile kernel (D <= 16) and, for the record, the general one-thread-per-item kernel
    Bj = 1 << 16
    jac = torch.randn(6, Bj, D, D, device=device)
    S, G = (torch.sign(torch.randn(1, Bj, D, device=device)) for _ in range(2))
    nbytes = jac.numel() * 4 + 4 * 6 * Bj
    rec("ff_trace_estimate, Hutch++ r = m = 1 on 6 x 2^16 Jacobians of 16 x 16 (1 KiB read per work item; LDS-tile kernel)", nbytes,
        ms_of(lambda: _native.trace_estimate(jac, "hutchpp", (S, G))))
    rec("ff_trace_estimate, XTrace m = 1, same Jacobians", nbytes, ms_of(lambda: _native.trace_estimate(jac, "xtrace", (S,))))
    os.environ["FF_TRACE_GENERIC"] = "1"
    try:
        rec("ff_trace_estimate, Hutch++ r = m = 1, general kernel (FF_TRACE_GENERIC=1: what shapes beyond the tile path take)", nbytes,
            ms_of(lambda: _native.trace_estimate(jac, "hutchpp", (S, G)), reps=3))
    finally:
        os.environ.pop("FF_TRACE_GENERIC", None)
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=1 << 20, help="samples per GPU")
    ap.add_argument("--cpu-batch", type=int, default=32768, help="samples for the CPU baseline (0 = skip)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend for N > 1 (gloo: rehearsal of the multi-rank path on one GPU)")
    ap.add_argument("--no-extras", dest="extras", action="store_false",
                    help="skip the one-launch timings of BASELINE configs 3, 4, 5 (N = 1 only)")
    ap.add_argument("--watchdog-seconds", type=float, nargs=2, default=(900.0, 300.0),
                    help="N > 1: patience with the sharded extras / the adaptive extra before the line is printed without them")
    ap.add_argument("--test-desert-rank", type=int, default=-1,
                    help="tests: this rank stays away from the adaptive extra, so the others wait for it in a collective")
    ap.add_argument("--group-of-one", action="store_true",
                    help="rehearsal: run the N > 1 branch with a one-rank process group (needs --gpus 1)")
    ap.add_argument("--single-device", action="store_true",
                    help="rehearsal: every rank uses cuda:0 (needs --backend gloo)")
    ap.add_argument("--c4-rows", type=int, default=1 << 22, help="N > 1: rows of BASELINE configs[3] over the node")
    ap.add_argument("--c5-rows", type=int, default=1 << 20, help="N > 1: rows of BASELINE configs[4] over the node")
    ap.add_argument("--adaptive-rows", type=int, default=1 << 18,
                    help="N > 1: points of the default-argument (adaptive) log_prob solved with whole-batch step control (0 = skip)")
    ap.add_argument("--c3-rows", type=int, default=1 << 20, help="N > 1: rows of BASELINE configs[2] (log_prob) over the node")
    ap.add_argument("--c3-steps", type=int, default=N_STEPS, help="N > 1: RK4 steps of configs[2]")
    ap.add_argument("--c4-steps", type=int, default=200, help="N > 1: fixed steps of configs[3] (rehearsals shorten it)")
    ap.add_argument("--c5-steps", type=int, default=1000, help="N > 1: Euler-Maruyama steps of configs[4]")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    # rehearsal on one GPU: take the multi-rank branch with a process group of ONE rank (RCCL refuses two ranks on one
    # device), so that every collective call of that branch has run under RCCL before a node runs it
    multi = world > 1 or args.group_of_one
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run", file=sys.stderr)
        sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py needs a GPU (there is no CPU path to time)", file=sys.stderr)
        sys.exit(2)
    dev_index = 0 if args.single_device else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    dist = None
    if multi:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")      # (torch.distributed.run sets both; the one-rank rehearsal may not)
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        # communicator set-up (RCCL builds its rings on the first collective) happens here, whatever --warmup says
        dist.all_reduce(torch.zeros(1, device=device if args.backend == "nccl" else "cpu"))

    # host threads: the ranks of a node share its cores (table building, packing, the gloo rehearsal's copies)
    host_threads = max(1, usable_cores() // world)
    torch.set_num_threads(host_threads)
    sm = build_model(device)
    eps = float(sm.sde.epsilon)
    opts = {"step_size": (1.0 - eps) / N_STEPS}
    B = args.batch
    # base samples from the library's counter-based stream keyed by the GLOBAL row (rank r holds rows [r B, (r + 1) B)):
    # a row's sample does not depend on how many ranks there are
    from flowfusion_amd import _native
    z = _native.normal_fill(B, DIM, 1234, rank * B, device)
    gather_dev = device if args.backend == "nccl" else torch.device("cpu")
    gathered = torch.empty(world * B, DIM, device=gather_dev) if multi else None

    def step(events=None):
        # HIP events on the stream the kernel is launched on (torch's current stream) bracket the fused launch
        # inside the timed region; the collective of N > 1 stays outside them
        if events is not None:
            events[0].record()
        x, _ = sm.sample_ode_from_base(z, method="rk4", options=opts)
        if events is not None:
            events[1].record()
        if multi:      # the single collective of the path: all shards meet on every rank
            dist.all_gather_into_tensor(gathered, x if args.backend == "nccl" else x.cpu())
            if events is not None:
                events[2].record()
        return x

    def barrier():
        if multi:
            dist.barrier()
        torch.cuda.synchronize(device)

    for _ in range(args.warmup):
        x = step()
    barrier()
    ev = [tuple(torch.cuda.Event(enable_timing=True) for _ in range(3)) for _ in range(args.steps)]
    net = sm._net()
    kernel_name = _native.lib().ff_kernel_name(net.plan(0).kernel_id).decode()
    table = sm._ode_table(torch.tensor([1.0, eps]), "rk4", opts, 0).to(device)
    n_evals = table.shape[0]
    t0 = time.perf_counter()
    for i in range(args.steps):
        x = step(ev[i])
    barrier()
    elapsed = time.perf_counter() - t0
    if multi:
        tmax = torch.tensor([elapsed], device=gather_dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # per-launch kernel time of the timed steps themselves
    kernel_ms = sorted(s.elapsed_time(e) for s, e, _ in ev)
    kernel_ms_avg = sum(kernel_ms) / len(kernel_ms)
    per_rank = None
    if multi:
        # what each rank spent in its kernel and in the collective (gloo rehearsal: the collective runs on the host
        # after a device-to-host copy, the events then only bracket that copy's enqueue)
        gather_ms_avg = sum(e.elapsed_time(g) for _, e, g in ev) / len(ev) if args.backend == "nccl" else float("nan")
        mine = torch.tensor([kernel_ms_avg, gather_ms_avg], device=gather_dev, dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = {"kernel_ms_avg": [float(t[0]) for t in allr], "allgather_ms_avg": [float(t[1]) for t in allr]}
    # rank invariance of the headline: this rank's first 256 rows as a launch of their own, and as they arrived in the
    # gathered tensor, equal the rows of the timed solve bit for bit
    head, _ = sm.sample_ode_from_base(z[:256].contiguous(), method="rk4", options=opts)
    rank_invariant = bool(torch.equal(head, x[:256]))
    sharded = None
    hung = False
    if multi:
        rank_invariant &= bool(torch.equal(gathered[rank * B: rank * B + 256].to(device), x[:256]))
        flag = torch.tensor([1.0 if rank_invariant else 0.0], device=gather_dev, dtype=torch.float64)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)                   # the headline's invariance, over all ranks
        rank_invariant = bool(flag.item() == 1.0)
        # Everything below is extra and runs behind a watchdog thread: the multi-rank collectives of these sections have
        # never run under RCCL before a node runs this line, and one that never completes must not take the measured
        # headline with it.  After a hang this rank issues no further collective, prints (rank 0) and leaves.
        if args.extras:
            res, why = run_guarded(lambda: sharded_extras(device, world, rank, dist, args.backend, args.c4_rows, args.c5_rows,
                                                          args.c4_steps, args.c5_steps, args.c3_rows, args.c3_steps),
                                   args.watchdog_seconds[0], device)
            if res is not None:
                sharded, ok = res
                rank_invariant &= ok                                   # (already MIN-reduced over the ranks inside)
            else:
                hung = why.startswith("no answer")
                sharded = [{"workload": "sharded BASELINE configs[2] / [3] / [4] (extras of the N > 1 run)", "error": why,
                            "where": {"section": "sharded_extras: barrier / all_gather_into_tensor / all_gather / all_reduce(MIN) "
                                                 "around configs[2], [3], [4]", "rank": rank, "watchdog_fired": hung}}]
                rank_invariant = False
        # last of all: the adaptive path's exchange step
        if args.extras and args.adaptive_rows > 0 and not hung:
            if rank == args.test_desert_rank:
                time.sleep(args.watchdog_seconds[1] + 6.0)            # alive but absent: the others' collective cannot finish
                res, why = None, "deserted (test)"
            else:
                res, why = run_guarded(lambda: adaptive_exchange_extra(device, world, rank, dist, args.backend,
                                                                       args.adaptive_rows), args.watchdog_seconds[1], device)
            hung = res is None and why is not None and why.startswith("no answer")
            entry = res if res is not None else {
                "workload": "default-argument log_prob sharded with whole-batch step control", "error": why,
                "where": {"section": "adaptive_exchange_extra: all_reduce of the 8 error-norm doubles (exchange hook) / barrier / "
                                     "all_gather of the per-rank report", "rank": rank, "watchdog_fired": hung}}
            sharded = (sharded or []) + [entry]

    if rank == 0:
        print(f"[bench] timed region {elapsed:.3f} s, kernel avg {kernel_ms_avg:.1f} ms", file=sys.stderr, flush=True)
        total = world * B * args.steps
        flop_per_launch = 2.0 * mac_per_eval(DIM, UNITS) * n_evals * B
        achieved = flop_per_launch / (kernel_ms_avg * 1e-3) / 1e12
        alg_bytes = 2 * DIM * 4 * B
        out = {
            "metric": "samples/sec (whole node), 16-dim VP-SDE 100-step RK4",
            "value": total / elapsed,
            "unit": "samples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: 16-dim VP-SDE score model, MLP 4x256, 100-step RK4 "
                                   "probability-flow ODE sampler (torchdiffeq rk4 = 3/8 rule, %d evals)" % n_evals,
                       "batch_per_gpu": B, "global_batch": world * B, "state_dim": DIM, "hidden": UNITS,
                       "sharding": f"batch x{world}, one RCCL all-gather per step" if multi else "single GPU"},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_FP32_MFMA_TFLOPS, "traffic": None,
                         "kernel": kernel_name, "kernel_ms_avg": kernel_ms_avg,
                         "flop_per_launch": flop_per_launch,
                         "algorithmic_hbm_bytes_per_launch": alg_bytes,
                         "algorithmic_hbm_GBps": alg_bytes / (kernel_ms_avg * 1e-3) / 1e9,
                         "hbm_frac_of_8TBps": alg_bytes / (kernel_ms_avg * 1e-3) / 1e9 / PEAK_HBM_GBS},
        }
        out["rank_invariant"] = rank_invariant
        out["host_threads_per_rank"] = host_threads
        if per_rank is not None:
            out["per_rank"] = per_rank
        if sharded is not None:
            out["extra_configs"] = sharded
        # `traffic` cannot be measured by this process (PMC counters need a rocprofv3 pass of their own): it is the
        # committed result of tools/profile_round.sh for this kernel at this batch, and says so
        traffic_file = ROOT / "profiles" / "hbm_traffic.json"
        if traffic_file.exists():
            tf = json.loads(traffic_file.read_text())
            if tf.get("batch") == B and kernel_name.split("_m")[-1].split("_")[0] in tf.get("kernel", ""):
                out["roofline"]["traffic"] = tf.get("bytes_per_launch")
                out["roofline"]["traffic_source"] = {
                    "file": "profiles/hbm_traffic.json", "from": tf.get("source"), "commit": tf.get("commit"),
                    "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command (tools/profile_round.sh), "
                            "not measured by this run"}
        if args.cpu_batch > 0 and not multi:      # CPU baseline and oracle parity: rank 0 of the 1-GPU run only
            ref, zc, cb = cpu_baseline(sm, args.cpu_batch, opts)
            out["cpu_baseline"] = cb
            # parity of the timed configuration: GPU vs oracle on the same base samples
            xg, _ = sm.sample_ode_from_base(zc.to(device), method="rk4", options=opts)
            out["parity_vs_cpu_oracle"] = {
                "max_abs_err_over_max_abs": float((xg.cpu() - ref).abs().max() / ref.abs().max()),
                "n": int(zc.shape[0]), "kernel": _native.kernel_name(sm._net().plan(0))}
            # log_prob relative error (second half of BASELINE's metric), Hutchinson, 100-step RK4.  Every accuracy figure
            # of this line names the kernel that produced it and says how many of its output elements differ from the f32
            # kernel's: identical maxima of different arithmetics are then visibly not the same numbers served three times.
            from oracle import flowfusion_oracle as O
            sd = {k: v.detach().cpu() for k, v in sm.state_dict().items()}
            so = O.ScoreOracle(O.mlp_params_from_state_dict(sd, "model."), O.VP(), no_sigma=True)
            sm.hutch = True
            torch.manual_seed(99)
            xq = torch.randn(128, DIM) * 0.9
            lp = sm.log_prob(xq.to(device), method="rk4", options=opts).cpu()
            lp_ref = so.log_prob(xq, None, "rk4", opts, "hutch", sm.e.cpu())
            rel = lambda got: ((got - lp_ref).abs() / lp_ref.abs().clamp_min(1.0))
            out["log_prob_rel_err"] = float(rel(lp).max())
            out["log_prob_evidence"] = {"kernel": _native.kernel_name(sm._net().plan(1)), "n": int(lp.numel()),
                                        "argmax": int(rel(lp).argmax())}
            split_lp_err = {}
            if args.extras:     # the same check on the split-precision kernels (same points, same probe)
                for prec in ("bf16x2",):      # (bf16x3: state-only kernels since round 3)
                    sm.precision = prec
                    torch.manual_seed(99)
                    assert torch.equal(torch.randn(128, DIM) * 0.9, xq)
                    lps = sm.log_prob(xq.to(device), method="rk4", options=opts).cpu()
                    split_lp_err[prec] = {"log_prob_rel_err": float(rel(lps).max()),
                                          "log_prob_evidence": {
                                              "kernel": _native.kernel_name(sm._net().plan(1)), "n": int(lps.numel()),
                                              "argmax": int(rel(lps).argmax()),
                                              "n_elements_differing_from_f32_result": int((lps != lp).sum()),
                                              "max_abs_diff_vs_f32_result": float((lps - lp).abs().max())}}
                sm.precision = "f32"
            sm.hutch = False
            _, _, cb1 = cpu_baseline(sm, 2048, opts, budget_s=10.0, threads=1)
            out["cpu_baseline_1thread"] = cb1
            torch.set_num_threads(host_threads)
        if args.extras and not multi:
            out["split_precision_record"] = split_precision_record(device, z, opts, args.steps, 1, x, "bf16x3")
            out["split_precision_record_bf16x2"] = split_precision_record(device, z, opts, args.steps, 1, x, "bf16x2")
            if args.cpu_batch > 0:
                out["split_precision_record_bf16x2"].update(split_lp_err["bf16x2"])
                # state error of each arithmetic against the float64 oracle on the CPU sample's base points
                from oracle import flowfusion_oracle as O64
                so64 = O64.ScoreOracle(O64.mlp_params_from_state_dict(sd, "model."), O64.VP(dtype=torch.float64), no_sigma=True,
                                       dtype=torch.float64)
                zs = zc[:256]
                r64 = so64.sample_ode_from_base(zs.double(), None, "rk4", opts)
                x_f32 = None
                for prec, key in (("f32", None), ("bf16x3", "split_precision_record"), ("bf16x2", "split_precision_record_bf16x2")):
                    sm.precision = prec
                    xg, _ = sm.sample_ode_from_base(zs.to(device), method="rk4", options=opts)
                    xg = xg.cpu()
                    err = (xg.double() - r64).abs()
                    e64 = float(err.max() / r64.abs().max())
                    ev = {"kernel": _native.kernel_name(sm._net().plan(0)), "n": int(xg.numel()), "argmax": int(err.argmax())}
                    if key is None:
                        x_f32 = xg
                        out["parity_vs_cpu_oracle"]["max_abs_err_over_max_abs_vs_float64_oracle"] = e64
                        out["parity_vs_cpu_oracle"]["float64_oracle_evidence"] = ev
                    else:
                        ev["n_elements_differing_from_f32_result"] = int((xg != x_f32).sum())
                        ev["max_abs_diff_vs_f32_result"] = float((xg - x_f32).abs().max())
                        out[key]["max_abs_err_over_max_abs_vs_float64_oracle"] = e64
                        out[key]["float64_oracle_evidence"] = ev
                sm.precision = "f32"
            out["extra_configs"] = extra_configs(device)
            out["streaming_helpers"] = streaming_helpers(device)
        print(json.dumps(out), flush=True)
    if multi:
        if hung or any("error" in e for e in (sharded or [])):
            # a rank stuck in a collective (or whose peers may be) cannot tear the group down: the line is out (its `error` /
            # `where` entries say which section, on which rank), leave -- with a NON-ZERO code, so that the launcher's return
            # code tells a run whose collective never completed from a clean one
            sys.stdout.flush()
            sys.stderr.flush()
            os._exit(EXIT_EXTRAS_FAILED)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
