"""ctypes binding of libflowfusion_amd.so and the PyTorch custom op that fronts it.

The shared library (C ABI: include/flowfusion_amd.h) is the product; this module only moves
pointers: tensors are torch-allocated device memory, the launch goes onto torch's current HIP
stream.  There is deliberately no fallback: a missing library or a tensor that is not on the
GPU is an error.
"""
from __future__ import annotations

import ctypes
import os
from pathlib import Path
from typing import List, Optional, Tuple

import torch

FF_OK = 0
FF_ERR_BADARG = -1
FF_ERR_UNSUPPORTED = -2
FF_ERR_HIP = -3
FF_ERR_EXCHANGE = -4

MODE_STATE = 0
MODE_HUTCH = 1
MODE_EXACT = 2

# FF_PREC_* of include/flowfusion_amd.h
PREC_F32, PREC_BF16X3, PREC_BF16X2 = 0, 1, 2
PRECISIONS = {"f32": PREC_F32, "bf16x3": PREC_BF16X3, "bf16x2": PREC_BF16X2}

# FF_ACT_* of include/flowfusion_amd.h
ACT_SILU, ACT_TANH, ACT_SIGMOID, ACT_RELU, ACT_LEAKY_RELU, ACT_ELU, ACT_SOFTPLUS, ACT_GELU, ACT_GELU_TANH = range(9)

_LIB_PATH = Path(__file__).resolve().parent / "lib" / "libflowfusion_amd.so"


class PlanStruct(ctypes.Structure):
    _fields_ = [
        ("dim", ctypes.c_int32),
        ("cond_dim", ctypes.c_int32),
        ("n_hidden", ctypes.c_int32),
        ("width", ctypes.c_int32),
        ("dregs", ctypes.c_int32),
        ("cregs", ctypes.c_int32),
        ("kernel_id", ctypes.c_int32),
        ("tile", ctypes.c_int32),
        ("activation", ctypes.c_int32),
        ("act_param", ctypes.c_float * 2),
        ("precision", ctypes.c_int32),
    ]


class OdeArgs(ctypes.Structure):
    _fields_ = [
        ("x_in", ctypes.c_void_p),
        ("x_out", ctypes.c_void_p),
        ("cond", ctypes.c_void_p),
        ("probe", ctypes.c_void_p),
        ("dlogp_out", ctypes.c_void_p),
        ("noise", ctypes.c_void_p),
        ("wpack", ctypes.c_void_p),
        ("etab", ctypes.c_void_p),
        ("in_shift", ctypes.c_void_p),
        ("in_scale", ctypes.c_void_p),
        ("out_scale", ctypes.c_void_p),
        ("out_shift", ctypes.c_void_p),
        ("status", ctypes.c_void_p),
        ("batch", ctypes.c_int64),
        ("noise_stride", ctypes.c_int64),
        ("n_evals", ctypes.c_int32),
        ("mode", ctypes.c_int32),
        ("tangent_first", ctypes.c_int32),
        ("tangent_count", ctypes.c_int32),
        ("k1_in", ctypes.c_void_p),
        ("kl1_in", ctypes.c_void_p),
        ("dlogp_in", ctypes.c_void_p),
        ("aux_out", ctypes.c_void_p * 4),
        ("aux_lp_out", ctypes.c_void_p * 4),
        ("n_aux", ctypes.c_int32),
        ("rng_noise_base", ctypes.c_int32),
        ("rng_seed", ctypes.c_uint64),
        ("rng_sample_offset", ctypes.c_int64),
        ("jac_out", ctypes.c_void_p),
        ("jac_all", ctypes.c_int32),
        ("stage_slots", ctypes.c_int32),
        ("gate", ctypes.c_void_p),
    ]


STATUS_NAN, STATUS_BAD_SLOT = 1, 2          # FF_STATUS_*
SCHED_FLOW, SCHED_VE, SCHED_VP, SCHED_SUBVP = 0, 1, 2, 3     # FF_SCHED_*
ADAPT_START, ADAPT_FINISH = 1, 2            # FF_ADAPT_START / FF_ADAPT_FINISH
ADAPT_ERR_UNDERFLOW, ADAPT_ERR_NONFINITE, ADAPT_ERR_MAXSTEPS = 1, 2, 3
ADAPT_MAX_PASSES = 8
MAX_SLOTS, MAX_AUX = 7, 4


class AdaptState(ctypes.Structure):       # ff_adapt_state (128 bytes of device memory)
    _fields_ = [
        ("t", ctypes.c_double), ("dt", ctypes.c_double), ("t_prev", ctypes.c_double), ("dt_prev", ctypes.c_double),
        ("t_end", ctypes.c_double), ("h0", ctypes.c_double), ("d0", ctypes.c_double), ("d1", ctypes.c_double),
        ("active", ctypes.c_int32), ("commit", ctypes.c_int32), ("done", ctypes.c_int32), ("error", ctypes.c_int32),
        ("n_attempts", ctypes.c_int32), ("n_accepted", ctypes.c_int32), ("n_steps", ctypes.c_int32),
        ("reserved0", ctypes.c_int32), ("last_ratio", ctypes.c_float), ("reserved1", ctypes.c_float * 7),
    ]


class AdaptConfig(ctypes.Structure):      # ff_adapt_config
    _fields_ = [
        ("n_stages", ctypes.c_int32), ("order", ctypes.c_int32),
        ("alpha", ctypes.c_float * (MAX_SLOTS - 1)), ("beta", (ctypes.c_float * 8) * (MAX_SLOTS - 1)),
        ("c_sol", ctypes.c_float * 8), ("c_mid", ctypes.c_float * 8), ("c_err", ctypes.c_float * 8),
        ("rtol", ctypes.c_float), ("atol", ctypes.c_float),
        ("min_step", ctypes.c_double), ("max_step", ctypes.c_double), ("first_step", ctypes.c_double),
        ("max_num_steps", ctypes.c_int32), ("sched", ctypes.c_int32), ("no_sigma", ctypes.c_int32), ("sign", ctypes.c_float),
        ("p", ctypes.c_double * 4),
        ("emb_w", ctypes.c_void_p), ("n_emb", ctypes.c_int32), ("pi", ctypes.c_float),
        ("w0t", ctypes.c_void_p), ("b0", ctypes.c_void_p), ("h_real", ctypes.c_int32), ("n_tcols", ctypes.c_int32),
    ]


class AdaptBuffers(ctypes.Structure):     # ff_adapt_buffers
    _fields_ = [
        ("y", ctypes.c_void_p), ("f0", ctypes.c_void_p), ("lp", ctypes.c_void_p), ("fl0", ctypes.c_void_p),
        ("aux", ctypes.c_void_p * MAX_AUX), ("aux_lp", ctypes.c_void_p * MAX_AUX), ("aux_lp_pass", ctypes.c_void_p),
        ("scratch_x", ctypes.c_void_p), ("scratch_lp", ctypes.c_void_p), ("etab", ctypes.c_void_p),
        ("out_y", ctypes.c_void_p), ("out_lp", ctypes.c_void_p), ("state", ctypes.c_void_p),
        ("norm_workspace", ctypes.c_void_p), ("norm_only", ctypes.c_void_p * 2), ("norm_only_n", ctypes.c_int64 * 2),
        ("n_passes", ctypes.c_int32), ("pass_first", ctypes.c_int32 * ADAPT_MAX_PASSES),
        ("pass_count", ctypes.c_int32 * ADAPT_MAX_PASSES),
        ("exchange_sums", ctypes.c_void_p), ("exchange", ctypes.c_void_p), ("exchange_user", ctypes.c_void_p),
        ("est_kind", ctypes.c_int32), ("est_r", ctypes.c_int32), ("est_m", ctypes.c_int32), ("est_reserved", ctypes.c_int32),
        ("est_probes0", ctypes.c_void_p), ("est_probes1", ctypes.c_void_p), ("est_jac", ctypes.c_void_p),
        ("est_div", ctypes.c_void_p), ("est_workspace", ctypes.c_void_p),
    ]


TRACE_HUTCHPP, TRACE_XTRACE = 1, 2        # FF_TRACE_*


class TraceArgs(ctypes.Structure):        # ff_trace_args
    _fields_ = [
        ("kind", ctypes.c_int32), ("dim", ctypes.c_int32), ("n_rows", ctypes.c_int32), ("r", ctypes.c_int32),
        ("m", ctypes.c_int32), ("reserved", ctypes.c_int32), ("batch", ctypes.c_int64),
        ("jac", ctypes.c_void_p), ("probes0", ctypes.c_void_p), ("probes1", ctypes.c_void_p), ("out", ctypes.c_void_p),
        ("workspace", ctypes.c_void_p), ("gate", ctypes.c_void_p),
    ]


EXCHANGE_DOUBLES = 8                      # FF_EXCHANGE_DOUBLES
EXCHANGE_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p)      # int (*exchange)(void* user, void* stream)


class CombineArgs(ctypes.Structure):      # ff_combine_args
    _fields_ = [
        ("x", ctypes.c_void_p),
        ("k", ctypes.c_void_p * 7),
        ("coef", ctypes.c_float * 7),
        ("x_coef", ctypes.c_float),
        ("out", ctypes.c_void_p),
        ("n", ctypes.c_int64),
    ]


class NormTerm(ctypes.Structure):         # ff_norm_term
    _fields_ = [
        ("num", ctypes.c_void_p),
        ("sub", ctypes.c_void_p),
        ("scale0", ctypes.c_void_p),
        ("scale1", ctypes.c_void_p),
        ("n", ctypes.c_int64),
    ]


NORM_TERMS = 3                      # FF_NORM_TERMS
PRIOR_NOISE_INDEX = 0xFFFFFFFF     # FF_PRIOR_NOISE_INDEX
PROBE_NOISE_INDEX = 0xFFFFFFFE     # FF_PROBE_NOISE_INDEX
TRACE_PROBE_NOISE_BASE = 0xFFFE0000  # FF_TRACE_PROBE_NOISE_BASE (second probe set: + 0x8000)

_lib = None


def library_path() -> Path:
    return Path(os.environ.get("FLOWFUSION_AMD_LIB", _LIB_PATH))


def lib() -> ctypes.CDLL:
    """Load libflowfusion_amd.so (once).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not path.exists():
        raise RuntimeError(
            f"{path} not found: the HIP library is not built. Run `python -m flowfusion_amd.build` "
            "(needs hipcc; cross-compiles gfx950 without a GPU). There is no CPU fallback.")
    _lib = _bind(ctypes.CDLL(str(path)))
    return _lib


def load_library(path) -> ctypes.CDLL:
    """A library with this ABI other than the product -- the test-only builds of flowfusion_amd/build.py VARIANTS
    (tests/test_gpu_skew.py) -- bound like the product library, NOT installed as the one the package launches through."""
    path = Path(path)
    if not path.exists():
        raise RuntimeError(f"{path} not found: run `python -m flowfusion_amd.build`")
    return _bind(ctypes.CDLL(str(path)))


def _bind(L: ctypes.CDLL) -> ctypes.CDLL:
    L.ff_version.restype = ctypes.c_char_p
    L.ff_kernel_count.restype = ctypes.c_int
    L.ff_kernel_name.restype = ctypes.c_char_p
    L.ff_kernel_name.argtypes = [ctypes.c_int]
    L.ff_mlp_plan.restype = ctypes.c_int
    L.ff_mlp_plan.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int),
                              ctypes.c_int, ctypes.POINTER(PlanStruct)]
    L.ff_mlp_plan_act.restype = ctypes.c_int
    L.ff_mlp_plan_act.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int),
                                  ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_float),
                                  ctypes.POINTER(PlanStruct)]
    L.ff_mlp_plan_prec.restype = ctypes.c_int
    L.ff_mlp_plan_prec.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int),
                                   ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_float), ctypes.c_int,
                                   ctypes.POINTER(PlanStruct)]
    L.ff_plan_kernel_name.restype = ctypes.c_char_p
    L.ff_plan_kernel_name.argtypes = [ctypes.POINTER(PlanStruct)]
    L.ff_mlp_wpack_floats.restype = ctypes.c_size_t
    L.ff_mlp_wpack_floats.argtypes = [ctypes.POINTER(PlanStruct)]
    L.ff_mlp_wpack.restype = ctypes.c_int
    L.ff_mlp_wpack.argtypes = [ctypes.POINTER(PlanStruct), ctypes.POINTER(ctypes.c_void_p),
                               ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int),
                               ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    L.ff_mlp_ode_launch.restype = ctypes.c_int
    L.ff_mlp_ode_launch.argtypes = [ctypes.POINTER(PlanStruct), ctypes.POINTER(OdeArgs), ctypes.c_void_p]
    L.ff_mlp_samples_per_workgroup.restype = ctypes.c_int
    L.ff_mlp_samples_per_workgroup.argtypes = [ctypes.POINTER(PlanStruct), ctypes.c_int]
    L.ff_last_hip_error.restype = ctypes.c_int
    L.ff_mlp_launch_kind.restype = ctypes.c_int
    L.ff_mlp_launch_kind.argtypes = [ctypes.POINTER(PlanStruct), ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32]
    L.ff_normal_fill.restype = ctypes.c_int
    L.ff_normal_fill.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_uint64, ctypes.c_int64,
                                 ctypes.c_uint32, ctypes.c_float, ctypes.c_void_p]
    L.ff_scaled_rms_workspace_bytes.restype = ctypes.c_size_t
    L.ff_scaled_rms.restype = ctypes.c_int
    L.ff_scaled_rms.argtypes = [ctypes.POINTER(NormTerm), ctypes.c_int32, ctypes.c_float, ctypes.c_float, ctypes.c_void_p,
                                ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    L.ff_stage_combine.restype = ctypes.c_int
    L.ff_stage_combine.argtypes = [ctypes.POINTER(CombineArgs), ctypes.c_void_p]
    L.ff_mlp_ode_adaptive.restype = ctypes.c_int
    L.ff_mlp_ode_adaptive.argtypes = [ctypes.POINTER(PlanStruct), ctypes.POINTER(OdeArgs), ctypes.POINTER(AdaptConfig),
                                      ctypes.POINTER(AdaptBuffers), ctypes.c_double, ctypes.c_double, ctypes.c_int32,
                                      ctypes.c_int32, ctypes.c_void_p]
    L.ff_trace_workspace_floats.restype = ctypes.c_size_t
    L.ff_trace_workspace_floats.argtypes = [ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int64]
    L.ff_trace_estimate.restype = ctypes.c_int
    L.ff_trace_estimate.argtypes = [ctypes.POINTER(TraceArgs), ctypes.c_void_p]
    L.ff_trace_estimate_host.restype = ctypes.c_int
    L.ff_trace_estimate_host.argtypes = [ctypes.POINTER(TraceArgs)]
    L.ff_adapt_host_row.restype = ctypes.c_int
    L.ff_adapt_host_row.argtypes = [ctypes.POINTER(AdaptConfig), ctypes.c_float, ctypes.POINTER(ctypes.c_float),
                                    ctypes.POINTER(ctypes.c_float), ctypes.c_void_p]
    L.ff_adapt_host_transition.restype = ctypes.c_int
    L.ff_adapt_host_transition.argtypes = [ctypes.POINTER(AdaptConfig), ctypes.POINTER(AdaptState), ctypes.c_int32,
                                           ctypes.POINTER(ctypes.c_float)]
    return L


def _err(rc: int, what: str) -> RuntimeError:
    names = {FF_ERR_BADARG: "FF_ERR_BADARG", FF_ERR_UNSUPPORTED: "FF_ERR_UNSUPPORTED", FF_ERR_HIP: "FF_ERR_HIP",
             FF_ERR_EXCHANGE: "FF_ERR_EXCHANGE"}
    extra = f" (hipError {lib().ff_last_hip_error()})" if rc == FF_ERR_HIP else ""
    return RuntimeError(f"{what} failed: {names.get(rc, rc)}{extra}")


def make_plan(dim: int, cond_dim: int, hidden: List[int], mode: int,
              act: Tuple[int, float, float] = (ACT_SILU, 0.0, 0.0), precision: int = PREC_F32) -> PlanStruct:
    """ff_mlp_plan_prec: pick the compiled kernel for this network shape (raises if none fits)."""
    p = PlanStruct()
    arr = (ctypes.c_int * len(hidden))(*hidden)
    prm = (ctypes.c_float * 2)(float(act[1]), float(act[2]))
    rc = lib().ff_mlp_plan_prec(dim, cond_dim, len(hidden), arr, mode, int(act[0]), prm, int(precision), ctypes.byref(p))
    if rc == FF_ERR_UNSUPPORTED and precision in (PREC_BF16X3, PREC_BF16X2):
        name = "bf16x3" if precision == PREC_BF16X3 else "bf16x2"
        raise NotImplementedError(
            f"precision='{name}' (one of the split-precision options 'bf16x3' / 'bf16x2') has no kernel for dim={dim}, cond_dim={cond_dim}, hidden={hidden}, mode={mode}, "
            f"activation={act[0]}: the split-precision family covers SiLU networks of 1-4 hidden layers up to 256 wide, "
            "cond_dim <= 16, dim <= 16: bf16x3 state-only solves (sampling, Euler-Maruyama; fixed grids and the adaptive "
            "methods); bf16x2 also Hutchinson and exact-trace log-densities and, state-only with at most 4 Runge-Kutta "
            "stages, dim <= 32; use precision='f32'")
    if rc == FF_ERR_UNSUPPORTED:
        raise NotImplementedError(
            f"no gfx950 kernel instantiation for dim={dim}, cond_dim={cond_dim}, hidden={hidden}, mode={mode}, "
            f"activation={act[0]}: "
            "compiled shapes cover dim<=128, cond_dim<=64, hidden width<=1024 (SiLU; other activations: dim<=64, "
            "cond_dim<=16, width<=512)")
    if rc != FF_OK:
        raise _err(rc, "ff_mlp_plan")
    return p


def pack_weights(plan: PlanStruct, weights: List[torch.Tensor], biases: List[Optional[torch.Tensor]],
                 hidden: List[int], x_col0: int, c_col0: int) -> torch.Tensor:
    """ff_mlp_wpack on host copies of the nn.Linear parameters; returns the packed CPU tensor."""
    L = lib()
    n = L.ff_mlp_wpack_floats(ctypes.byref(plan))
    out = torch.empty(n, dtype=torch.float32)
    ws = [w.detach().to("cpu", torch.float32).contiguous() for w in weights]
    bs = [None if b is None else b.detach().to("cpu", torch.float32).contiguous() for b in biases]
    wp = (ctypes.c_void_p * len(ws))(*[w.data_ptr() for w in ws])
    bp = (ctypes.c_void_p * len(bs))(*[0 if b is None else b.data_ptr() for b in bs])
    hw = (ctypes.c_int * len(hidden))(*hidden)
    rc = L.ff_mlp_wpack(ctypes.byref(plan), wp, bp, hw, int(ws[0].shape[1]), x_col0, c_col0, out.data_ptr())
    if rc != FF_OK:
        raise _err(rc, "ff_mlp_wpack")
    return out


def kernel_name(plan: PlanStruct) -> str:
    """ff_plan_kernel_name: the instantiation a plan selects."""
    n = lib().ff_plan_kernel_name(ctypes.byref(plan))
    return n.decode() if n else "?"


LAUNCH_ONE_WAVE, LAUNCH_TWIN, LAUNCH_ONE_WAVE_AND_TWIN = 0, 1, 2      # FF_LAUNCH_*


def launch_kind(plan: PlanStruct, batch: int, mode: int, tangent_count: int = 0, jac_out: bool = False) -> int:
    """ff_mlp_launch_kind: which kernel(s) a launch of ``batch`` samples takes (LAUNCH_*)."""
    rc = int(lib().ff_mlp_launch_kind(ctypes.byref(plan), batch, mode, tangent_count, int(jac_out)))
    if rc < 0:
        raise _err(rc, "ff_mlp_launch_kind")
    return rc


def samples_per_workgroup(plan: PlanStruct, mode: int) -> int:
    return int(lib().ff_mlp_samples_per_workgroup(ctypes.byref(plan), mode))


def normal_fill(batch: int, dim: int, seed: int, sample_offset: int, device, noise_index: int = PRIOR_NOISE_INDEX,
                scale: float = 1.0) -> torch.Tensor:
    """ff_normal_fill: [batch, dim] normals of the counter-based stream for global rows
    sample_offset .. sample_offset + batch - 1 (the prior draw of the sharded Euler-Maruyama sampler)."""
    device = torch.device(device)
    if device.type != "cuda":
        raise RuntimeError("flowfusion_amd: ff_normal_fill fills device memory (there is no CPU path)")
    out = torch.empty(batch, dim, dtype=torch.float32, device=device)
    if batch == 0:
        return out
    with torch.cuda.device(device):
        stream = torch.cuda.current_stream(device).cuda_stream
        rc = lib().ff_normal_fill(out.data_ptr(), batch, dim, seed & 0xFFFFFFFFFFFFFFFF, sample_offset,
                                  noise_index & 0xFFFFFFFF, float(scale), ctypes.c_void_p(stream))
    if rc != FF_OK:
        raise _err(rc, "ff_normal_fill")
    return out


def stage_combine(out: torch.Tensor, x: Optional[torch.Tensor], ks, coefs, x_coef: float = 1.0) -> torch.Tensor:
    """ff_stage_combine: out = x_coef * x + sum_s coefs[s] * ks[s] in one pass (flat fp32 device tensors of
    equal numel; `out` may alias an input).  Terms with a zero coefficient are not read."""
    if not out.is_cuda:
        raise RuntimeError("flowfusion_amd: ff_stage_combine works on device memory (there is no CPU path)")
    dev = out.device
    a = CombineArgs()
    n = out.numel()
    a.out = _chk(out, "out", dev)
    a.x = _chk(x, "x", dev)
    if x is not None and x.numel() != n:
        raise RuntimeError("stage_combine: x and out differ in size")
    if len(ks) > 7 or len(ks) != len(coefs):
        raise RuntimeError("stage_combine: at most FF_MAX_SLOTS = 7 terms, one coefficient each")
    for s, (k, c) in enumerate(zip(ks, coefs)):
        c = float(c)
        if k is None or c == 0.0:
            continue
        if k.numel() != n:
            raise RuntimeError("stage_combine: term and out differ in size")
        a.k[s] = _chk(k, f"k[{s}]", dev)
        a.coef[s] = c
    a.x_coef = float(x_coef)
    a.n = n
    if n == 0:
        return out
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream(dev).cuda_stream
        rc = lib().ff_stage_combine(ctypes.byref(a), ctypes.c_void_p(stream))
    if rc != FF_OK:
        raise _err(rc, "ff_stage_combine")
    return out


def trace_kind_and_probes(kind: str, probes):
    """(FF_TRACE_*, probes0, probes1, r, m) from ``kind`` ("hutchpp": probes = (S, G); "xtrace": probes = (O,)), each
    probe tensor [n, B, D] as the reference stores them (diffusion.py:708-719)."""
    if kind == "hutchpp":
        S, G = probes
        return TRACE_HUTCHPP, S, G, int(S.shape[0]), int(G.shape[0])
    if kind == "xtrace":
        (O,) = probes
        return TRACE_XTRACE, O, None, int(O.shape[0]), 0
    raise ValueError(f"trace estimator {kind!r}: expected 'hutchpp' or 'xtrace'")


def trace_estimate(jac: torch.Tensor, kind: str, probes, host: bool = False) -> torch.Tensor:
    """ff_trace_estimate: Hutch++ / XTrace divergence estimates [n_rows, B] from recorded Jacobians ``jac``
    [n_rows, B, D, D] (A = J^T per evaluation row and sample) in ONE launch; ``host=True`` runs the same arithmetic on CPU
    tensors through ff_trace_estimate_host (tests without a GPU)."""
    n_rows, B, D, D2 = jac.shape
    code, p0, p1, r, m = trace_kind_and_probes(kind, probes)
    if D != D2 or tuple(p0.shape) != (r, B, D) or (p1 is not None and tuple(p1.shape) != (m, B, D)):
        raise RuntimeError(f"trace_estimate: jac {tuple(jac.shape)} and probes {tuple(p0.shape)} do not match")
    dev = jac.device
    if host != (dev.type == "cpu"):
        raise RuntimeError("flowfusion_amd: ff_trace_estimate works on device memory (host=True: CPU tensors, tests only)")
    f32 = lambda t: None if t is None else t.detach().to(dev, torch.float32).contiguous()
    jac, p0, p1 = f32(jac), f32(p0), f32(p1)
    out = torch.empty(n_rows, B, dtype=torch.float32, device=dev)
    items = 1 if host else n_rows * B
    ws = torch.empty(max(1, int(lib().ff_trace_workspace_floats(code, D, r, items))), dtype=torch.float32, device=dev)
    a = TraceArgs()
    a.kind, a.dim, a.n_rows, a.r, a.m, a.batch = code, D, n_rows, r, m, B
    a.jac, a.probes0, a.probes1 = jac.data_ptr(), p0.data_ptr(), (0 if p1 is None else p1.data_ptr())
    a.out, a.workspace = out.data_ptr(), ws.data_ptr()
    if host:
        rc = lib().ff_trace_estimate_host(ctypes.byref(a))
    else:
        with torch.cuda.device(dev):
            rc = lib().ff_trace_estimate(ctypes.byref(a), ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    if rc != FF_OK:
        raise _err(rc, "ff_trace_estimate")
    return out


_norm_ws = {}     # (device index, stream) -> (workspace, out) of ff_scaled_rms


def scaled_rms(terms, atol: float, rtol: float, check: Optional[torch.Tensor] = None, read: bool = True):
    """ff_scaled_rms: ``terms`` = up to 3 tuples (num, sub or None, scale0, scale1 or None) of equal-sized fp32 device
    tensors; returns [rms_0, .., rms_{n-1}, nonfinite(check)] as Python floats -- one launch, one read-back (the single
    host synchronisation of an attempted step under the host controller).  ``read=False`` returns the device tensor the
    kernel wrote instead (no synchronisation; valid until the next call on this stream)."""
    dev = terms[0][0].device
    if dev.type != "cuda":
        raise RuntimeError("flowfusion_amd: ff_scaled_rms works on device memory (there is no CPU path)")
    if not 1 <= len(terms) <= NORM_TERMS:
        raise RuntimeError(f"scaled_rms: 1..{NORM_TERMS} terms")
    arr = (NormTerm * NORM_TERMS)()
    keep = []
    for i, (num, sub, s0, s1) in enumerate(terms):
        for t in (num, sub, s0, s1):
            if t is not None and t.numel() != num.numel():
                raise RuntimeError("scaled_rms: the arrays of a term differ in size")
        arr[i].num, arr[i].sub = _chk(num, "num", dev), _chk(sub, "sub", dev)
        arr[i].scale0, arr[i].scale1 = _chk(s0, "scale0", dev), _chk(s1, "scale1", dev)
        arr[i].n = num.numel()
        keep.append((num, sub, s0, s1))
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream(dev).cuda_stream
        ws, out = norm_workspace(dev, stream)
        rc = lib().ff_scaled_rms(arr, len(terms), float(atol), float(rtol), _chk(check, "check", dev),
                                 0 if check is None else check.numel(), out.data_ptr(), ws.data_ptr(), ctypes.c_void_p(stream))
    if rc != FF_OK:
        ws[:4].zero_()      # whatever happened to the launch, the arrival counter starts the next one from zero
        raise _err(rc, "ff_scaled_rms")
    return out[: len(terms) + 1].tolist() if read else out[: len(terms) + 1]


def norm_workspace(dev, stream: int):
    """(workspace, out) of the norm reductions for this device and stream: the kernels leave the arrival counter at
    zero, so sequential launches on one stream share a workspace (ff_scaled_rms, the adaptive controller)."""
    index = dev.index if dev.index is not None else torch.cuda.current_device()
    key = (index, stream)
    if key not in _norm_ws:
        nbytes = int(lib().ff_scaled_rms_workspace_bytes())
        _norm_ws[key] = (torch.zeros((nbytes + 3) // 4, dtype=torch.int32, device=dev),
                         torch.empty(NORM_TERMS + 1, dtype=torch.float32, device=dev))
    return _norm_ws[key]


_PLAN_WORDS = ctypes.sizeof(PlanStruct) // 4


def _plan_from_words(words: List[int]) -> PlanStruct:
    return PlanStruct.from_buffer_copy((ctypes.c_int32 * _PLAN_WORDS)(*words[:_PLAN_WORDS]))


def _slots_hint(words: List[int]) -> int:
    """ff_ode_args.stage_slots travelling behind the plan's words (0 = unknown)."""
    return int(words[_PLAN_WORDS]) if len(words) > _PLAN_WORDS else 0


def plan_words(plan: PlanStruct, stage_slots: int = 0) -> List[int]:
    """The plan as 32-bit words (how it travels through the custom op's integer-list argument), followed by the number of
    stage slots the launch's table uses (ff_ode_args.stage_slots; 0 = unknown)."""
    return list((ctypes.c_int32 * _PLAN_WORDS).from_buffer_copy(plan)) + [int(stage_slots)]


def _chk(t: Optional[torch.Tensor], name: str, dev) -> int:
    if t is None:
        return 0
    if t.device != dev:
        raise RuntimeError(f"{name} is on {t.device}, expected {dev}")
    if t.dtype != torch.float32 or not t.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous float32")
    return t.data_ptr()


@torch.library.custom_op("flowfusion_amd::mlp_ode", mutates_args=())
def mlp_ode(x: torch.Tensor, cond: Optional[torch.Tensor], probe: Optional[torch.Tensor],
            noise: Optional[torch.Tensor], wpack: torch.Tensor, etab: torch.Tensor,
            in_shift: Optional[torch.Tensor], in_scale: Optional[torch.Tensor],
            out_scale: Optional[torch.Tensor], out_shift: Optional[torch.Tensor],
            plan: List[int], mode: int, tangent_first: int = 0, tangent_count: int = 0,
            rng_seed: int = 0, rng_sample_offset: int = 0, rng_noise_base: int = 0
            ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Fused integration on the GPU: returns (final state [B,D], integrated divergence [B], status[1]).
    With `noise=None`, rows flagged FF_ROW_NOISE draw their normals in the kernel (Philox4x32-10 keyed by
    `rng_seed` and the global sample index `rng_sample_offset + row`; see ff_ode_args)."""
    if not x.is_cuda:
        raise RuntimeError("flowfusion_amd::mlp_ode needs tensors on the GPU (there is no CPU path)")
    dev = x.device
    p = _plan_from_words(plan)
    B, D = x.shape
    if D != p.dim:
        raise RuntimeError(f"state has {D} columns, plan was made for {p.dim}")
    x_out = torch.empty_like(x)
    dlogp = torch.zeros(B if mode != MODE_STATE else 0, dtype=torch.float32, device=dev)
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    if B == 0:                      # nothing to integrate (zero-size tensors have no storage to point at)
        return x_out, dlogp, status
    a = OdeArgs()
    a.x_in = _chk(x, "x", dev)
    a.x_out = x_out.data_ptr()
    a.cond = _chk(cond, "cond", dev)
    a.probe = _chk(probe, "probe", dev)
    a.dlogp_out = dlogp.data_ptr() if mode != MODE_STATE else 0
    a.noise = _chk(noise, "noise", dev)
    a.wpack = _chk(wpack, "wpack", dev)
    a.etab = _chk(etab, "etab", dev)
    a.in_shift = _chk(in_shift, "in_shift", dev)
    a.in_scale = _chk(in_scale, "in_scale", dev)
    a.out_scale = _chk(out_scale, "out_scale", dev)
    a.out_shift = _chk(out_shift, "out_shift", dev)
    a.status = status.data_ptr()
    a.batch = B
    a.noise_stride = B * D
    a.n_evals = etab.shape[0]
    a.mode = mode
    a.tangent_first = tangent_first
    a.tangent_count = tangent_count
    a.rng_seed = rng_seed & 0xFFFFFFFFFFFFFFFF
    a.rng_sample_offset = rng_sample_offset
    a.rng_noise_base = rng_noise_base
    a.stage_slots = _slots_hint(plan)
    if cond is not None and tuple(cond.shape) != (B, p.cond_dim):
        raise RuntimeError(f"cond has shape {tuple(cond.shape)}, expected {(B, p.cond_dim)}")
    if probe is not None and tuple(probe.shape) != (B, D):
        raise RuntimeError(f"probe has shape {tuple(probe.shape)}, expected {(B, D)}")
    if etab.shape[1] != 32 + p.width:
        raise RuntimeError("evaluation table width does not match the plan")
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream(dev).cuda_stream
        rc = lib().ff_mlp_ode_launch(ctypes.byref(p), ctypes.byref(a), ctypes.c_void_p(stream))
    if rc != FF_OK:
        raise _err(rc, "ff_mlp_ode_launch")
    return x_out, dlogp, status


@mlp_ode.register_fake
def _(x, cond, probe, noise, wpack, etab, in_shift, in_scale, out_scale, out_shift, plan, mode,
      tangent_first=0, tangent_count=0, rng_seed=0, rng_sample_offset=0, rng_noise_base=0):
    B = x.shape[0]
    return (torch.empty_like(x), x.new_empty(B if mode != MODE_STATE else 0),
            torch.empty(1, dtype=torch.int32, device=x.device))


@torch.library.custom_op("flowfusion_amd::mlp_ode_step", mutates_args=())
def mlp_ode_step(x: torch.Tensor, cond: Optional[torch.Tensor], probe: Optional[torch.Tensor],
                 k1: Optional[torch.Tensor], kl1: Optional[torch.Tensor], dlogp0: Optional[torch.Tensor],
                 wpack: torch.Tensor, etab: torch.Tensor, plan: List[int], mode: int, n_aux: int,
                 tangent_first: int = 0, tangent_count: int = 0) -> Tuple[torch.Tensor, torch.Tensor]:
    """One attempt of an embedded Runge-Kutta step (adaptive solvers): the evaluation rows of `etab`
    fill the stage slots (slot 0 preloaded from `k1`/`kl1`), and `n_aux` linear combinations of the
    slots, described by the two trailing rows of `etab`, are returned: (aux [n_aux, B, D],
    aux_lp [n_aux, B])."""
    if not x.is_cuda:
        raise RuntimeError("flowfusion_amd::mlp_ode_step needs tensors on the GPU (there is no CPU path)")
    dev = x.device
    p = _plan_from_words(plan)
    B, D = x.shape
    aux = torch.empty(n_aux, B, D, dtype=torch.float32, device=dev)
    aux_lp = torch.zeros(n_aux, B if mode != MODE_STATE else 0, dtype=torch.float32, device=dev)
    if B == 0:
        return aux, aux_lp
    scratch = torch.empty_like(x)
    dl = torch.empty(B if mode != MODE_STATE else 0, dtype=torch.float32, device=dev)
    a = OdeArgs()
    a.x_in = _chk(x, "x", dev)
    a.x_out = scratch.data_ptr()
    a.cond = _chk(cond, "cond", dev)
    a.probe = _chk(probe, "probe", dev)
    a.dlogp_out = dl.data_ptr() if mode != MODE_STATE else 0
    a.wpack = _chk(wpack, "wpack", dev)
    a.etab = _chk(etab, "etab", dev)
    a.batch = B
    a.n_evals = etab.shape[0] - 2
    a.mode = mode
    a.tangent_first = tangent_first
    a.tangent_count = tangent_count
    a.k1_in = _chk(k1, "k1", dev)
    a.kl1_in = _chk(kl1, "kl1", dev)
    a.dlogp_in = _chk(dlogp0, "dlogp0", dev)
    for j in range(n_aux):
        a.aux_out[j] = aux[j].data_ptr()
        a.aux_lp_out[j] = aux_lp[j].data_ptr() if mode != MODE_STATE else 0
    a.n_aux = n_aux
    a.stage_slots = _slots_hint(plan)
    if etab.shape[1] != 32 + p.width or a.n_evals < 0:
        raise RuntimeError("evaluation table does not match the plan")
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream(dev).cuda_stream
        rc = lib().ff_mlp_ode_launch(ctypes.byref(p), ctypes.byref(a), ctypes.c_void_p(stream))
    if rc != FF_OK:
        raise _err(rc, "ff_mlp_ode_launch")
    return aux, aux_lp


@mlp_ode_step.register_fake
def _(x, cond, probe, k1, kl1, dlogp0, wpack, etab, plan, mode, n_aux, tangent_first=0, tangent_count=0):
    B = x.shape[0]
    return (x.new_empty(n_aux, B, x.shape[1]), x.new_empty(n_aux, B if mode != MODE_STATE else 0))


@torch.library.custom_op("flowfusion_amd::mlp_rhs_jac", mutates_args=("jac",))
def mlp_rhs_jac(x: torch.Tensor, cond: Optional[torch.Tensor], wpack: torch.Tensor, etab: torch.Tensor,
                plan: List[int], tangent_first: int, tangent_count: int, jac: torch.Tensor) -> torch.Tensor:
    """One right-hand-side evaluation with its Jacobian (ff_ode_args.jac_out): `etab` is one evaluation row
    followed by the two auxiliary rows; returns rhs [B, D] and fills rows [tangent_first, +count) of
    ``jac[b, j, i] = d rhs_i / d y_j``."""
    if not x.is_cuda:
        raise RuntimeError("flowfusion_amd::mlp_rhs_jac needs tensors on the GPU (there is no CPU path)")
    dev = x.device
    p = _plan_from_words(plan)
    B, D = x.shape
    rhs = torch.empty_like(x)
    if B == 0:
        return rhs
    if tuple(jac.shape) != (B, D, D) or etab.shape[0] != 3 or etab.shape[1] != 32 + p.width:
        raise RuntimeError("mlp_rhs_jac: jac must be [B, D, D] and etab one evaluation row + two auxiliary rows")
    scratch = torch.empty_like(x)
    dl = torch.empty(B, dtype=torch.float32, device=dev)
    a = OdeArgs()
    a.x_in = _chk(x, "x", dev)
    a.x_out = scratch.data_ptr()
    a.cond = _chk(cond, "cond", dev)
    a.dlogp_out = dl.data_ptr()
    a.wpack = _chk(wpack, "wpack", dev)
    a.etab = _chk(etab, "etab", dev)
    a.batch = B
    a.n_evals = 1
    a.mode = MODE_EXACT
    a.tangent_first = tangent_first
    a.tangent_count = tangent_count
    a.aux_out[0] = rhs.data_ptr()
    a.n_aux = 1
    a.jac_out = _chk(jac, "jac", dev)
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream(dev).cuda_stream
        rc = lib().ff_mlp_ode_launch(ctypes.byref(p), ctypes.byref(a), ctypes.c_void_p(stream))
    if rc != FF_OK:
        raise _err(rc, "ff_mlp_ode_launch")
    return rhs


@mlp_rhs_jac.register_fake
def _(x, cond, wpack, etab, plan, tangent_first, tangent_count, jac):
    return torch.empty_like(x)


@torch.library.custom_op("flowfusion_amd::mlp_ode_jacobians", mutates_args=("jac",))
def mlp_ode_jacobians(x: torch.Tensor, cond: Optional[torch.Tensor], wpack: torch.Tensor, etab: torch.Tensor,
                      plan: List[int], tangent_first: int, tangent_count: int, jac: torch.Tensor) -> torch.Tensor:
    """Integrate a whole fixed-grid table in exact mode and record the Jacobian of EVERY evaluation row
    (ff_ode_args.jac_all): fills rows [tangent_first, +count) of ``jac[e, b, j, i] = d rhs_i / d y_j`` and returns the
    final state [B, D].  The state does not depend on the divergence, so the estimators that need per-evaluation
    Jacobians (Hutch++, XTrace) can run after the fact, for all rows at once."""
    if not x.is_cuda:
        raise RuntimeError("flowfusion_amd::mlp_ode_jacobians needs tensors on the GPU (there is no CPU path)")
    dev = x.device
    p = _plan_from_words(plan)
    B, D = x.shape
    x_out = torch.empty_like(x)
    if B == 0:
        return x_out
    n_evals = etab.shape[0]
    if tuple(jac.shape) != (n_evals, B, D, D) or etab.shape[1] != 32 + p.width:
        raise RuntimeError("mlp_ode_jacobians: jac must be [n_evals, B, D, D] and etab must match the plan")
    dl = torch.empty(B, dtype=torch.float32, device=dev)
    a = OdeArgs()
    a.x_in = _chk(x, "x", dev)
    a.x_out = x_out.data_ptr()
    a.cond = _chk(cond, "cond", dev)
    a.dlogp_out = dl.data_ptr()
    a.wpack = _chk(wpack, "wpack", dev)
    a.etab = _chk(etab, "etab", dev)
    a.batch = B
    a.n_evals = n_evals
    a.mode = MODE_EXACT
    a.tangent_first = tangent_first
    a.tangent_count = tangent_count
    a.jac_out = _chk(jac, "jac", dev)
    a.jac_all = 1
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream(dev).cuda_stream
        rc = lib().ff_mlp_ode_launch(ctypes.byref(p), ctypes.byref(a), ctypes.c_void_p(stream))
    if rc != FF_OK:
        raise _err(rc, "ff_mlp_ode_launch")
    return x_out


@mlp_ode_jacobians.register_fake
def _(x, cond, wpack, etab, plan, tangent_first, tangent_count, jac):
    return torch.empty_like(x)
