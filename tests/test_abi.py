"""The C ABI pinned independently of flowfusion_amd/_native.py.

CPU tier: include/flowfusion_amd.h compiles as C11 (-Wall -Werror), a C program links against
libflowfusion_amd.so, calls the plan / packing entry points without a GPU and prints sizeof / offsetof of every
struct field; those must equal the ctypes mirrors in _native.py AND the ones in INTEGRATION.md's stub.
GPU tier: INTEGRATION.md's stub B -- the binding a reference maintainer would paste into flowfusion/diffusion.py
-- is executed verbatim (its own ctypes structs, no flowfusion_amd._native) against the golden hybrid fixtures.
"""
import ctypes
import os
import re
import subprocess
import sys
from pathlib import Path

import pytest
import torch

from tests._util import golden_names, load_golden, max_rel, score_model

ROOT = Path(__file__).resolve().parents[1]


def _stub_source():
    text = (ROOT / "INTEGRATION.md").read_text()
    sec = text[text.index("## B."):]
    m = re.search(r"```python\n(.*?)```", sec, re.S)
    assert m, "INTEGRATION.md section B holds no python block"
    return m.group(1)


def _stub_namespace(built_library):
    from flowfusion_amd import _native
    os.environ["FLOWFUSION_AMD_LIB"] = str(_native.library_path())
    src = _stub_source()
    assert "flowfusion_amd._native" not in src and "import _native" not in src, "the stub must bind the library on its own"
    ns = {}
    exec(compile(src, "INTEGRATION.md#B", "exec"), ns)
    return ns


def _probe_layout(tmp_path, built_library):
    from flowfusion_amd import _native
    lib = _native.library_path()
    exe = tmp_path / "abi_probe"
    cmd = ["gcc", "-x", "c", "-std=c11", "-Wall", "-Wextra", "-Werror", "-pedantic", f"-I{ROOT / 'include'}",
           str(ROOT / "tests" / "abi" / "abi_probe.c"), "-x", "none", str(lib), f"-Wl,-rpath,{lib.parent}",
           "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    env = dict(os.environ, LD_LIBRARY_PATH=f"{lib.parent}:/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run([str(exe)], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 0, r.stderr
    sizes, offsets, other = {}, {}, {}
    for line in r.stdout.splitlines():
        w = line.split()
        if w[0] == "sizeof":
            sizes[w[1]] = int(w[2])
        elif w[0] == "offsetof":
            offsets[(w[1], w[2])] = int(w[3])
        else:
            other[w[0]] = line
    return sizes, offsets, other


def test_header_is_c11_and_layouts_match_both_ctypes_mirrors(tmp_path, built_library):
    from flowfusion_amd import _native
    sizes, offsets, other = _probe_layout(tmp_path, built_library)
    ns = _stub_namespace(built_library)
    mirrors = {
        "ff_mlp_plan_t": (_native.PlanStruct, ns["_Plan"]),
        "ff_ode_args": (_native.OdeArgs, ns["_Args"]),
        "ff_combine_args": (_native.CombineArgs,),
        "ff_norm_term": (_native.NormTerm,),
        "ff_adapt_state": (_native.AdaptState,),
        "ff_adapt_config": (_native.AdaptConfig,),
        "ff_adapt_buffers": (_native.AdaptBuffers,),
        "ff_trace_args": (_native.TraceArgs,),
    }
    for cname, structs in mirrors.items():
        fields_c = {f for (s, f) in offsets if s == cname}
        for st in structs:
            assert ctypes.sizeof(st) == sizes[cname], (cname, st)
            names = [n for n, *_ in st._fields_]
            assert set(names) == fields_c, (cname, st, set(names) ^ fields_c)      # no field missing on either side
            for n in names:
                assert getattr(st, n).offset == offsets[(cname, n)], (cname, st, n)
    # the C program planned BASELINE config 2's network and asked for its packed size, without a GPU
    assert "rc 0 tile 16 width 256 dregs 4 cregs 0" in other["plan"]
    p = _native.make_plan(16, 0, [256] * 4, 0)
    assert int(other["wpack_floats"].split()[1]) == built_library.ff_mlp_wpack_floats(ctypes.byref(p))
    assert other["spw"].split()[1] == "64"
    assert other["badmode"].split()[2] == "-1" and other["nullargs"].split()[2] == "-1"
    assert other["nulladapt"].split()[2] == "-1"
    assert other["trace"].split()[2] == "0" and abs(float(other["trace"].split()[4]) - 5.0) < 1e-5      # tr [[1, 2], [3, 4]]
    assert int(other["tracews"].split()[1]) <= 16 and other["badtrace"].split()[2] == "-1"
    assert "gfx950" in other["version"]


def test_kernel_args_header_matches_public_header():
    """The public constants the kernels are compiled against (csrc/ff_kernel_args.h, ff_layout.h) are static_assert-ed
    against include/flowfusion_amd.h in ff_api.cpp; check the assertion lines are still there."""
    src = (ROOT / "flowfusion_amd" / "csrc" / "ff_api.cpp").read_text()
    for needle in ("FF_MAX_SLOTS == ff::kSlots", "FF_MAX_AUX == ff::kAux", "FF_ROW_HDR * 4 == sizeof(ff::RowHdr)",
                   "FF_STATUS_NAN == ff::kStatusNaN", "sizeof(ff_adapt_state) == 128"):
        assert needle in src


@pytest.mark.gpu
@pytest.mark.parametrize("name", golden_names("hybrid_score_"))
def test_integration_stub_runs_verbatim_against_golden(name, built_library):
    """Stub B of INTEGRATION.md, executed as written, reproduces the golden hybrids (reference RHS x restated
    stepper) -- the ABI as a third party would bind it."""
    assert torch.cuda.is_available()
    ns = _stub_namespace(built_library)
    meta, a = load_golden(name)
    sm = score_model(meta, a, "cuda")
    cond = a.get("cond")
    cond_d = None if cond is None else cond.to("cuda").contiguous()
    ran = 0
    for run in meta["runs"]:
        m, opts = run["method"], {"step_size": run["step_size"]}
        z = a["base"].to("cuda")
        if hasattr(sm.sde, "sigma_max"):
            z = z * sm.sde.sigma_max                                   # diffusion.py:605-606, done by the caller of odeint
        out = ns["_native_sample"](sm, z.contiguous(), cond_d, m, opts)
        assert out is not None
        exp = a[f"sample_{m}"]
        assert max_rel(out.cpu(), exp, floor=exp.abs().max().item()) < 2e-5, (name, m)
        ran += 1
    assert ran > 0
