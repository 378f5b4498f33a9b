"""Build libflowfusion_amd.so (gfx950) in-tree with hipcc.

``python -m flowfusion_amd.build`` generates one translation unit per kernel instantiation
(so they compile in parallel), compiles them with ``hipcc --offload-arch=gfx950`` and links
``flowfusion_amd/lib/libflowfusion_amd.so``.  hipcc cross-compiles, so this works on a
machine without a GPU.  Objects are cached by content hash under ``flowfusion_amd/_build``.
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG = Path(__file__).resolve().parent
ROOT = PKG.parent
CSRC = PKG / "csrc"
# FF_BUILD_FULL=1 (round 2's instance set, for A/B measurements) builds beside the product library, not over it
FULL = os.environ.get("FF_BUILD_FULL", "") not in ("", "0")
GEN = PKG / "_build" / ("gen_full" if FULL else "gen")
OBJ = PKG / "_build" / ("obj_full" if FULL else "obj")
LIBDIR = PKG / "lib"
LIB = LIBDIR / ("libflowfusion_amd_full.so" if FULL else "libflowfusion_amd.so")

ARCH = "gfx950"

# (TILE, H, DREGS, CREGS, TANGENTS, WPS, RING, ACT):
#   TILE     MFMA columns per wavefront: 32 = v_mfma_f32_32x32x2, 16 = v_mfma_f32_16x16x4
#   H        hidden width on chip;  DREGS / CREGS state / conditional registers per lane
#            (TILE 32: dim <= 2*DREGS; TILE 16: dim <= 4*DREGS)
#   TANGENTS divergence-capable;  WPS waves per SIMD (launch bound);  RING weight chunks in flight
#   ACT      FF_ACT_* code of the hidden activation compiled in: 0 = SiLU (the reference default and the only activation
#            its code, docs and notebooks ever use), 1..8 the others, 9 = any non-SiLU activation, chosen at run time
#
# Round 3: the clean build was cut from 265 translation units / 68 CPU-minutes to what a BASELINE configuration, a
# reference demo shape or a documented option of the reference's constructors reaches.
# FF_BUILD_FULL=1 restores the round-2 set (compiled-in activations, 5-6 hidden layers on the split family, ...) for
# A/B measurements.
INSTANCES = [
    # narrow networks: 32 samples per wavefront, one wavefront per SIMD (width 64; width 128 beyond 16 dimensions)
    (32, h, d, c, t, 1, 8, 0)
    for (h, ds) in ((64, (4, 8, 16)), (128, (16,)))
    for d in ds
    for c in (0, 8)
    for t in (0, 1)
] + [
    # 128-wide networks on the 16-column tile (round 3: the reference's own notebook networks are 3x128; the 32-column
    # kernels ran them at ~65 % of the MFMA rate with one wavefront per SIMD -- measured scratch/tile16_h128.py: +8..12 %
    # at 2^20 samples with two wavefronts per SIMD, notebook log_prob 19.1 -> 16.7 ms).  THREE wavefronts per SIMD with four
    # chunks in flight (149 registers, no spill; eight chunks spill at the 168-register cap): another +1..5 %, 16.1 ms --
    # at this width the activations' VALU time is half of the MFMA time, and a third wavefront hides more of it
    (16, 128, 4, c, t, 3, 4, 0) for c in (0, 4) for t in (0, 1)
] + [
    # Two wavefronts per SIMD: with 16 samples per wavefront a 256-wide network needs only 64 + 64
    # activation/accumulator registers, so two wavefronts share a SIMD and one's VALU work (SiLU,
    # stage bookkeeping) overlaps the other's MFMAs -- a single wavefront cannot overlap its own
    # VALU with f32 MFMAs (measured: scratch/bench_proto3.hip).
    (16, 256, d, c, t, 2, 8, 0) for (d, c) in ((4, 0), (4, 4), (8, 4)) for t in (0, 1)
] + [
    # up to 32 conditional inputs (and 32 dimensions) at width 256 -- in round 2 a 512-wide instance caught these at 4x
    # the matrix work and two minutes of compile time
    (16, 256, 8, 8, t, 2, 8, 0) for t in (0, 1)
] + [
    # up to 64 dimensions on a 256-wide network (one wavefront per SIMD: the stage slots of 64 dimensions take
    # 114 KB of LDS per workgroup) -- otherwise such a model would run on the 512-wide kernels at 4x the FLOPs
    (16, 256, 16, 4, t, 1, 8, 0) for t in (0, 1)
] + [
    # wide / high-dimensional networks (BASELINE config 4: 64-dim flow, 5x512): 512 features x 16
    # samples fill the register file of one wavefront per SIMD
    (16, 512, 16, 4, t, 1, 4, 0) for t in (0, 1)
] + [
    # Non-default activations (`activation=` of the reference constructors; its code, docs and notebooks only ever use
    # SiLU).  ACT = 9 chooses the function at run time (the chosen kind's stages run back to back behind a wave-uniform
    # switch): measured against the compiled-in variants (scratch/act_bench.py, profiles/r03/act_bench.txt) it costs
    # 0-3 % for state-only solves at width 256, 9-14 % at width 128 and is what the 512-wide kernels (over a minute of
    # compile time each) use; with tangent columns at width 256 it costs 16 % (the switch at every activation site spills
    # 187 registers at two wavefronts per SIMD, so it would run one): THAT shape keeps a compiled-in instantiation per
    # activation (ACT = 1..8), without cooperative twins.  The 128-wide ones were compiled in too until the clean build
    # sat at 3 m 30-50 s of the 4 minutes allowed: 16 translation units for 10 % on networks nobody has shown to exist.
    (32, 128, 16, 8, 0, 1, 8, 9), (32, 128, 16, 8, 1, 1, 8, 9),
    (16, 256, 8, 4, 0, 2, 8, 9), (16, 512, 16, 4, 0, 1, 4, 9), (16, 512, 16, 4, 1, 1, 4, 9),
] + [
    (16, 256, 8, 4, 1, 2, 8, act) for act in range(1, 9)
] + ([
    # round-2 extras (FF_BUILD_FULL): width 256 on the 32-wide tile (A/B runs, FF_TILE=32); up to 32 conditional inputs
    # at width 512 (the wide catch-alls serve them otherwise); the narrow 32-column shapes the 16-column tile replaced;
    # every non-SiLU activation compiled in per width <= 256 and mode, and run-time choice everywhere
    (32, 256, 8, 0, 0, 1, 8, 0), (32, 256, 8, 0, 1, 1, 8, 0),
    (16, 512, 16, 8, 0, 1, 4, 0), (16, 512, 16, 8, 1, 1, 4, 0),
    (16, 256, 8, 4, 1, 1, 8, 9),
] + [
    (32, 128, d, c, t, 1, 8, 0) for d in (4, 8) for c in (0, 8) for t in (0, 1)
] + [
    (tile, h, d, c, t, wps, 8, act)
    for act in range(1, 9)
    for (tile, h, d, c, t, wps) in ((32, 64, 16, 8, 0, 1), (32, 64, 16, 8, 1, 1), (32, 128, 16, 8, 0, 1), (32, 128, 16, 8, 1, 1),
                                    (16, 256, 8, 4, 0, 2))
] if FULL else [])

# Wide catch-alls (kernel template WIDE: cooperative at every batch size, hidden operands read from LDS): networks up
# to 1024 wide, states up to 128 dimensions, up to 64 conditional inputs.  (TILE, H, DREGS, CREGS, TANGENTS)
WIDE_INSTANCES = [(16, 1024, d, c, t) for (d, c) in (((16, 8), (32, 16)) if FULL else ((32, 16),)) for t in (0, 1)]


def _wide_name(tile, h, d, c, t) -> str:
    return f"mlp_ode_m{tile}_h{h}_d{d}_c{c}_t{t}_wide"


# split-precision family (opt-in `precision=`): (hidden layers, TANGENTS, bf16 parts per operand: 3 = FF_PREC_BF16X3,
# 2 = FF_PREC_BF16X2, 16-dimension tiles of the state: 1 = dim <= 16; 2 = dim <= 32 (two-part kernels only), on-chip
# width: 256, or 128 for networks up to 128 wide).  TANGENTS: 0 state only, 1 Hutchinson column pairs, 2 exact trace.
# Frozen in round 3 at what the BASELINE configurations and the reference's demo networks reach: 1-4 hidden layers
# (configs 2, 3 and 5 are 4x256, the notebooks 3x128); two parts: every mode for states of up to 16 dimensions, state-only
# (Euler-Maruyama: config 5) for 17-32 dimensions, 128-wide instances; three parts: state-only, 256 wide.
if FULL:
    SPLIT_INSTANCES = [(nh, t, parts, 1, 256) for parts in (3, 2) for nh in (1, 2, 3, 4, 5, 6) for t in (0, 1, 2)] + \
                      [(nh, t, 2, 2, 256) for nh in (1, 2, 3, 4, 5, 6) for t in (0, 1, 2)] + \
                      [(nh, t, parts, dt, 128) for (parts, dt) in ((3, 1), (2, 1), (2, 2)) for nh in (1, 2, 3, 4) for t in (0, 1, 2)]
else:
    # three parts (fp32-class): the state-only kernels -- the second record of the headline workload is what round 1 asked
    # of this option; log-densities under precision= take the two-part kernels or f32
    SPLIT_INSTANCES = [(nh, 0, 3, 1, 256) for nh in (1, 2, 3, 4)] + \
                      [(nh, t, 2, 1, 256) for nh in (1, 2, 3, 4) for t in (0, 1, 2)] + \
                      [(nh, 0, 2, 2, 256) for nh in (1, 2, 3, 4)] + \
                      [(nh, t, 2, 1, 128) for nh in (1, 2, 3, 4) for t in (0, 1, 2)]


# ---- test-only libraries (tests/test_gpu_skew.py; csrc/ff_skew.h) -------------------------------------------------------
# A handful of instantiations compiled with -DFF_DEBUG_SKEW (one wavefront of every workgroup held back where a missing
# barrier would show) beside the product library, never in it:
#   skew        HEAD's kernels: the 256-wide cooperative twin (state-only and with tangent columns) and its one-wavefront
#               kernel, the wide catch-all's code path instantiated at width 256 (a tenth of the 1024-wide unit's compile
#               time; the product serves that shape with a one-wavefront kernel, which gives the test its reference), one
#               split-precision kernel
#   skew_unfix  the same twin with round 3's two synchronisation fixes removed again (-DFF_DEBUG_UNFIX): what the test must
#               see FAIL, or it guards nothing
_SKEW_TWIN = [(16, 256, 4, 0, t, 2, 8, 0) for t in (0, 1)]
VARIANTS = {
    "skew": dict(defines=["-DFF_DEBUG_SKEW=0"], instances=_SKEW_TWIN, wide=[(16, 256, 8, 4, 0)], split=[(3, 0, 2, 1, 128)]),
    "skew_unfix": dict(defines=["-DFF_DEBUG_SKEW=0", "-DFF_DEBUG_UNFIX=1"], instances=_SKEW_TWIN, wide=[], split=[]),
}


if os.environ.get("FF_BUILD_NOSLP", "") not in ("", "0"):
    # one-off A/B (round 4; scratch/slp_ab.py): the headline, config-3 and notebook kernels with hipcc's SLP
    # vectoriser off -- it packs the SiLU tail's scalar products into v_pk_mul_f32 / v_pk_fma_f32 between the MFMAs
    VARIANTS["noslp"] = dict(defines=["-fno-slp-vectorize"],
                             instances=[(16, 256, 4, 0, 0, 2, 8, 0), (16, 256, 4, 0, 1, 2, 8, 0), (16, 128, 4, 0, 0, 3, 4, 0),
                                        (16, 128, 4, 0, 1, 3, 4, 0)], wide=[], split=[])


if os.environ.get("FF_BUILD_EXP_DPP", "") not in ("", "0"):
    VARIANTS["dpp"] = dict(defines=["-DFF_EXP_DPP=1"], instances=[(16, 256, 4, 0, 1, 2, 8, 0), (16, 128, 4, 0, 1, 3, 4, 0)], wide=[], split=[])


def variant_lib(name: str) -> Path:
    return LIBDIR / f"libflowfusion_amd_{name}.so"


def _has_four_slot_twin(inst) -> bool:
    """128-wide kernels for states of up to 16 dimensions: a twin with four stage slots (two workgroups per CU, +26 % on the
    opt-in arithmetic at notebook widths).  Built with FF_BUILD_FULL only since round 3 (12 translation units of a frozen,
    opt-in family); the launcher's `launch4` pointer is then null and every launch takes the seven-slot kernel."""
    return FULL and inst[2] in (2, 3) and inst[3] == 1 and inst[4] == 128


def _split_name(nh, t, parts=3, dt=1, width=256) -> str:
    return f"mlp_ode_split{'' if parts == 3 else parts}_h{width}{'' if dt == 1 else '_d' + str(dt)}_n{nh}_t{t}"


def _has_coop(h, act=0) -> bool:
    """Instances that get a cooperative twin: the blocks of a layer (H / 32) must split four ways.  (Of the
    non-SiLU instantiations only the run-time-choice one at width 256 gets one, and every one of round 2's set: build time.)"""
    return (h // 32) % 4 == 0 and (act == 0 or (h <= 256 and (act == 9 or FULL)))


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm under /opt/rocm)")


def _inst_name(tile, h, d, c, t, wps, ring, act) -> str:
    # (the name carries _w<n> only where it always has -- the two-waves-per-SIMD SiLU kernels)
    return f"mlp_ode_m{tile}_h{h}_d{d}_c{c}_t{t}" + (f"_w{wps}" if (wps != 1 and act == 0) else "") + (f"_a{act}" if act else "")


def _launcher_unit(name: str, header: str, kernel_expr: str) -> str:
    """One translation unit: the instantiation `kernel_expr` behind `int launch_<name>(args, grid, lds, stream)`."""
    return f"""// generated by flowfusion_amd/build.py -- do not edit
#include <atomic>
#include "ff_registry.h"
#include "{header}"
namespace ff {{
int launch_{name}(const KernelArgs* a, unsigned grid, unsigned lds, hipStream_t s)
{{
    auto kern = {kernel_expr};
    // the dynamic-LDS limit is a per-device attribute of the function: set it once per device of this process
    static std::atomic<unsigned char> ready[kMaxDevices];
    int dev = 0;
    hipError_t err = hipGetDevice(&dev);
    if (err != hipSuccess) return (int)err;
    if (dev < 0 || dev >= kMaxDevices || !ready[dev].load(std::memory_order_acquire)) {{
        err = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (err != hipSuccess) return (int)err;
        if (dev >= 0 && dev < kMaxDevices) ready[dev].store(1, std::memory_order_release);
    }}
    void* params[] = {{(void*)a}};
    return (int)hipLaunchKernel((const void*)kern, dim3(grid), dim3(256), params, lds, s);
}}
}}
"""


def _write(path: Path, text: str) -> Path:
    if not path.exists() or path.read_text() != text:
        path.write_text(text)
    return path


def _gen_sources(gen: Path | None = None, instances=None, wide=None, split=None) -> list[Path]:
    """The translation units of a library: the product's (default arguments) or a test variant's instance lists."""
    GEN = gen if gen is not None else globals()["GEN"]
    INSTANCES = instances if instances is not None else globals()["INSTANCES"]
    WIDE_INSTANCES = wide if wide is not None else globals()["WIDE_INSTANCES"]
    SPLIT_INSTANCES = split if split is not None else globals()["SPLIT_INSTANCES"]
    GEN.mkdir(parents=True, exist_ok=True)
    files = []
    tf = lambda t: "true" if t else "false"
    for tile, h, d, c, t, wps, ring, act in INSTANCES:
        name = _inst_name(tile, h, d, c, t, wps, ring, act)
        files.append(_write(GEN / f"{name}.hip", _launcher_unit(
            name, "ff_mlp_ode.hpp", f"mlp_ode_kernel<{tile}, {h}, {d}, {c}, {tf(t)}, {wps}, {ring}, {act}>")))
        if _has_coop(h, act):
            # cooperative twin (small batches): same layout, one tile per workgroup; a translation unit of its own
            cwps = 2 if h <= 256 else 1
            files.append(_write(GEN / f"{name}_coop.hip", _launcher_unit(
                f"{name}_coop", "ff_mlp_ode.hpp", f"mlp_ode_kernel<{tile}, {h}, {d}, {c}, {tf(t)}, {cwps}, 4, {act}, true>")))
    for tile, h, d, c, t in WIDE_INSTANCES:
        name = _wide_name(tile, h, d, c, t)
        files.append(_write(GEN / f"{name}.hip", _launcher_unit(
            name, "ff_mlp_ode.hpp", f"mlp_ode_kernel<{tile}, {h}, {d}, {c}, {tf(t)}, 1, 4, 0, true, true>")))
    split_units = [(i, False) for i in SPLIT_INSTANCES] + [(i, True) for i in SPLIT_INSTANCES if _has_four_slot_twin(i)]
    for (nh, t, parts, dt, width), four in split_units:
        name = _split_name(nh, t, parts, dt, width) + ("_s4" if four else "")
        files.append(_write(GEN / f"{name}.hip", _launcher_unit(
            name, "ff_mlp_ode_split.hpp", f"split::mlp_ode_split_kernel<{nh}, {t}, {parts}, {dt}, {width}{', 4' if four else ''}>")))
    decls = "\n".join(
        [f"int launch_{_inst_name(*i)}(const KernelArgs*, unsigned, unsigned, hipStream_t);" for i in INSTANCES] +
        [f"int launch_{_inst_name(*i)}_coop(const KernelArgs*, unsigned, unsigned, hipStream_t);" for i in INSTANCES
         if _has_coop(i[1], i[7])] +
        [f"int launch_{_wide_name(*i)}(const KernelArgs*, unsigned, unsigned, hipStream_t);" for i in WIDE_INSTANCES] +
        [f"int launch_{_split_name(*i)}(const KernelArgs*, unsigned, unsigned, hipStream_t);" for i in SPLIT_INSTANCES] +
        [f"int launch_{_split_name(*i)}_s4(const KernelArgs*, unsigned, unsigned, hipStream_t);" for i in SPLIT_INSTANCES
         if _has_four_slot_twin(i)]
    )
    split_rows = ",\n".join(
        f'    {{{i[0]}, {i[1]}, {i[2]}, {i[3]}, {i[4]}, launch_{_split_name(*i)}, "{_split_name(*i)}", '
        + (f"launch_{_split_name(*i)}_s4" if _has_four_slot_twin(i) else "nullptr") + "}" for i in SPLIT_INSTANCES
    )
    rows = ",\n".join(
        f'    {{{i[0]}, {i[1]}, {i[2]}, {i[3]}, {i[4]}, {i[7]}, launch_{_inst_name(*i)}, "{_inst_name(*i)}", '
        + (f"launch_{_inst_name(*i)}_coop" if _has_coop(i[1], i[7]) else "nullptr") + f", {i[5]}}}"
        for i in INSTANCES
    )
    # wide catch-alls: no one-wavefront kernel (launch = nullptr), the cooperative launcher serves every batch size
    if WIDE_INSTANCES:
        rows += ",\n" + ",\n".join(
            f'    {{{i[0]}, {i[1]}, {i[2]}, {i[3]}, {i[4]}, 0, nullptr, "{_wide_name(*i)}", launch_{_wide_name(*i)}, 0}}'
            for i in WIDE_INSTANCES
        )
    if not SPLIT_INSTANCES:          # (a test variant without the family: one inert row, count 0 -- no zero-length array)
        split_rows = '    {0, 0, 0, 0, 0, nullptr, "", nullptr}'
    table = f"""// generated by flowfusion_amd/build.py -- do not edit
#include "ff_registry.h"
namespace ff {{
{decls}
const KernelEntry g_kernels[] = {{
{rows}
}};
const int g_n_kernels = {len(INSTANCES) + len(WIDE_INSTANCES)};
const SplitKernelEntry g_split_kernels[] = {{
{split_rows}
}};
const int g_n_split_kernels = {len(SPLIT_INSTANCES)};
}}
"""
    files.append(_write(GEN / "ff_table.cpp", table))
    files.append(CSRC / "ff_api.cpp")
    files.append(CSRC / "ff_aux.hip")
    files.append(CSRC / "ff_adaptive.hip")
    files.append(CSRC / "ff_trace.hip")
    return files


def _cost(src: Path) -> float:
    """Rough compile seconds of a translation unit (measured in round 3, /tmp timing of a clean build): the pool starts
    the longest first so that none of them trails the others."""
    import re
    n = src.name
    m = re.match(r"mlp_ode_split(\d?)_h(\d+)(_d2)?_n(\d)_t(\d)", n)
    if m:
        parts, h, d2, nh, t = m.group(1) or "3", int(m.group(2)), bool(m.group(3)), int(m.group(4)), int(m.group(5))
        base = (4.5 if parts == "3" else 3.2) * (2.2 if t else 1.0) * (1.4 if d2 else 1.0) * (1.0 if h == 256 else 0.3)
        return base * (1 + nh)
    if "_h1024_" in n:
        return 95 if "_t1" in n else 55
    if "_h512_" in n:
        return 12 if "_coop" in n else (90 if "_t1" in n else 58)
    if "_h256_" in n:
        return 5 if "_coop" in n else 16
    return 6


_INCLUDE_RE = None


def _local_includes(path: Path, seen: dict) -> None:
    """Transitive closure of the `#include "..."` headers of `path` that live under csrc/ or include/."""
    global _INCLUDE_RE
    import re
    if _INCLUDE_RE is None:
        _INCLUDE_RE = re.compile(r'^\s*#\s*include\s*"([^"]+)"', re.M)
    for name in _INCLUDE_RE.findall(path.read_text()):
        for base in (CSRC, ROOT / "include"):
            h = base / name
            if h.exists() and h not in seen:
                seen[h] = True
                _local_includes(h, seen)


def _deps_hash(src: Path) -> str:
    """Hash of the project headers a translation unit actually includes (transitively): a change to the split-precision
    kernel does not recompile the f32 kernels, a change to the public header only the two files that implement it."""
    seen: dict = {}
    _local_includes(src, seen)
    h = hashlib.sha256()
    for p in sorted(seen):
        h.update(p.name.encode())
        h.update(p.read_bytes())
    return h.hexdigest()


def _compile(src: Path, dep_hash: str, verbose: bool, obj_dir: Path | None = None, defines=()) -> Path:
    OBJ = obj_dir if obj_dir is not None else globals()["OBJ"]
    key = hashlib.sha256(dep_hash.encode() + src.read_bytes() + src.name.encode() + " ".join(defines).encode()).hexdigest()[:16]
    obj = OBJ / f"{src.stem}.{key}.o"
    if obj.exists():
        return obj
    for old in OBJ.glob(f"{src.stem}.*.o"):
        old.unlink()
    if src.suffix == ".hip":   # device code: one kernel instantiation
        lang = ["-x", "hip", f"--offload-arch={ARCH}"]
    else:                       # host-only C++ (C ABI, packer, kernel table)
        lang = ["-x", "c++", "-D__HIP_PLATFORM_AMD__=1", f"-I{Path(_hipcc()).resolve().parents[1] / 'include'}"]
    cmd = [_hipcc(), "-O3", "-std=c++17", "-fPIC", "-Wno-inline-asm", *lang, *defines,
           f"-I{CSRC}", f"-I{ROOT / 'include'}", "-c", str(src), "-o", str(obj)]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src.name}:\n{r.stderr}")
    return obj


def _link(objs, lib: Path, stamp_file: Path, verbose: bool) -> Path:
    stamp = hashlib.sha256("".join(sorted(o.name for o in objs)).encode()).hexdigest()
    if lib.exists() and stamp_file.exists() and stamp_file.read_text() == stamp:
        return lib
    cmd = [_hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", str(lib)] + [str(o) for o in objs]
    if verbose:
        print(" ".join(cmd[:6]), "...", flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stderr}")
    stamp_file.write_text(stamp)
    return lib


def build(verbose: bool = False, jobs: int | None = None, variants: bool = True) -> Path:
    """Compile every kernel instantiation for gfx950 and link the shared library -- and, beside it, the small test-only
    libraries of VARIANTS (``variants=False`` skips them).  All translation units share one pool, longest first."""
    OBJ.mkdir(parents=True, exist_ok=True)
    LIBDIR.mkdir(parents=True, exist_ok=True)
    srcs = _gen_sources()
    units = [(s, OBJ, ()) for s in srcs]
    extra = {}
    if variants and not FULL:
        for name, v in VARIANTS.items():
            gen, obj = PKG / "_build" / f"gen_{name}", PKG / "_build" / f"obj_{name}"
            obj.mkdir(parents=True, exist_ok=True)
            vs = _gen_sources(gen, v["instances"], v["wide"], v["split"])
            extra[name] = (vs, obj, tuple(v["defines"]))
            units += [(s, obj, tuple(v["defines"])) for s in vs]
    jobs = jobs or min(8, os.cpu_count() or 1)
    # longest first (estimated), so that no slow translation unit trails the pool
    order = sorted(units, key=lambda u: (-_cost(u[0]), u[0].name, str(u[1])))
    with ThreadPoolExecutor(max_workers=jobs) as ex:
        done = dict(zip([(u[0], u[1]) for u in order],
                        ex.map(lambda u: _compile(u[0], _deps_hash(u[0]), verbose, u[1], u[2]), order)))
    _link([done[(s, OBJ)] for s in srcs], LIB, PKG / "_build" / ("link_full.stamp" if FULL else "link.stamp"), verbose)
    for name, (vs, obj, _) in extra.items():
        _link([done[(s, obj)] for s in vs], variant_lib(name), PKG / "_build" / f"link_{name}.stamp", verbose)
    return LIB


if __name__ == "__main__":
    lib = build(verbose="-v" in sys.argv)
    print(lib)
