// ff_registry.h -- table of compiled kernel instantiations (filled by generated code).
#pragma once
#include <hip/hip_runtime_api.h>
#include "ff_kernel_args.h"

namespace ff {

constexpr int kMaxDevices = 64;   // launchers keep one "attribute set" flag per device of the process

typedef int (*LaunchFn)(const KernelArgs* args, unsigned grid, unsigned lds_bytes, hipStream_t stream);

struct KernelEntry {
    int tile;       // MFMA columns per wavefront: 32 (32x32x2) or 16 (16x16x4)
    int H;          // hidden width on chip
    int dregs;      // state registers  (covers dim <= dregs * 64/tile)
    int cregs;      // conditional registers
    int tangents;   // 1: divergence-capable instantiation
    int act;        // FF_ACT_* code of the hidden activation this instantiation has compiled in (0 = SiLU; 9 = any
                    // non-SiLU activation, chosen at run time)
    LaunchFn launch;
    const char* name;
    LaunchFn launch_coop;   // cooperative twin for small batches (one tile per workgroup, ff_mlp_ode.hpp COOP) or NULL
    int wps;        // wavefronts per SIMD the one-wavefront kernel is built for (its launch bound): 1024 * wps tiles run at once
};

// defined in the generated ff_table.cpp
extern const KernelEntry g_kernels[];
extern const int g_n_kernels;

// split-precision family (FF_PREC_BF16X3 / FF_PREC_BF16X2, ff_mlp_ode_split.hpp): width <= 256, dim <= 16, cond <= 16
struct SplitKernelEntry {
    int n_hidden;   // hidden layers (compile-time in this family)
    int tangents;   // 0: state only; 1: Hutchinson (value / tangent column pairs); 2: exact trace (a value column + unit tangents)
    int parts;      // bf16 parts per fp32 operand: 3 (truncation, six products) or 2 (round to nearest, three products)
    int dt;         // 16-dimension tiles of the state: 1 (dim <= 16, 7 stage slots) or 2 (dim <= 32, 4 stage slots)
    int width;      // on-chip layer width: 256 or 128
    LaunchFn launch;
    const char* name;
    LaunchFn launch4;   // four-slot twin (two workgroups per CU; ff_split_layout.h has_four_slot_twin) or NULL
};
extern const SplitKernelEntry g_split_kernels[];
extern const int g_n_split_kernels;

} // namespace ff
