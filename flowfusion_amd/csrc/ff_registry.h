// ff_registry.h -- table of compiled kernel instantiations (filled by generated code).
#pragma once
#include <hip/hip_runtime_api.h>
#include "ff_kernel_args.h"

namespace ff {

typedef int (*LaunchFn)(const KernelArgs* args, unsigned grid, unsigned lds_bytes, hipStream_t stream);

struct KernelEntry {
    int H;          // hidden width on chip
    int dregs;      // state registers  (covers dim <= 2*dregs)
    int cregs;      // conditional registers (covers cond_dim <= 2*cregs)
    int tangents;   // 1: divergence-capable instantiation
    LaunchFn launch;
    const char* name;
};

// defined in the generated ff_table.cpp
extern const KernelEntry g_kernels[];
extern const int g_n_kernels;

} // namespace ff
