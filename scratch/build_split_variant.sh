#!/bin/bash
# usage: build_split_variant.sh OUT.so "-DFLAG ..." [instance] [row]  -- ONE split-precision instance + the api in a private .so
#   (select it with FLOWFUSION_AMD_LIB=OUT.so); timing ablations: -DFF_SPLIT_NODMA -DFF_SPLIT_NOACT -DFF_SPLIT_NOBARRIER
set -e
OUT=$1; FLAGS=$2; NAME=${3:-mlp_ode_split_h256_n4_t0}; ROW=${4:-4,0,3,1,256}     # ROW = n_hidden, tangents, parts, state tiles, width
# FLAGS may carry -I<dir> to take an older ff_mlp_ode_split.hpp first (e.g. -Iscratch/split_v1: the kernel before the pinned gaps)
R=${GRAFT_REPO_ROOT:-/root/repo}
T=$(mktemp -d)
hipcc -O3 -std=c++17 -fPIC -Wno-inline-asm -x hip --offload-arch=gfx950 $FLAGS -I$R/flowfusion_amd/csrc -I$R/include -c $R/flowfusion_amd/_build/gen/$NAME.hip -o $T/k.o
hipcc -O3 -std=c++17 -fPIC -x hip --offload-arch=gfx950 -I$R/flowfusion_amd/csrc -I$R/include -c $R/flowfusion_amd/csrc/ff_aux.hip -o $T/x.o
cat > $T/table.cpp <<EOT
#include "ff_registry.h"
namespace ff {
int launch_$NAME(const KernelArgs*, unsigned, unsigned, hipStream_t);
const KernelEntry g_kernels[] = { {0, 0, 0, 0, 0, 0, nullptr, "none"} };
const int g_n_kernels = 0;
const SplitKernelEntry g_split_kernels[] = { {$ROW, launch_$NAME, "$NAME"} };
const int g_n_split_kernels = 1;
}
EOT
hipcc -O3 -std=c++17 -fPIC -x c++ -D__HIP_PLATFORM_AMD__=1 -I/opt/rocm/include -I$R/flowfusion_amd/csrc -I$R/include -c $T/table.cpp -o $T/t.o
hipcc -O3 -std=c++17 -fPIC -x c++ -D__HIP_PLATFORM_AMD__=1 $FLAGS -I/opt/rocm/include -I$R/flowfusion_amd/csrc -I$R/include -c $R/flowfusion_amd/csrc/ff_api.cpp -o $T/a.o
hipcc -shared -fPIC --offload-arch=gfx950 -o $OUT $T/k.o $T/x.o $T/t.o $T/a.o
rm -rf $T
echo built $OUT
