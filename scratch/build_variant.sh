#!/bin/bash
# usage: build_variant.sh OUT.so "-DFLAG ..." [instance]   -- builds ONE kernel instance + the api into a private .so
#   instance = a generated source under flowfusion_amd/_build/gen (default mlp_ode_m16_h256_d4_c0_t0_w2) and its
#   registry row "tile,H,dregs,cregs,tangents,any_act" (default 16,256,4,0,0,0), e.g.
#   build_variant.sh /tmp/v.so "-DFF_DEBUG_LINEAR_ACT" mlp_ode_m16_h512_d16_c4_t0 16,512,16,4,0,0
set -e
OUT=$1; FLAGS=$2; NAME=${3:-mlp_ode_m16_h256_d4_c0_t0_w2}; ROW=${4:-16,256,4,0,0,0}
R=/root/repo
T=$(mktemp -d)
hipcc -O3 -std=c++17 -fPIC -x hip --offload-arch=gfx950 $FLAGS -I$R/flowfusion_amd/csrc -I$R/include -c $R/flowfusion_amd/_build/gen/$NAME.hip -o $T/k.o
cat > $T/table.cpp <<EOT
#include "ff_registry.h"
namespace ff {
int launch_$NAME(const KernelArgs*, unsigned, unsigned, hipStream_t);
const KernelEntry g_kernels[] = { {$ROW, launch_$NAME, "$NAME"} };
const int g_n_kernels = 1;
}
EOT
hipcc -O3 -std=c++17 -fPIC -x c++ -D__HIP_PLATFORM_AMD__=1 -I/opt/rocm/include -I$R/flowfusion_amd/csrc -I$R/include -c $T/table.cpp -o $T/t.o
hipcc -O3 -std=c++17 -fPIC -x c++ -D__HIP_PLATFORM_AMD__=1 -I/opt/rocm/include -I$R/flowfusion_amd/csrc -I$R/include -c $R/flowfusion_amd/csrc/ff_api.cpp -o $T/a.o
hipcc -shared -fPIC --offload-arch=gfx950 -o $OUT $T/k.o $T/t.o $T/a.o
rm -rf $T
echo built $OUT
