#!/bin/bash
# usage: build_variant.sh OUT.so "-DFLAG ..."   -- builds only the h256_d8_c0_t0 kernel + api into a private .so
set -e
OUT=$1; FLAGS=$2
R=/root/repo
T=$(mktemp -d)
hipcc -O3 -std=c++17 -fPIC -x hip --offload-arch=gfx950 $FLAGS -I$R/flowfusion_amd/csrc -I$R/include -c $R/flowfusion_amd/_build/gen/mlp_ode_m16_h256_d4_c0_t0_w2.hip -o $T/k.o
cat > $T/table.cpp <<EOT
#include "ff_registry.h"
namespace ff {
int launch_mlp_ode_m16_h256_d4_c0_t0_w2(const KernelArgs*, unsigned, unsigned, hipStream_t);
const KernelEntry g_kernels[] = { {16, 256, 4, 0, 0, 0, launch_mlp_ode_m16_h256_d4_c0_t0_w2, "mlp_ode_m16_h256_d4_c0_t0_w2"} };
const int g_n_kernels = 1;
}
EOT
hipcc -O3 -std=c++17 -fPIC -x c++ -D__HIP_PLATFORM_AMD__=1 -I/opt/rocm/include -I$R/flowfusion_amd/csrc -I$R/include -c $T/table.cpp -o $T/t.o
hipcc -O3 -std=c++17 -fPIC -x c++ -D__HIP_PLATFORM_AMD__=1 -I/opt/rocm/include -I$R/flowfusion_amd/csrc -I$R/include -c $R/flowfusion_amd/csrc/ff_api.cpp -o $T/a.o
hipcc -shared -fPIC --offload-arch=gfx950 -o $OUT $T/k.o $T/t.o $T/a.o
rm -rf $T
echo built $OUT
