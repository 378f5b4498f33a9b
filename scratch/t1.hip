#include "../flowfusion_amd/csrc/ff_mlp_ode.hpp"
namespace ff {
template __global__ void mlp_ode_kernel<256,8,0,false>(const KernelArgs);
template __global__ void mlp_ode_kernel<256,8,0,true>(const KernelArgs);
template __global__ void mlp_ode_kernel<256,16,8,false>(const KernelArgs);
template __global__ void mlp_ode_kernel<256,16,8,true>(const KernelArgs);
}
