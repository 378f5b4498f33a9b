#include "../flowfusion_amd/csrc/ff_mlp_ode.hpp"
namespace ff {
template __global__ void mlp_ode_kernel<16,256,4,0,false,3,2>(const KernelArgs);
template __global__ void mlp_ode_kernel<16,256,4,0,false,3,4>(const KernelArgs);
template __global__ void mlp_ode_kernel<16,256,4,0,false,2,4>(const KernelArgs);
}
