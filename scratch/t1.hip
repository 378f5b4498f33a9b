#include "../flowfusion_amd/csrc/ff_mlp_ode.hpp"
namespace ff {
template __global__ void mlp_ode_kernel<32,256,8,0,false,1>(const KernelArgs);
template __global__ void mlp_ode_kernel<32,256,8,0,true,1>(const KernelArgs);
template __global__ void mlp_ode_kernel<16,256,4,0,false,2>(const KernelArgs);
template __global__ void mlp_ode_kernel<16,512,16,0,false,1>(const KernelArgs);
template __global__ void mlp_ode_kernel<16,512,16,4,true,1>(const KernelArgs);
}
