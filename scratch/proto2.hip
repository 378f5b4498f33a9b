#include <hip/hip_runtime.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template<int H>
__global__ __launch_bounds__(256, 1) void proto(const float* __restrict__ wp, const float* __restrict__ xin, float* __restrict__ xout, int n_layers, int n_evals)
{
    constexpr int NB = H/32;
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    float P[NB*16];
#pragma unroll
    for (int i = 0; i < NB*16; ++i) P[i] = xin[(size_t)wave*64*NB*16 + i*64 + lane];
    for (int e = 0; e < n_evals; ++e) {
    const float* w = wp;
    for (int l = 0; l < n_layers; ++l) {
        f32x16 acc[NB];
#pragma unroll
        for (int ob = 0; ob < NB; ++ob) {
            const f32x4* bias = (const f32x4*)(w + (size_t)NB*NB*4*256 + ob*32 + 4*(lane>>5));
#pragma unroll
            for (int j = 0; j < 4; ++j) { f32x4 b = bias[2*j]; acc[ob][4*j]=b[0]; acc[ob][4*j+1]=b[1]; acc[ob][4*j+2]=b[2]; acc[ob][4*j+3]=b[3]; }
        }
        // packed: [g][ob][lane][4]
        const f32x4* wb = (const f32x4*)w + lane;
#pragma unroll
        for (int g = 0; g < NB*4; ++g) {
#pragma unroll
            for (int ob = 0; ob < NB; ++ob) {
                f32x4 a = wb[(g*NB+ob)*64];
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    acc[ob] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q], P[g*4+q], acc[ob], 0, 0, 0);
            }
        }
#pragma unroll
        for (int ob = 0; ob < NB; ++ob) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float a = acc[ob][r];
                float ex = __expf(-a);
                P[ob*16+r] = a * __builtin_amdgcn_rcpf(1.0f + ex);
            }
        }
        w += (size_t)NB*NB*4*256 + H;
    }
    }
#pragma unroll
    for (int i = 0; i < NB*16; ++i) xout[(size_t)wave*64*NB*16 + i*64 + lane] = P[i];
}
template __global__ void proto<256>(const float*, const float*, float*, int, int);
