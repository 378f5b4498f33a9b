// Scratch: practical ceiling of fp32 MFMA streams on gfx950 as a function of tile shape and wavefronts
// per SIMD, and what independent VALU work from the SAME or the OTHER wavefront costs.
//   TILE 16: v_mfma_f32_16x16x4_f32 (8 passes)    TILE 32: v_mfma_f32_32x32x2_f32 (16 passes)
//   WPS: wavefronts per SIMD (blocks of 256 threads, WPS blocks per CU)
//   NV:  independent v_fma_f32 issued after every MFMA by the same wavefront
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)

template <int TILE, int WPS, int NV>
__global__ __launch_bounds__(256, WPS) void k(const float* __restrict__ xin, float* __restrict__ xout, int iters)
{
    // operand pattern of the product kernel: A from a ring of 16-byte weight registers, B from the
    // activation registers P[64], accumulators rotating (16 of 16x16 or 4 of 32x32)
    const int lane = threadIdx.x & 63;
    float P[64], v[8];
    f32x4 ring[8];
#pragma unroll
    for (int i = 0; i < 64; ++i) P[i] = xin[lane + 64 * (i & 15)] + i;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        v[i] = xin[1024 + lane + 64 * i];
        ring[i] = f32x4{xin[lane + i], xin[lane + i + 64], xin[lane + i + 128], xin[lane + i + 192]};
    }
    float s = 0.f;
    if constexpr (TILE == 16) {
        f32x4 acc[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = f32x4{0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int g = 0; g < 16; ++g)                 // 16 k-groups x 4 k-steps x 2 tiles = 128 MFMAs
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int p = 0; p < 2; ++p) {
                        const int i = (2 * g + p) & 15;
                        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(ring[(2 * g + p) & 7][q], P[4 * g + q], acc[i], 0, 0, 0);
#pragma unroll
                        for (int j = 0; j < NV; ++j) v[(q + j) & 7] = __builtin_fmaf(v[(q + j) & 7], 1.0001f, 0.5f);
                    }
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else {
        f32x16 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int g = 0; g < 16; ++g)                 // 64 MFMAs
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    acc[g & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(ring[g & 7][q], P[4 * g + q], acc[g & 3], 0, 0, 0);
#pragma unroll
                    for (int j = 0; j < NV; ++j) v[(q + j) & 7] = __builtin_fmaf(v[(q + j) & 7], 1.0001f, 0.5f);
                }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += acc[i][r];
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i];
    xout[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

template <int TILE, int WPS, int NV>
void run(float* dx, float* dy, int iters)
{
    const int nwg = 256 * WPS * 4;       // 4 rounds of full occupancy
    auto kern = k<TILE, WPS, NV>;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), 0, 0, dx, dy, iters / 8);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), 0, 0, dx, dy, iters);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double mfma = (double)nwg * 4 * iters * (TILE == 16 ? 128 : 64);
    const double flop = mfma * (TILE == 16 ? 2.0 * 16 * 16 * 4 : 2.0 * 32 * 32 * 2);
    const double cyc_per_mfma = TILE == 16 ? 32.0 : 64.0;
    printf("tile %2d  waves/SIMD %d  VALU/MFMA %d : %8.2f ms  %7.2f TFLOP/s   (%.1f%% of 157.3)\n", TILE, WPS, NV, ms,
           flop / ms / 1e9, 100.0 * flop / ms / 1e9 / 157.3);
    (void)cyc_per_mfma;
}

int main(int argc, char** argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    float *dx, *dy;
    CK(hipMalloc(&dx, 4096 * 4)); CK(hipMalloc(&dy, (size_t)256 * 8 * 4 * 256 * 4));
    float hx[4096];
    for (int i = 0; i < 4096; ++i) hx[i] = (rand() / (float)RAND_MAX - 0.5f) * 0.01f;
    CK(hipMemcpy(dx, hx, sizeof(hx), hipMemcpyHostToDevice));
    run<32, 1, 0>(dx, dy, iters);
    run<32, 2, 0>(dx, dy, iters);
    run<16, 1, 0>(dx, dy, iters);
    run<16, 2, 0>(dx, dy, iters);
    run<16, 4, 0>(dx, dy, iters);
    run<16, 1, 1>(dx, dy, iters);
    run<16, 2, 1>(dx, dy, iters);
    run<16, 2, 2>(dx, dy, iters);
    run<16, 2, 4>(dx, dy, iters);
    run<16, 4, 2>(dx, dy, iters);
    run<32, 1, 2>(dx, dy, iters);
    run<32, 2, 2>(dx, dy, iters);
    run<32, 2, 4>(dx, dy, iters);
    return 0;
}
