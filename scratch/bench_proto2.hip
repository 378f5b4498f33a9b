// Scratch: where does the fp32 MFMA rate go?  Variants of a 1024-MFMA loop body (one 256x256 layer
// worth) with 8 accumulators, all operands in registers, then with the weight loads added.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)

// MODE 0: pure MFMA, q-major over 8 independent accumulators, A from 4 regs, B from P[128]
// MODE 1: pure MFMA, 4 dependent per accumulator then next accumulator
// MODE 2: MODE 0 + one 16-B buffer load per 4 MFMAs (ring of 8), loads feed A
// MODE 3: MODE 1 + loads
template<int MODE>
__global__ __launch_bounds__(256, 1) void k(const float* __restrict__ wp, const float* __restrict__ xin, float* __restrict__ xout, int iters)
{
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    float P[128];
#pragma unroll
    for (int i = 0; i < 128; ++i) P[i] = xin[(size_t)(wave % 64) * 64 * 128 + i * 64 + lane];
    f32x16 acc[8];
#pragma unroll
    for (int ob = 0; ob < 8; ++ob)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ob][r] = 0.f;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)wp, 0, 256 * 1024 * 4, 0x00020000);
    f32x4 ring[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) ring[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, i * 1024, 0));
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 32; ++g) {
            if (MODE == 0 || MODE == 2) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int ob = 0; ob < 8; ++ob)
                        acc[ob] = __builtin_amdgcn_mfma_f32_32x32x2f32(ring[ob][q], P[g * 4 + q], acc[ob], 0, 0, 0);
                if (MODE == 2) {
#pragma unroll
                    for (int ob = 0; ob < 8; ++ob)
                        ring[ob] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, ((g + 1) & 31) * 8192 + ob * 1024, 0));
                }
                __builtin_amdgcn_sched_barrier(0);
            } else {
#pragma unroll
                for (int ob = 0; ob < 8; ++ob) {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        acc[ob] = __builtin_amdgcn_mfma_f32_32x32x2f32(ring[ob][q], P[g * 4 + q], acc[ob], 0, 0, 0);
                    if (MODE == 3)
                        ring[ob] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, ((g + 1) & 31) * 8192 + ob * 1024, 0));
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int ob = 0; ob < 8; ++ob)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[ob][r];
    xout[(size_t)wave * 64 + lane] = s;
}

int main(int argc, char** argv)
{
    int iters = argc > 1 ? atoi(argv[1]) : 400;
    int nwg = argc > 2 ? atoi(argv[2]) : 1024;
    std::vector<float> hw(256 * 1024);
    srand(1);
    for (auto& v : hw) v = ((rand() / (float)RAND_MAX) - 0.5f) * 0.01f;
    float *dw, *dx, *dy;
    size_t nx = (size_t)64 * 64 * 128;
    CK(hipMalloc(&dw, hw.size() * 4)); CK(hipMalloc(&dx, nx * 4)); CK(hipMalloc(&dy, (size_t)nwg * 256 * 4));
    CK(hipMemcpy(dw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    std::vector<float> hx(nx); for (auto& v : hx) v = ((rand() / (float)RAND_MAX) - 0.5f) * 2.f;
    CK(hipMemcpy(dx, hx.data(), nx * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char* name, auto kern) {
        hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), 0, 0, dw, dx, dy, iters); CK(hipDeviceSynchronize());
        float best = 1e30f;
        for (int t = 0; t < 3; ++t) {
            CK(hipEventRecord(e0)); hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), 0, 0, dw, dx, dy, iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        double flop = (double)nwg * 4 * iters * 1024.0 * 4096.0;
        printf("%-44s %9.3f ms  %7.2f TFLOP/s\n", name, best, flop / best / 1e9); fflush(stdout);
    };
    run("0 pure MFMA q-major (8 indep acc)", k<0>);
    run("1 pure MFMA 4-dependent chains", k<1>);
    run("2 q-major + L2 loads (ring 8)", k<2>);
    run("3 dependent chains + L2 loads", k<3>);
    return 0;
}
