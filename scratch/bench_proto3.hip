// Scratch: how much VALU work hides behind v_mfma_f32_32x32x2_f32?  Per MFMA slot: NV plain FMAs and
// NT transcendentals (independent of the MFMAs), pinned with sched_barrier(0).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)

template<int NV, int NT, bool CHAIN, bool ACCREAD>
__global__ __launch_bounds__(256, 1) void k(const float* __restrict__ xin, float* __restrict__ xout, int iters)
{
    const int lane = threadIdx.x & 63;
    float P[16], A[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { P[i] = xin[i * 64 + lane]; A[i] = xin[1024 + i * 64 + lane]; }
    f32x16 acc[8];
#pragma unroll
    for (int ob = 0; ob < 8; ++ob)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ob][r] = 0.f;
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = xin[2048 + i * 64 + lane];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 128; ++m) {
            const int ob = CHAIN ? (m / 16) : (m % 8);
            acc[ob] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[m % 16], P[(m * 7) % 16], acc[ob], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NV; ++j) v[j % 8] = __builtin_fmaf(v[j % 8], 1.0001f, 0.5f);
#pragma unroll
            for (int j = 0; j < NT; ++j) v[(j + 4) % 8] = __builtin_amdgcn_exp2f(v[(j + 4) % 8]);
            if (ACCREAD) v[m % 8] += acc[(ob + 7) % 8][m % 16] * 1e-30f;     // read an accumulator that is NOT in flight
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int ob = 0; ob < 8; ++ob)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[ob][r];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i];
    xout[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

int main(int argc, char** argv)
{
    int iters = argc > 1 ? atoi(argv[1]) : 2000;
    int nwg = 1024;
    float *dx, *dy;
    std::vector<float> hx(4096); srand(1); for (auto& v : hx) v = ((rand() / (float)RAND_MAX) - 0.5f) * 0.01f;
    CK(hipMalloc(&dx, 4096 * 4)); CK(hipMalloc(&dy, (size_t)nwg * 256 * 4));
    CK(hipMemcpy(dx, hx.data(), 4096 * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char* name, auto kern) {
        hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), 0, 0, dx, dy, iters); CK(hipDeviceSynchronize());
        float best = 1e30f;
        for (int t = 0; t < 3; ++t) {
            CK(hipEventRecord(e0)); hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), 0, 0, dx, dy, iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        double nm = (double)iters * 128 * 4;      // MFMAs per SIMD-slot sequence: 4 waves-in-sequence per SIMD (1024 WG / 256 CU)
        double cyc = best * 1e-3 * 2.38e9 / nm;
        printf("%-46s %8.3f ms  -> %6.1f cycles per MFMA slot (@2.38 GHz)\n", name, best, cyc); fflush(stdout);
    };
    run("chain, 0 VALU", (k<0, 0, true, false>));
    run("chain, 4 fma", (k<4, 0, true, false>));
    run("chain, 8 fma", (k<8, 0, true, false>));
    run("chain, 12 fma", (k<12, 0, true, false>));
    run("chain, 2 exp", (k<0, 2, true, false>));
    run("chain, 4 exp", (k<0, 4, true, false>));
    run("chain, 4 fma + 2 exp", (k<4, 2, true, false>));
    run("chain, 4 fma + 2 exp + accread", (k<4, 2, true, true>));
    run("indep(8 acc), 0 VALU", (k<0, 0, false, false>));
    run("indep(8 acc), 8 fma", (k<8, 0, false, false>));
    run("indep(8 acc), 4 exp", (k<0, 4, false, false>));
    run("indep(8 acc), 4 fma + 2 exp", (k<4, 2, false, false>));
    return 0;
}
