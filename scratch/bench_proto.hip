// Scratch microbenchmark: which register/LDS structure sustains the FP32 MFMA rate
// for a chain of 256x256 layers with in-register SiLU? Not product code.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)

__device__ __forceinline__ float silu(float a){ float ex = __expf(-a); return a * __builtin_amdgcn_rcpf(1.0f + ex); }

// V1: 32x32x2, k-outer, direct global loads. stream order [g][ob][lane][4], bias after.
template<int H>
__global__ __launch_bounds__(256, 1) void v1(const float* __restrict__ wp, const float* __restrict__ xin, float* __restrict__ xout, int n_layers, int n_evals)
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
        for (int ob = 0; ob < NB; ++ob)
#pragma unroll
            for (int r = 0; r < 16; ++r) P[ob*16+r] = silu(acc[ob][r]);
        w += (size_t)NB*NB*4*256 + H;
    }
    }
#pragma unroll
    for (int i = 0; i < NB*16; ++i) xout[(size_t)wave*64*NB*16 + i*64 + lane] = P[i];
}

// V2: same but weights staged through LDS: chunk = 4 groups (NB*4 KiB), 2 slots.
template<int H>
__global__ __launch_bounds__(256, 1) void v2(const float* __restrict__ wp, const float* __restrict__ xin, float* __restrict__ xout, int n_layers, int n_evals)
{
    constexpr int NB = H/32;
    constexpr int CHUNK_F = 4*NB*256;          // floats per chunk (4 groups)
    constexpr int NCH = NB;                     // chunks per layer
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    const int wid = threadIdx.x >> 6;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    float P[NB*16];
#pragma unroll
    for (int i = 0; i < NB*16; ++i) P[i] = xin[(size_t)wave*64*NB*16 + i*64 + lane];
    const size_t layer_f = (size_t)NB*NB*4*256 + H;
    // issue chunk loader: each wave loads CHUNK_F/4 floats = NB KiB -> NB glds16
    auto issue = [&](const float* src, int slot) {
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const float* g = src + (size_t)(wid*NB + i)*256 + lane*4;
            float* d = lds + slot*CHUNK_F + (wid*NB + i)*256;   // wave-uniform base
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)d, 16, 0, 0);
        }
    };
    const int total_chunks = n_evals * n_layers * NCH;
    issue(wp, 0);
    __syncthreads();
    int cidx = 0;
    for (int e = 0; e < n_evals; ++e) {
    for (int l = 0; l < n_layers; ++l) {
        const float* w = wp + l*layer_f;
        f32x16 acc[NB];
#pragma unroll
        for (int ob = 0; ob < NB; ++ob) {
            const f32x4* bias = (const f32x4*)(w + (size_t)NB*NB*4*256 + ob*32 + 4*(lane>>5));
#pragma unroll
            for (int j = 0; j < 4; ++j) { f32x4 b = bias[2*j]; acc[ob][4*j]=b[0]; acc[ob][4*j+1]=b[1]; acc[ob][4*j+2]=b[2]; acc[ob][4*j+3]=b[3]; }
        }
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            // prefetch next chunk
            int nc = cidx + 1;
            if (nc < total_chunks) {
                int nl = (l*NCH + c + 1);           // next chunk linear index within eval
                int ll = nl / NCH, cc = nl % NCH;
                if (ll == n_layers) { ll = 0; }
                issue(wp + ll*layer_f + (size_t)cc*CHUNK_F, nc & 1);
            }
            const f32x4* wb = (const f32x4*)(lds + (cidx & 1)*CHUNK_F) + lane;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
#pragma unroll
                for (int ob = 0; ob < NB; ++ob) {
                    f32x4 a = wb[(g*NB+ob)*64];
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        acc[ob] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q], P[(c*4+g)*4+q], acc[ob], 0, 0, 0);
                }
            }
            __syncthreads();
            ++cidx;
        }
#pragma unroll
        for (int ob = 0; ob < NB; ++ob)
#pragma unroll
            for (int r = 0; r < 16; ++r) P[ob*16+r] = silu(acc[ob][r]);
    }
    }
#pragma unroll
    for (int i = 0; i < NB*16; ++i) xout[(size_t)wave*64*NB*16 + i*64 + lane] = P[i];
}

// V3: 16x16x4, 16 samples per wave, 8 waves / WG (2 per SIMD), LDS staged. A frag: lane holds A[l&15][k=l>>4]
// stream order per chunk: [g][ob16][lane][4] where ob16 indexes 16-row blocks (H/16 of them), g = group of 4 k-steps (16 k each).
template<int H, int WAVES>
__global__ __launch_bounds__(WAVES*64, WAVES/4) void v3(const float* __restrict__ wp, const float* __restrict__ xin, float* __restrict__ xout, int n_layers, int n_evals)
{
    constexpr int NB = H/16;                    // 16-row blocks (16)
    constexpr int NG = H/16;                    // k groups per layer: each group = 4 ksteps x 4 k = 16 k
    constexpr int GPC = 2;                      // groups per chunk
    constexpr int CHUNK_F = GPC*NB*256;         // floats per chunk = 2*16*256*4B = 32 KiB
    constexpr int NCH = NG/GPC;
    constexpr int PER_WAVE = CHUNK_F/256/WAVES; // glds per wave per chunk
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    const int wid = threadIdx.x >> 6;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    float P[NB*4];
#pragma unroll
    for (int i = 0; i < NB*4; ++i) P[i] = xin[(size_t)wave*64*NB*4 + i*64 + lane];
    const size_t layer_f = (size_t)NG*NB*256 + H;
    auto issue = [&](const float* src, int slot) {
#pragma unroll
        for (int i = 0; i < PER_WAVE; ++i) {
            const float* g = src + (size_t)(wid*PER_WAVE + i)*256 + lane*4;
            float* d = lds + slot*CHUNK_F + (wid*PER_WAVE + i)*256;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)d, 16, 0, 0);
        }
    };
    const int total_chunks = n_evals * n_layers * NCH;
    issue(wp, 0);
    __syncthreads();
    int cidx = 0;
    for (int e = 0; e < n_evals; ++e) {
    for (int l = 0; l < n_layers; ++l) {
        const float* w = wp + l*layer_f;
        f32x4 acc[NB];
#pragma unroll
        for (int ob = 0; ob < NB; ++ob) {
            const f32x4* bias = (const f32x4*)(w + (size_t)NG*NB*256 + ob*16 + 4*(lane>>4));
            acc[ob] = bias[0];
        }
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            int nc = cidx + 1;
            if (nc < total_chunks) {
                int nl = (l*NCH + c + 1);
                int ll = nl / NCH, cc = nl % NCH;
                if (ll == n_layers) { ll = 0; }
                issue(wp + ll*layer_f + (size_t)cc*CHUNK_F, nc & 1);
            }
            const f32x4* wb = (const f32x4*)(lds + (cidx & 1)*CHUNK_F) + lane;
#pragma unroll
            for (int g = 0; g < GPC; ++g) {
#pragma unroll
                for (int ob = 0; ob < NB; ++ob) {
                    f32x4 a = wb[(g*NB+ob)*64];
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        acc[ob] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q], P[(c*GPC+g)*4+q], acc[ob], 0, 0, 0);
                }
            }
            __syncthreads();
            ++cidx;
        }
#pragma unroll
        for (int ob = 0; ob < NB; ++ob)
#pragma unroll
            for (int r = 0; r < 4; ++r) P[ob*4+r] = silu(acc[ob][r]);
    }
    }
#pragma unroll
    for (int i = 0; i < NB*4; ++i) xout[(size_t)wave*64*NB*4 + i*64 + lane] = P[i];
}

int main(int argc, char** argv)
{
    const int H = 256, NL = 3;
    int n_evals = argc > 1 ? atoi(argv[1]) : 40;
    int nwg = argc > 2 ? atoi(argv[2]) : 1024;
    size_t layer_f = (size_t)H*H + H;
    std::vector<float> hw(layer_f*NL);
    srand(1);
    for (auto& v : hw) v = ((rand() / (float)RAND_MAX) - 0.5f) * 0.2f;
    float *dw, *dx, *dy;
    size_t nx = (size_t)nwg * 8 * 64 * 128;  // generous
    CK(hipMalloc(&dw, hw.size()*4)); CK(hipMalloc(&dx, nx*4)); CK(hipMalloc(&dy, nx*4));
    CK(hipMemcpy(dw, hw.data(), hw.size()*4, hipMemcpyHostToDevice));
    std::vector<float> hx(nx); for (auto& v : hx) v = ((rand() / (float)RAND_MAX) - 0.5f) * 2.f;
    CK(hipMemcpy(dx, hx.data(), nx*4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char* name, auto launch, double samples_per_wg) {
        launch(); CK(hipDeviceSynchronize());
        float best = 1e30f;
        for (int it = 0; it < 3; ++it) {
            CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        CK(hipGetLastError());
        double flop = samples_per_wg * nwg * (double)n_evals * NL * 2.0 * H * H;
        std::vector<float> out(64); CK(hipMemcpy(out.data(), dy, 256, hipMemcpyDeviceToHost));
        printf("%-28s %8.3f ms  %7.2f TFLOP/s  (out[0]=%g out[5]=%g)\n", name, best, flop / best / 1e9, out[0], out[5]);
        fflush(stdout);
    };
    CK(hipFuncSetAttribute((const void*)v2<256>, hipFuncAttributeMaxDynamicSharedMemorySize, 2*32768));
    CK(hipFuncSetAttribute((const void*)v3<256,8>, hipFuncAttributeMaxDynamicSharedMemorySize, 2*32768));
    CK(hipFuncSetAttribute((const void*)v3<256,12>, hipFuncAttributeMaxDynamicSharedMemorySize, 2*32768*3/2));
    run("v1 32x32x2 direct", [&]{ hipLaunchKernelGGL(v1<256>, dim3(nwg), dim3(256), 0, 0, dw, dx, dy, NL, n_evals); }, 128);
    run("v2 32x32x2 lds", [&]{ hipLaunchKernelGGL(v2<256>, dim3(nwg), dim3(256), 2*32768, 0, dw, dx, dy, NL, n_evals); }, 128);
    run("v3 16x16x4 lds 8 waves", [&]{ hipLaunchKernelGGL((v3<256,8>), dim3(nwg), dim3(512), 2*32768, 0, dw, dx, dy, NL, n_evals); }, 128);
    return 0;
}
