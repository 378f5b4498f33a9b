// data_bench.hip -- does the DATA change the s_memtime cycles of a bare v_mfma_f32_16x16x32_bf16 stream?  (scratch)
// Same instruction stream three times: constant operands / operands drawn once at random / operands refreshed by an
// xorshift (integer VALU: free beside the MFMA, see gap_bench) behind every MFMA.  One wavefront per SIMD on every CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <utility>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)
template <class F, int... I> __device__ __forceinline__ void sfor_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F> __device__ __forceinline__ void sfor(F&& f) { sfor_impl(f, std::make_integer_sequence<int, N>{}); }
__device__ __forceinline__ f32x4 mm(u32x4 a, u32x4 b, f32x4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// MODE 0: all operand words equal (1.0); 1: random bf16 in [-1, 1), fixed; 2: random, one word of A and of B refreshed per MFMA;
// 3: random fixed + the same xorshift instructions running on unrelated registers (cost of the integer work itself)
template <int MODE>
__global__ __launch_bounds__(256, 1) void data_kernel(float* out, unsigned long long* cycles, int n_iter, const unsigned* in)
{
    const int lane = threadIdx.x & 63;
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    u32x4 a[3], b[6];
    unsigned seed = in[threadIdx.x + blockIdx.x % 7] | 1u, other = seed * 747796405u;
    auto rnd = [&](unsigned& s) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return (s & 0x807F807Fu) | 0x3F003F00u; };   // two bf16 of magnitude 0.5..1, random sign / mantissa
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) a[i][j] = MODE == 0 ? 0x3F803F80u : rnd(seed);
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) b[i][j] = MODE == 0 ? 0x3F803F80u : rnd(seed);
    unsigned long long c0 = 0, c1 = 0;
    for (int it = 0; it < n_iter + 1; ++it) {
        if (it == 1) c0 = __builtin_readcyclecounter();
        sfor<192>([&](auto gg) {
            constexpr int G = decltype(gg)::value;
            acc[G & 1] = mm(a[(G / 2) % 3], b[G % 6], acc[G & 1]);
            if constexpr (MODE == 2) {
                if constexpr (G % 2 == 0) a[((G / 2) + 1) % 3][(G / 6) % 4] = rnd(seed);
                else b[(G + 3) % 6][(G / 6) % 4] = rnd(seed);
            } else if constexpr (MODE == 3) {
                other ^= rnd(seed);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        if (it % 64 == 63) { acc[0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[1] = acc[0]; }      // keep the sums finite
    }
    c1 = __builtin_readcyclecounter();
    out[blockIdx.x * 256 + threadIdx.x] = acc[0][0] + acc[1][1] + __builtin_bit_cast(float, other & 0x3FFFFFFFu);
    if (blockIdx.x == 0 && threadIdx.x == 0) cycles[0] = c1 - c0;
}
static const char* kNames[] = {"all operand words 1.0", "random operands, fixed", "random operands, one word of A / B refreshed behind every MFMA",
                               "random fixed operands + the refresh instructions on unrelated registers"};
template <int MODE> void run(float* out, unsigned long long* cyc, const unsigned* in, int n_iter)
{
    auto k = data_kernel<MODE>;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, out, cyc, 8, in); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, out, cyc, n_iter, in);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
    const double n = 192.0 * n_iter;
    printf("%-80s %6.2f s_memtime ticks/MFMA  %6.2f ns/MFMA  (ticks per ns %.3f)\n", kNames[MODE], c / n, ms * 1e6 / n, (c / n) / (ms * 1e6 / n));
    if constexpr (MODE < 3) run<MODE + 1>(out, cyc, in, n_iter);
}
int main()
{
    float* out; unsigned* in; unsigned long long* cyc;
    CK(hipMalloc(&out, 256 * 256 * 4)); CK(hipMalloc(&in, 4096)); CK(hipMalloc(&cyc, 8));
    unsigned h[1024]; srand(5); for (int i = 0; i < 1024; ++i) h[i] = (unsigned)rand() * 2654435761u;
    CK(hipMemcpy(in, h, 4096, hipMemcpyHostToDevice));
    run<0>(out, cyc, in, 20000);        // ~60 ms per launch: long enough for the power management to settle
    return 0;
}
