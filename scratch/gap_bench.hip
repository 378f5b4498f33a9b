// gap_bench.hip -- what does one instruction cost behind a v_mfma_f32_16x16x32_bf16?  (scratch, timing only)
// One wavefront per SIMD on every CU runs N x 192 MFMAs (two accumulation chains, as the split kernel's groups) with a
// chosen filler pinned behind every MFMA / every k-th MFMA (full scheduling barriers: the stream is the source order)
// and reports shader cycles per MFMA (s_memtime), wavefront 0 of workgroup 0.
//   hipcc -O3 --offload-arch=gfx950 scratch/gap_bench.hip -o /tmp/gap_bench && /tmp/gap_bench
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

enum { F_NONE, F_ADD1, F_ADD2, F_ADD3, F_ADD4, F_EXP1, F_EXP1_ADD1, F_EXP_EVERY2, F_EXP_EVERY4, F_EXP2_EVERY4, F_RCP1, F_ACCREAD1, F_ACCREAD2,
       F_AND_LIT2, F_PERM2, F_DSREAD_EVERY4, F_DSREAD_EVERY4_ADD2, F_MIX, F_MIX_DEP, F_FMA2, F_MUL2, F_ADD8_EVERY8, F_ADD16_EVERY16, F_ADD8_EVERY8_EXP, F_ADD8_EVERY4, F_MUL8_EVERY8_DEP_EXP, F_EXPDEP1, F_EXPDEP2, F_EXPDEP3, F_EXPDEP4, F_EXPDEP6, F_EXPDEP8, F_EXPDEP12, F_ADDDEP1, F_ADDDEP2, F_EXP_AFTER_ADD1, F_EXP_AFTER_ADD2, F_COUNT };
static const char* kNames[F_COUNT] = {
    "none", "1 v_add_f32", "2 v_add_f32", "3 v_add_f32", "4 v_add_f32", "1 v_exp_f32", "1 v_exp_f32 + 1 v_add_f32",
    "1 v_exp_f32 behind every 2nd MFMA", "1 v_exp_f32 behind every 4th MFMA", "2 v_exp_f32 behind every 4th MFMA", "1 v_rcp_f32",
    "1 v_accvgpr_read", "2 v_accvgpr_read", "2 v_and_b32 with a literal", "2 v_perm_b32",
    "1 ds_read_b128 behind every 4th MFMA", "1 ds_read_b128 every 4th + 2 v_add_f32 in the others",
    "the split kernel's mix (14 micro-ops per 24 gaps, independent)", "the same mix, each micro-op fed by the one two gaps before",
    "2 v_fma_f32", "2 v_mul_f32", "8 v_add_f32 behind every 8th MFMA", "16 v_add_f32 behind every 16th MFMA",
    "8 v_add_f32 behind every 8th MFMA, 1 v_exp_f32 behind the others", "8 v_add_f32 behind every 4th MFMA",
    "8 v_mul_f32 every 8th, fed by the 7 v_exp_f32 of the gaps before (lockstep plan)",
    "every 2nd gap: v_exp_f32; v_mul_f32 of its result 1 gap later", "... 2 gaps later", "... 3 gaps later", "... 4 gaps later", "... 6 gaps later", "... 8 gaps later", "... 12 gaps later",
    "every 2nd gap: v_add_f32; v_mul_f32 of its result 1 gap later", "... 2 gaps later",
    "every 2nd gap: v_add_f32; v_exp_f32 of its result 1 gap later", "... 2 gaps later"};

template <int FILL, int NCHAIN = 2>
__global__ __launch_bounds__(256, 1) void gap_kernel(float* out, unsigned long long* cycles, int n_iter, const float* in)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63;
    f32x4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 big[8];                                             // lives in AGPRs under pressure; read by the ACCREAD fillers
#pragma unroll
    for (int i = 0; i < 8; ++i) big[i] = f32x4{in[lane + i], in[lane + 8 + i], in[lane + 16 + i], in[lane + 24 + i]};
    u32x4 w = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    u32x4 b = u32x4{(unsigned)lane, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = in[lane + 64 + i];
    u32x4 wl = u32x4{0u, 0u, 0u, 0u};
    for (int i = threadIdx.x; i < 4096; i += 256) ((float*)lds)[i] = 1.0f;
    __syncthreads();
    unsigned long long t0 = 0, t1 = 0;
    for (int it = 0; it < n_iter + 1; ++it) {
        if (it == 1) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)"); t0 = __builtin_readcyclecounter(); }
        sfor<192>([&](auto gg) {
            constexpr int G = decltype(gg)::value;
            acc[G % NCHAIN] = mm(w, b, acc[G % NCHAIN]);
            constexpr int r = G % 16;
            if constexpr (FILL == F_ADD1) { v[r] += 1.0f; }
            else if constexpr (FILL == F_ADD2) { v[r] += 1.0f; v[(r + 5) % 16] += 1.0f; }
            else if constexpr (FILL == F_ADD3) { v[r] += 1.0f; v[(r + 5) % 16] += 1.0f; v[(r + 10) % 16] += 1.0f; }
            else if constexpr (FILL == F_ADD4) { v[r] += 1.0f; v[(r + 4) % 16] += 1.0f; v[(r + 8) % 16] += 1.0f; v[(r + 12) % 16] += 1.0f; }
            else if constexpr (FILL == F_FMA2) { v[r] = __builtin_fmaf(v[r], 1.0001f, 0.5f); v[(r + 5) % 16] = __builtin_fmaf(v[(r + 5) % 16], 1.0001f, 0.5f); }
            else if constexpr (FILL == F_MUL2) { v[r] *= 1.0001f; v[(r + 5) % 16] *= 1.0001f; }
            else if constexpr (FILL == F_ADD8_EVERY8) { if constexpr (G % 8 == 0) { sfor<8>([&](auto k) { v[decltype(k)::value] += 1.0f; }); } }
            else if constexpr (FILL == F_ADD8_EVERY4) { if constexpr (G % 4 == 0) { sfor<8>([&](auto k) { v[decltype(k)::value + 8 * ((G / 4) % 2)] += 1.0f; }); } }
            else if constexpr (FILL == F_ADD16_EVERY16) { if constexpr (G % 16 == 0) { sfor<16>([&](auto k) { v[decltype(k)::value] += 1.0f; }); } }
            else if constexpr (FILL == F_ADD8_EVERY8_EXP) {
                if constexpr (G % 8 == 0) { sfor<8>([&](auto k) { v[decltype(k)::value] += 1.0f; }); }
                else v[8 + G % 8] = __builtin_amdgcn_exp2f(v[8 + G % 8]);
            }
            else if constexpr (FILL == F_MUL8_EVERY8_DEP_EXP) {
                if constexpr (G % 8 == 0) { sfor<8>([&](auto k) { constexpr int K = decltype(k)::value; v[K] = v[K] * v[8 + K]; }); }
                else v[8 + G % 8] = __builtin_amdgcn_exp2f(v[G % 8]);
            }
            else if constexpr (FILL >= F_EXPDEP1 && FILL <= F_EXP_AFTER_ADD2) {
                // producers on the even gaps (register (G/2) % 16), one consumer of the producer K gaps back on the gaps in between
                constexpr int K = FILL == F_EXPDEP1 ? 1 : FILL == F_EXPDEP2 ? 2 : FILL == F_EXPDEP3 ? 3 : FILL == F_EXPDEP4 ? 4 : FILL == F_EXPDEP6 ? 6
                                  : FILL == F_EXPDEP8 ? 8 : FILL == F_EXPDEP12 ? 12 : (FILL == F_ADDDEP1 || FILL == F_EXP_AFTER_ADD1) ? 1 : 2;
                constexpr bool prod_exp = FILL <= F_EXPDEP12, cons_exp = FILL >= F_EXP_AFTER_ADD1;
                if constexpr (G % 2 == 0) {
                    constexpr int r = (G / 2) % 16;
                    if constexpr (prod_exp) v[r] = __builtin_amdgcn_exp2f(v[r]); else v[r] = v[r] + 1.0f;
                }
                if constexpr ((G - K) % 2 == 0 && G >= K) {             // (K even: producer and consumer share a gap)
                    constexpr int r = ((G - K) / 2) % 16;
                    if constexpr (cons_exp) v[r] = __builtin_amdgcn_exp2f(v[r]); else v[r] = v[r] * 0.5f;
                }
            }
            else if constexpr (FILL == F_EXP1) { v[r] = __builtin_amdgcn_exp2f(v[r]); }
            else if constexpr (FILL == F_EXP1_ADD1) { v[r] = __builtin_amdgcn_exp2f(v[r]); v[(r + 5) % 16] += 1.0f; }
            else if constexpr (FILL == F_EXP_EVERY2) { if constexpr (G % 2 == 0) v[r] = __builtin_amdgcn_exp2f(v[r]); }
            else if constexpr (FILL == F_EXP_EVERY4) { if constexpr (G % 4 == 0) v[r] = __builtin_amdgcn_exp2f(v[r]); }
            else if constexpr (FILL == F_EXP2_EVERY4) { if constexpr (G % 4 == 0) { v[r] = __builtin_amdgcn_exp2f(v[r]); v[r + 1] = __builtin_amdgcn_exp2f(v[r + 1]); } }
            else if constexpr (FILL == F_RCP1) { v[r] = __builtin_amdgcn_rcpf(v[r]); }
            else if constexpr (FILL == F_ACCREAD1) { float x; asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x) : "a"(big[G % 8][0])); v[r] = x; }
            else if constexpr (FILL == F_ACCREAD2) {
                float x, y;
                asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x) : "a"(big[G % 8][0]));
                asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(y) : "a"(big[G % 8][1]));
                v[r] = x; v[(r + 5) % 16] = y;
            }
            else if constexpr (FILL == F_AND_LIT2) {
                v[r] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v[r]) & 0xFFFF0000u);
                v[(r + 5) % 16] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v[(r + 5) % 16]) & 0xFFFF0000u);
            }
            else if constexpr (FILL == F_PERM2) {
                v[r] = __builtin_bit_cast(float, __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, v[r]), __builtin_bit_cast(unsigned, v[(r + 1) % 16]), 0x07060302u));
                v[(r + 5) % 16] = __builtin_bit_cast(float, __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, v[(r + 5) % 16]), __builtin_bit_cast(unsigned, v[(r + 6) % 16]), 0x07060302u));
            }
            else if constexpr (FILL == F_DSREAD_EVERY4) { if constexpr (G % 4 == 1) wl = *(const u32x4*)(lds + lane * 16 + (G % 8) * 1024); }
            else if constexpr (FILL == F_DSREAD_EVERY4_ADD2) {
                if constexpr (G % 4 == 1) wl = *(const u32x4*)(lds + lane * 16 + (G % 8) * 1024);
                else { v[r] += 1.0f; v[(r + 5) % 16] += 1.0f; }
            }
            else if constexpr (FILL == F_MIX || FILL == F_MIX_DEP) {
                // 14 micro-ops per 24 gaps on the odd gaps + the fragment reads: what the split kernel pins (state-only mode)
                constexpr int ph = G % 24;
                constexpr int u = FILL == F_MIX_DEP ? 0 : (ph % 16);        // dependent: one register chain; independent: rotating
                if constexpr (G % 4 == 1 && (G / 4) % 1 == 0 && (G % 12) / 4 < 3) wl = *(const u32x4*)(lds + lane * 16 + (G % 8) * 1024);
                if constexpr (ph % 2 == 0 && ph / 2 < 14) {
                    constexpr int j = ph / 2;
                    float& a0 = v[u]; float& a1 = v[(u + 1) % 16]; float& c0 = v[(u + 2) % 16]; float& c1 = v[(u + 3) % 16];
                    if constexpr (j == 0) { float x, y; asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x) : "a"(big[G % 8][0])); asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(y) : "a"(big[G % 8][1])); a0 = x; a1 = y; }
                    else if constexpr (j == 1) { c0 = a0 * -1.4426950f; c1 = a1 * -1.4426950f; }
                    else if constexpr (j == 2) { c0 = __builtin_amdgcn_exp2f(c0); }
                    else if constexpr (j == 3) { c1 = __builtin_amdgcn_exp2f(c1); }
                    else if constexpr (j == 4) { c0 += 1.0f; c1 += 1.0f; }
                    else if constexpr (j == 5) { c0 = __builtin_amdgcn_rcpf(c0); }
                    else if constexpr (j == 6) { c1 = __builtin_amdgcn_rcpf(c1); }
                    else if constexpr (j == 7) { a0 *= c0; a1 *= c1; }
                    else if constexpr (j == 8) { c0 = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, a0) & 0xFFFF0000u); c1 = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, a1) & 0xFFFF0000u); }
                    else if constexpr (j == 9) { c0 = a0 - c0; c1 = a1 - c1; }
                    else if constexpr (j == 10) { b[1] = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, a1), __builtin_bit_cast(unsigned, a0), 0x07060302u); a0 = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, c0) & 0xFFFF0000u); }
                    else if constexpr (j == 11) { a1 = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, c1) & 0xFFFF0000u); b[2] = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, c1), __builtin_bit_cast(unsigned, c0), 0x07060302u); }
                    else if constexpr (j == 12) { a0 = c0 - a0; a1 = c1 - a1; }
                    else { b[3] = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, a1), __builtin_bit_cast(unsigned, a0), 0x07060302u); }
                }
            }
#ifdef ALIGN_WAVES     // the product kernel's barrier every 8 groups keeps the four wavefronts of a workgroup in step
            if constexpr (G % 96 == 95) __builtin_amdgcn_s_barrier();
#endif
            __builtin_amdgcn_sched_barrier(0);
        });
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)");
    t1 = __builtin_readcyclecounter();
    float s = __builtin_bit_cast(float, wl[0]);
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0];
#pragma unroll
    for (int i = 0; i < 16; ++i) s += v[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += big[i][2];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) cycles[0] = t1 - t0;
}

template <int FILL, int NCHAIN = 2> void run1(float* out, unsigned long long* cyc, const float* in, int n_iter)
{
    auto k = gap_kernel<FILL, NCHAIN>;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k, dim3(256), dim3(256), 16384, 0, out, cyc, 8, in);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(256), dim3(256), 16384, 0, out, cyc, n_iter, in);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
    const double n = 192.0 * n_iter;
    printf("[%d chains] %-72s %6.2f cycles/MFMA   (%.2f ns/MFMA wall, %.2f GHz implied)\n", NCHAIN, kNames[FILL], c / n, ms * 1e6 / n, (c / n) / (ms * 1e6 / n));
}
template <int FILL> void run(float* out, unsigned long long* cyc, const float* in, int n_iter)
{
    run1<FILL, 2>(out, cyc, in, n_iter);
    if constexpr (FILL + 1 < F_COUNT) run<FILL + 1>(out, cyc, in, n_iter);
}

int main(int argc, char**)
{
    float *out, *in; unsigned long long* cyc;
    CK(hipMalloc(&out, 256 * 256 * 4)); CK(hipMalloc(&in, 4096)); CK(hipMalloc(&cyc, 8));
    float h[1024]; for (int i = 0; i < 1024; ++i) h[i] = 0.001f * (i % 97) + 0.5f;
    CK(hipMemcpy(in, h, 4096, hipMemcpyHostToDevice));
    if (argc > 1) {       // chains: the same fillers with 1, 4 and 8 independent accumulators
        run1<F_NONE, 1>(out, cyc, in, 2000); run1<F_NONE, 4>(out, cyc, in, 2000); run1<F_NONE, 8>(out, cyc, in, 2000);
        run1<F_ADD2, 1>(out, cyc, in, 2000); run1<F_ADD2, 2>(out, cyc, in, 2000); run1<F_ADD2, 4>(out, cyc, in, 2000); run1<F_ADD2, 8>(out, cyc, in, 2000);
        run1<F_ADD1, 4>(out, cyc, in, 2000); run1<F_ADD1, 8>(out, cyc, in, 2000);
        run1<F_EXP1_ADD1, 4>(out, cyc, in, 2000); run1<F_EXP1_ADD1, 8>(out, cyc, in, 2000);
        run1<F_MIX, 4>(out, cyc, in, 2000); run1<F_MIX, 8>(out, cyc, in, 2000);
        return 0;
    }
    run<0>(out, cyc, in, 2000);
    return 0;
}
