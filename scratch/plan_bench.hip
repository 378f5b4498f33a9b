// plan_bench.hip -- which PLACEMENT of the split kernel's activation instructions hides best behind
// v_mfma_f32_16x16x32_bf16?  (scratch, timing only.)  Two activation units (A, B: two pre-activations each -> one word of
// the hi / mid / lo fragments: 23 instructions per unit) per period of 48 MFMAs -- the density of the kernel's hidden
// layers (8 units per 192 MFMAs) -- placed behind the MFMAs by a table; full scheduling barriers pin the stream.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 scratch/plan_bench.hip -o /tmp/plan_bench && /tmp/plan_bench
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
__device__ __forceinline__ float top(float x) { return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, x) & 0xFFFF0000u); }
__device__ __forceinline__ unsigned pack_hi(float a, float b)
{
    return __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, b), __builtin_bit_cast(unsigned, a), 0x07060302u);
}

// the 23 instructions of a unit, in dependency order
enum { R0, R1, S0, S1, E0, E1, A0, A1, C0, C1, V0, V1, T0, T1, M0, M1, PH, U0, U1, PM, L0, L1, PL, NI };
constexpr int kPeriod = 48, kMaxPer = 16;
struct Plan { signed char n[kPeriod]; signed char unit[kPeriod][kMaxPer]; signed char ins[kPeriod][kMaxPer]; };
constexpr void put(Plan& p, int gap, int unit, int ins)
{
    gap %= kPeriod;
    p.unit[gap][p.n[gap]] = (signed char)unit;
    p.ins[gap][p.n[gap]] = (signed char)ins;
    ++p.n[gap];
}
// micro-op j (0..13) of the kernel's current cut: its instructions
constexpr void put_micro(Plan& p, int gap, int u, int j)
{
    constexpr int m[14][3] = {{R0, R1, -1}, {S0, S1, -1}, {E0, -1, -1}, {E1, -1, -1}, {A0, A1, -1}, {C0, -1, -1}, {C1, -1, -1}, {V0, V1, -1},
                              {T0, T1, -1}, {M0, M1, -1}, {PH, U0, -1}, {U1, PM, -1}, {L0, L1, -1}, {PL, -1, -1}};
    for (int k = 0; k < 3; ++k)
        if (m[j][k] >= 0) put(p, gap, u, m[j][k]);
}
constexpr Plan make_plan(int id)
{
    Plan p{};
    if (id == 0) return p;                                            // no activation work at all
    if (id == 1) {                                                    // the kernel today: A on gaps 0,2,..,26, B on 1,3,..,27
        for (int j = 0; j < 14; ++j) { put_micro(p, 2 * j, 0, j); put_micro(p, 2 * j + 1, 1, j); }
    } else if (id == 2) {                                             // one unit after the other, every gap
        for (int j = 0; j < 14; ++j) { put_micro(p, j, 0, j); put_micro(p, 24 + j, 1, j); }
    } else if (id == 3) {                                             // one unit after the other, even gaps only (A 0..26, B 20..46 overlap on 20..26: shift B to odd)
        for (int j = 0; j < 14; ++j) { put_micro(p, 2 * j, 0, j); put_micro(p, (2 * j + 24) % 48 + ((2 * j + 24) < 28 ? 1 : 0), 1, j); }
    } else if (id == 4) {                                             // fat micro-ops (trans + its neighbours' fp), even gaps, A then B
        // f0 R0 R1 | f1 S0 S1 E0 | f2 E1 A0 | f3 A1 C0 | f4 C1 V0 | f5 V1 T0 M0 | f6 T1 M1 PH U0 | f7 U1 PM L0 | f8 L1 PL  (9 per unit)
        constexpr int f[9][4] = {{R0, R1, -1, -1}, {S0, S1, E0, -1}, {E1, A0, -1, -1}, {A1, C0, -1, -1}, {C1, V0, -1, -1}, {V1, T0, M0, -1},
                                 {T1, M1, PH, U0}, {U1, PM, L0, -1}, {L1, PL, -1, -1}};
        for (int u = 0; u < 2; ++u)
            for (int j = 0; j < 9; ++j)
                for (int k = 0; k < 4; ++k)
                    if (f[j][k] >= 0) put(p, 24 * u + 2 * j, u, f[j][k]);
    } else if (id == 5) {                                             // the same fat micro-ops on every 3rd gap, A and B interleaved
        constexpr int f[9][4] = {{R0, R1, -1, -1}, {S0, S1, E0, -1}, {E1, A0, -1, -1}, {A1, C0, -1, -1}, {C1, V0, -1, -1}, {V1, T0, M0, -1},
                                 {T1, M1, PH, U0}, {U1, PM, L0, -1}, {L1, PL, -1, -1}};
        for (int u = 0; u < 2; ++u)
            for (int j = 0; j < 9; ++j)
                for (int k = 0; k < 4; ++k)
                    if (f[j][k] >= 0) put(p, 3 * j + u + (u ? 27 - 27 : 0), u, f[j][k]);   // A at 0,3,..,24; B at 1,4,..,25
    } else if (id == 6) {                                             // lockstep A + B: fp instructions in bursts of four, everything else one per gap
        int g = 0;
        auto one = [&](int u, int i) { put(p, g, u, i); ++g; };
        auto burst = [&](int i0, int i1) { put(p, g, 0, i0); put(p, g, 0, i1); put(p, g, 1, i0); put(p, g, 1, i1); g += 2; };   // + an empty gap
        one(0, R0); one(0, R1); one(1, R0); one(1, R1);
        burst(S0, S1);
        one(0, E0); one(0, E1); one(1, E0); one(1, E1);
        burst(A0, A1);
        one(0, C0); one(0, C1); one(1, C0); one(1, C1);
        burst(V0, V1);
        put(p, g, 0, T0); put(p, g, 0, T1); ++g; put(p, g, 1, T0); put(p, g, 1, T1); ++g;
        burst(M0, M1);
        put(p, g, 0, PH); put(p, g, 0, U0); ++g; put(p, g, 0, U1); put(p, g, 0, PM); ++g;
        put(p, g, 1, PH); put(p, g, 1, U0); ++g; put(p, g, 1, U1); put(p, g, 1, PM); ++g;
        burst(L0, L1);
        put(p, g, 0, PL); put(p, g, 1, PL); ++g;
    } else if (id == 7) {                                             // everything of a unit in two gaps: [R S E | A C V T M PH U PM L PL] (the list scheduler's extreme)
        for (int u = 0; u < 2; ++u) {
            for (int i = R0; i <= E1; ++i) put(p, 24 * u, u, i);
            for (int i = A0; i <= C1; ++i) put(p, 24 * u + 4, u, i);
            for (int i = V0; i < NI; ++i) put(p, 24 * u + 8, u, i);
        }
    } else if (id == 8) {                                             // fat micro-ops, even gaps, but the two fp of a pair split: at most 1 fp + 1 trans + ints per gap
        // g0 R0 | g1 R1 S0 | g2 S1 E0 | g3 E1 | g4 A0 | g5 A1 C0 | g6 C1 | g7 V0 | g8 V1 T0 | g9 M0 T1 | g10 M1 PH | g11 U0 U1 | g12 L0 PM | g13 L1 | g14 PL
        constexpr int f[15][3] = {{R0, -1, -1}, {R1, S0, -1}, {S1, E0, -1}, {E1, -1, -1}, {A0, -1, -1}, {A1, C0, -1}, {C1, -1, -1}, {V0, -1, -1}, {V1, T0, -1},
                                  {M0, T1, -1}, {M1, PH, -1}, {U0, U1, -1}, {L0, PM, -1}, {L1, -1, -1}, {PL, -1, -1}};
        for (int u = 0; u < 2; ++u)
            for (int j = 0; j < 15; ++j)
                for (int k = 0; k < 3; ++k)
                    if (f[j][k] >= 0) put(p, (24 * u + (u ? 1 : 0) + 2 * j) % 48, u, f[j][k]);      // A even gaps 0..28, B odd gaps 25..(53 -> 5)
    }
    return p;
}
static const char* kPlanNames[] = {
    "no activation instructions", "today: micro-ops of <= 2 instructions, units A / B on alternate gaps (28 busy gaps, 20 idle)",
    "micro-ops on consecutive gaps, A then B (14 busy, 10 idle, twice)", "micro-ops on every 2nd gap, A then B",
    "fat micro-ops (a transcendental with the fp instructions around it), every 2nd gap, A then B",
    "fat micro-ops on every 3rd gap, A / B interleaved", "A + B in lockstep: fp in bursts of four with an idle gap behind, the rest one per gap",
    "all of a unit in three gaps (clumped)", "<= 1 fp + 1 transcendental + ints per gap, every 2nd gap per unit"};
constexpr int kPlans = 9;

template <int PLAN>
struct PlanHolder { static constexpr Plan p = make_plan(PLAN); };

template <int PLAN>
__global__ __launch_bounds__(256, 1) void plan_kernel(float* out, unsigned long long* cycles, int n_iter, const float* in)
{
    const int lane = threadIdx.x & 63;
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    f32x4 tiles[8];                                           // "accumulator tiles of the layer before": live in AGPRs
#pragma unroll
    for (int i = 0; i < 8; ++i) tiles[i] = f32x4{in[lane + i], in[lane + 8 + i], in[lane + 16 + i], in[lane + 24 + i]};
    u32x4 w = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    u32x4 b[3] = {u32x4{(unsigned)lane, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u}, u32x4{1u, 2u, 3u, 4u}, u32x4{5u, 6u, 7u, 8u}};
#ifdef TILE_ACCS
    f32x4 Cc[16][2], P[16][2];
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            Cc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            P[i][j] = f32x4{in[lane + i], in[lane + 8 + i + j], in[lane + 16 + i], in[lane + 24 + i]};
        }
#endif
    u32x4 fr[2][2][3];
#pragma unroll
    for (int i = 0; i < 12; ++i) fr[i / 6][(i / 3) % 2][i % 3] = u32x4{(unsigned)lane + i, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
    u32x4 wpool[6], bpool[2][3];
#pragma unroll
    for (int i = 0; i < 6; ++i) wpool[i] = u32x4{0x3f803f80u + i, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
#pragma unroll
    for (int i = 0; i < 6; ++i) bpool[i / 3][i % 3] = u32x4{(unsigned)lane + i, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
    asm volatile("" : "+v"(wpool[0]), "+v"(wpool[1]), "+v"(wpool[2]), "+v"(wpool[3]), "+v"(wpool[4]), "+v"(wpool[5]));
    asm volatile("" : "+v"(bpool[0][0]), "+v"(bpool[0][1]), "+v"(bpool[0][2]), "+v"(bpool[1][0]), "+v"(bpool[1][1]), "+v"(bpool[1][2]));
    float a0[2] = {0.f, 0.f}, a1[2] = {0.f, 0.f}, t0[2] = {0.f, 0.f}, t1[2] = {0.f, 0.f};
    unsigned long long c0 = 0, c1 = 0;
    for (int it = 0; it < n_iter + 1; ++it) {
        if (it == 1) c0 = __builtin_readcyclecounter();
        sfor<192>([&](auto gg) {
            constexpr int G = decltype(gg)::value;
#ifdef PINGPONG         // the kernel's operand hand-over: the MFMAs of span q read fragment set q & 1, the units write set (q + 1) & 1
            {
                constexpr int pr = (G % 12) >> 1, cb = G & 1, q = (G / 96) & 1;
                constexpr int wp = (pr == 2 || pr == 4) ? 1 : (pr == 5 ? 2 : 0), bp = (pr == 1 || pr == 4) ? 1 : (pr == 3 ? 2 : 0);
                acc[cb] = mm(wpool[wp + 3 * ((G / 12) % 2)], fr[q][cb][bp], acc[cb]);
            }
#elif defined(TILE_ACCS)        // the kernel's accumulator traffic: 16 row tiles x 2 column blocks written (128 registers), another 128 read by the units
            {
                constexpr int pr = (G % 12) >> 1, cb = G & 1, rt = (G / 12) % 16;
                constexpr int wp = (pr == 2 || pr == 4) ? 1 : (pr == 5 ? 2 : 0), bp = (pr == 1 || pr == 4) ? 1 : (pr == 3 ? 2 : 0);
                Cc[rt][cb] = mm(wpool[wp + 3 * ((G / 12) % 2)], bpool[cb][bp], Cc[rt][cb]);
            }
#elif defined(VARY_OPERANDS)    // a new (A, B) register pair for every MFMA, as in the kernel's groups (3 weight parts x 6 operand fragments)
            {
                constexpr int pr = (G % 12) >> 1, cb = G & 1;
                constexpr int wp = (pr == 2 || pr == 4) ? 1 : (pr == 5 ? 2 : 0), bp = (pr == 1 || pr == 4) ? 1 : (pr == 3 ? 2 : 0);
                acc[cb] = mm(wpool[wp + 3 * ((G / 12) % 2)], bpool[cb][bp], acc[cb]);
            }
#else
            acc[G & 1] = mm(w, b[G % 3], acc[G & 1]);
#endif
            constexpr int ph = G % kPeriod;
            sfor<kMaxPer>([&](auto kk) {
                constexpr int K = decltype(kk)::value;
                if constexpr (K < PlanHolder<PLAN>::p.n[ph]) {
                    constexpr int u = PlanHolder<PLAN>::p.unit[ph][K], I = PlanHolder<PLAN>::p.ins[ph][K];
                    constexpr int tile = (G / kPeriod * 2 + u) % 8;
#ifdef TILE_ACCS
                    if constexpr (I == R0) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(a0[u]) : "a"(P[(G / kPeriod * 2 + u) % 16][u][0]));
                    else if constexpr (I == R1) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(a1[u]) : "a"(P[(G / kPeriod * 2 + u) % 16][u][1]));
#else
                    if constexpr (I == R0) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(a0[u]) : "a"(tiles[tile][0]));
                    else if constexpr (I == R1) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(a1[u]) : "a"(tiles[tile][1]));
#endif
                    else if constexpr (I == S0) t0[u] = a0[u] * -1.44269504f;
                    else if constexpr (I == S1) t1[u] = a1[u] * -1.44269504f;
                    else if constexpr (I == E0) t0[u] = __builtin_amdgcn_exp2f(t0[u]);
                    else if constexpr (I == E1) t1[u] = __builtin_amdgcn_exp2f(t1[u]);
                    else if constexpr (I == A0) t0[u] = 1.0f + t0[u];
                    else if constexpr (I == A1) t1[u] = 1.0f + t1[u];
                    else if constexpr (I == C0) t0[u] = __builtin_amdgcn_rcpf(t0[u]);
                    else if constexpr (I == C1) t1[u] = __builtin_amdgcn_rcpf(t1[u]);
                    else if constexpr (I == V0) a0[u] = a0[u] * t0[u];
                    else if constexpr (I == V1) a1[u] = a1[u] * t1[u];
                    else if constexpr (I == T0) t0[u] = top(a0[u]);
                    else if constexpr (I == T1) t1[u] = top(a1[u]);
                    else if constexpr (I == M0) t0[u] = a0[u] - t0[u];
                    else if constexpr (I == M1) t1[u] = a1[u] - t1[u];
                    else if constexpr (I == PH) { b[0][1 + u] = pack_hi(a0[u], a1[u]); bpool[u][0][1] = b[0][1 + u]; fr[((G / 96) + 1) & 1][u][0][(G / kPeriod) % 2 + 1] = b[0][1 + u]; }
                    else if constexpr (I == U0) a0[u] = top(t0[u]);
                    else if constexpr (I == U1) a1[u] = top(t1[u]);
                    else if constexpr (I == PM) { b[1][1 + u] = pack_hi(t0[u], t1[u]); bpool[u][1][1] = b[1][1 + u]; fr[((G / 96) + 1) & 1][u][1][(G / kPeriod) % 2 + 1] = b[1][1 + u]; }
                    else if constexpr (I == L0) a0[u] = t0[u] - a0[u];
                    else if constexpr (I == L1) a1[u] = t1[u] - a1[u];
                    else { b[2][1 + u] = pack_hi(a0[u], a1[u]); bpool[u][2][1] = b[2][1 + u]; fr[((G / 96) + 1) & 1][u][2][(G / kPeriod) % 2 + 1] = b[2][1 + u]; }
                }
            });
            __builtin_amdgcn_sched_barrier(0);
        });
    }
    c1 = __builtin_readcyclecounter();
    float s = acc[0][0] + acc[1][0] + a0[0] + a0[1] + t1[0] + t1[1] + __builtin_bit_cast(float, fr[0][0][0][1] ^ fr[1][1][2][2]);
#ifdef TILE_ACCS
#pragma unroll
    for (int i = 0; i < 16; ++i) s += Cc[i][0][0] + Cc[i][1][1] + P[i][0][2] + P[i][1][3];
#endif
#pragma unroll
    for (int i = 0; i < 8; ++i) s += tiles[i][2];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) cycles[0] = c1 - c0;
}

template <int PLAN> void run(float* out, unsigned long long* cyc, const float* in, int n_iter)
{
    auto k = plan_kernel<PLAN>;
    hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, out, cyc, 8, in);
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, out, cyc, n_iter, in);
    CK(hipDeviceSynchronize());
    unsigned long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
    const double n = 192.0 * n_iter;
    printf("plan %d  %6.2f cycles/MFMA  (+%5.1f cycles per unit)  %s\n", PLAN, c / n, (c / n - 16.25) * 24, kPlanNames[PLAN]);
    if constexpr (PLAN + 1 < kPlans) run<PLAN + 1>(out, cyc, in, n_iter);
}

int main()
{
    float *out, *in; unsigned long long* cyc;
    CK(hipMalloc(&out, 256 * 256 * 4)); CK(hipMalloc(&in, 4096)); CK(hipMalloc(&cyc, 8));
    float h[1024]; for (int i = 0; i < 1024; ++i) h[i] = 0.001f * (i % 97) + 0.5f;
    CK(hipMemcpy(in, h, 4096, hipMemcpyHostToDevice));
    run<0>(out, cyc, in, 2000);
    return 0;
}
