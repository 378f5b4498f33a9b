// ff_mlp_ode.hpp -- fused MLP-ODE integrator for gfx950 (MI355X).
//
// One launch integrates  dy/ds = a_e * y + b_e * NET(y, cond; c1_e)  for every sample over
// all evaluation rows e (explicit Runge-Kutta stages or Euler-Maruyama steps) without
// leaving the chip.  It replaces the reference's per-step Python / torchdiffeq loop:
//   RHS                  flowfusion/diffusion.py:258-279 (ode_drift), :281-334 (forward)
//   network              flowfusion/diffusion.py:82-121 (MLP.forward); flow.py:89-120, 553-596
//   stepping             torchdiffeq fixed-grid solvers behind diffusion.py:631-639, 744-752
//                        and flow.py:299-303, 371-382; EM loop diffusion.py:543-562
//
// Mapping (see ff_layout.h): a wavefront owns TILE = 32 (v_mfma_f32_32x32x2) or 16
// (v_mfma_f32_16x16x4) MFMA columns.  In FF_MODE_STATE a column is a sample.  In the divergence modes a sample owns 1 + T adjacent columns: its value
// column and T tangent columns that carry forward-mode derivatives J.v through the same
// weight operands (v = Hutchinson probe, or the D unit vectors for the exact trace); the
// per-column contribution v . (J v) is integrated per lane and reduced once at the end.
// Activations never leave registers: the f32 accumulator tile of layer l (which starts from
// the layer's bias), after the activation, is the B operand of layer l+1.  Weights stream from L2 as the A operand
// (one 16-byte load per lane per 4 MFMAs) through a register ring that runs a fixed number
// of chunks ahead of the MFMAs, across layer and evaluation boundaries.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <utility>
#include "ff_layout.h"
#include "ff_kernel_args.h"
#include "ff_philox.h"

#include "ff_skew.h"

namespace ff {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// RING (template parameter below) = weight chunks kept in flight per wavefront; it must divide
// kChunkPad so that every layer starts at ring slot 0.

template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>)
{
    (f(std::integral_constant<int, I>{}), ...);
}
// compile-time loop: f(std::integral_constant<int, i>) for i in [0, N)
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f)
{
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// Weight / bias / table streams are read through buffer resources: the per-lane part of
// the address is one VGPR for the whole kernel and everything else is scalar (soffset),
// so the loads cost no vector address arithmetic and no 64-bit address registers.
struct Stream {
    __amdgpu_buffer_rsrc_t rsrc;
};
__device__ __forceinline__ Stream make_stream(const float* base, long long floats)
{
    Stream s;
    s.rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)(floats * 4), 0x00020000);
    return s;
}
// 16 bytes at byte offset  soff (wave-uniform) + voff (per lane)
__device__ __forceinline__ f32x4 sload(const Stream& s, int voff, int soff)
{
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(s.rsrc, voff, soff, 0));
}

// MFMA shape traits: one LOGICAL block = 32 feature rows = PHYS accumulator tiles.
template <int TILE>
struct Tile;
template <>
struct Tile<32> {
    static constexpr int NQ = 2, PHYS = 1, RB = 16, SHIFT = 5;
    typedef f32x16 Vec;
    static __device__ __forceinline__ Vec zero()
    {
        return Vec{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    }
    static __device__ __forceinline__ Vec mfma(float a, float b, Vec c)
    {
        return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
    }
};
template <>
struct Tile<16> {
    static constexpr int NQ = 4, PHYS = 2, RB = 8, SHIFT = 4;
    typedef f32x4 Vec;
    static __device__ __forceinline__ Vec zero() { return Vec{0.f, 0.f, 0.f, 0.f}; }
    static __device__ __forceinline__ Vec mfma(float a, float b, Vec c)
    {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
};
// accumulators of one logical block; reg(r) = logical register r in [0, RB)
template <int TILE>
struct BlockAcc {
    typename Tile<TILE>::Vec v[Tile<TILE>::PHYS];
    __device__ __forceinline__ float reg(int r) const
    {
        if constexpr (TILE == 32) return v[0][r];
        else return v[r >> 2][r & 3];
    }
};

// bias rows of one logical block in accumulator-register order (RB/4 x 16 bytes per lane group)
template <int TILE>
struct BiasBlk {
    f32x4 v[Tile<TILE>::RB / 4];
    __device__ __forceinline__ float reg(int r) const { return v[r >> 2][r & 3]; }
};
template <int TILE>
__device__ __forceinline__ BiasBlk<TILE> load_bias(const Stream& s, int q16, int byte_off)
{
    BiasBlk<TILE> b;
#pragma unroll
    for (int j = 0; j < Tile<TILE>::RB / 4; ++j) b.v[j] = sload(s, q16, byte_off + 16 * Tile<TILE>::NQ * j);
    return b;
}

// A bias block as the initial value of the block's accumulator tile(s) (same register order).
template <int TILE>
__device__ __forceinline__ BlockAcc<TILE> load_bias_acc(const Stream& s, int q16, int byte_off)
{
    const BiasBlk<TILE> b = load_bias<TILE>(s, q16, byte_off);
    BlockAcc<TILE> a;
    if constexpr (TILE == 32) {
#pragma unroll
        for (int r = 0; r < 16; ++r) a.v[0][r] = b.reg(r);
    } else {
#pragma unroll
        for (int j = 0; j < Tile<TILE>::PHYS; ++j) a.v[j] = b.v[j];
    }
    return a;
}

// Activation of 4 registers, cut into stages that are issued one MFMA apart, so that no
// instruction waits on the one just before it (a lone SiLU chain  add -> mul -> exp -> add -> rcp ->
// mul  costs ~100 cycles of back-to-back latency; the in-order wavefront would hold the next
// MFMA behind it).  Value columns: P = SiLU(pre).  Tangent columns: P = SiLU'(pre of the
// sample's value column) * pre, with SiLU' fetched across lanes by ds_bpermute (stage 3) and
// consumed one stage later.  pre = accumulator + bias (zero bias on tangent columns).
struct ActGroup {
    float pre[4], t[4], r[4], h[4];
    int dv[4];
};
// Parameters of the activation (FF_ACT_LEAKY_RELU: p0 = negative slope; FF_ACT_ELU: p0 = alpha; FF_ACT_SOFTPLUS:
// p0 = beta, p1 = threshold; include/flowfusion_amd.h).  Wave-uniform.
struct ActSpec {
    float p0, p1;
    float ip0;      // 1 / p0 (softplus divides by beta: one division per kernel instead of one per element)
    int kind;       // FF_ACT_* code, read only by the run-time-choice instantiations (ACT = kActAny)
};

// Activations other than SiLU (template parameter ACT = FF_ACT_* code, compiled in): value h = act(a) and slope
// d = act'(a) of 4 pre-activations, cut into the same five stages as the SiLU path -- argument, exponential,
// reciprocal / logarithm, value and slope, select -- so that no instruction waits on a transcendental issued just
// before it.  Formulas follow torch.nn's definitions; exp / log / rcp are the hardware approximations (1 ulp), erf of
// GELU is Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7).
template <bool TANGENTS, int ACT, int STAGE>
__device__ __forceinline__ void act_stage_kind(ActGroup& g, float* __restrict__ dst, bool is_tangent,
                                               int value_lane_bytes, const ActSpec& s)
{
    constexpr float LOG2E = 1.44269504088896340736f, LN2 = 0.69314718055994530942f;
    constexpr float SQ2PI = 0.79788456080286535588f;
    float d[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float a = g.pre[i];
        d[i] = 0.f;
        if constexpr (ACT == 1) {                      // tanh(a) = 2 sigmoid(2a) - 1
            if constexpr (STAGE == 0) g.t[i] = a * (-2.0f * LOG2E);
            else if constexpr (STAGE == 1) g.t[i] = __builtin_amdgcn_exp2f(g.t[i]);
            else if constexpr (STAGE == 2) g.r[i] = __builtin_amdgcn_rcpf(1.0f + g.t[i]);
            else if constexpr (STAGE == 3) {
                g.h[i] = __builtin_fmaf(2.0f, g.r[i], -1.0f);
                d[i] = __builtin_fmaf(-g.h[i], g.h[i], 1.0f);
            }
        } else if constexpr (ACT == 2) {               // sigmoid
            if constexpr (STAGE == 0) g.t[i] = a * -LOG2E;
            else if constexpr (STAGE == 1) g.t[i] = __builtin_amdgcn_exp2f(g.t[i]);
            else if constexpr (STAGE == 2) g.r[i] = __builtin_amdgcn_rcpf(1.0f + g.t[i]);
            else if constexpr (STAGE == 3) {
                g.h[i] = g.r[i];
                d[i] = g.r[i] * (1.0f - g.r[i]);
            }
        } else if constexpr (ACT == 3) {               // relu
            if constexpr (STAGE == 3) {
                g.h[i] = a > 0.f ? a : 0.f;
                d[i] = a > 0.f ? 1.f : 0.f;
            }
        } else if constexpr (ACT == 4) {               // leaky relu
            if constexpr (STAGE == 3) {
                g.h[i] = a > 0.f ? a : a * s.p0;
                d[i] = a > 0.f ? 1.f : s.p0;
            }
        } else if constexpr (ACT == 5) {               // elu
            if constexpr (STAGE == 0) g.t[i] = fminf(a, 0.f) * LOG2E;
            else if constexpr (STAGE == 1) g.t[i] = __builtin_amdgcn_exp2f(g.t[i]);
            else if constexpr (STAGE == 3) {
                g.h[i] = a > 0.f ? a : s.p0 * (g.t[i] - 1.0f);
                d[i] = a > 0.f ? 1.f : s.p0 * g.t[i];
            }
        } else if constexpr (ACT == 6) {               // softplus: log(1 + exp(beta a)) / beta, linear above the threshold
            if constexpr (STAGE == 0) g.t[i] = -fabsf(a * s.p0) * LOG2E;
            else if constexpr (STAGE == 1) g.t[i] = __builtin_amdgcn_exp2f(g.t[i]);            // e = exp(-|z|)
            else if constexpr (STAGE == 2) {
                g.r[i] = __builtin_amdgcn_rcpf(1.0f + g.t[i]);
                g.h[i] = __builtin_amdgcn_logf(1.0f + g.t[i]);                                  // log2(1 + e)
            } else if constexpr (STAGE == 3) {
                const float z = a * s.p0;
                const float sp = __builtin_fmaf(g.h[i], LN2, fmaxf(z, 0.f));
                const bool lin = z > s.p1;
                g.h[i] = lin ? a : sp * s.ip0;
                d[i] = lin ? 1.f : (z >= 0.f ? g.r[i] : g.t[i] * g.r[i]);
            }
        } else if constexpr (ACT == 7) {               // gelu (erf form): a Phi(a)
            if constexpr (STAGE == 0) {
                const float u = fabsf(a) * 0.70710678118654752440f;
                g.t[i] = -u * u * LOG2E;
                g.r[i] = __builtin_fmaf(0.3275911f, u, 1.0f);
            } else if constexpr (STAGE == 1) {
                g.t[i] = __builtin_amdgcn_exp2f(g.t[i]);                                        // exp(-a^2 / 2)
                g.r[i] = __builtin_amdgcn_rcpf(g.r[i]);
            } else if constexpr (STAGE == 2) {
                const float t = g.r[i];
                float poly = __builtin_fmaf(1.061405429f, t, -1.453152027f);
                poly = __builtin_fmaf(poly, t, 1.421413741f);
                poly = __builtin_fmaf(poly, t, -0.284496736f);
                poly = __builtin_fmaf(poly, t, 0.254829592f);
                g.h[i] = poly * t * g.t[i];                                                      // 1 - erf(u), u >= 0
            } else if constexpr (STAGE == 3) {
                const float phi = a >= 0.f ? 1.0f - 0.5f * g.h[i] : 0.5f * g.h[i];               // Phi(a)
                g.h[i] = a * phi;
                d[i] = __builtin_fmaf(a * 0.39894228040143267794f, g.t[i], phi);
            }
        } else {                                       // gelu, tanh form: a sigmoid(2 w), w = sqrt(2/pi) (a + 0.044715 a^3)
            static_assert(ACT == 8, "unknown activation code");
            if constexpr (STAGE == 0) g.t[i] = SQ2PI * a * __builtin_fmaf(0.044715f, a * a, 1.0f) * (-2.0f * LOG2E);
            else if constexpr (STAGE == 1) g.t[i] = __builtin_amdgcn_exp2f(g.t[i]);
            else if constexpr (STAGE == 2) g.r[i] = __builtin_amdgcn_rcpf(1.0f + g.t[i]);
            else if constexpr (STAGE == 3) {
                const float dw = SQ2PI * __builtin_fmaf(3.0f * 0.044715f, a * a, 1.0f);
                g.h[i] = a * g.r[i];
                d[i] = __builtin_fmaf(2.0f * dw * g.h[i], 1.0f - g.r[i], g.r[i]);
            }
        }
        // common tail: the tangent columns take the slope of their value column
        if constexpr (STAGE == 3) {
            if constexpr (TANGENTS) g.dv[i] = __builtin_amdgcn_ds_bpermute(value_lane_bytes, __builtin_bit_cast(int, d[i]));
            else dst[i] = g.h[i];
        } else if constexpr (STAGE == 4 && TANGENTS) {
            dst[i] = is_tangent ? __builtin_bit_cast(float, g.dv[i]) * a : g.h[i];
        }
    }
}
constexpr int kActAny = 9;      // ACT code of the instantiations that choose the activation at run time (ActSpec::kind)
template <bool TANGENTS, int ACT, int STAGE>
__device__ __forceinline__ void act_stage(ActGroup& g, float* __restrict__ dst, bool is_tangent,
                                          int value_lane_bytes, const ActSpec& spec)
{
    if constexpr (ACT == kActAny) {
        // Run-time choice (one instantiation per mode serves every non-SiLU activation at width 512, where a kernel
        // takes minutes to compile): the chosen kind's stages 0..3 run back to back in stage 3 -- a wave-uniform
        // switch, not latency-tuned -- and the common select follows in stage 4.
        if constexpr (STAGE == 3) {
            auto run = [&](auto kind) __attribute__((always_inline)) {
                constexpr int K = decltype(kind)::value;
                act_stage_kind<TANGENTS, K, 0>(g, dst, is_tangent, value_lane_bytes, spec);
                act_stage_kind<TANGENTS, K, 1>(g, dst, is_tangent, value_lane_bytes, spec);
                act_stage_kind<TANGENTS, K, 2>(g, dst, is_tangent, value_lane_bytes, spec);
                act_stage_kind<TANGENTS, K, 3>(g, dst, is_tangent, value_lane_bytes, spec);
            };
            switch (spec.kind) {
            case 1: run(std::integral_constant<int, 1>{}); break;
            case 2: run(std::integral_constant<int, 2>{}); break;
            case 3: run(std::integral_constant<int, 3>{}); break;
            case 4: run(std::integral_constant<int, 4>{}); break;
            case 5: run(std::integral_constant<int, 5>{}); break;
            case 6: run(std::integral_constant<int, 6>{}); break;
            case 7: run(std::integral_constant<int, 7>{}); break;
            default: run(std::integral_constant<int, 8>{}); break;
            }
        } else if constexpr (STAGE == 4) {
            act_stage_kind<TANGENTS, 8, 4>(g, dst, is_tangent, value_lane_bytes, spec);     // the select is the same for every kind
        }
        return;
    } else if constexpr (ACT != 0) {
        act_stage_kind<TANGENTS, ACT, STAGE>(g, dst, is_tangent, value_lane_bytes, spec);
        return;
    } else {
#ifdef FF_DEBUG_LINEAR_ACT      // timing experiment only: identity activation (wrong results)
    if constexpr (STAGE == 3) {
#pragma unroll
        for (int i = 0; i < 4; ++i) dst[i] = g.pre[i];
    }
    return;
#endif
    // Written as scalar code; what the compiler makes of it is its own business: hipcc's SLP vectoriser packs part of this
    // tail into v_pk_mul_f32 / v_pk_fma_f32 (headline kernel: 138 packed instructions, 38 of them between MFMAs; with
    // -fno-slp-vectorize 92 remain -- the f32x4 stage algebra -- and 2 between MFMAs).  Measured both ways in round 4
    // (scratch/slp_ab.py, profiles/r04/slp_ab.txt): headline 1224.0 vs 1226.5 ms, config 3 2521.4 vs 2520.8 ms, the
    // 128-wide notebook kernels 59.88 vs 59.97 and 199.31 vs 199.33 ms, bitwise the same results -- packing neither costs
    // nor saves here, so the build keeps the compiler's default.
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if constexpr (STAGE == 0) {
            g.t[i] = g.pre[i] * -1.44269504088896340736f;      // exp(-pre) = exp2(-pre * log2 e)
        } else if constexpr (STAGE == 1) {
            g.t[i] = __builtin_amdgcn_exp2f(g.t[i]);
        } else if constexpr (STAGE == 2) {
            g.r[i] = __builtin_amdgcn_rcpf(1.0f + g.t[i]);
        } else if constexpr (STAGE == 3) {
            if constexpr (TANGENTS) {
                // silu'(a) = s + a s (1 - s) = s (1 + a (1 - s)),  s = sigmoid(a): two fmas
                const float w = __builtin_fmaf(-g.pre[i], g.r[i], g.pre[i]);
                const float d = __builtin_fmaf(g.r[i], w, g.r[i]);
#ifdef FF_EXP_DPP       // timing experiment only (Hutchinson pairs: the value column is the lane to the left)
                g.dv[i] = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, d), 0x111, 0xf, 0xf, false);
#else
                g.dv[i] = __builtin_amdgcn_ds_bpermute(value_lane_bytes, __builtin_bit_cast(int, d));
#endif
            } else {
                dst[i] = g.pre[i] * g.r[i];
            }
        } else {
            // value columns: a * s;  tangent columns: a' * silu'(a of the value column)
            if constexpr (TANGENTS) dst[i] = g.pre[i] * (is_tangent ? __builtin_bit_cast(float, g.dv[i]) : g.r[i]);
        }
    }
    }   // SiLU
}
constexpr int kActStages = 5;
constexpr int kTailSlots = 2 * kActStages;   // slots granted after a layer's last MFMA

#ifdef FF_DEBUG_STAMPS
// Diagnostic build: cycle stamps (s_memtime) of wavefront 0, written to a buffer nothing else reads.
__device__ __forceinline__ void ff_stamp(unsigned long long* buf, int& n, bool on)
{
    __builtin_amdgcn_sched_barrier(0);
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    buf[(on && n < 4095) ? n : 4095] = t;      // branch-free: everyone else scribbles on the last slot
    ++n;
}
#endif

// One layer:  acc[ob] = W[ob,:] . B  over the layer's chunk stream, consumed in the order of
// ff_layout.h.  `ring` holds the next RING chunks on entry and on exit (of the following
// layer); chunk c + RING is requested right after chunk c's MFMAs have been issued.  With
// TILE 16 a chunk feeds two accumulator tiles whose MFMAs alternate, which also keeps dependent
// MFMAs (40-cycle latency, 32-cycle issue) from following each other directly.
//
//   sbyte          byte offset of this layer's chunk 0 in the stream (wave-uniform)
//   WRAP           the stream ends with this layer: prefetch wraps to offset 0 (next evaluation)
//   pre_block(ob)  called at the first phase-B chunk of block ob (bias prefetch)
//   slot(M, acc)   called after the M-th MFMA of the layer has been issued, and kTailSlots more
//                  times after the last one: the caller hangs the activation stages of finished
//                  blocks on these slots
//   last(acc)      called once the last block's accumulator is complete
//   LAST_PHYS      physical tiles of the last block that carry rows anybody reads (the output layer of a
//                  <= 16-dimensional state on the 16-row tile needs one of two): the MFMAs of the others
//                  are not issued and their accumulators read as zero
//   acc            the layer's accumulators.  ZERO_INIT: they start from zero (the first MFMA of a block takes a
//                  zero C operand); otherwise they come in holding the layer's bias, which the MFMA chain then
//                  adds for free (the caller loads the next layer's bias into a block once it has been consumed)
template <int TILE, int RING, int KR, int NOB, bool WRAP, bool ZERO_INIT, int LAST_PHYS = Tile<TILE>::PHYS, class PreFn,
          class SlotFn, class LastFn, class DbgFn>
__device__ __forceinline__ void run_layer(f32x4 (&ring)[RING][Tile<TILE>::PHYS], const Stream& ws, int lane16,
                                          int sbyte, const float (&B)[KR], BlockAcc<TILE> (&acc)[NOB],
                                          PreFn&& pre_block, SlotFn&& slot_fn, LastFn&& last, DbgFn&& dbg)
{
    typedef Tile<TILE> T;
    constexpr LayerGeom L = layer_geom(KR, NOB, T::RB / 4);
    constexpr int CB = 1024 * T::PHYS;              // bytes per chunk
    dbg();
    static_for<L.CPAD>([&](auto cc) {
        constexpr int c = decltype(cc)::value;
        constexpr int slot = c % RING;
        if constexpr (c == L.GA * L.NOB && c > 0) dbg();
        if constexpr (c < L.NC) {
            constexpr int g = chunk_group(L, c);
            constexpr int ob = chunk_block(L, c);
            constexpr bool phase_b = c >= L.GA * L.NOB;
            if constexpr (phase_b && (c - L.GA * L.NOB) % L.GB == 0) pre_block(std::integral_constant<int, ob>{});
            static_for<4>([&](auto qq) {
                constexpr int q = decltype(qq)::value;
                static_for<T::PHYS>([&](auto pp) {
                    constexpr int p = decltype(pp)::value;
                    if constexpr (ob == NOB - 1 && p >= LAST_PHYS) {
                        if constexpr (g == 0 && q == 0 && ZERO_INIT) acc[ob].v[p] = T::zero();
                    } else if constexpr (g == 0 && q == 0 && ZERO_INIT)
                        acc[ob].v[p] = T::mfma(ring[slot][p][q], B[4 * g + q], T::zero());
                    else
                        acc[ob].v[p] = T::mfma(ring[slot][p][q], B[4 * g + q], acc[ob].v[p]);
                    // activation stages only ever hang on the head of a layer (the previous layer's
                    // parked block) and on phase B; skipping the call elsewhere saves the compiler
                    // tens of thousands of empty template instantiations
                    constexpr int M = (4 * c + q) * T::PHYS + p;
                    if constexpr (phase_b || M < (16 + kActStages) * T::PHYS) slot_fn(std::integral_constant<int, M>{}, acc);
                });
            });
            if constexpr (c == L.NC - 1) {
                dbg();
                last(acc[NOB - 1]);
                static_for<kTailSlots>([&](auto tt) {
                    slot_fn(std::integral_constant<int, 4 * L.NC * T::PHYS + decltype(tt)::value>{}, acc);
                });
                dbg();
            }
        }
        constexpr int nxt = c + RING;
#ifndef FF_DEBUG_NO_WLOADS      // timing experiment only: never refill the ring (wrong results)
        static_for<T::PHYS>([&](auto pp) {
            constexpr int p = decltype(pp)::value;
            if constexpr (WRAP && nxt >= L.CPAD)
                ring[slot][p] = sload(ws, lane16, (nxt - L.CPAD) * CB + p * 1024);
            else
                ring[slot][p] = sload(ws, lane16, sbyte + nxt * CB + p * 1024);
        });
#endif
        // Pin the stream order: MFMAs and vector-memory loads may not be scheduled across this
        // point (VALU / SALU / transcendental / DS work of the activations may), so every load
        // is issued exactly one ring length ahead of its use.
        __builtin_amdgcn_sched_barrier(0x2 | 0x4 | 0x400 | 0x80);
    });
}

template <int TILE, int KR, int NOB>
struct GeomTag {
    static constexpr LayerGeom value = layer_geom(KR, NOB, Tile<TILE>::RB / 4);
};

// Activation schedule.  A logical block has GPB = RB/4 groups of 4 registers and spans
// SPB = 4*GB*PHYS MFMA slots in phase B.  Group gi of block blk (< NOB-1) starts its stages at slot
//     pb + (blk+1)*SPB + gi*(SPB/GPB)          (pb = first phase-B slot)
// i.e. once block blk+1 has started and its own accumulator is complete; stage k follows k*PHYS
// slots (one 64-cycle MFMA time) later.  Returns the flat group id blk*GPB + gi whose stage k falls
// on slot M, or -1.
FF_HD constexpr int act_group_at(const LayerGeom& L, int phys, int gpb, int M, int k)
{
    const int spb = 4 * L.GB * phys, step = spb / gpb;
    const int s0 = M - 4 * L.GA * L.NOB * phys - k * phys;
    if (s0 < spb || s0 % step != 0) return -1;
    const int q = s0 / step;
    const int blk = q / gpb - 1, gi = q % gpb;
    return blk <= L.NOB - 2 ? blk * gpb + gi : -1;
}

// COOP (small batches): the four wavefronts of a workgroup share ONE tile of TILE columns and split every hidden layer's
// output rows between them (NB / 4 logical blocks each), exchanging the activations through LDS after each layer --
// one evaluation then takes about a third of a lone wavefront's time, which is what counts when the batch is too small
// to give every SIMD a tile of its own.  Same packed weights, same fp32 FMA chains in the same order (results equal the
// one-wavefront kernel's bit for bit); state, stage slots and bookkeeping are replicated in the four wavefronts and
// wavefront 0 writes the outputs.  The launcher picks the twin by batch size (ff_api.cpp).
//
// WIDE (implies COOP; networks wider than 512 or with more than 64 dimensions / 32 conditional inputs, at ANY batch
// size): the operand vector of a hidden layer no longer fits a wavefront's registers beside its accumulators, so the
// B operands are read from the LDS exchange buffer group by group (one ds_read_b128 per 4 operand registers, one group
// ahead of the MFMAs that use it) instead of all at once after the exchange; one exchange buffer and a second barrier
// per layer (everybody has finished reading before anybody overwrites).  Same packed layout and FMA chains.
template <int TILE, int H, int DREGS, int CREGS, bool TANGENTS, int WPS, int RING, int ACT = 0, bool COOP = false,
          bool WIDE = false>
__global__ __launch_bounds__(256, WPS) void mlp_ode_kernel(const KernelArgs args)
{
    static_assert(!WIDE || COOP, "the wide variant is a cooperative kernel");
    static_assert(kChunkPad % RING == 0, "ring must divide the chunk padding");
    static_assert(!COOP || (H / 32) % 4 == 0, "the cooperative twin splits the blocks of a layer four ways");
    typedef Tile<TILE> T;
    constexpr int NB = H / 32;                       // logical blocks per hidden layer
    constexpr int RB = T::RB;                        // registers per logical block
    constexpr int GPB = RB / 4;                      // activation groups per logical block
    constexpr int NOB_OUT = blocks_for_regs(TILE, DREGS);
    constexpr int K1 = DREGS + CREGS;
    constexpr int KH = NB * RB;                      // operand registers of a hidden layer
    constexpr int R4 = DREGS / 4;
    typedef const __attribute__((address_space(4))) RowHdr* HdrPtr;   // scalar (SMEM) loads

    // launches enqueued ahead of a device-side decision (the adaptive driver, ff_adaptive.hip): a cleared gate word makes
    // this launch a no-op -- uniform over the grid, read before any barrier
    if (args.gate && *(const volatile int*)args.gate == 0) return;

    const int lane = threadIdx.x & 63;
    const int qd = lane >> T::SHIFT;                 // lane group (k index of the MFMA operands)
    const int col = lane & (TILE - 1);
    const int lane16 = lane * 16;
    const int q16 = qd * 16;
    // tile index: one per wavefront, or one per workgroup in the cooperative twin
    const long long wave = COOP ? (long long)blockIdx.x : (((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);       // wavefront of the workgroup
    const int D = args.dim;
    const int C = args.cond_dim;
    const ActSpec aspec = {args.act_p0, args.act_p1, (ACT == 6 || ACT == kActAny) ? 1.0f / args.act_p0 : 0.f, args.act_kind};

    // ---- column roles -----------------------------------------------------------------
    long long sample;
    bool is_tangent = false;
    bool col_live = true;        // column carries a sample that must be written back
    int role = 0;                // 0 = value column, j >= 1 = tangent j-1
    int value_lane_bytes = lane * 4;
    if constexpr (TANGENTS) {
        const int ncol = 1 + args.n_tangent;
        const int samples_per_wave = TILE / ncol;
        int s_in_wave = col / ncol;
        role = col - s_in_wave * ncol;
        if (s_in_wave >= samples_per_wave) { s_in_wave = 0; role = 0; col_live = false; }
        is_tangent = role != 0;
        value_lane_bytes = ((qd << T::SHIFT) | (s_in_wave * ncol)) * 4;
        sample = wave * samples_per_wave + s_in_wave;
    } else {
        sample = wave * TILE + col;
    }
    if (sample >= args.batch) { sample = args.batch - 1; col_live = false; }
    const int q16b = (TANGENTS && is_tangent) ? 0x7ffffff0 : q16;     // bias offset: out of range = zero

    // ---- load state (value columns) / tangent vectors -----------------------------------
    float x[DREGS];
#pragma unroll
    for (int r = 0; r < DREGS; ++r) {
        const int d = feat_of_reg(TILE, r, qd);
        float v = 0.f;
        if (d < D) {
            if (!is_tangent) {
                v = args.x_in[sample * D + d];
                if (args.in_shift) v = v - args.in_shift[d];
                if (args.in_scale) v = v / args.in_scale[d];
            } else if (args.unit_tangents) {
                v = (d == args.tangent_first + role - 1) ? 1.0f : 0.0f;
            } else {
                v = args.probe[sample * D + d];
            }
        }
        x[r] = v;
    }
    float cnd[CREGS > 0 ? CREGS : 1];
    if constexpr (CREGS > 0) {
#pragma unroll
        for (int r = 0; r < CREGS; ++r) {
            const int d = feat_of_reg(TILE, r, qd);
            cnd[r] = (d < C && !is_tangent) ? args.cond[sample * C + d] : 0.f;
        }
    }

    // tangent lanes: e.e restricted to this lane's features
    float ee = 0.f;
    if constexpr (TANGENTS) {
#pragma unroll
        for (int r = 0; r < DREGS; ++r) ee = __builtin_fmaf(x[r], x[r], ee);
    }

    // Runge-Kutta stage slots k[s] live in LDS (each lane only ever touches its own words,
    // so no barrier is needed); this keeps 6*DREGS registers free for the weight pipeline.
    extern __shared__ __attribute__((aligned(16))) f32x4 lds_slots[];
    // (cooperative twin: one copy -- the four wavefronts hold the same tile and write the same values)
    f32x4* const ks = lds_slots + (size_t)(COOP ? 0 : (threadIdx.x >> 6)) * kSlots * R4 * 64 + lane;
    f32x4* const exch = lds_slots + (size_t)kSlots * R4 * 64 + lane;         // COOP: 2 (WIDE: 1) x (KH / 4) x 64 exchange slots
    FF_SKEW_HOLD(COOP && wv == kSkewWave, 1);          // (test builds: this wavefront starts late ...)
#pragma unroll
    for (int s = 0; s < kSlots; ++s)
#pragma unroll
        for (int j = 0; j < R4; ++j) ks[(s * R4 + j) * 64] = f32x4{0.f, 0.f, 0.f, 0.f};
    FF_SKEW_HOLD(COOP && wv == kSkewWave, 2);          // (... and lingers between its zero fill and its first store)
    float kl[kSlots];
#pragma unroll
    for (int s = 0; s < kSlots; ++s) kl[s] = 0.f;
    float lp = 0.f;
    // Cooperative twin: the four wavefronts share the slots.  Everything they store there later is the same value from
    // each of them, so late or repeated stores are harmless -- except this zero fill: a wavefront that starts late would
    // wipe the caller's first stage (below) between another wavefront's store and its first read.  All fills first.
#if !defined(FF_DEBUG_UNFIX)
    if constexpr (COOP) __syncthreads();
#endif
    if (args.k1_in) {            // first stage supplied by the caller (FSAL of the previous step)
#pragma unroll
        for (int j = 0; j < R4; ++j) {
            f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int d = feat_of_reg(TILE, 4 * j + i, qd);
                if (d < D && !is_tangent) v[i] = args.k1_in[sample * D + d];
            }
            ks[j * 64] = v;
        }
    }
    FF_SKEW_HOLD(COOP && wv != kSkewWave, 2);          // (test builds: the others wait between that store and their first read)
    if constexpr (TANGENTS) {
        // the per-sample divergence of stage 0 is carried by ONE lane of the sample (first tangent
        // column, lane group 0); all other lanes integrate their own partial sums from zero
        if (args.kl1_in && role == 1 && qd == 0) kl[0] = args.kl1_in[sample];
    }

    const Layout L = make_layout(TILE, H, DREGS, CREGS, args.n_hidden);
    constexpr int CB = 1024 * T::PHYS;               // bytes per chunk
    const Stream ws = make_stream(args.wpack, args.wpack_floats);
    const Stream ts = make_stream(args.etab, (long long)(args.n_evals + (args.n_aux > 0 ? 2 : 0)) * args.etab_stride);
    const int out_sbyte = L.chunk_off_out() * CB;
    const int out_bias_byte = (int)(L.bias_off_out() * 4);

    // ---- cooperative twin: this wavefront's share of a layer --------------------------------------------------------
    // Wavefront wv owns logical blocks [wv * NBW, (wv + 1) * NBW) of every hidden layer and visits its chunks of a layer
    // group-major: visit i = (group i / NM, own block i % NM) -> chunk_index() of ff_layout.h in the SAME packed stream.
    // Every layer's visiting list is padded to a multiple of RING, so a layer always starts at ring slot 0.
    constexpr int NBW = COOP ? NB / 4 : NB;
    const int ob0 = wv * NBW;
    constexpr LayerGeom CG1 = layer_geom(K1, NB, RB / 4), CGH = layer_geom(KH, NB, RB / 4), CGO = layer_geom(KH, NOB_OUT, RB / 4);
    constexpr int CN1 = (CG1.G * NBW + RING - 1) / RING * RING;          // padded visits of layer 1
    auto coop_byte = [&](const LayerGeom& G, int sbyte, int i, int nm, int b0) __attribute__((always_inline)) {
        const int g = i / nm, j = i % nm;
        return sbyte + chunk_index(G, g < G.G ? g : 0, b0 + j) * CB;      // (padding visits re-read a real chunk)
    };

    // prefetch ring: the first RING chunks of layer 1 (of this wavefront's visiting list in the cooperative twin)
    f32x4 ring[RING][T::PHYS];
#pragma unroll
    for (int i = 0; i < RING; ++i)
#pragma unroll
        for (int p = 0; p < T::PHYS; ++p)
            ring[i][p] = sload(ws, lane16, (COOP ? coop_byte(CG1, 0, i, NBW, ob0) : i * CB) + p * 1024);

    float P[WIDE ? 4 : KH];          // operand registers of a hidden layer (WIDE: they stay in LDS)
    // Accumulators of the hidden layers.  They always hold the bias of the layer about to run: a block is
    // refilled with the next layer's bias (straight from the bias stream, in accumulator order) as soon as its
    // pre-activations have been consumed, so the MFMA chain adds the bias and the activation path does not.
    // Tangent lanes fetch through an out-of-range offset (q16b) and start from zero.
    BlockAcc<TILE> hacc[COOP ? 1 : NB];
    if constexpr (!COOP) {
#pragma unroll
        for (int o = 0; o < NB; ++o) hacc[o] = load_bias_acc<TILE>(ts, q16b, 128 + o * 128);
    }
#ifdef FF_DEBUG_STAMPS
    int stamp_n = 0;
    const bool stamp_on = (blockIdx.x == 0 && threadIdx.x == 0 && args.debug_stamps != nullptr);
#endif

    bool bad_slot = false;
    // Cooperative twin: the exchange buffer of the NEXT activation hand-over.  It alternates over the whole launch, not per
    // evaluation: with an odd number of hidden layers a per-evaluation count would end one evaluation and start the next on
    // the same buffer, with a single barrier between a slow wavefront's reads of the old contents and a fast one's stores.
    [[maybe_unused]] int xbuf = 0;
    for (int e = 0; e < args.n_evals; ++e) {
#if defined(FF_DEBUG_UNFIX)
        xbuf = 0;                  // (round 3's second hole, restored for the test: the count restarts with every evaluation)
#endif
        const int row_byte = e * args.etab_stride * 4;
        HdrPtr hdr = (HdrPtr)(args.etab + (size_t)e * args.etab_stride);
        const float a_e = hdr->a, b_e = hdr->b;
        const uint32_t flags = hdr->flags;
        const int slot = hdr->slot;

        // stage input  y = x + sum_s cin[s] * k[s]
        float y[K1];
#pragma unroll
        for (int j = 0; j < R4; ++j) {
            f32x4 v = f32x4{x[4 * j], x[4 * j + 1], x[4 * j + 2], x[4 * j + 3]};
#pragma unroll
            for (int s = 0; s < kSlots; ++s) v += hdr->cin[s] * ks[(s * R4 + j) * 64];
#pragma unroll
            for (int i = 0; i < 4; ++i) y[4 * j + i] = v[i];
        }
        if constexpr (CREGS > 0) {
#pragma unroll
            for (int r = 0; r < CREGS; ++r) y[DREGS + r] = cnd[r];
        }

        // noise for this row (requested early, consumed after the network)
        float nz[DREGS];
        if ((flags & 2u) && args.noise) {
            const float* np = args.noise + (size_t)hdr->noise_idx * args.noise_stride + sample * D;
#pragma unroll
            for (int r = 0; r < DREGS; ++r) {
                const int d = feat_of_reg(TILE, r, qd);
                nz[r] = (d < D) ? np[d] : 0.f;
            }
        }

#ifdef FF_DEBUG_STAMPS
        auto dbg = [&]() { ff_stamp(args.debug_stamps, stamp_n, stamp_on); };
        dbg();
#else
        auto dbg = []() {};
#endif
        float net[NOB_OUT * RB];
        if constexpr (COOP) {
            // ---- cooperative evaluation: NB / 4 blocks of every layer per wavefront, activations exchanged through LDS --
            constexpr int RBQ = RB / 4;
            const int c1_byte = row_byte + 128, c1_next = row_byte + args.etab_stride * 4 + 128;
            // one layer of this wavefront's share: acc[j] += W[block b0 + j, :] . Bop over the layer's groups, in
            // ascending group order (the order of the one-wavefront kernel: same FMA chain per output row).
            // next(k, slot): request visit k of the NEXT layer into ring slot `slot`.
            auto coop_layer = [&](auto tag, const auto& Bop, auto& acc, int sbyte, int b0, auto&& next) __attribute__((always_inline)) {
                constexpr int KIND = decltype(tag)::value;          // 0 = layer 1, 1 = hidden, 2 = output
                constexpr LayerGeom G = KIND == 0 ? CG1 : (KIND == 1 ? CGH : CGO);
                constexpr int NM = KIND == 2 ? NOB_OUT : NBW;
                constexpr int NV = KIND == 0 ? CN1 : G.G * NM;
                static_assert(NV % RING == 0, "visiting lists are multiples of the ring length");
                constexpr int OUT_LAST = (DREGS * T::NQ - (NOB_OUT - 1) * 32 + TILE - 1) / TILE;
                constexpr int LAST_PHYS = KIND == 2 ? (OUT_LAST < T::PHYS ? OUT_LAST : T::PHYS) : T::PHYS;
                // WIDE: the operands of group g come from the exchange buffer, requested one group ahead
                constexpr bool LDS_B = WIDE && KIND != 0;
                f32x4 bq = f32x4{0.f, 0.f, 0.f, 0.f}, bq_next = f32x4{0.f, 0.f, 0.f, 0.f};
                if constexpr (LDS_B) bq_next = exch[0];
                static_for<NV>([&](auto ii) {
                    constexpr int i = decltype(ii)::value;
                    constexpr int slot = i % RING, g = i / NM, j = i % NM;
                    if constexpr (g < G.G) {
                        if constexpr (LDS_B && j == 0) {
                            bq = bq_next;
                            if constexpr (g + 1 < G.G) bq_next = exch[(g + 1) * 64];
                        }
                        static_for<4>([&](auto qq) {
                            constexpr int q = decltype(qq)::value;
                            float bop;
                            if constexpr (LDS_B) bop = bq[q];
                            else bop = Bop[4 * g + q];
                            static_for<T::PHYS>([&](auto pp) {
                                constexpr int p = decltype(pp)::value;
                                if constexpr (KIND == 2 && j == NM - 1 && p >= LAST_PHYS) {
                                    if constexpr (g == 0 && q == 0) acc[j].v[p] = T::zero();
                                } else if constexpr (KIND == 2 && g == 0 && q == 0)
                                    acc[j].v[p] = T::mfma(ring[slot][p][q], bop, T::zero());
                                else
                                    acc[j].v[p] = T::mfma(ring[slot][p][q], bop, acc[j].v[p]);
                            });
                        });
                    }
                    constexpr int nxt = i + RING;
                    if constexpr (nxt < NV) {
                        static_for<T::PHYS>([&](auto pp) {
                            constexpr int p = decltype(pp)::value;
                            ring[slot][p] = sload(ws, lane16, coop_byte(G, sbyte, nxt, NM, b0) + p * 1024);
                        });
                    } else {
                        next(std::integral_constant<int, nxt - NV>{}, std::integral_constant<int, slot>{});
                    }
                    __builtin_amdgcn_sched_barrier(0x2 | 0x4 | 0x400 | 0x80);
                });
            };
            // activate this wavefront's blocks and trade them for everybody else's: P <- all KH operand registers
            auto coop_exchange = [&](const BlockAcc<TILE> (&acc)[NBW], int buf) __attribute__((always_inline)) {
                f32x4* const xb = exch + (size_t)(WIDE ? 0 : buf) * (KH / 4) * 64;
                if constexpr (WIDE) __syncthreads();         // one buffer: the layer that read it has finished everywhere
                static_for<NBW>([&](auto jj) {
                    constexpr int j = decltype(jj)::value;
                    static_for<RBQ>([&](auto rr) {
                        constexpr int r4 = decltype(rr)::value;
                        ActGroup ag;
                        float out[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) ag.pre[i] = acc[j].reg(4 * r4 + i);
                        static_for<kActStages>([&](auto kk) {
                            act_stage<TANGENTS, ACT, decltype(kk)::value>(ag, out, is_tangent, value_lane_bytes, aspec);
                        });
                        xb[((ob0 + j) * RBQ + r4) * 64] = f32x4{out[0], out[1], out[2], out[3]};
                    });
                });
                __syncthreads();
                FF_SKEW_HOLD(wv == kSkewWave, 2);       // (test builds: late to read what the others are about to overwrite)
                if constexpr (!WIDE) {
#pragma unroll
                    for (int k4 = 0; k4 < KH / 4; ++k4) {
                        const f32x4 v = xb[k4 * 64];
#pragma unroll
                        for (int i = 0; i < 4; ++i) P[4 * k4 + i] = v[i];
                    }
                }
            };
            auto next_hidden_or_out = [&](int l_next, auto kk, auto sl) __attribute__((always_inline)) {
                // visit k of the layer after a hidden-side layer: hidden layer l_next, or the output layer
                constexpr int k = decltype(kk)::value, slot = decltype(sl)::value;
                const bool is_hid = l_next < args.n_hidden - 1;
                const int byte = is_hid ? coop_byte(CGH, L.chunk_off_hid(l_next) * CB, k, NBW, ob0)
                                        : coop_byte(CGO, out_sbyte, k, NOB_OUT, 0);
                static_for<T::PHYS>([&](auto pp) {
                    constexpr int p = decltype(pp)::value;
                    ring[slot][p] = sload(ws, lane16, byte + p * 1024);
                });
            };
            BlockAcc<TILE> cacc[NBW];
            // layer 1
#pragma unroll
            for (int j = 0; j < NBW; ++j) cacc[j] = load_bias_acc<TILE>(ts, q16b, c1_byte + (ob0 + j) * 128);
            coop_layer(std::integral_constant<int, 0>{}, y, cacc, 0, ob0,
                       [&](auto kk, auto sl) { next_hidden_or_out(0, kk, sl); });
            for (int l = 0; l < args.n_hidden - 1; ++l) {
                BlockAcc<TILE> nacc[NBW];                            // this layer's bias: requested before the exchange
                const int bbyte = (int)(L.bias_off_hid(l) * 4);
#pragma unroll
                for (int j = 0; j < NBW; ++j) nacc[j] = load_bias_acc<TILE>(ws, q16b, bbyte + (ob0 + j) * 128);
                coop_exchange(cacc, xbuf);
                xbuf ^= 1;
#pragma unroll
                for (int j = 0; j < NBW; ++j) cacc[j] = nacc[j];
                coop_layer(std::integral_constant<int, 1>{}, P, cacc, L.chunk_off_hid(l) * CB, ob0,
                           [&](auto kk, auto sl) { next_hidden_or_out(l + 1, kk, sl); });
            }
            BiasBlk<TILE> obias[NOB_OUT];
#pragma unroll
            for (int o = 0; o < NOB_OUT; ++o) obias[o] = load_bias<TILE>(ws, q16b, out_bias_byte + o * 128);
            coop_exchange(cacc, xbuf);
            xbuf ^= 1;
            // output layer: every wavefront computes all of it (a handful of rows; no exchange, and the same chain as the
            // one-wavefront kernel); the ring moves on to layer 1 of the next evaluation
            BlockAcc<TILE> oacc[NOB_OUT];
            coop_layer(std::integral_constant<int, 2>{}, P, oacc, out_sbyte, 0, [&](auto kk, auto sl) {
                constexpr int k = decltype(kk)::value, slot = decltype(sl)::value;
                static_for<T::PHYS>([&](auto pp) {
                    constexpr int p = decltype(pp)::value;
                    ring[slot][p] = sload(ws, lane16, coop_byte(CG1, 0, k, NBW, ob0) + p * 1024);
                });
            });
#pragma unroll
            for (int o = 0; o < NOB_OUT; ++o)
#pragma unroll
                for (int r = 0; r < RB; ++r) net[o * RB + r] = oacc[o].reg(r) + obias[o].reg(r);
            (void)c1_next;
        } else {
        // `pend` = pre-activations of the previous layer's last block; they
        // are activated into P[(NB-1)*RB ..] behind the first MFMAs of the next layer, whose
        // phase A does not read the last k-block.
        float pend[RB];
        BiasBlk<TILE> bias[2];   // output layer only (tangent lanes read zeros through q16b)
        ActGroup ag[12];         // in-flight groups: [0,4) parked block of the previous layer, 4 + id % 8 own blocks
        // stages of the previous layer's parked block: group gi starts at slot gi*(16*PHYS/GPB)
        auto prev_slot = [&](auto mm) {
            constexpr int M = decltype(mm)::value;
            static_for<kActStages>([&](auto kk) {
                constexpr int k = kActStages - 1 - decltype(kk)::value;     // oldest group first
                constexpr int step = 16 * T::PHYS / GPB;
                constexpr int s0 = M - k * T::PHYS;
                if constexpr (s0 >= 0 && s0 < 16 * T::PHYS && s0 % step == 0) {
                    constexpr int gi = s0 / step;
                    if constexpr (k == 0) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) ag[gi].pre[i] = pend[4 * gi + i];
                    }
                    act_stage<TANGENTS, ACT, k>(ag[gi], &P[(NB - 1) * RB + 4 * gi], is_tangent, value_lane_bytes, aspec);
                }
            });
        };
        // stages of this layer's own blocks 0 .. NB-2 (geometry G), hung on slot M
        auto own_slot = [&](auto geom, auto mm, const BlockAcc<TILE> (&acc)[NB]) {
            constexpr LayerGeom G = decltype(geom)::value;
            constexpr int M = decltype(mm)::value;
            static_for<kActStages>([&](auto kk) {
                constexpr int k = kActStages - 1 - decltype(kk)::value;     // oldest group first
                constexpr int id = act_group_at(G, T::PHYS, GPB, M, k);
                if constexpr (id >= 0) {
                    constexpr int blk = id / GPB, gi = id % GPB;
                    if constexpr (k == 0) {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            ag[4 + id % 8].pre[i] = acc[blk].reg(4 * gi + i);
                    }
                    act_stage<TANGENTS, ACT, k>(ag[4 + id % 8], &P[blk * RB + 4 * gi], is_tangent, value_lane_bytes, aspec);
                }
            });
        };
        // Refill schedule: at the first phase-B chunk of block ob, block ob-2 has been consumed (its groups
        // start their stages while block ob-1 runs); the last two blocks follow at the end of the layer, the
        // last one after its pre-activations have been parked.
        auto refill = [&](const Stream& st, int byte, auto ob) {
            constexpr int o = decltype(ob)::value;
            if constexpr (o >= 2) hacc[o - 2] = load_bias_acc<TILE>(st, q16b, byte + (o - 2) * 128);
        };
        auto park_and_refill = [&](const Stream& st, int byte, const BlockAcc<TILE>& acc) {
#pragma unroll
            for (int r = 0; r < RB; ++r) pend[r] = acc.reg(r);
            if constexpr (NB >= 2) hacc[NB - 2] = load_bias_acc<TILE>(st, q16b, byte + (NB - 2) * 128);
            hacc[NB - 1] = load_bias_acc<TILE>(st, q16b, byte + (NB - 1) * 128);
        };
        // ---- layer 1: [x | cond] -> H, bias c1_e (already in hacc) ------------------------
        {
            using G1 = GeomTag<TILE, K1, NB>;
            const int nbyte = (int)(L.bias_off_hid(0) * 4);          // next: first hidden->hidden layer
            run_layer<TILE, RING, K1, NB, false, false>(
                ring, ws, lane16, 0, y, hacc, [&](auto ob) { refill(ws, nbyte, ob); },
                [&](auto mm, const BlockAcc<TILE> (&acc)[NB]) { own_slot(G1{}, mm, acc); },
                [&](const BlockAcc<TILE>& acc) { park_and_refill(ws, nbyte, acc); }, dbg);
        }
        // ---- hidden -> hidden ------------------------------------------------------------
        for (int l = 0; l < args.n_hidden - 1; ++l) {
            using GH = GeomTag<TILE, KH, NB>;
            const int sbyte = L.chunk_off_hid(l) * CB;
            // next layer's bias; after the last hidden layer nothing reads hacc before the output layer
            // replaces it with the next row's c1, so whatever this fetches then is ignored
            const int nbyte = (int)(L.bias_off_hid(l + 1) * 4);
            run_layer<TILE, RING, KH, NB, false, false>(
                ring, ws, lane16, sbyte, P, hacc, [&](auto ob) { refill(ws, nbyte, ob); },
                [&](auto mm, const BlockAcc<TILE> (&acc)[NB]) {
                    prev_slot(mm);
                    own_slot(GH{}, mm, acc);
                },
                [&](const BlockAcc<TILE>& acc) { park_and_refill(ws, nbyte, acc); }, dbg);
        }
        // ---- output layer ----------------------------------------------------------------
        // the hidden accumulators are idle from here to the next evaluation's first layer: fetch its c1 now
        // (a row past the table reads as zeros)
#pragma unroll
        for (int o = 0; o < NB; ++o) hacc[o] = load_bias_acc<TILE>(ts, q16b, row_byte + args.etab_stride * 4 + 128 + o * 128);
        BlockAcc<TILE> oacc[NOB_OUT];
        constexpr int OUT_LAST_PHYS = (DREGS * T::NQ - (NOB_OUT - 1) * 32 + TILE - 1) / TILE;   // tiles with state rows
        run_layer<TILE, RING, KH, NOB_OUT, true, true, (OUT_LAST_PHYS < T::PHYS ? OUT_LAST_PHYS : T::PHYS)>(
            ring, ws, lane16, out_sbyte, P, oacc,
            [&](auto ob) {
                constexpr int o = decltype(ob)::value;
                bias[o & 1] = load_bias<TILE>(ws, q16b, out_bias_byte + o * 128);
            },
            [&](auto mm, const BlockAcc<TILE> (&acc)[NOB_OUT]) {
                prev_slot(mm);
                if constexpr (NOB_OUT > 1) {          // finished output blocks: plain bias add
                    constexpr LayerGeom GO = layer_geom(KH, NOB_OUT, RB / 4);
                    constexpr int M = decltype(mm)::value;
                    constexpr int id = act_group_at(GO, T::PHYS, GPB, M, 0);
                    if constexpr (id >= 0) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            constexpr int blk = id / GPB, r0 = 4 * (id % GPB);
                            net[blk * RB + r0 + i] = acc[blk].reg(r0 + i) + bias[blk & 1].reg(r0 + i);
                        }
                    }
                }
            },
            [&](const BlockAcc<TILE>& acc) {
#pragma unroll
                for (int r = 0; r < RB; ++r)
                    net[(NOB_OUT - 1) * RB + r] = acc.reg(r) + bias[(NOB_OUT - 1) & 1].reg(r);
            },
            dbg);
        }   // !COOP

        // ---- RHS and stage bookkeeping -----------------------------------------------------
        float rhs[DREGS];
        float div = 0.f;
        if constexpr (TANGENTS) {
            float dot = 0.f;
#pragma unroll
            for (int r = 0; r < DREGS; ++r) dot = __builtin_fmaf(x[r], net[r], dot);
            div = is_tangent ? __builtin_fmaf(a_e, ee, b_e * dot) : 0.f;
            // full Jacobian of the last evaluation's right-hand side (unit tangents): tangent column j holds
            // d rhs / d y_j = a_e e_j + b_e dNET/dy_j; stored as row j of jac_out[sample]
            if (args.jac_out && (args.jac_all || e == args.n_evals - 1) && is_tangent && col_live && (!COOP || wv == 0)) {
                const int tj = args.tangent_first + role - 1;
                float* jp = args.jac_out + (((size_t)(args.jac_all ? e : 0) * args.batch + sample) * D + tj) * D;
#pragma unroll
                for (int r = 0; r < DREGS; ++r) {
                    const int d = feat_of_reg(TILE, r, qd);
                    if (d < D) jp[d] = __builtin_fmaf(b_e, net[r], d == tj ? a_e : 0.f);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < DREGS; ++r) {
            const float v = __builtin_fmaf(a_e, y[r], b_e * net[r]);
            rhs[r] = is_tangent ? 0.f : v;
        }
        // (a row naming a slot beyond the kSlots on chip is refused: nothing is stored, the status word says so)
        const bool slot_ok = (unsigned)slot < (unsigned)kSlots;
        bad_slot |= !slot_ok;
        if (slot_ok) {
#pragma unroll
            for (int j = 0; j < R4; ++j)
                ks[(slot * R4 + j) * 64] = f32x4{rhs[4 * j], rhs[4 * j + 1], rhs[4 * j + 2], rhs[4 * j + 3]};
        }
        if constexpr (TANGENTS) {
#pragma unroll
            for (int s = 0; s < kSlots; ++s) kl[s] = (slot == s) ? div : kl[s];
        }
        if (flags & 1u) {
#pragma unroll
            for (int j = 0; j < R4; ++j) {
                f32x4 v = f32x4{x[4 * j], x[4 * j + 1], x[4 * j + 2], x[4 * j + 3]};
#pragma unroll
                for (int s = 0; s < kSlots; ++s) v += hdr->cout[s] * ks[(s * R4 + j) * 64];
#pragma unroll
                for (int i = 0; i < 4; ++i) x[4 * j + i] = v[i];
            }
            if constexpr (TANGENTS) {
#pragma unroll
                for (int s = 0; s < kSlots; ++s) lp = __builtin_fmaf(hdr->cout[s], kl[s], lp);
            }
        }
        if (flags & 2u) {
            if constexpr (!TANGENTS) {
                if (!args.noise) {      // in-kernel noise: registers 4j..4j+3 of a lane are dimensions 4*blk..4*blk+3
                    const unsigned long long gs = (unsigned long long)(sample + args.rng_sample_offset);
#pragma unroll
                    for (int j = 0; j < R4; ++j) {
                        uint32_t c[4] = {(uint32_t)gs, (uint32_t)(gs >> 32), (uint32_t)(hdr->noise_idx + args.rng_noise_base),
                                         (uint32_t)(feat_of_reg(TILE, 4 * j, qd) >> 2)};
                        philox4x32_10(c, (uint32_t)args.rng_seed, (uint32_t)(args.rng_seed >> 32));
                        box_muller(c[0], c[1], nz[4 * j], nz[4 * j + 1]);
                        box_muller(c[2], c[3], nz[4 * j + 2], nz[4 * j + 3]);
                    }
                }
            }
            const float gn = hdr->gn;
#pragma unroll
            for (int r = 0; r < DREGS; ++r) x[r] = __builtin_fmaf(gn, nz[r], x[r]);
        }
    }

    // ---- epilogue -----------------------------------------------------------------------------
    // sum of a per-lane partial over all lane groups and over the sample's tangent columns
    auto reduce_tangents = [&](float part) {
        float tot = 0.f;
        for (int j = 1; j <= args.n_tangent; ++j) {
#pragma unroll
            for (int g = 0; g < T::NQ; ++g) {
                const int src = ((g << T::SHIFT) | ((col + j) & (TILE - 1))) * 4;
                tot += __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, part)));
            }
        }
        return tot;
    };
    const bool writer = col_live && !is_tangent && (!COOP || wv == 0);
    float lp0 = 0.f;
    if constexpr (TANGENTS) {
        if (args.dlogp_in) lp0 = args.dlogp_in[sample];
        const float tot = reduce_tangents(lp);
        if (writer && qd == 0 && args.dlogp_out) args.dlogp_out[sample] = lp0 + tot;
    }
    // auxiliary outputs: aux_j = use_y_j * y + sum_s coef_j[s] * k[s]  (coefficients in the two
    // rows that follow the evaluation rows)
    if (args.n_aux > 0) {
        HdrPtr t0h = (HdrPtr)(args.etab + (size_t)args.n_evals * args.etab_stride);
        HdrPtr t1h = (HdrPtr)(args.etab + (size_t)(args.n_evals + 1) * args.etab_stride);
        const uint32_t use_y = t0h->flags;
        static_for<kAux>([&](auto jj) {
            constexpr int j = decltype(jj)::value;
            if (j < args.n_aux) {
                HdrPtr th = (j < 2) ? t0h : t1h;
                float c[kSlots];
#pragma unroll
                for (int s = 0; s < kSlots; ++s) c[s] = (j & 1) ? th->cout[s] : th->cin[s];
                const float uy = ((use_y >> j) & 1u) ? 1.f : 0.f;
#pragma unroll
                for (int q4 = 0; q4 < R4; ++q4) {
                    f32x4 v = uy * f32x4{x[4 * q4], x[4 * q4 + 1], x[4 * q4 + 2], x[4 * q4 + 3]};
#pragma unroll
                    for (int s = 0; s < kSlots; ++s) v += c[s] * ks[(s * R4 + q4) * 64];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int d = feat_of_reg(TILE, 4 * q4 + i, qd);
                        if (writer && d < D && args.aux_out[j]) args.aux_out[j][sample * D + d] = v[i];
                    }
                }
                if constexpr (TANGENTS) {
                    float part = 0.f;
#pragma unroll
                    for (int s = 0; s < kSlots; ++s) part = __builtin_fmaf(c[s], kl[s], part);
                    const float tot = reduce_tangents(part);
                    if (writer && qd == 0 && args.aux_lp_out[j]) args.aux_lp_out[j][sample] = uy * lp0 + tot;
                }
            }
        });
    }
    bool bad = false;
    if (writer) {
#pragma unroll
        for (int r = 0; r < DREGS; ++r) {
            const int d = feat_of_reg(TILE, r, qd);
            if (d < D) {
#pragma clang fp contract(off)      // x * scale + shift as two roundings, like the reference's torch expression
                float v = x[r];
                bad |= (v != v);
                if (args.out_scale) v = v * args.out_scale[d];
                if (args.out_shift) v = v + args.out_shift[d];
                args.x_out[sample * D + d] = v;
            }
        }
    }
    if (args.status && __any(bad)) {
        if (lane == 0) atomicOr(args.status, kStatusNaN);
    }
    if (args.status && bad_slot) {
        if (lane == 0) atomicOr(args.status, kStatusBadSlot);
    }
}

} // namespace ff
