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
// Mapping (see ff_layout.h): a wavefront owns 32 MFMA columns.  In FF_MODE_STATE a column
// is a sample.  In the divergence modes a sample owns 1 + T adjacent columns: its value
// column and T tangent columns that carry forward-mode derivatives J.v through the same
// weight operands (v = Hutchinson probe, or the D unit vectors for the exact trace); the
// per-column contribution v . (J v) is integrated per lane and reduced once at the end.
// Activations never leave registers: the 32x32 f32 accumulator tile of layer l, after
// bias and SiLU, is the B operand of layer l+1.  Weights stream from L2 as the A operand
// (one 16-byte load per lane per 4 MFMAs) through a register ring that runs a fixed number
// of chunks ahead of the MFMAs, across layer and evaluation boundaries.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <utility>
#include "ff_layout.h"
#include "ff_kernel_args.h"

namespace ff {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kRing = 8;    // weight chunks (1 KiB each, 4 VGPRs per lane) kept in flight per wavefront
static_assert(kChunkPad % kRing == 0, "ring must divide the chunk padding");

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

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c)
{
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// sigmoid from the hardware exp2 / rcp units (each within 1 ulp): silu(a) = a*s,
// silu'(a) = s + a*s*(1-s).
__device__ __forceinline__ float sigmoidf_fast(float a)
{
    return __builtin_amdgcn_rcpf(1.0f + __expf(-a));
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

// 32 bias rows of one output block in accumulator-register order (4 x 16 bytes per lane half)
struct Bias16 {
    f32x4 v[4];
};
__device__ __forceinline__ Bias16 load_bias(const Stream& s, int half16, int byte_off)
{
    Bias16 b;
#pragma unroll
    for (int j = 0; j < 4; ++j) b.v[j] = sload(s, half16, byte_off + 32 * j);
    return b;
}

// One activation unit: SiLU(pre) on value columns; SiLU'(pre of the value column) * pre on
// tangent columns (pre = accumulator + bias; tangent columns carry zero bias).
template <bool TANGENTS>
__device__ __forceinline__ float act_unit(float pre, bool is_tangent, int value_lane_bytes)
{
    const float s = sigmoidf_fast(pre);
    const float h = pre * s;
    if constexpr (TANGENTS) {
        const float d = __builtin_fmaf(h, 1.0f - s, s);
        const float dv =
            __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(value_lane_bytes, __builtin_bit_cast(int, d)));
        return is_tangent ? dv * pre : h;
    } else {
        return h;
    }
}

// One layer:  acc[ob] = W[ob,:] . B  over the layer's chunk stream, consumed in the order of
// ff_layout.h.  `ring` holds the next kRing chunks on entry and on exit (of the following
// layer); chunk c + kRing is requested right after chunk c's four MFMAs have been issued.
//
// The VALU work of the activations is spread one register per MFMA so that it runs in the
// shadow of the matrix pipe:
//   * while block ob (>= 1) runs its 16 phase-B MFMAs, `unit(ob-1, i, acc)` finishes register i
//     of block ob-1;
//   * the LAST block has no MFMAs behind it in this layer: `last(acc)` only parks it, and the
//     next layer runs `prev_unit(i)` behind its first 16 MFMAs (its phase A does not read the
//     last k-block, so those registers are not needed before).
//   sbyte      byte offset of this layer's chunk 0 in the stream (wave-uniform)
//   WRAP       the stream ends with this layer: prefetch wraps to offset 0 (next evaluation)
//   pre_block(ob)  called at the first phase-B chunk of block ob (bias prefetch)
template <int KR, int NOB, bool WRAP, bool PREV, class PrevFn, class PreFn, class UnitFn, class LastFn>
__device__ __forceinline__ void run_layer(f32x4 (&ring)[kRing], const Stream& ws, int lane16, int sbyte,
                                          const float (&B)[KR], PrevFn&& prev_unit, PreFn&& pre_block,
                                          UnitFn&& unit, LastFn&& last)
{
    constexpr LayerGeom L = layer_geom(KR, NOB);
    static_assert(!PREV || L.GA * L.NOB * 4 >= 16, "deferred activation needs 16 phase-A MFMAs");
    f32x16 acc[NOB];
    static_for<L.CPAD>([&](auto cc) {
        constexpr int c = decltype(cc)::value;
        constexpr int slot = c % kRing;
        if constexpr (c < L.NC) {
            constexpr int g = chunk_group(L, c);
            constexpr int ob = chunk_block(L, c);
            constexpr bool phase_b = c >= L.GA * L.NOB;
            constexpr int cb = phase_b ? (c - L.GA * L.NOB) % L.GB : 0;     // chunk within the block's tail
            if constexpr (phase_b && cb == 0) pre_block(std::integral_constant<int, ob>{});
            const f32x4 a = ring[slot];
            static_for<4>([&](auto qq) {
                constexpr int q = decltype(qq)::value;
                if constexpr (g == 0 && q == 0)
                    acc[ob] = mfma32(a[q], B[4 * g + q], f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f,
                                                                 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f});
                else
                    acc[ob] = mfma32(a[q], B[4 * g + q], acc[ob]);
                constexpr int mi = 4 * c + q;                                  // MFMA index in the layer
                if constexpr (PREV && mi < 16) prev_unit(std::integral_constant<int, mi>{});
                if constexpr (phase_b && ob >= 1) {
                    // the 16 registers of block ob-1 are spread over the 4*GB MFMAs of block ob
                    constexpr int nm = 4 * L.GB, m = 4 * cb + q;
                    constexpr int lo = (16 * m + nm - 1) / nm, hi = (16 * (m + 1) + nm - 1) / nm;
                    static_for<hi - lo>([&](auto uu) {
                        constexpr int i = lo + decltype(uu)::value;
                        unit(std::integral_constant<int, ob - 1>{}, std::integral_constant<int, i>{}, acc[ob - 1]);
                    });
                }
            });
            if constexpr (c == L.NC - 1) last(acc[NOB - 1]);
        }
        constexpr int nxt = c + kRing;
        if constexpr (WRAP && nxt >= L.CPAD)
            ring[slot] = sload(ws, lane16, (nxt - L.CPAD) * 1024);
        else
            ring[slot] = sload(ws, lane16, sbyte + nxt * 1024);
        // Pin the stream order: MFMAs and vector-memory loads may not be scheduled across this
        // point (VALU / SALU / transcendental / DS work of the activations may), so every load
        // is issued exactly one ring length ahead of its use.
        __builtin_amdgcn_sched_barrier(0x2 | 0x4 | 0x400 | 0x80);
    });
}

template <int H, int DREGS, int CREGS, bool TANGENTS>
__global__ __launch_bounds__(256, 1) void mlp_ode_kernel(const KernelArgs args)
{
    constexpr int NB = H / 32;
    constexpr int NOB_OUT = (DREGS + 15) / 16;
    constexpr int K1 = DREGS + CREGS;
    constexpr int R4 = DREGS / 4;
    typedef const __attribute__((address_space(4))) RowHdr* HdrPtr;   // scalar (SMEM) loads

    const int lane = threadIdx.x & 63;
    const int half = lane >> 5;
    const int col = lane & 31;
    const int lane16 = lane * 16;
    const int half16 = half * 16;
    const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int D = args.dim;
    const int C = args.cond_dim;

    // ---- column roles -----------------------------------------------------------------
    long long sample;
    bool is_tangent = false;
    bool col_live = true;        // column carries a sample that must be written back
    int role = 0;                // 0 = value column, j >= 1 = tangent j-1
    int value_lane_bytes = lane * 4;
    if constexpr (TANGENTS) {
        const int ncol = 1 + args.n_tangent;
        const int samples_per_wave = 32 / ncol;
        int s_in_wave = col / ncol;
        role = col - s_in_wave * ncol;
        if (s_in_wave >= samples_per_wave) { s_in_wave = 0; role = 0; col_live = false; }
        is_tangent = role != 0;
        value_lane_bytes = ((half << 5) | (s_in_wave * ncol)) * 4;
        sample = wave * samples_per_wave + s_in_wave;
    } else {
        sample = wave * 32 + col;
    }
    if (sample >= args.batch) { sample = args.batch - 1; col_live = false; }

    // ---- load state (value columns) / tangent vectors -----------------------------------
    float x[DREGS];
#pragma unroll
    for (int r = 0; r < DREGS; ++r) {
        const int d = feat_of_reg(r, half);
        float v = 0.f;
        if (d < D) {
            if (!is_tangent) {
                v = args.x_in[sample * D + d];
                if (args.in_shift) v = v - args.in_shift[d];
                if (args.in_scale) v = v / args.in_scale[d];
            } else if (args.unit_tangents) {
                v = (d == role - 1) ? 1.0f : 0.0f;
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
            const int d = feat_of_reg(r, half);
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
    f32x4* const ks = lds_slots + (size_t)(threadIdx.x >> 6) * kSlots * R4 * 64 + lane;
#pragma unroll
    for (int s = 0; s < kSlots; ++s)
#pragma unroll
        for (int j = 0; j < R4; ++j) ks[(s * R4 + j) * 64] = f32x4{0.f, 0.f, 0.f, 0.f};
    float kl[kSlots];
#pragma unroll
    for (int s = 0; s < kSlots; ++s) kl[s] = 0.f;
    float lp = 0.f;

    const Layout L = make_layout(H, DREGS, CREGS, args.n_hidden);
    const Stream ws = make_stream(args.wpack, args.wpack_floats);
    const Stream ts = make_stream(args.etab, (long long)args.n_evals * args.etab_stride);
    const int out_sbyte = L.chunk_off_out() * 1024;
    const int out_bias_byte = (int)(L.bias_off_out() * 4);

    // prefetch ring: the first kRing chunks of layer 1
    f32x4 ring[kRing];
#pragma unroll
    for (int i = 0; i < kRing; ++i) ring[i] = sload(ws, lane16, i * 1024);

    float P[NB * 16];

    for (int e = 0; e < args.n_evals; ++e) {
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
        if (flags & 2u) {
            const float* np = args.noise + (size_t)hdr->noise_idx * args.noise_stride + sample * D;
#pragma unroll
            for (int r = 0; r < DREGS; ++r) {
                const int d = feat_of_reg(r, half);
                nz[r] = (d < D) ? np[d] : 0.f;
            }
        }

        // `pend` = pre-activations (accumulator + bias) of the previous layer's last block; they
        // are activated into P[(NB-1)*16 ..] behind the first MFMAs of the next layer.
        float pend[16];
        Bias16 bias[2];
        auto bias_reg = [&](const Bias16& b, int r) {
            const float v = b.v[r >> 2][r & 3];
            return TANGENTS ? (is_tangent ? 0.f : v) : v;
        };
        auto prev_unit = [&](auto ii) {
            constexpr int i = decltype(ii)::value;
            P[(NB - 1) * 16 + i] = act_unit<TANGENTS>(pend[i], is_tangent, value_lane_bytes);
        };
        auto unit = [&](auto ob, auto ii, const f32x16& acc) {
            constexpr int o = decltype(ob)::value;
            constexpr int i = decltype(ii)::value;
            P[o * 16 + i] = act_unit<TANGENTS>(acc[i] + bias_reg(bias[o & 1], i), is_tangent, value_lane_bytes);
        };
        auto park_last = [&](const f32x16& acc) {
#pragma unroll
            for (int r = 0; r < 16; ++r) pend[r] = acc[r] + bias_reg(bias[(NB - 1) & 1], r);
        };
        // ---- layer 1: [x | cond] -> H, bias c1_e ---------------------------------------
        run_layer<K1, NB, false, false>(
            ring, ws, lane16, 0, y, [](auto) {},
            [&](auto ob) {
                constexpr int o = decltype(ob)::value;
                bias[o & 1] = load_bias(ts, half16, row_byte + 128 + o * 128);
            },
            unit, park_last);
        // ---- hidden -> hidden ------------------------------------------------------------
        for (int l = 0; l < args.n_hidden - 1; ++l) {
            const int sbyte = L.chunk_off_hid(l) * 1024;
            const int bbyte = (int)(L.bias_off_hid(l) * 4);
            run_layer<NB * 16, NB, false, true>(
                ring, ws, lane16, sbyte, P, prev_unit,
                [&](auto ob) {
                    constexpr int o = decltype(ob)::value;
                    bias[o & 1] = load_bias(ws, half16, bbyte + o * 128);
                },
                unit, park_last);
        }
        // ---- output layer ----------------------------------------------------------------
        float net[NOB_OUT * 16];
        run_layer<NB * 16, NOB_OUT, true, true>(
            ring, ws, lane16, out_sbyte, P, prev_unit,
            [&](auto ob) {
                constexpr int o = decltype(ob)::value;
                bias[o & 1] = load_bias(ws, half16, out_bias_byte + o * 128);
            },
            [&](auto ob, auto ii, const f32x16& acc) {
                constexpr int o = decltype(ob)::value;
                constexpr int i = decltype(ii)::value;
                net[o * 16 + i] = acc[i] + bias_reg(bias[o & 1], i);
            },
            [&](const f32x16& acc) {
#pragma unroll
                for (int r = 0; r < 16; ++r) net[(NOB_OUT - 1) * 16 + r] = acc[r] + bias_reg(bias[(NOB_OUT - 1) & 1], r);
            });

        // ---- RHS and stage bookkeeping -----------------------------------------------------
        float rhs[DREGS];
        float div = 0.f;
        if constexpr (TANGENTS) {
            float dot = 0.f;
#pragma unroll
            for (int r = 0; r < DREGS; ++r) dot = __builtin_fmaf(x[r], net[r], dot);
            div = is_tangent ? __builtin_fmaf(a_e, ee, b_e * dot) : 0.f;
        }
#pragma unroll
        for (int r = 0; r < DREGS; ++r) {
            const float v = __builtin_fmaf(a_e, y[r], b_e * net[r]);
            rhs[r] = is_tangent ? 0.f : v;
        }
#pragma unroll
        for (int j = 0; j < R4; ++j)
            ks[(slot * R4 + j) * 64] = f32x4{rhs[4 * j], rhs[4 * j + 1], rhs[4 * j + 2], rhs[4 * j + 3]};
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
            const float gn = hdr->gn;
#pragma unroll
            for (int r = 0; r < DREGS; ++r) x[r] = __builtin_fmaf(gn, nz[r], x[r]);
        }
    }

    // ---- epilogue -----------------------------------------------------------------------------
    if constexpr (TANGENTS) {
        // sum the per-lane integrals over both lane halves and over the sample's tangent columns
        float tot = 0.f;
        for (int j = 1; j <= args.n_tangent; ++j) {
            const int src_lo = ((lane + j) & 63) * 4;
            const int src_hi = (((lane ^ 32) + j) & 63) * 4;
            tot += __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src_lo, __builtin_bit_cast(int, lp)));
            tot += __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src_hi, __builtin_bit_cast(int, lp)));
        }
        if (col_live && !is_tangent && half == 0 && args.dlogp_out) args.dlogp_out[sample] = tot;
    }
    bool bad = false;
    if (col_live && !is_tangent) {
#pragma unroll
        for (int r = 0; r < DREGS; ++r) {
            const int d = feat_of_reg(r, half);
            if (d < D) {
                float v = x[r];
                bad |= (v != v);
                if (args.out_scale) v = v * args.out_scale[d];
                if (args.out_shift) v = v + args.out_shift[d];
                args.x_out[sample * D + d] = v;
            }
        }
    }
    if (args.status && __any(bad)) {
        if (lane == 0) atomicOr(args.status, 1u);
    }
}

} // namespace ff
