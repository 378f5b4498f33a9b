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
// bias (folded into the accumulator's initial value) and SiLU, is the B operand of layer
// l+1.  Weights stream from L2 as the A operand, one 16-byte load per lane per 4 MFMAs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "ff_layout.h"
#include "ff_kernel_args.h"

namespace ff {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

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

// acc[ob] += W[ob, :] . breg over KR operand registers.  The A stream is consumed in groups
// of NOB x 16-byte loads (4 k-steps each); group g+1 is requested before the 4*NOB MFMAs of
// group g are issued, and the compiler is fenced from hoisting loads any further ahead
// (it would otherwise fill all 512 registers with prefetched weights and spill the state).
template <int KR, int NOB>
__device__ __forceinline__ void gemm_kouter(f32x16 (&acc)[NOB], const Stream& ws, int lane16, int wbyte,
                                            const float (&breg)[KR])
{
    constexpr int G = KR / 4;
    f32x4 A[2][NOB];
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob) A[0][ob] = sload(ws, lane16, wbyte + ob * 1024);
#pragma unroll
    for (int g = 0; g < G; ++g) {
        if (g + 1 < G) {
#pragma unroll
            for (int ob = 0; ob < NOB; ++ob)
                A[(g + 1) & 1][ob] = sload(ws, lane16, wbyte + ((g + 1) * NOB + ob) * 1024);
        }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob) {
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[ob] = mfma32(A[g & 1][ob][q], breg[g * 4 + q], acc[ob]);
        }
    }
}

// acc[ob] <- bias rows of block ob in accumulator order (zero on tangent columns).
template <int NOB>
__device__ __forceinline__ void init_acc(f32x16 (&acc)[NOB], const Stream& bs, int half16, int bbyte,
                                         bool zero)
{
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x4 b = sload(bs, half16, bbyte + (ob * 32 + 8 * j) * 4);
            if (zero) b = f32x4{0.f, 0.f, 0.f, 0.f};
            acc[ob][4 * j + 0] = b[0];
            acc[ob][4 * j + 1] = b[1];
            acc[ob][4 * j + 2] = b[2];
            acc[ob][4 * j + 3] = b[3];
        }
    }
}

// P <- SiLU(acc) on value columns;  P <- SiLU'(a_value) * acc on tangent columns.
template <int NB, bool TANGENTS>
__device__ __forceinline__ void activate(float (&P)[NB * 16], const f32x16 (&acc)[NB], bool is_tangent,
                                         int value_lane_bytes)
{
#pragma unroll
    for (int ob = 0; ob < NB; ++ob) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float a = acc[ob][r];
            const float s = sigmoidf_fast(a);
            const float h = a * s;
            if constexpr (TANGENTS) {
                const float d = __builtin_fmaf(h, 1.0f - s, s);
                const float dv = __builtin_bit_cast(
                    float, __builtin_amdgcn_ds_bpermute(value_lane_bytes, __builtin_bit_cast(int, d)));
                P[ob * 16 + r] = is_tangent ? dv * a : h;
            } else {
                P[ob * 16 + r] = h;
            }
        }
    }
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
    const int out_wbyte = (int)(L.off_out() * 4);

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

        // ---- layer 1: [x | cond] -> H, bias c1_e ---------------------------------------
        {
            f32x16 acc[NB];
            init_acc<NB>(acc, ts, half16, row_byte + 128, is_tangent);
            gemm_kouter<K1, NB>(acc, ws, lane16, 0, y);
            activate<NB, TANGENTS>(P, acc, is_tangent, value_lane_bytes);
        }
        // ---- hidden -> hidden ------------------------------------------------------------
        for (int l = 0; l < args.n_hidden - 1; ++l) {
            const int wbyte = (int)(L.off_hid(l) * 4);
            f32x16 acc[NB];
            init_acc<NB>(acc, ws, half16, wbyte + (int)(L.hid_w_floats * 4), is_tangent);
            gemm_kouter<NB * 16, NB>(acc, ws, lane16, wbyte, P);
            activate<NB, TANGENTS>(P, acc, is_tangent, value_lane_bytes);
        }
        // ---- output layer ----------------------------------------------------------------
        float net[NOB_OUT * 16];
        {
            f32x16 acc[NOB_OUT];
            init_acc<NOB_OUT>(acc, ws, half16, out_wbyte + (int)(L.out_w_floats * 4), is_tangent);
            gemm_kouter<NB * 16, NOB_OUT>(acc, ws, lane16, out_wbyte, P);
#pragma unroll
            for (int ob = 0; ob < NOB_OUT; ++ob)
#pragma unroll
                for (int r = 0; r < 16; ++r) net[ob * 16 + r] = acc[ob][r];
        }

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
