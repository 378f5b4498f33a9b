// ff_mlp_ode_split.hpp -- fused MLP-ODE integrator on the bf16 matrix cores with fp32-class accuracy (gfx950).
//
// Same contract as ff_mlp_ode.hpp (one launch integrates dy/ds = a_e y + b_e NET(y, cond; c1_e) over all evaluation
// rows; reference call sites flowfusion/diffusion.py:631-639, 744-752, RHS :258-334, network :82-121), different
// arithmetic (FF_PREC_BF16X3, opt-in): every fp32 operand of a Linear layer is cut into three bf16 parts
//     v = hi + mid + lo      hi = top 16 bits of v, mid = top 16 bits of (v - hi), lo = top 16 bits of the rest
// (24 significand bits in all: exact for the weights, which are split on the host), and a product is the sum of the six
// v_mfma_f32_32x32x16_bf16 whose parts' magnitudes reach 2^-16 of the full product,
//     w.x ~= wh.xh + wh.xm + wm.xh + wh.xl + wm.xm + wl.xh          (dropped terms: <= 2^-24 relative)
// accumulated in fp32 by the MFMA.  Error of a layer ~1e-7 relative to sum |w||x| -- what an fp32 dot product has --
// at 16/6 = 2.7x the MFMA rate of v_mfma_f32_32x32x2_f32.
//
// Mapping.  A wavefront owns 32 MFMA columns (32 samples; in Hutchinson mode 16 samples x (value, tangent)), a
// workgroup is 4 wavefronts, one per SIMD (the kernel needs ~400 of the 512 registers).  Features sit on MFMA rows:
// register i of an accumulator tile holds row (i & 3) + 8 (i >> 2) + 4 h on lane half h = lane >> 5, and the eight
// registers 8u .. 8u+7 of a tile are, element for element, the k-slice the B operand of k-step 2 tile + u wants on that
// lane half -- so after SiLU and the split an accumulator tile IS two k-steps of the next layer's B fragments and
// activations never leave registers.  (The weights are packed on the host in the matching k order, kidx() below.)
//
// Loop order: k-major.  A layer keeps ALL its NT = H/32 accumulator tiles live (NT x 16 registers) and walks the
// k-steps in pairs: pair p needs exactly the previous layer's tile p, whose SiLU + split (VALU, ~55 instructions per
// register) is issued one pair ahead, in the shadow of the 12 NT MFMAs of the pair before (a bf16 MFMA holds the
// vector issue port for a quarter of its duration).  Two sets of NT tiles ping-pong between consecutive layers; a tile
// is refilled with the bias of its next use (from LDS) as soon as it has been consumed, so the MFMA chain adds the bias.
//
// Weights: every wavefront needs every fragment, so they are staged through LDS and shared by the workgroup.  The
// fragments of one evaluation form a linear stream of 24 KiB granules (8 groups x [hi, mid, lo] x 1 KiB; one group =
// one (row tile, k-step) = 6 MFMAs) in consumption order, periodic over evaluations.  Three LDS buffers: at the start
// of granule g each wavefront starts the LDS-DMA (global_load_lds_dwordx4, no registers) of its quarter of granule
// g+2; before the last group of granule g a counted wait (vmcnt(6): all but the DMAs just issued) and ONE barrier
// publish granule g+1 and retire granule g-1's buffer.  The per-evaluation first-layer bias c1_e travels the same way.
//
// Scope of this family: width 256 (NT = 8), dim <= 16, cond <= 16, SiLU, FF_MODE_STATE and FF_MODE_HUTCH, any
// fixed-grid table (<= 7 stage slots, kept in LDS); no noise rows, no adaptive-step outputs, no Jacobian output.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <utility>
#include "ff_layout.h"
#include "ff_kernel_args.h"
#include "ff_split_layout.h"

namespace ff {
namespace split {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <class F, int... I>
__device__ __forceinline__ void sfor_impl(F&& f, std::integer_sequence<int, I...>)
{
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void sfor(F&& f)
{
    sfor_impl(f, std::make_integer_sequence<int, N>{});
}

__device__ __forceinline__ f32x16 mm(u32x4 a, u32x4 b, f32x16 c)
{
#ifdef FF_SPLIT_MFMA16          // timing experiment only: the same FLOPs as two 16x16x32 MFMAs (wrong results)
    f32x4 lo = f32x4{c[0], c[1], c[2], c[3]}, hi = f32x4{c[4], c[5], c[6], c[7]};
    lo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), lo, 0, 0, 0);
    hi = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), hi, 0, 0, 0);
    c[0] = lo[0]; c[1] = lo[1]; c[2] = lo[2]; c[3] = lo[3];
    c[4] = hi[0]; c[5] = hi[1]; c[6] = hi[2]; c[7] = hi[3];
    return c;
#else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
#endif
}
// one group: the six products of (w_hi, w_mid, w_lo) x (x_hi, x_mid, x_lo) that matter
__device__ __forceinline__ void group6(f32x16& acc, const u32x4 (&w)[3], const u32x4 (&b)[3])
{
    acc = mm(w[0], b[0], acc);
    acc = mm(w[0], b[1], acc);
    acc = mm(w[1], b[0], acc);
    acc = mm(w[0], b[2], acc);
    acc = mm(w[1], b[1], acc);
    acc = mm(w[2], b[0], acc);
}
// top halves of (a, b) -> one register of two bf16 (a in the low half): truncation split
__device__ __forceinline__ unsigned pack_hi(float a, float b)
{
    return __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, b), __builtin_bit_cast(unsigned, a), 0x07060302u);
}
__device__ __forceinline__ float top(float x) { return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, x) & 0xFFFF0000u); }
// two fp32 values -> word j of the three fragments
__device__ __forceinline__ void split2(float v0, float v1, u32x4 (&o)[3], int j)
{
    const float m0 = v0 - top(v0), m1 = v1 - top(v1);
    const float l0 = m0 - top(m0), l1 = m1 - top(m1);
    o[0][j] = pack_hi(v0, v1);
    o[1][j] = pack_hi(m0, m1);
    o[2][j] = pack_hi(l0, l1);
}
// value of the even neighbour lane (the value column of a (value, tangent) column pair): DPP quad_perm [0, 0, 2, 2]
__device__ __forceinline__ float from_value_lane(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xA0, 0xF, 0xF, true));
}
// hidden activation of one pre-activation.  Value columns: SiLU(a).  Tangent columns (forward-mode derivative through
// the same weights): a' * SiLU'(a of the sample's value column),  SiLU'(a) = s + a s (1 - s),  s = sigmoid(a).
template <bool TANGENTS>
__device__ __forceinline__ float act1(float a, bool is_tangent)
{
#ifdef FF_SPLIT_NOACT           // timing experiment only: identity activation (wrong results)
    return a;
#endif
    const float s = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(a * -1.44269504088896340736f));
    const float h = a * s;
    if constexpr (!TANGENTS) return h;
    const float d = __builtin_fmaf(h, 1.0f - s, s);
    const float dv = from_value_lane(d);
    return is_tangent ? a * dv : h;
}
// unit U (0..7) of a tile: registers 2U, 2U+1 -> word U & 3 of the fragments of k-step U >> 2
template <bool TANGENTS, int U>
__device__ __forceinline__ void act_unit(const f32x16& acc, u32x4 (&bf)[2][3], bool is_tangent)
{
    split2(act1<TANGENTS>(acc[2 * U], is_tangent), act1<TANGENTS>(acc[2 * U + 1], is_tangent), bf[U >> 2], U & 3);
}

// The same unit cut into four stages that are issued one GROUP (6 MFMAs, ~190 cycles) apart, so that no instruction
// waits on a transcendental issued just before it: the wavefront issues in order, and a VALU instruction stalled on
// its operand holds the next MFMA behind it.  Between stages a unit lives in four registers.
struct UnitState {
    float a0, a1, t0, t1;
};
template <bool TANGENTS, int U, int STAGE>
__device__ __forceinline__ void act_unit_stage(const f32x16& acc, UnitState& u, u32x4 (&bf)[2][3], bool is_tangent)
{
    constexpr float NLOG2E = -1.44269504088896340736f;
    if constexpr (STAGE == 0) {            // pre-activations out of the accumulator; exp(-a)
        u.a0 = acc[2 * U];
        u.a1 = acc[2 * U + 1];
#ifdef FF_SPLIT_NOACT
        u.t0 = u.a0; u.t1 = u.a1;
#else
        u.t0 = __builtin_amdgcn_exp2f(u.a0 * NLOG2E);
        u.t1 = __builtin_amdgcn_exp2f(u.a1 * NLOG2E);
#endif
    } else if constexpr (STAGE == 1) {     // sigmoid
#ifndef FF_SPLIT_NOACT
        u.t0 = __builtin_amdgcn_rcpf(1.0f + u.t0);
        u.t1 = __builtin_amdgcn_rcpf(1.0f + u.t1);
#endif
    } else if constexpr (STAGE == 2) {     // activation value (tangent columns: a' * SiLU'(a of the value column)); first residual
#ifdef FF_SPLIT_NOACT
        const float h0 = u.a0, h1 = u.a1;
#else
        float h0 = u.a0 * u.t0, h1 = u.a1 * u.t1;
        if constexpr (TANGENTS) {
            const float d0 = from_value_lane(__builtin_fmaf(h0, 1.0f - u.t0, u.t0));
            const float d1 = from_value_lane(__builtin_fmaf(h1, 1.0f - u.t1, u.t1));
            h0 = is_tangent ? u.a0 * d0 : h0;
            h1 = is_tangent ? u.a1 * d1 : h1;
        }
#endif
        u.a0 = h0;
        u.a1 = h1;
#ifndef FF_SPLIT_NOSPLIT        // timing experiment only: the three parts are all the top half (wrong results)
        u.t0 = h0 - top(h0);
        u.t1 = h1 - top(h1);
#endif
    } else {                               // second residual and the three packed words
#ifdef FF_SPLIT_NOSPLIT
        bf[U >> 2][0][U & 3] = bf[U >> 2][1][U & 3] = bf[U >> 2][2][U & 3] = pack_hi(u.a0, u.a1);
#else
        const float l0 = u.t0 - top(u.t0), l1 = u.t1 - top(u.t1);
        bf[U >> 2][0][U & 3] = pack_hi(u.a0, u.a1);
        bf[U >> 2][1][U & 3] = pack_hi(u.t0, u.t1);
        bf[U >> 2][2][U & 3] = pack_hi(l0, l1);
#endif
    }
}
// Schedule of a tile's 8 units over the groups of a span: unit u starts at group `start(u)`, stage k runs at group
// start(u) + k.  kind 0: a whole k-step pair of a hidden layer (16 groups: starts 0,1,3,4,6,7,9,10);
// kind 1: the 14 groups after a tile's own completion (starts 2 + u);  kind 2: the 7 groups after layer 1's tile 0
// (two units per group: starts 1 + u / 2).
FF_HD constexpr int unit_start(int kind, int u) { return kind == 0 ? (3 * u) / 2 : (kind == 1 ? 2 + u : 1 + u / 2); }
// run whatever stages of the tile's units fall on group `G` of the span
template <bool TANGENTS, int KIND, int G>
__device__ __forceinline__ void act_stages_at(const f32x16& acc, UnitState (&us)[8], u32x4 (&bf)[2][3], bool is_tangent)
{
    sfor<8>([&](auto uu) {
        constexpr int U = decltype(uu)::value;
        constexpr int k = G - unit_start(KIND, U);
        if constexpr (k >= 0 && k < 4) act_unit_stage<TANGENTS, U, k>(acc, us[U], bf, is_tangent);
    });
}

// LDS-DMA of one fragment: 64 lanes x 16 bytes from `g` (wave-uniform) + lane * 16 to LDS byte `lds_byte` + lane * 16.
// Inline asm on purpose: as a builtin the DMA makes hipcc spill, and every spill reload then queues behind it.
__device__ __forceinline__ void dma_fragment(unsigned lds_byte, const void* g, int lane16)
{
    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_byte), "v"(lane16), "s"(g) : "m0", "memory");
}

// NH (hidden layers) is a compile-time parameter: with the layer sequence unrolled the evaluation loop is one
// straight-line body and the 2 x NT accumulator tiles keep their registers (a run-time layer loop made hipcc shuffle
// all 256 accumulator registers at every control-flow join).
template <int NT, int K1S, int NH, bool TANGENTS>
__global__ __launch_bounds__(256, 1) void mlp_ode_split_kernel(const KernelArgs args)
{
    static_assert(NT == 8, "the granule schedule below is written for width 256 (8 row tiles)");
    constexpr int H = 32 * NT;
    constexpr int GB = kGranuleBytes;                  // 24 KiB
    constexpr int R = 8;                               // state registers per lane (16 dimensions over two lane halves)
    typedef const __attribute__((address_space(4))) RowHdr* HdrPtr;

    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int hf = lane >> 5;                          // lane half = k half of the fragments
    const int col = lane & 31;
    const int lane16 = lane * 16;
    const int D = args.dim, C = args.cond_dim;
    const LdsMap M = lds_map(H, NH);

    // ---- column roles ---------------------------------------------------------------------------------------------
    const long long wave = (long long)blockIdx.x * 4 + wv;
    long long sample;
    bool is_tangent = false, col_live = true;
    if constexpr (TANGENTS) {
        sample = wave * 16 + (col >> 1);
        is_tangent = (col & 1) != 0;
    } else {
        sample = wave * 32 + col;
    }
    if (sample >= args.batch) { sample = args.batch - 1; col_live = false; }

    // ---- state / probe / conditional ------------------------------------------------------------------------------
    // register j holds dimension kidx(0, hf, j) = 8 (j >> 2) + 4 hf + (j & 3)
    float x[R];
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const int d = kidx(0, hf, j);
        float v = 0.f;
        if (d < D) {
            if (!is_tangent) {
                v = args.x_in[sample * D + d];
                if (args.in_shift) v = v - args.in_shift[d];
                if (args.in_scale) v = v / args.in_scale[d];
            } else {
                v = args.probe[sample * D + d];
            }
        }
        x[j] = v;
    }
    u32x4 yf[K1S][3];                                  // B fragments of the first layer: k-step 0 = state, 1 = conditional
    if constexpr (K1S > 1) {
        float cnd[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int d = kidx(0, hf, j);
            cnd[j] = (d < C && !is_tangent) ? args.cond[sample * C + d] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) split2(cnd[2 * j], cnd[2 * j + 1], yf[1], j);
    }
    float ee = 0.f;                                    // tangent lanes: e.e restricted to this lane's dimensions
    if constexpr (TANGENTS) {
#pragma unroll
        for (int j = 0; j < R; ++j) ee = __builtin_fmaf(x[j], x[j], ee);
    }

    // ---- LDS set-up: stage slots, zero page, static biases --------------------------------------------------------
    f32x4* const ks = (f32x4*)(lds + M.slots) + threadIdx.x;          // slot s, quad j4: ks[(s * 2 + j4) * 256]
#pragma unroll
    for (int s = 0; s < kSlots; ++s)
#pragma unroll
        for (int j4 = 0; j4 < 2; ++j4) ks[(s * 2 + j4) * 256] = f32x4{0.f, 0.f, 0.f, 0.f};
    // the state x lives in LDS as well (slot kSlots + 1; slot kSlots parks the stage input y): it is touched twice per
    // evaluation, and 8 more live registers would spill inside the loop
#pragma unroll
    for (int j4 = 0; j4 < 2; ++j4)
        ks[((kSlots + 1) * 2 + j4) * 256] = f32x4{x[4 * j4], x[4 * j4 + 1], x[4 * j4 + 2], x[4 * j4 + 3]};
    for (int i = threadIdx.x; i < H; i += 256) ((float*)(lds + M.zero))[i] = 0.f;
    {
        const float* bsrc = args.wpack + (size_t)stream_words(NT, K1S, NH);
        const int nb = (NH - 1) * H + 32;
        for (int i = threadIdx.x; i < nb; i += 256) ((float*)(lds + M.hbias))[i] = bsrc[i];
    }
    float kl[kSlots];
#pragma unroll
    for (int s = 0; s < kSlots; ++s) kl[s] = 0.f;
    float lp = 0.f;

    // ---- weight pipeline state (all wave-uniform) -----------------------------------------------------------------
    const unsigned char* const wbase = (const unsigned char*)args.wpack;
    const long long wbytes = (long long)granules_per_eval(NT, K1S, NH) * GB;
    long long dpos = 0;                                // byte position in the stream of the NEXT granule to fetch
    unsigned rbuf = 0;                                 // LDS byte offset of the buffer the current granule is read from
    const int my_frag = wv * 6 * 1024;                 // this wavefront's quarter of a granule
    auto fetch_granule = [&](unsigned wbuf) __attribute__((always_inline)) {
        // The stream position is periodic in the evaluation loop, so hipcc would compute all ~300 fragment addresses
        // of an evaluation once, ahead of the loop, and keep them in (spilled) SGPRs: hide the two bases from it.
        const unsigned char* src = wbase + dpos + my_frag;
        unsigned dst = wbuf + my_frag;
        asm volatile("" : "+s"(src), "+s"(dst));
#pragma unroll
        for (int f = 0; f < 6; ++f) dma_fragment(dst + f * 1024, src + f * 1024, lane16);
        dpos += GB;
        if (dpos >= wbytes) dpos = 0;
    };
    auto fetch_c1 = [&](int e) __attribute__((always_inline)) {                       // c1 of evaluation e -> its LDS buffer (every wavefront issues the
        const int ee_ = e < args.n_evals ? e : 0;      // same 1 KiB copy: equal VMEM counts keep the counted waits uniform)
        dma_fragment(M.c1 + (e & 1) * 1024, (const unsigned char*)(args.etab + (size_t)ee_ * args.etab_stride + 32), lane16);
    };
    fetch_c1(0);
    fetch_granule(0);
    fetch_granule(GB);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();

    // bias tile t of the vector at LDS byte `base` (tangent lanes read the zero page): accumulator register order
    const int zsel = M.zero + hf * 16;
    auto bias_tile = [&](int base, int t) __attribute__((always_inline)) {
        const int a = (TANGENTS && is_tangent) ? zsel : base + hf * 16;
        f32x16 r;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 v = *(const f32x4*)(lds + a + (32 * t + 8 * g) * 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) r[4 * g + i] = v[i];
        }
        return r;
    };

    f32x16 A[NT], B[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        A[t] = bias_tile(M.c1, t);
        B[t] = bias_tile(M.hbias, t);                  // (unused when there is a single hidden layer)
    }

    // weight fragments of the current group (w) and the next one (wn), read from LDS one group ahead
    u32x4 w[3], wn[3];
    int wa = lane16;                                   // LDS address of this lane's 16 bytes of fragment 0 of the granule
    int wa_next = lane16;
#pragma unroll
    for (int p = 0; p < 3; ++p) w[p] = *(const u32x4*)(lds + wa + p * 1024);

    // Called before the MFMAs of group GQ (0..7) of a granule.
    auto pre = [&](auto gq) __attribute__((always_inline)) {
        constexpr int GQ = decltype(gq)::value;
        if constexpr (GQ == 0) {
            asm volatile("" : "+s"(rbuf));             // (periodic over evaluations as well: keep it a run-time value)
            unsigned wb = rbuf + 2 * GB;
            if (wb >= 3 * GB) wb -= 3 * GB;
#ifndef FF_SPLIT_NODMA          // timing experiment only: never refresh the weight buffers (wrong results)
            fetch_granule(wb);                         // granule + 2 -> the buffer read before this one
#endif
        }
        if constexpr (GQ == 7) {
            // everything but the six DMAs just issued has landed, and this wavefront's reads of the current buffer
            // have returned: after the barrier the next granule is visible to all and the previous buffer is free
            asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
#ifndef FF_SPLIT_NOBARRIER      // timing experiment only (with FF_SPLIT_NODMA)
            __builtin_amdgcn_s_barrier();
#endif
            rbuf += GB;
            if (rbuf >= 3 * GB) rbuf = 0;
            wa_next = lane16 + rbuf;
#pragma unroll
            for (int p = 0; p < 3; ++p) wn[p] = *(const u32x4*)(lds + wa_next + p * 1024);
        } else {
#ifndef FF_SPLIT_NOWREAD        // timing experiment only: one group's fragments serve the whole granule (wrong results)
#pragma unroll
            for (int p = 0; p < 3; ++p) wn[p] = *(const u32x4*)(lds + wa + ((GQ + 1) * 3 + p) * 1024);
#else
#pragma unroll
            for (int p = 0; p < 3; ++p) wn[p] = w[p];
#endif
        }
    };
    auto post = [&](auto gq) __attribute__((always_inline)) {
        constexpr int GQ = decltype(gq)::value;
#pragma unroll
        for (int p = 0; p < 3; ++p) w[p] = wn[p];
        if constexpr (GQ == 7) wa = wa_next;
        // MFMAs, LDS and vector-memory instructions keep their order; VALU / SALU / transcendentals may move
        __builtin_amdgcn_sched_barrier(0x2 | 0x4 | 0x400);
    };

    u32x4 bf[2][2][3];                                 // B fragments of the k-step pair in use / in preparation
    UnitState us[8];                                   // activation units in flight (four stages, one group apart)

    // A hidden -> hidden layer (reads P, writes C) or, with OUT, the output layer (reads P, writes O[0]).
    // `refill` = LDS byte address of the bias vector the tiles of P are refilled with once consumed (their next use).
    auto layer = [&](auto is_out, f32x16 (&P)[NT], f32x16 (&Cc)[NT], f32x16& O, int refill) __attribute__((always_inline)) {
        constexpr bool OUT = decltype(is_out)::value;
        P[0] = bias_tile(refill, 0);                   // tile 0 was consumed at the end of the layer before
        sfor<NT>([&](auto pp) {
            constexpr int p = decltype(pp)::value;
            constexpr int NTILE = OUT ? 1 : NT;
            sfor<NTILE>([&](auto tt) {
                constexpr int t = decltype(tt)::value;
                sfor<2>([&](auto ss) {
                    constexpr int s = decltype(ss)::value;
                    constexpr int q = (p * NTILE + t) * 2 + s;
                    pre(std::integral_constant<int, q % 8>{});
                    if constexpr (OUT) group6(O, w, bf[p & 1][s]);
                    else group6(Cc[t], w, bf[p & 1][s]);
                    // activation stages in the shadow of the MFMAs just issued
                    if constexpr (OUT) {
                        if constexpr (p < NT - 1) {    // four whole units per group (the output layer is VALU-bound anyway)
                            sfor<4>([&](auto uu) { act_unit<TANGENTS, 4 * s + decltype(uu)::value>(P[p + 1], bf[(p + 1) & 1], is_tangent); });
                        }
                    } else if constexpr (p < NT - 1) {
                        act_stages_at<TANGENTS, 0, 2 * t + s>(P[p + 1], us, bf[(p + 1) & 1], is_tangent);
                    } else {                           // last pair: the first tile of THIS layer's output, complete after group 1
                        act_stages_at<TANGENTS, 1, 2 * t + s>(Cc[0], us, bf[0], is_tangent);
                    }
                    post(std::integral_constant<int, q % 8>{});
                });
            });
            if constexpr (p < NT - 1) P[p + 1] = bias_tile(refill, p + 1);     // consumed during this pair
        });
    };

    for (int e = 0; e < args.n_evals; ++e) {
        HdrPtr hdr = (HdrPtr)(args.etab + (size_t)e * args.etab_stride);
        const float a_e = hdr->a, b_e = hdr->b;
        const uint32_t flags = hdr->flags;
        const int slot = hdr->slot;

        // stage input  y = x + sum_s cin[s] k[s]
        float y[R];
#pragma unroll
        for (int j4 = 0; j4 < 2; ++j4) {
            f32x4 v = ks[((kSlots + 1) * 2 + j4) * 256];
#pragma unroll
            for (int s = 0; s < kSlots; ++s) v += hdr->cin[s] * ks[(s * 2 + j4) * 256];
#pragma unroll
            for (int i = 0; i < 4; ++i) y[4 * j4 + i] = v[i];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) split2(y[2 * j], y[2 * j + 1], yf[0], j);
        // y is needed again for the right-hand side, a whole network evaluation later: park it in LDS (slot kSlots)
#pragma unroll
        for (int j4 = 0; j4 < 2; ++j4)
            ks[(kSlots * 2 + j4) * 256] = f32x4{y[4 * j4], y[4 * j4 + 1], y[4 * j4 + 2], y[4 * j4 + 3]};

        // ---- layer 1: [state | conditional] -> H, accumulators A already hold c1_e ---------------------------------
        sfor<K1S>([&](auto ss) {
            constexpr int s = decltype(ss)::value;
            sfor<NT>([&](auto tt) {
                constexpr int t = decltype(tt)::value;
                constexpr int q = s * NT + t;
                pre(std::integral_constant<int, q % 8>{});
#ifndef FF_SPLIT_NODMA
                if constexpr (q == 0) fetch_c1(e + 1);                 // (after the weight DMAs of this granule)
#endif
                group6(A[t], w, yf[s]);
                if constexpr (s == K1S - 1) {                           // tile 0 is complete after group 0: its activation rides here
                    act_stages_at<TANGENTS, 2, t>(A[0], us, bf[0], is_tangent);
                }
                post(std::integral_constant<int, q % 8>{});
            });
        });

        // ---- hidden -> hidden layers, ping-pong A -> B -> A ..., then the output layer ------------------------------
        // refill address of the set being READ by layer j (1-based; j = NH is the output layer): the bias of the next
        // layer that writes that set -- layer j+1 of this evaluation if it is a hidden one, else the first layer of the
        // next evaluation (set A: c1 of e+1) or its first hidden->hidden layer (set B)
        const int c1_next = M.c1 + ((e + 1) & 1) * 1024;
        auto refill_for = [&](int j, bool reads_A) __attribute__((always_inline)) {
            if (j + 1 <= NH - 1) return M.hbias + j * H * 4;
            return reads_A ? c1_next : M.hbias;
        };
        f32x16 O;
        sfor<NH>([&](auto jj) {
            constexpr int j = decltype(jj)::value + 1;                  // layers 1 .. NH-1 hidden -> hidden, NH = output
            constexpr bool reads_A = (j & 1) != 0;
            if constexpr (j < NH) {
                if constexpr (reads_A) layer(std::false_type{}, A, B, O, refill_for(j, true));
                else layer(std::false_type{}, B, A, O, refill_for(j, false));
            } else {
                O = bias_tile(M.hbias + (NH - 1) * H * 4, 0);           // output bias (32 rows)
                if constexpr (reads_A) layer(std::true_type{}, A, B, O, refill_for(j, true));
                else layer(std::true_type{}, B, A, O, refill_for(j, false));
            }
        });

        // ---- right-hand side and stage bookkeeping ------------------------------------------------------------------
        float rhs[R];
        float div = 0.f;
        if constexpr (TANGENTS) {
            float dot = 0.f;
#pragma unroll
            for (int j4 = 0; j4 < 2; ++j4) {
                const f32x4 xv = ks[((kSlots + 1) * 2 + j4) * 256];          // tangent lanes: the probe e
#pragma unroll
                for (int i = 0; i < 4; ++i) dot = __builtin_fmaf(xv[i], O[4 * j4 + i], dot);
            }
            div = is_tangent ? __builtin_fmaf(a_e, ee, b_e * dot) : 0.f;
        }
#pragma unroll
        for (int j4 = 0; j4 < 2; ++j4) {
            const f32x4 yv = ks[(kSlots * 2 + j4) * 256];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float v = __builtin_fmaf(a_e, yv[i], b_e * O[4 * j4 + i]);
                rhs[4 * j4 + i] = is_tangent ? 0.f : v;
            }
        }
#pragma unroll
        for (int j4 = 0; j4 < 2; ++j4)
            ks[(slot * 2 + j4) * 256] = f32x4{rhs[4 * j4], rhs[4 * j4 + 1], rhs[4 * j4 + 2], rhs[4 * j4 + 3]};
        if constexpr (TANGENTS) {
#pragma unroll
            for (int s = 0; s < kSlots; ++s) kl[s] = (slot == s) ? div : kl[s];
        }
        if (flags & 1u) {
#pragma unroll
            for (int j4 = 0; j4 < 2; ++j4) {
                f32x4 v = ks[((kSlots + 1) * 2 + j4) * 256];
#pragma unroll
                for (int s = 0; s < kSlots; ++s) v += hdr->cout[s] * ks[(s * 2 + j4) * 256];
                ks[((kSlots + 1) * 2 + j4) * 256] = v;
            }
            if constexpr (TANGENTS) {
#pragma unroll
                for (int s = 0; s < kSlots; ++s) lp = __builtin_fmaf(hdr->cout[s], kl[s], lp);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the pipeline runs two granules ahead: let it drain

    // ---- epilogue -----------------------------------------------------------------------------------------------------
    const bool writer = col_live && !is_tangent;
    if constexpr (TANGENTS) {
        // the sample's divergence = sum over both lane halves of its tangent column (the next lane)
        float tot = 0.f;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const int src = ((g << 5) | ((col + 1) & 31)) * 4;
            tot += __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, lp)));
        }
        if (writer && hf == 0 && args.dlogp_out) args.dlogp_out[sample] = (args.dlogp_in ? args.dlogp_in[sample] : 0.f) + tot;
    }
    bool bad = false;
    float xfin[R];
#pragma unroll
    for (int j4 = 0; j4 < 2; ++j4) {
        const f32x4 v = ks[((kSlots + 1) * 2 + j4) * 256];
#pragma unroll
        for (int i = 0; i < 4; ++i) xfin[4 * j4 + i] = v[i];
    }
    if (writer) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int d = kidx(0, hf, r);
            if (d < D) {
#pragma clang fp contract(off)      // x * scale + shift as two roundings, like the reference's torch expression
                float v = xfin[r];
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

} // namespace split
} // namespace ff
