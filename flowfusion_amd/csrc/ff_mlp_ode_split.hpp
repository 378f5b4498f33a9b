// ff_mlp_ode_split.hpp -- fused MLP-ODE integrator on the bf16 matrix cores with fp32-class accuracy (gfx950).
//
// Same contract as ff_mlp_ode.hpp (one launch integrates dy/ds = a_e y + b_e NET(y, cond; c1_e) over all evaluation
// rows; reference call sites flowfusion/diffusion.py:631-639, 744-752, RHS :258-334, network :82-121), different
// arithmetic (FF_PREC_BF16X3, opt-in): every fp32 operand of a Linear layer is cut into three bf16 parts
//     v = hi + mid + lo      hi = top 16 bits of v, mid = top 16 bits of (v - hi), lo = top 16 bits of the rest
// (24 significand bits in all: exact for the weights, which are split on the host), and a product is the sum of the six
// bf16 MFMAs whose parts' magnitudes reach 2^-16 of the full product,
//     w.x ~= wh.xh + wh.xm + wm.xh + wh.xl + wm.xm + wl.xh          (dropped terms: <= 2^-24 relative)
// accumulated in fp32 by the MFMA.  Error of a layer ~1e-7 relative to sum |w||x| -- what an fp32 dot product has --
// at 16/6 = 2.7x the MFMA rate of the f32 MFMAs.
//
// MFMA shape: v_mfma_f32_16x16x32_bf16 (16 cycles).  The 32x32x16 form does the same FLOPs per cycle, but this
// kernel is power-limited and the chip holds a markedly higher clock on the 16x16 form (measured in this kernel by
// swapping the instruction: 2.27 vs 1.80 GHz; MI355X_MICROARCH.md, DVFS give-back item 7).
//
// Mapping.  A wavefront owns 32 MFMA columns = two column blocks of 16 (32 samples; in Hutchinson mode 16 samples x
// (value, tangent)); a workgroup is 4 wavefronts, one per SIMD (the kernel needs ~480 of the 512 registers).  Lane l
// of a 16x16 accumulator tile holds column l & 15 and rows 4 (l >> 4) + i in its 4 registers; the B operand of the
// MFMA wants, on the same lane, 8 consecutive k slots 8 (l >> 4) + j.  So the accumulator tiles of TWO consecutive
// 16-row tiles (2s, 2s+1) are, register for register, the B operand of k-step s of the next layer if the weights'
// k slots are laid out as kidx() below says: after SiLU and the split, activations never leave registers.
//
// Loop order: k-major.  A layer keeps ALL its 16 row tiles x 2 column blocks live (128 registers) and walks the
// k-steps (32 features each); k-step s needs exactly row tiles 2s, 2s+1 of the previous layer, whose SiLU + split
// (~12 VALU instructions per element) is done one k-step ahead, in the shadow of that k-step's 192 MFMAs: cut into
// micro-ops of at most two instructions, each pinned behind one particular MFMA (GapPlan below; full scheduling
// barriers make the instruction stream the one written here).  Two sets of tiles ping-pong between consecutive layers; a tile is refilled with
// the bias of its next use (from LDS) as soon as it has been consumed, so the MFMA chain adds the bias.
//
// Weights: every wavefront needs every fragment, so they are staged through LDS and shared by the workgroup.  The
// fragments of one evaluation form a linear stream of 24 KiB granules (8 groups x [hi, mid, lo] x 1 KiB; one group =
// one (16-row tile, k-step) = 12 MFMAs: six products x two column blocks) in consumption order, periodic over
// evaluations.  Three LDS buffers: at the start of granule g each wavefront starts the LDS-DMA
// (global_load_lds_dwordx4, no registers) of its quarter of granule g+2; before the last group of granule g a counted
// wait (vmcnt(6): all but the DMAs just issued) and ONE barrier publish granule g+1 and retire granule g-1's buffer.
// The per-evaluation first-layer bias c1_e travels the same way.
//
// Scope of this family: width 256, dim <= 16, cond <= 16 (state and conditional inputs share the first layer's single
// k-step), SiLU, 1-6 hidden layers, FF_MODE_STATE and FF_MODE_HUTCH, any fixed-grid table (<= 7 stage slots, kept in
// LDS with the state), the adaptive-step inputs / outputs (k1_in, aux_out: one attempted step per launch) and, in the
// state-only kernels, Euler-Maruyama noise rows (buffer or in-kernel Philox); no Jacobian output.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <utility>
#include "ff_layout.h"
#include "ff_kernel_args.h"
#include "ff_split_layout.h"
#include "ff_philox.h"
#include "ff_skew.h"

namespace ff {
namespace split {

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

__device__ __forceinline__ f32x4 mm(u32x4 a, u32x4 b, f32x4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// top halves of (a, b) -> one register of two bf16 (a in the low half): truncation split
__device__ __forceinline__ unsigned pack_hi(float a, float b)
{
    return __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, b), __builtin_bit_cast(unsigned, a), 0x07060302u);
}
__device__ __forceinline__ float top(float x) { return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, x) & 0xFFFF0000u); }
// (a, b) rounded to nearest-even bf16, a in the low half: one v_cvt_pk_bf16_f32
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_rne(float a, float b)
{
    return __builtin_bit_cast(unsigned, bf16x2{(__bf16)a, (__bf16)b});
}
__device__ __forceinline__ float lo_half(unsigned p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float hi_half(unsigned p) { return __builtin_bit_cast(float, p & 0xFFFF0000u); }
// two fp32 values -> word j of the NP fragments: three parts by truncation (exact), two by round-to-nearest
template <int NP>
__device__ __forceinline__ void split2(float v0, float v1, u32x4 (&o)[NP], int j)
{
    if constexpr (NP == 3) {
        const float m0 = v0 - top(v0), m1 = v1 - top(v1);
        const float l0 = m0 - top(m0), l1 = m1 - top(m1);
        o[0][j] = pack_hi(v0, v1);
        o[1][j] = pack_hi(m0, m1);
        o[2][j] = pack_hi(l0, l1);
    } else {
        const unsigned h = pack_rne(v0, v1);
        o[0][j] = h;
        o[1][j] = pack_rne(v0 - lo_half(h), v1 - hi_half(h));
    }
}
// value of the even neighbour lane (the value column of a (value, tangent) column pair): DPP quad_perm [0, 0, 2, 2]
__device__ __forceinline__ float from_value_lane(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xA0, 0xF, 0xF, true));
}

// Activation + split of one UNIT = two pre-activations -> one word of the NP fragments, cut into MICRO-OPS of at
// most two plain VALU instructions or one transcendental: what fits the 8 cycles of vector issue a 16-cycle MFMA
// leaves free (MI355X_MICROARCH.md, issue costs).  Every micro-op is pinned behind one particular MFMA (a GAP; plan
// below): left to the list scheduler the same instructions clump -- four or five behind one MFMA, none behind the next
// six.  Value columns: SiLU(a).  Tangent columns (forward-mode derivative through the same weights): a' * SiLU'(a of
// the sample's value column), SiLU'(a) = s + a s (1 - s).
#ifndef FF_SPLIT_AHEAD
#define FF_SPLIT_AHEAD 1
#endif
constexpr int kAhead = FF_SPLIT_AHEAD;      // groups between the LDS read of a weight fragment and its MFMAs

struct UnitState {
    float a0, a1, t0, t1, h0, h1, o0, o1;      // (h, o: tangent kernels only)
};
enum MicroOp {
    M_LOAD, M_SCALE, M_EXP0, M_EXP1, M_ADD1, M_RCP0, M_RCP1, M_VALUE,
    M_TV_H, M_TV_OMS, M_TV_SLOPE, M_TV_DPP, M_TV_MUL, M_TV_SEL,           // tangent kernels: in place of M_VALUE
    M_TOPH, M_RESM, M_PACKH, M_PACKM, M_RESL, M_PACKL,                    // three parts by truncation
    M_RHI, M_RRES, M_RMID                                                 // two parts by round-to-nearest
};
FF_HD constexpr int micro_count(bool tangents, int parts) { return (tangents ? 13 : 8) + (parts == 3 ? 6 : 3); }
FF_HD constexpr int micro_op(bool tangents, int parts, int j)
{
    if (j < 7) return j;                                   // M_LOAD .. M_RCP1
    const int nv = tangents ? 6 : 1;                       // micro-ops that form the activation value
    if (j < 7 + nv) return tangents ? M_TV_H + (j - 7) : M_VALUE;
    return (parts == 3 ? M_TOPH : M_RHI) + (j - 7 - nv);
}
// TM: 0 = no tangent columns, 1 = (value, tangent) column pairs (Hutchinson: the slope comes from the neighbour lane by
// DPP), 2 = a value column followed by up to 15 unit-tangent columns (exact trace: the slope comes from lane `vsrc / 4`
// through the LDS crossbar, ds_bpermute)
template <int TM, int NP, int J>
__device__ __forceinline__ void unit_micro(float pre0, float pre1, UnitState& u, u32x4 (&frag)[NP], int word, bool is_tangent,
                                           int vsrc)
{
    constexpr float NLOG2E = -1.44269504088896340736f;
    constexpr bool TANGENTS = TM != 0;
    constexpr int OP = micro_op(TANGENTS, NP, J);
#ifdef FF_SPLIT_SKIP_OPS        // timing experiments only: bit OP set = that micro-op is left out (wrong results)
    if constexpr (((FF_SPLIT_SKIP_OPS) >> OP) & 1) return;
#endif
    if constexpr (OP == M_LOAD) {              // pre-activations out of the accumulator tile (AGPR -> VGPR copies, here)
        u.a0 = pre0;
        u.a1 = pre1;
        asm volatile("" : "+v"(u.a0), "+v"(u.a1));
    } else if constexpr (OP == M_SCALE) {
        u.t0 = u.a0 * NLOG2E;
        u.t1 = u.a1 * NLOG2E;
    } else if constexpr (OP == M_EXP0) {
        u.t0 = __builtin_amdgcn_exp2f(u.t0);
    } else if constexpr (OP == M_EXP1) {
        u.t1 = __builtin_amdgcn_exp2f(u.t1);
    } else if constexpr (OP == M_ADD1) {
        u.t0 = 1.0f + u.t0;
        u.t1 = 1.0f + u.t1;
    } else if constexpr (OP == M_RCP0) {       // sigmoid
        u.t0 = __builtin_amdgcn_rcpf(u.t0);
    } else if constexpr (OP == M_RCP1) {
        u.t1 = __builtin_amdgcn_rcpf(u.t1);
    } else if constexpr (OP == M_VALUE) {      // SiLU
        u.a0 = u.a0 * u.t0;
        u.a1 = u.a1 * u.t1;
    } else if constexpr (OP == M_TV_H) {
        u.h0 = u.a0 * u.t0;
        u.h1 = u.a1 * u.t1;
    } else if constexpr (OP == M_TV_OMS) {     // h (1 - s)
        u.o0 = __builtin_fmaf(-u.h0, u.t0, u.h0);
        u.o1 = __builtin_fmaf(-u.h1, u.t1, u.h1);
    } else if constexpr (OP == M_TV_SLOPE) {   // SiLU'(a) = s + h (1 - s)
        u.t0 = u.o0 + u.t0;
        u.t1 = u.o1 + u.t1;
    } else if constexpr (OP == M_TV_DPP) {     // tangent lanes: the slope of their sample's value column
        if constexpr (TM == 1) {
            u.t0 = from_value_lane(u.t0);
            u.t1 = from_value_lane(u.t1);
        } else {
            u.t0 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(vsrc, __builtin_bit_cast(int, u.t0)));
            u.t1 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(vsrc, __builtin_bit_cast(int, u.t1)));
        }
    } else if constexpr (OP == M_TV_MUL) {
        u.o0 = u.a0 * u.t0;
        u.o1 = u.a1 * u.t1;
    } else if constexpr (OP == M_TV_SEL) {
        u.a0 = is_tangent ? u.o0 : u.h0;
        u.a1 = is_tangent ? u.o1 : u.h1;
    } else if constexpr (OP == M_TOPH) {       // three-way split of the activation value (a0, a1)
        u.t0 = top(u.a0);
        u.t1 = top(u.a1);
    } else if constexpr (OP == M_RESM) {
        u.t0 = u.a0 - u.t0;
        u.t1 = u.a1 - u.t1;
    } else if constexpr (OP == M_PACKH) {
        frag[0][word] = pack_hi(u.a0, u.a1);
        u.a0 = top(u.t0);
    } else if constexpr (OP == M_PACKM) {
        u.a1 = top(u.t1);
        frag[1][word] = pack_hi(u.t0, u.t1);
    } else if constexpr (OP == M_RESL) {
        u.a0 = u.t0 - u.a0;
        u.a1 = u.t1 - u.a1;
    } else if constexpr (OP == M_PACKL) {
        frag[NP - 1][word] = pack_hi(u.a0, u.a1);
    } else if constexpr (OP == M_RHI) {        // two-way split, round to nearest: hi = bf16(a), mid = bf16(a - hi)
        const unsigned h = pack_rne(u.a0, u.a1);
        frag[0][word] = h;
        u.t0 = lo_half(h);
        u.t1 = hi_half(h);
    } else if constexpr (OP == M_RRES) {
        u.t0 = u.a0 - u.t0;
        u.t1 = u.a1 - u.t1;
    } else {
        frag[NP - 1][word] = pack_rne(u.t0, u.t1);
    }
}

// GAP PLAN of a span of 16 groups (one k-step of a hidden layer, or the whole first layer; 12 MFMAs per group with three
// parts, 6 with two): which micro-ops of which of the 8 units (a k-step's operands: 2 column blocks x 4 words) sit
// behind MFMA g of the span.  Built greedily: a unit's micro-ops are at least two gaps apart (nothing waits on the
// instruction before it), the gaps that carry an LDS-DMA hold none, and a gap holds as few micro-ops as the span's room
// allows (one with three parts; up to two with two parts, whose k-steps are half as long).  Units of span kind 1 read
// row tiles 0 and 1 of the span's OWN output, complete after groups 0 and 1: they start behind group 1.
constexpr int kMaxSpanGaps = 16 * 12, kMaxPerGap = 5;
#ifndef FF_SPLIT_SPACING
#define FF_SPLIT_SPACING 2
#endif
constexpr int kMicroSpacing = FF_SPLIT_SPACING;      // gaps between consecutive micro-ops of a unit
FF_HD constexpr int gaps_per_group(int parts) { return 2 * products_of(parts); }
// gap i of a group that opens a granule issues one of the wavefront's `ndma` LDS-DMAs (its quarter of the granule:
// parts * granule_groups / 4 fragments -- 6 / 4 at width 256, 3 / 2 at width 128)
FF_HD constexpr int dma_count(int parts, int w) { return parts * granule_groups(w) / 4; }
FF_HD constexpr int dma_index(int parts, int i) { return parts == 3 ? i / 2 : i; }
FF_HD constexpr bool dma_gap(int parts, int w, int i)
{
    return (parts == 3 ? (i % 2 == 0) : true) && dma_index(parts, i) < dma_count(parts, w);
}
// gap i of a group reads fragment `part` of the group kAhead on from LDS (-1: none)
FF_HD constexpr int read_part(int parts, int i) { return parts == 3 ? (i % 4 == 1 ? i / 4 : -1) : (i == 1 ? 0 : (i == 4 ? 1 : -1)); }
struct GapPlan {
    signed char n[kMaxSpanGaps];
    signed char unit[kMaxSpanGaps][kMaxPerGap];
    signed char micro[kMaxSpanGaps][kMaxPerGap];
    bool ok;
};
FF_HD constexpr GapPlan make_gap_plan(bool tangents, int parts, int first_gap, int w)
{
    GapPlan p{};
    const int per = gaps_per_group(parts), span = row_tiles(w) * per, nm = micro_count(tangents, parts);
    bool dma[kMaxSpanGaps] = {};
    int room = 0;
    for (int g = 0; g < span; ++g) {
        dma[g] = ((g / per) % granule_groups(w) == 0) && dma_gap(parts, w, g % per);
        room += (g >= first_gap && !dma[g]) ? 1 : 0;
    }
    const int cap = (8 * nm + room - 1) / room;
    p.ok = cap <= kMaxPerGap;
    for (int cap_now = cap; p.ok; ++cap_now) {           // (the spacing rule can strand a unit at the span's end: allow one more per gap)
        for (int g = 0; g < kMaxSpanGaps; ++g) p.n[g] = 0;
        bool fits = true;
        for (int u = 0; u < 8 && fits; ++u) {
            int g = first_gap;
            for (int j = 0; j < nm && fits; ++j) {
                while (g < span && (dma[g] || p.n[g] >= cap_now)) ++g;
                if (g >= span) {
                    fits = false;
                    break;
                }
                p.unit[g][p.n[g]] = (signed char)u;
                p.micro[g][p.n[g]] = (signed char)j;
                ++p.n[g];
                g += kMicroSpacing;
            }
        }
        if (fits) break;
        if (cap_now >= kMaxPerGap) p.ok = false;
    }
    return p;
}
template <bool TANGENTS, int NP, int HW>
struct GapPlans {
    static constexpr GapPlan k0 = make_gap_plan(TANGENTS, NP, 0, HW);
    static constexpr GapPlan k1 = make_gap_plan(TANGENTS, NP, 2 * gaps_per_group(NP), HW);
    static_assert(k0.ok && k1.ok, "the activation micro-ops of a k-step do not fit behind its MFMAs");
};

// LDS-DMA of one fragment: 64 lanes x 16 bytes from `g` (wave-uniform) + lane * 16 to LDS byte `lds_byte` + lane * 16.
// Inline asm on purpose: as a builtin the DMA makes hipcc spill, and every spill reload then queues behind it.
__device__ __forceinline__ void dma_fragment(unsigned lds_byte, const void* g, int lane16)
{
    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_byte), "v"(lane16), "s"(g) : "m0", "memory");
}

// NH (hidden layers) is a compile-time parameter: with the layer sequence unrolled the evaluation loop is one
// straight-line body and the accumulator tiles keep their registers (a run-time layer loop made hipcc shuffle all 256
// accumulator registers at every control-flow join).
// NSL = stage slots kept in LDS: slots_on_chip(DT), or 4 for the four-slot twins (two workgroups per CU)
template <int NH, int TM, int NP = 3, int DT = 1, int HW = 256, int NSL = slots_on_chip(DT)>
__global__ __launch_bounds__(256, (NSL == 4 && has_four_slot_twin(NP, DT, HW)) ? 2 : 1) void mlp_ode_split_kernel(const KernelArgs args)
{
    static_assert(HW == 256 || HW == 128, "on-chip width: 256 or 128");
    static_assert(TM >= 0 && TM <= 2, "0: state only, 1: Hutchinson column pairs, 2: exact trace (unit tangents)");
    constexpr bool TANGENTS = TM != 0;
    static_assert(NP == 2 || NP == 3, "two (round-to-nearest) or three (truncation) bf16 parts per operand");
    static_assert(DT == 1 || (DT == 2 && NP == 2), "states of up to 32 dimensions: two-part kernels only (LDS)");
    constexpr int NSLOT = NSL;                         // stage slots kept in LDS: 7 / 4
    constexpr int NR = row_tiles(HW);                  // row tiles of 16 rows: 16 (width 256) / 8
    constexpr int NS = ksteps(HW);                     // k-steps of 32 features: 8 / 4
    constexpr int H = HW;
    constexpr int GRG = granule_groups(HW);            // groups per granule: 8 / 4
    constexpr int GB = granule_bytes(NP, HW);          // 24 KiB (three parts) / 16 KiB (two) at width 256, half that at 128
    constexpr int GG = gaps_per_group(NP);             // MFMAs of a group: 12 / 6
    constexpr int NDMA = dma_count(NP, HW);            // fragments of a granule this wavefront fetches
    // weight buffers in LDS: 3 (the DMA runs two granules ahead) or, in the three-part four-slot twins, 2 (one granule ahead)
    constexpr int NBUF = (NSL == 4 && DT == 1 && has_four_slot_twin(NP, DT, HW)) ? weight_buffers(NP, 4) : kBuffers;
    typedef const __attribute__((address_space(4))) RowHdr* HdrPtr;

    // a cleared gate word (launches enqueued ahead of a device-side decision: ff_adaptive.hip) makes this launch a no-op
    if (args.gate && *(const volatile int*)args.gate == 0) return;

    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int qd = lane >> 4;                          // quad: rows 4 qd .. 4 qd + 3 of a tile, k slots 8 qd .. 8 qd + 7
    const int col = lane & 15;                         // column within a column block
    const int lane16 = lane * 16;
    const int D = args.dim, C = args.cond_dim;
    const LdsMap M = lds_map(H, NH, NP, DT, NSL);

    // ---- column roles: this lane serves one column of each of the two column blocks ---------------------------------
    const long long wave = (long long)blockIdx.x * 4 + wv;
    long long sample[2];
    bool live[2];
    bool is_tangent = false;
    int role = 0;                                      // 0: value column; j > 0: the j-th tangent column of its sample
    int vsrc = 0;                                      // byte index (lane * 4) of the sample's value lane in this lane's quad row
    const int NT = TM == 2 ? args.n_tangent : (TM == 1 ? 1 : 0);   // tangent columns per sample
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        if constexpr (TM == 1) {
            sample[cb] = wave * 16 + cb * 8 + (col >> 1);
            is_tangent = (col & 1) != 0;
            role = col & 1;
            live[cb] = sample[cb] < args.batch;
        } else if constexpr (TM == 2) {
            // a column block of 16 holds floor(16 / (1 + NT)) samples, each a value column and its NT unit tangents
            const int per = 1 + NT, spb = 16 / per, sb = col / per;
            role = col - sb * per;
            is_tangent = role > 0;
            vsrc = ((lane & 0x30) | (col - role)) * 4;
            sample[cb] = (wave * 2 + cb) * spb + sb;
            live[cb] = sb < spb && sample[cb] < args.batch;
        } else {
            sample[cb] = wave * 32 + cb * 16 + col;
            live[cb] = sample[cb] < args.batch;
        }
        if (!live[cb]) sample[cb] = args.batch - 1;
    }

    // ---- state / probe / conditional: register i of tile t of column block cb holds dimension 16 t + 4 qd + i ----------
    // slot s, column block cb, tile t: ks[SL(s, cb, t)]; slot NSLOT parks the stage input y, slot NSLOT + 1 is the state x
    f32x4* const ks = (f32x4*)(lds + M.slots) + threadIdx.x;
    auto SL = [](int s, int cb, int t) __attribute__((always_inline)) { return ((s * 2 + cb) * DT + t) * 256; };
    float ee[2] = {0.f, 0.f};                          // tangent lanes: e.e restricted to this lane's dimensions
    // B fragments of the first layer.  DT = 1: one k-step, words 0,1 = state, 2,3 = conditional inputs.  DT = 2: k-step 0 =
    // the state (words 0,1 = tile 0, words 2,3 = tile 1), k-step 1 = the conditional inputs (words 0,1; 2,3 zero)
    u32x4 yf[DT][2][NP];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        f32x4 cv = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < DT; ++t) {
            f32x4 xv = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int d = 16 * t + 4 * qd + i;
                if (d < D) {
                    if (!is_tangent) {
                        float v = args.x_in[sample[cb] * D + d];
                        if (args.in_shift) v = v - args.in_shift[d];
                        if (args.in_scale) v = v / args.in_scale[d];
                        xv[i] = v;
                    } else if constexpr (TM == 2) {
                        xv[i] = (d == args.tangent_first + role - 1) ? 1.f : 0.f;      // unit tangent e_j
                    } else {
                        xv[i] = args.probe[sample[cb] * D + d];
                    }
                }
            }
            if constexpr (TANGENTS) {
#pragma unroll
                for (int i = 0; i < 4; ++i) ee[cb] = __builtin_fmaf(xv[i], xv[i], ee[cb]);
            }
            // the state x lives in LDS: it is touched twice per evaluation, and the live registers would spill inside the loop
            ks[SL(NSLOT + 1, cb, t)] = xv;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int d = 4 * qd + i;
            if (d < C && !is_tangent) cv[i] = args.cond[sample[cb] * C + d];
        }
        if constexpr (DT == 1) {
            split2<NP>(cv[0], cv[1], yf[0][cb], 2);
            split2<NP>(cv[2], cv[3], yf[0][cb], 3);
        } else {
            split2<NP>(cv[0], cv[1], yf[1][cb], 0);
            split2<NP>(cv[2], cv[3], yf[1][cb], 1);
            split2<NP>(0.f, 0.f, yf[1][cb], 2);
            split2<NP>(0.f, 0.f, yf[1][cb], 3);
        }
    }
#pragma unroll
    for (int s = 0; s < NSLOT; ++s)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int t = 0; t < DT; ++t) ks[SL(s, cb, t)] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int i = threadIdx.x; i < H + kZeroSkew / 4; i += 256) ((float*)(lds + M.zero))[i] = 0.f;
    {
        const float* bsrc = args.wpack + (size_t)stream_words(NH, NP, DT, HW);
        const int nb = (NH - 1) * H + 16 * DT;
        for (int i = threadIdx.x; i < nb; i += 256) ((float*)(lds + M.hbias))[i] = bsrc[i];
    }
    float kl[NSLOT][2];
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) kl[s][0] = kl[s][1] = 0.f;
    float lp[2] = {0.f, 0.f};
    // adaptive stepping (one attempted step per launch): stage slot 0 is the derivative at the step start, supplied by
    // the caller (FSAL of the step before); its divergence is carried by ONE lane of the sample's tangent column
    if (args.k1_in) {
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
#pragma unroll
            for (int t = 0; t < DT; ++t) {
                f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int d = 16 * t + 4 * qd + i;
                    if (d < D && !is_tangent) v[i] = args.k1_in[sample[cb] * D + d];
                }
                ks[SL(0, cb, t)] = v;
            }
            if constexpr (TANGENTS) {
                if (args.kl1_in && role == 1 && qd == 0) kl[0][cb] = args.kl1_in[sample[cb]];
            }
        }
    }
#ifdef FF_SPLIT_STAMPS          // diagnostic builds only (scratch/kbench_split.hip): cycle stamps of wavefront 0, evaluations 2 and 3
    int stamp_i = 0;
    bool stamp_on = false;
#define FF_STAMP()                                                                                         \
    do {                                                                                                   \
        if (stamp_on) {                                                                                    \
            const unsigned long long t_ = __builtin_readcyclecounter();                                    \
            if (lane == 0) args.debug_stamps[stamp_i] = t_;                                                \
            ++stamp_i;                                                                                     \
        }                                                                                                  \
    } while (0)
#else
#define FF_STAMP() do {} while (0)
#endif

    // ---- weight pipeline state (all wave-uniform) -----------------------------------------------------------------
    const unsigned char* const wbase = (const unsigned char*)args.wpack;
    const long long wbytes = (long long)granules_per_eval(NH, DT, HW) * GB;
    long long dpos = 0;                                // byte position in the stream of the NEXT granule to fetch
    unsigned rbuf = 0;                                 // LDS byte offset of the buffer the current granule is read from
    const int my_frag = wv * NDMA * 1024;              // this wavefront's quarter of a granule
    auto fetch_c1 = [&](int e) __attribute__((always_inline)) {      // c1 of evaluation e -> its LDS buffer (every wavefront issues
        const int ee_ = e < args.n_evals ? e : 0;                      // the same 1 KiB copy: equal VMEM counts keep the counted waits uniform)
        // (width 128: a row holds 512 bytes of c1 -- the upper lanes re-read its last 16 bytes instead of the next row)
        dma_fragment(M.c1 + (e & 1) * 1024, (const unsigned char*)(args.etab + (size_t)ee_ * args.etab_stride + 32),
                     lane16 < H * 4 - 16 ? lane16 : H * 4 - 16);
    };
    FF_SKEW_HOLD(wv == kSkewWave, 1);                  // (test builds, ff_skew.h: this wavefront starts late ...)
    fetch_c1(0);
    for (int g = 0; g < NBUF - 1; ++g) {               // the first NBUF - 1 granules into their buffers
#pragma unroll
        for (int f = 0; f < NDMA; ++f) dma_fragment(g * GB + my_frag + f * 1024, wbase + dpos + my_frag + f * 1024, lane16);
        dpos += GB;
        if (dpos >= wbytes) dpos = 0;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();

    // bias of row tile rt of the vector at LDS byte `base` (tangent lanes read the zero page): accumulator register order
    const int zsel = M.zero + kZeroSkew + qd * 16;     // (skewed against the bias vectors' banks: ff_split_layout.h)
    auto bias_tile = [&](int base, int rt) __attribute__((always_inline)) {
        const int a = (TANGENTS && is_tangent) ? zsel : base + qd * 16;
        return *(const f32x4*)(lds + a + 64 * rt);
    };

    f32x4 A[NR][2], B[NR][2];
#pragma unroll
    for (int rt = 0; rt < NR; ++rt) {
        A[rt][0] = A[rt][1] = bias_tile(M.c1, rt);
        B[rt][0] = B[rt][1] = bias_tile(M.hbias, rt);  // (unused when there is a single hidden layer)
    }

    // weight fragments of the current group (wq[0]) and of the kAhead groups after it, read from LDS kAhead groups
    // ahead (one group = 192 cycles is enough: distances 2 and 3 measured the same to 0.1 %)
    u32x4 wq[kAhead + 1][NP];
    int wa = lane16;                                   // LDS address of this lane's 16 bytes of fragment 0 of the granule
    int wa_next = lane16;
#pragma unroll
    for (int a = 0; a < kAhead; ++a)
#pragma unroll
        for (int p = 0; p < NP; ++p) wq[a][p] = *(const u32x4*)(lds + wa + (a * NP + p) * 1024);

    // ONE GROUP: the products of (w_hi, w_mid[, w_lo]) x (x_hi, x_mid[, x_lo]) that matter -- hh hm mh hl mm lh with
    // three parts (dropped terms <= 2^-24 relative), hh hm mh with two -- for both column blocks, in an order where
    // consecutive MFMAs never depend on each other, each followed by what is pinned to its gap: a fragment read of the
    // next group, in the group that opens a granule an LDS-DMA of granule + 2, and `fill(gap)`, the caller's activation
    // micro-ops -- and a full scheduling barrier, so that the instruction stream is the one written here.  GQ = the
    // group's index in its granule.
    auto group = [&](auto gq, f32x4 (&acc)[2], const u32x4 (&b)[2][NP], auto&& fill) __attribute__((always_inline)) {
        constexpr int GQ = decltype(gq)::value;
        const unsigned char* dsrc = nullptr;
        unsigned ddst = 0;
        if constexpr (GQ == 0) {
            // The stream position is periodic in the evaluation loop, so hipcc would compute all ~300 fragment
            // addresses of an evaluation once, ahead of the loop, and keep them in (spilled) SGPRs: hide the bases.
            asm volatile("" : "+s"(rbuf));
            unsigned wb = rbuf + (NBUF - 1) * GB;
            if (wb >= NBUF * GB) wb -= NBUF * GB;      // granule + NBUF - 1 -> the buffer read before this one
            dsrc = wbase + dpos + my_frag;
            ddst = wb + my_frag;
            asm volatile("" : "+s"(dsrc), "+s"(ddst));
            dpos += GB;
            if (dpos >= wbytes) dpos = 0;
        }
        if constexpr (GQ == GRG - kAhead) {
            // everything but the NDMA DMAs issued in this granule has landed, and this wavefront's reads of the current
            // buffer have returned: after the barrier the next granule is visible to all and the previous buffer is free
            if constexpr (NBUF == 2) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     // (one granule ahead: all of it)
            else if constexpr (NDMA == 6) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
            else if constexpr (NDMA == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
            else if constexpr (NDMA == 3) asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
#ifndef FF_SPLIT_NOBARRIER       // timing experiment only (with FF_SPLIT_NODMA)
            __builtin_amdgcn_s_barrier();
#endif
            FF_SKEW_HOLD(wv == kSkewWave, 1);          // (... and is late behind every granule barrier)
            rbuf += GB;
            if (rbuf >= NBUF * GB) rbuf = 0;
            wa_next = lane16 + rbuf;
        }
        sfor<GG>([&](auto ii) {
            constexpr int I = decltype(ii)::value;
            constexpr int cb = I & 1, pr = I >> 1;
            constexpr int wp = (pr == 2 || pr == 4) ? 1 : (pr == 5 ? 2 : 0);        // hh hm mh hl mm lh
            constexpr int bp = (pr == 1 || pr == 4) ? 1 : (pr == 3 ? 2 : 0);
            acc[cb] = mm(wq[0][wp], b[cb][bp], acc[cb]);
            constexpr int part = read_part(NP, I);
#ifdef FF_SPLIT_NOWREAD          // timing experiment only: one group's fragments serve the whole granule (wrong results)
            if constexpr (part >= 0 && GQ == 0) {
#else
            if constexpr (part >= 0) {
#endif
                constexpr int T = GQ + kAhead;                          // fragments of the group kAhead groups on
                if constexpr (T >= GRG) wq[kAhead][part] = *(const u32x4*)(lds + wa_next + ((T - GRG) * NP + part) * 1024);
                else wq[kAhead][part] = *(const u32x4*)(lds + wa + (T * NP + part) * 1024);
            }
#ifndef FF_SPLIT_NODMA           // timing experiment only: never refresh the weight buffers (wrong results)
            if constexpr (GQ == 0 && dma_gap(NP, HW, I))
                dma_fragment(ddst + dma_index(NP, I) * 1024, dsrc + dma_index(NP, I) * 1024, lane16);
#endif
            fill(ii);
            __builtin_amdgcn_sched_barrier(0);
        });
#ifdef FF_SPLIT_NOWREAD
        if constexpr (GQ == 0)
#endif
#pragma unroll
        for (int a = 0; a < kAhead; ++a)
#pragma unroll
            for (int p = 0; p < NP; ++p) wq[a][p] = wq[a + 1][p];
        if constexpr (GQ == GRG - 1) wa = wa_next;
    };
    auto no_fill = [](auto) {};

    u32x4 bf[2][2][NP];                                // B fragments [k-step parity][column block][part] in use / in preparation
    UnitState us[8];                                   // activation units in flight
#ifdef FF_SPLIT_SKIP_OPS
#pragma unroll
    for (int i = 0; i < 8; ++i) us[i] = UnitState{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int k = 0; k < NP; ++k) bf[i][j][k] = u32x4{0u, 0u, 0u, 0u};
#endif

    // The micro-ops the plan `KIND` pins to gap G of a span: the units turn row tiles (2 sn, 2 sn + 1) of `T` into the
    // fragments of k-step sn.  Unit u: column block u >> 2, word u & 3 = registers 2 (u & 1), 2 (u & 1) + 1 of row tile
    // 2 sn + ((u >> 1) & 1).
    auto act_gap = [&](auto kind, auto gg, auto snn, f32x4 (&T)[NR][2], u32x4 (&dst)[2][NP]) __attribute__((always_inline)) {
        constexpr int KIND = decltype(kind)::value, G = decltype(gg)::value, sn = decltype(snn)::value;
        sfor<kMaxPerGap>([&](auto kk) {
            constexpr int K = decltype(kk)::value;
            constexpr int N = KIND == 0 ? GapPlans<TANGENTS, NP, HW>::k0.n[G] : GapPlans<TANGENTS, NP, HW>::k1.n[G];
            if constexpr (K < N) {
                constexpr int U = KIND == 0 ? GapPlans<TANGENTS, NP, HW>::k0.unit[G][K] : GapPlans<TANGENTS, NP, HW>::k1.unit[G][K];
                constexpr int J = KIND == 0 ? GapPlans<TANGENTS, NP, HW>::k0.micro[G][K] : GapPlans<TANGENTS, NP, HW>::k1.micro[G][K];
                constexpr int cb = U >> 2, word = U & 3, rt = 2 * sn + (word >> 1), r0 = 2 * (word & 1);
                unit_micro<TM, NP, J>(T[rt][cb][r0], T[rt][cb][r0 + 1], us[U], dst[cb], word, is_tangent, vsrc);
            }
        });
    };

    // A hidden -> hidden layer (reads P, writes Cc) or, with OUT, the output layer (reads P, writes O).
    // `refill` = LDS byte address of the bias vector the tiles of P are refilled with once consumed (their next use).
    auto layer = [&](auto is_out, f32x4 (&P)[NR][2], f32x4 (&Cc)[NR][2], f32x4 (&O)[DT][2], int refill) __attribute__((always_inline)) {
        constexpr bool OUT = decltype(is_out)::value;
        P[0][0] = P[0][1] = bias_tile(refill, 0);      // row tiles 0, 1 were consumed at the end of the layer before
        P[1][0] = P[1][1] = bias_tile(refill, 1);
        sfor<NS>([&](auto ss) {
            constexpr int s = decltype(ss)::value;
            if constexpr (OUT) {
                // DT groups per k-step: the output layer is VALU-bound, its activation units run whole behind them,
                // interleaved so that eight independent chains cover each other's latencies
                sfor<DT>([&](auto tt) {
                    constexpr int t = decltype(tt)::value;
                    group(std::integral_constant<int, (s * DT + t) % GRG>{}, O[t], bf[s & 1], no_fill);
                });
                if constexpr (s < NS - 1) {
                    sfor<micro_count(TANGENTS, NP)>([&](auto jj) {
                        sfor<8>([&](auto uu) {
                            constexpr int U = decltype(uu)::value;
                            constexpr int cb = U >> 2, word = U & 3, prt = 2 * (s + 1) + (word >> 1), r0 = 2 * (word & 1);
                            unit_micro<TM, NP, decltype(jj)::value>(P[prt][cb][r0], P[prt][cb][r0 + 1], us[U],
                                                                   bf[(s + 1) & 1][cb], word, is_tangent, vsrc);
                        });
                    });
                }
            } else {
                sfor<NR>([&](auto tt) {
                    constexpr int rt = decltype(tt)::value;
#ifdef FF_SPLIT_STAMP_GROUPS    // diagnostic: a stamp per group of k-step 3 (every hidden layer)
                    if constexpr (s == 3) FF_STAMP();
#endif
                    group(std::integral_constant<int, rt % GRG>{}, Cc[rt], bf[s & 1], [&](auto ii) {
                        constexpr int G = rt * GG + decltype(ii)::value;
#ifdef FF_SPLIT_STAMP_GAPS      // diagnostic: a stamp behind every MFMA of groups 0 and 1 of k-step 3
                        if constexpr (s == 3 && rt < 2) FF_STAMP();
#endif
#ifdef FF_SPLIT_UNITS_PARITY      // diagnostic: the activation units run in the k-steps of one parity only (wrong results)
                        if constexpr (s % 2 != (FF_SPLIT_UNITS_PARITY)) return;
#endif
                        if constexpr (s < NS - 1)       // the operands of the next k-step, out of the layer before
                            act_gap(std::integral_constant<int, 0>{}, std::integral_constant<int, G>{},
                                    std::integral_constant<int, s + 1>{}, P, bf[(s + 1) & 1]);
                        else                            // last k-step: row tiles 0, 1 of THIS layer's output
                            act_gap(std::integral_constant<int, 1>{}, std::integral_constant<int, G>{},
                                    std::integral_constant<int, 0>{}, Cc, bf[0]);
                    });
                });
            }
            if constexpr (s < NS - 1) {                // row tiles 2s+2, 2s+3 were consumed during this k-step
                P[2 * s + 2][0] = P[2 * s + 2][1] = bias_tile(refill, 2 * s + 2);
                P[2 * s + 3][0] = P[2 * s + 3][1] = bias_tile(refill, 2 * s + 3);
            }
            FF_STAMP();
        });
    };

    bool bad_slot = false;
    for (int e = 0; e < args.n_evals; ++e) {
#ifdef FF_SPLIT_STAMPS
        stamp_on = args.debug_stamps && blockIdx.x == 0 && wv == 0 && (e == 2 || e == 3);
#endif
        FF_STAMP();
        HdrPtr hdr = (HdrPtr)(args.etab + (size_t)e * args.etab_stride);
        const float a_e = hdr->a, b_e = hdr->b;
        const uint32_t flags = hdr->flags;
        const int slot = hdr->slot;

        // Euler-Maruyama rows (state-only kernels): the row's slab of standard normals, requested now and used after the
        // network (include/flowfusion_amd.h: noise[noise_idx * noise_stride + sample * dim + d], or drawn in the kernel)
        f32x4 nz[2][DT];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int t = 0; t < DT; ++t) nz[cb][t] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (!TANGENTS) {
            if ((flags & 2u) && args.noise) {
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) {
                    const float* np = args.noise + (size_t)hdr->noise_idx * args.noise_stride + sample[cb] * D;
#pragma unroll
                    for (int t = 0; t < DT; ++t)
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            if (16 * t + 4 * qd + i < D) nz[cb][t][i] = np[16 * t + 4 * qd + i];
                }
            }
        }

        // stage input  y = x + sum_s cin[s] k[s]  (parked in LDS slot NSLOT for the right-hand side); its fragments
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int t = 0; t < DT; ++t) {
                f32x4 v = ks[SL(NSLOT + 1, cb, t)];
#pragma unroll
                for (int s = 0; s < NSLOT; ++s) v += hdr->cin[s] * ks[SL(s, cb, t)];
                ks[SL(NSLOT, cb, t)] = v;
                split2<NP>(v[0], v[1], yf[0][cb], 2 * t);
                split2<NP>(v[2], v[3], yf[0][cb], 2 * t + 1);
            }

        FF_STAMP();
        // ---- layer 1: [state | conditional] (one k-step) -> H, accumulators A already hold c1_e --------------------
        sfor<DT>([&](auto kk) {
            constexpr int k1 = decltype(kk)::value;                      // k-step of the first layer
            sfor<NR>([&](auto tt) {
                constexpr int rt = decltype(tt)::value;
                group(std::integral_constant<int, rt % GRG>{}, A[rt], yf[k1], [&](auto ii) {
                    constexpr int I = decltype(ii)::value;
                    if constexpr (k1 == 0 && rt == 0 && I == GG - 1) fetch_c1(e + 1);     // (after the weight DMAs of this granule)
                    // last k-step: row tiles 0, 1 are complete after groups 0, 1: their activation rides on the other 14
                    if constexpr (k1 == DT - 1)
                        act_gap(std::integral_constant<int, 1>{}, std::integral_constant<int, rt * GG + I>{},
                                std::integral_constant<int, 0>{}, A, bf[0]);
                });
            });
        });

        FF_STAMP();
        // ---- hidden -> hidden layers, ping-pong A -> B -> A ..., then the output layer ------------------------------
        // refill address of the set being READ by layer j (1-based; j = NH is the output layer): the bias of the next
        // layer that writes that set -- layer j+1 of this evaluation if it is a hidden one, else the first layer of the
        // next evaluation (set A: c1 of e+1) or its first hidden->hidden layer (set B)
        const int c1_next = M.c1 + ((e + 1) & 1) * 1024;
        auto refill_for = [&](int j, bool reads_A) __attribute__((always_inline)) {
            if (j + 1 <= NH - 1) return M.hbias + j * H * 4;
            return reads_A ? c1_next : M.hbias;
        };
        f32x4 O[DT][2];
        sfor<NH>([&](auto jj) {
            constexpr int j = decltype(jj)::value + 1;                  // layers 1 .. NH-1 hidden -> hidden, NH = output
            constexpr bool reads_A = (j & 1) != 0;
            if constexpr (j < NH) {
                if constexpr (reads_A) layer(std::false_type{}, A, B, O, refill_for(j, true));
                else layer(std::false_type{}, B, A, O, refill_for(j, false));
            } else {
#pragma unroll
                for (int t = 0; t < DT; ++t) O[t][0] = O[t][1] = bias_tile(M.hbias + (NH - 1) * H * 4, t);   // output bias (16 DT rows)
                if constexpr (reads_A) layer(std::true_type{}, A, B, O, refill_for(j, true));
                else layer(std::true_type{}, B, A, O, refill_for(j, false));
            }
        });

        FF_STAMP();
        // ---- right-hand side and stage bookkeeping ------------------------------------------------------------------
        // (a row naming a slot beyond the NSLOT on chip -- behind them sit the parked stage input and the state -- is
        // refused: nothing is stored, the status word says so)
        const bool slot_ok = (unsigned)slot < (unsigned)NSLOT;
        bad_slot |= !slot_ok;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            float dot = 0.f;
#pragma unroll
            for (int t = 0; t < DT; ++t) {
                const f32x4 yv = ks[SL(NSLOT, cb, t)];
                f32x4 rhs;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float v = __builtin_fmaf(a_e, yv[i], b_e * O[t][cb][i]);
                    rhs[i] = is_tangent ? 0.f : v;
                }
                if (slot_ok) ks[SL(slot, cb, t)] = rhs;
                if constexpr (TANGENTS) {
                    const f32x4 xv = ks[SL(NSLOT + 1, cb, t)];            // tangent lanes: the probe e
#pragma unroll
                    for (int i = 0; i < 4; ++i) dot = __builtin_fmaf(xv[i], O[t][cb][i], dot);
                }
            }
            if constexpr (TANGENTS) {
                const float div = is_tangent ? __builtin_fmaf(a_e, ee[cb], b_e * dot) : 0.f;
#pragma unroll
                for (int s = 0; s < NSLOT; ++s) kl[s][cb] = (slot == s) ? div : kl[s][cb];
            }
        }
        if (flags & 1u) {
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
#pragma unroll
                for (int t = 0; t < DT; ++t) {
                    f32x4 v = ks[SL(NSLOT + 1, cb, t)];
#pragma unroll
                    for (int s = 0; s < NSLOT; ++s) v += hdr->cout[s] * ks[SL(s, cb, t)];
                    ks[SL(NSLOT + 1, cb, t)] = v;
                }
                if constexpr (TANGENTS) {
#pragma unroll
                    for (int s = 0; s < NSLOT; ++s) lp[cb] = __builtin_fmaf(hdr->cout[s], kl[s][cb], lp[cb]);
                }
            }
        }
        if constexpr (!TANGENTS) {
            if (flags & 2u) {                          // x += g sqrt(|dt|) z
                const float gn = hdr->gn;
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                    for (int t = 0; t < DT; ++t) {
                        if (!args.noise) {             // in-kernel noise: a lane's four registers of tile t are dimensions 16 t + 4 qd ..
                            const unsigned long long gs = (unsigned long long)(sample[cb] + args.rng_sample_offset);
                            uint32_t c[4] = {(uint32_t)gs, (uint32_t)(gs >> 32), (uint32_t)(hdr->noise_idx + args.rng_noise_base),
                                             (uint32_t)(4 * t + qd)};
                            philox4x32_10(c, (uint32_t)args.rng_seed, (uint32_t)(args.rng_seed >> 32));
                            float z0, z1, z2, z3;
                            box_muller(c[0], c[1], z0, z1);
                            box_muller(c[2], c[3], z2, z3);
                            nz[cb][t] = f32x4{z0, z1, z2, z3};
                        }
                        f32x4 v = ks[SL(NSLOT + 1, cb, t)];
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] = __builtin_fmaf(gn, nz[cb][t][i], v[i]);
                        ks[SL(NSLOT + 1, cb, t)] = v;
                    }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the pipeline runs two granules ahead: let it drain

    // ---- epilogue -----------------------------------------------------------------------------------------------------
    bool bad = false;
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        const bool writer = live[cb] && !is_tangent;
        // a sample's divergence = sum over the four quads of its tangent columns (the NT lanes after its value lane)
        auto reduce_tangents = [&](float part) __attribute__((always_inline)) {
            float tot = 0.f;
            for (int j = 1; j <= NT; ++j) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int src = ((g << 4) | ((col + j) & 15)) * 4;
                    tot += __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, part)));
                }
            }
            return tot;
        };
        if constexpr (TANGENTS) {
            const float tot = reduce_tangents(lp[cb]);
            if (writer && qd == 0 && args.dlogp_out)
                args.dlogp_out[sample[cb]] = (args.dlogp_in ? args.dlogp_in[sample[cb]] : 0.f) + tot;
        }
        // auxiliary outputs of an adaptive step: aux_j = use_y_j * y + sum_s coef_j[s] * k[s] (new state, last stage,
        // dense-output midpoint, error estimate); coefficients in the two rows that follow the evaluation rows
        if (args.n_aux > 0) {
            HdrPtr t0h = (HdrPtr)(args.etab + (size_t)args.n_evals * args.etab_stride);
            HdrPtr t1h = (HdrPtr)(args.etab + (size_t)(args.n_evals + 1) * args.etab_stride);
            const uint32_t use_y = t0h->flags;
            sfor<kAux>([&](auto jj) {
                constexpr int j = decltype(jj)::value;
                if (j < args.n_aux) {
                    HdrPtr th = (j < 2) ? t0h : t1h;
                    float c[NSLOT];
#pragma unroll
                    for (int s = 0; s < NSLOT; ++s) c[s] = (j & 1) ? th->cout[s] : th->cin[s];
                    const float uy = ((use_y >> j) & 1u) ? 1.f : 0.f;
#pragma unroll
                    for (int t = 0; t < DT; ++t) {
                        f32x4 v = uy * ks[SL(NSLOT + 1, cb, t)];
#pragma unroll
                        for (int s = 0; s < NSLOT; ++s) v += c[s] * ks[SL(s, cb, t)];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int d = 16 * t + 4 * qd + i;
                            if (writer && d < D && args.aux_out[j]) args.aux_out[j][sample[cb] * D + d] = v[i];
                        }
                    }
                    if constexpr (TANGENTS) {
                        float part = 0.f;
#pragma unroll
                        for (int s = 0; s < NSLOT; ++s) part = __builtin_fmaf(c[s], kl[s][cb], part);
                        const float tot = reduce_tangents(part);
                        if (writer && qd == 0 && args.aux_lp_out[j])
                            args.aux_lp_out[j][sample[cb]] = uy * (args.dlogp_in ? args.dlogp_in[sample[cb]] : 0.f) + tot;
                    }
                }
            });
        }
#pragma unroll
        for (int t = 0; t < DT; ++t) {
            const f32x4 xv = ks[SL(NSLOT + 1, cb, t)];
            if (writer) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int d = 16 * t + 4 * qd + i;
                    if (d < D) {
#pragma clang fp contract(off)      // x * scale + shift as two roundings, like the reference's torch expression
                        float v = xv[i];
                        bad |= (v != v);
                        if (args.out_scale) v = v * args.out_scale[d];
                        if (args.out_shift) v = v + args.out_shift[d];
                        args.x_out[sample[cb] * D + d] = v;
                    }
                }
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

} // namespace split
} // namespace ff
