// ff_split_layout.h -- weight stream and LDS map of the split-precision (FF_PREC_BF16X3) kernels, shared by the
// host-side packer (ff_api.cpp) and the gfx950 kernel (ff_mlp_ode_split.hpp).
//
// MFMA: v_mfma_f32_16x16x32_bf16.  The A operand (weights) of one (16-row tile, k-step of 32 features) is a FRAGMENT:
// 64 lanes x 8 bf16 = 1 KiB, lane l = (row l & 15 of the tile, quad q = l >> 4), element j of the lane = input feature
// kidx(s, q, j) of k-step s.  kidx is the order in which the registers of two fp32 accumulator tiles line up as the B
// operand of the next layer: a 16x16 accumulator tile keeps rows 4 q + i (i < 4) on quad q, the B operand wants k slots
// 8 q + j (j < 8) there, so slots j < 4 of k-step s are rows 4 q + j of row tile 2s and slots j >= 4 are rows
// 4 q + (j - 4) of row tile 2s + 1.  A GROUP = the fragments [hi, mid(, lo)] of one (row tile, k-step) = 12 MFMAs with three
// parts (six products x two column blocks of 16 samples), 6 with two; a GRANULE = 8 groups, the unit of the LDS pipeline.
// Stream of one evaluation, in consumption order (width 256: 16 row tiles, 8 k-steps):
//     layer 1          one k-step (features 0..15 = state dimensions, 16..31 = conditional inputs): for row tile rt
//                      (dim <= 32: two k-steps -- features 0..31 the state, then 0..15 the conditional inputs)
//     hidden layer l   for k-step s < 8:  for row tile rt < 16:  group (rt, s)                       l = 1 .. NH-1
//     output layer     one row tile (the state's 16 dimensions):  for k-step s < 8:  group (0, s)
//                      (dim <= 32: two row tiles: for k-step s: group (0, s), group (1, s))
// Behind the stream: the fp32 biases of the hidden->hidden layers [(NH-1)][256] and of the output layer [16] (the
// first layer's bias travels in the evaluation table as c1_e).
#pragma once
#include <stdint.h>
#include <stddef.h>
#include "ff_layout.h"

namespace ff {
namespace split {

constexpr int kWidth = 256;           // widest network of the family
constexpr int kFragBytes = 1024;
constexpr int kBuffers = 3;

// WIDTH of the on-chip layers: 256 (16 row tiles of 16 rows, 8 k-steps of 32 features) or 128 (8 row tiles, 4 k-steps:
// networks up to 128 wide -- the reference's demo sizes -- do a quarter of the matrix work of the 256-wide kernels).
FF_HD constexpr int row_tiles(int w) { return w / 16; }
FF_HD constexpr int ksteps(int w) { return w / 32; }
FF_HD constexpr int granule_groups(int w) { return w >= 256 ? 8 : 4; }     // groups per granule: layer boundaries stay granule-aligned

// PARTS = bf16 parts per fp32 operand: 3 (FF_PREC_BF16X3: hi / mid / lo by truncation, exact; six products per term) or
// 2 (FF_PREC_BF16X2: hi / mid by round-to-nearest, 16 significand bits; three products per term).  A group carries
// `parts` fragments.
FF_HD constexpr int products_of(int parts) { return parts == 3 ? 6 : 3; }
FF_HD constexpr int granule_bytes(int parts, int w = kWidth) { return kFragBytes * parts * granule_groups(w); }   // 24 / 16 KiB at width 256

// input feature held by element j (0..7) of quad q in the fragment of k-step s
FF_HD constexpr int kidx(int s, int q, int j) { return 32 * s + 16 * (j >> 2) + 4 * q + (j & 3); }

// DT = 16-dimension tiles of the state: 1 (dim <= 16: state and conditional inputs share the first layer's single
// k-step) or 2 (dim <= 32: the first layer takes two k-steps, state then conditional inputs; the output layer two row
// tiles).  Stage slots kept on chip: 7 (DT = 1: everything up to Dormand-Prince) or 4 (DT = 2: up to the Runge-Kutta
// 4 schemes and Euler-Maruyama -- the slots of 32 dimensions take twice the LDS).
FF_HD constexpr int groups_hid(int w = kWidth) { return row_tiles(w) * ksteps(w); }            // 128 / 32
FF_HD constexpr int groups_l1(int dt, int w = kWidth) { return row_tiles(w) * dt; }             // 16 / 32 (width 256)
FF_HD constexpr int groups_out(int dt, int w = kWidth) { return ksteps(w) * dt; }               // 8 / 16 (width 256)
FF_HD constexpr int slots_on_chip(int dt) { return dt == 1 ? 7 : 4; }
// ... and a four-slot twin of the 128-wide two-part kernels for states of up to 16 dimensions: with 48 KiB of slots
// instead of 72 a workgroup takes under half of a CU's LDS and 256 registers per lane suffice, so TWO workgroups share a
// CU (two wavefronts per SIMD) and one's activation instructions fill the other's MFMA gaps (+26 %, measured).
// (the three-part twin also drops the third weight buffer -- the DMA then runs one granule ahead instead of two -- to get
// under half of the LDS)
FF_HD constexpr bool has_four_slot_twin(int parts, int dt, int w) { return (parts == 2 || parts == 3) && dt == 1 && w == 128; }
FF_HD constexpr int weight_buffers(int parts, int slots) { return (slots == 4 && parts == 3) ? 2 : kBuffers; }
FF_HD constexpr int granules_per_eval(int n_hidden, int dt = 1, int w = kWidth)
{
    return (groups_l1(dt, w) + (n_hidden - 1) * groups_hid(w) + groups_out(dt, w)) / granule_groups(w);
}
// 4-byte words of the fragment stream / of the whole packed buffer (behind the stream: the hidden->hidden biases and the
// 16 dt output biases)
FF_HD constexpr size_t stream_words(int n_hidden, int parts, int dt = 1, int w = kWidth)
{
    return (size_t)granules_per_eval(n_hidden, dt, w) * (granule_bytes(parts, w) / 4);
}
FF_HD constexpr size_t total_words(int n_hidden, int parts, int dt = 1, int w = kWidth)
{
    return stream_words(n_hidden, parts, dt, w) + (size_t)(n_hidden - 1) * w + 16 * dt;
}

// Tangent lanes read zeros where value lanes read a bias.  Every bias vector starts at a multiple of 256 B (one LDS bank
// row), and so did the zero page: in each 16-lane group of a ds_read_b128 the value lanes of a quad and the tangent lanes
// of the same quad then hit the same four banks at two addresses -- a 2-way conflict on every bias read of the
// divergence kernels (round 2 PMC: SQ_LDS_BANK_CONFLICT 6.8e9 of 1.03e11 LDS cycles in the Hutchinson kernel, 0 in the
// state-only one).  The tangent lanes' reads are therefore skewed by a quarter of the bank row.
constexpr int kZeroSkew = 64;

// LDS map (byte offsets) of a workgroup of 4 wavefronts
struct LdsMap {
    int wbuf;    // kBuffers weight granules
    int slots;   // Runge-Kutta stage slots + the parked stage input y + the state x: (slots + 2) x 2 column blocks x dt x 256 threads x 16 B
    int c1;      // 2 x 1 KiB: first-layer bias of the current / next evaluation (one LDS-DMA fragment each)
    int hbias;   // (NH-1) x H floats + 16 dt: hidden->hidden and output biases
    int zero;    // H + 16 floats of zeros (what tangent columns read instead of a bias, 64 B into the page: see kZeroSkew)
    int total;
};
FF_HD constexpr LdsMap lds_map(int H, int n_hidden, int parts, int dt = 1, int slots = 0)
{
    LdsMap m{};
    m.wbuf = 0;
    m.slots = ((slots == 4 && dt == 1) ? weight_buffers(parts, 4) : kBuffers) * granule_bytes(parts, H);
    m.c1 = m.slots + ((slots > 0 ? slots : slots_on_chip(dt)) + 2) * 2 * dt * 256 * 16;
    m.hbias = m.c1 + 2 * 1024;
    const int nh1 = n_hidden - 1 > 1 ? n_hidden - 1 : 1;
    m.zero = m.hbias + nh1 * H * 4 + 256 * 4;    // (a spare KiB: a tile read of the 16 dt output biases stays inside)
    m.total = m.zero + 256 * 4 + 256;
    return m;
}

} // namespace split
} // namespace ff
