// ff_split_layout.h -- weight stream and LDS map of the split-precision (FF_PREC_BF16X3) kernels, shared by the
// host-side packer (ff_api.cpp) and the gfx950 kernel (ff_mlp_ode_split.hpp).
//
// MFMA: v_mfma_f32_32x32x16_bf16.  The A operand (weights) of one k-step is a FRAGMENT: 64 lanes x 8 bf16 = 1 KiB,
// lane l = (row l & 31 of a 32-row tile, k half l >> 5), element j of the lane = input feature kidx(s, l >> 5, j) of
// k-step s.  kidx is the order in which the registers of an fp32 accumulator tile (register i of lane half h = row
// (i & 3) + 8 (i >> 2) + 4 h) line up as B operands of the next layer: registers 8u .. 8u+7 of row tile t are k-step
// 2t + u.  A GROUP = the three fragments [hi, mid, lo] of one (row tile, k-step) = 6 MFMAs; a GRANULE = 8 groups =
// 24 KiB, the unit of the LDS pipeline.  Stream of one evaluation, in consumption order (NT = H / 32 row tiles):
//     layer 1          for k-step s < K1S (0: state dimensions, 1: conditional inputs):  for tile t:  group (t, s)
//     hidden layer l   for pair p < NT:  for tile t < NT:  groups (t, 2p), (t, 2p + 1)               l = 1 .. NH-1
//     output layer     for k-step s < 2 NT:  group (tile 0, s)
// every layer padded to whole granules.  Behind the stream: the fp32 biases of the hidden->hidden layers
// [(NH-1)][H] and of the output layer [32] (the first layer's bias travels in the evaluation table as c1_e).
#pragma once
#include <stdint.h>
#include <stddef.h>
#include "ff_layout.h"

namespace ff {
namespace split {

constexpr int kFragBytes = 1024;
constexpr int kGroupFrags = 3;
constexpr int kGranuleGroups = 8;
constexpr int kGranuleBytes = kFragBytes * kGroupFrags * kGranuleGroups;     // 24 KiB
constexpr int kBuffers = 3;

// input feature held by element j (0..7) of lane half h in the fragment of k-step s
FF_HD constexpr int kidx(int s, int h, int j) { return 16 * s + 8 * (j >> 2) + 4 * h + (j & 3); }

FF_HD constexpr int pad8(int groups) { return (groups + kGranuleGroups - 1) / kGranuleGroups * kGranuleGroups; }
FF_HD constexpr int groups_l1(int nt, int k1s) { return pad8(nt * k1s); }
FF_HD constexpr int groups_hid(int nt) { return pad8(2 * nt * nt); }
FF_HD constexpr int groups_out(int nt) { return pad8(2 * nt); }
FF_HD constexpr int granules_per_eval(int nt, int k1s, int n_hidden)
{
    return (groups_l1(nt, k1s) + (n_hidden - 1) * groups_hid(nt) + groups_out(nt)) / kGranuleGroups;
}
// 4-byte words of the fragment stream / of the whole packed buffer
FF_HD constexpr size_t stream_words(int nt, int k1s, int n_hidden)
{
    return (size_t)granules_per_eval(nt, k1s, n_hidden) * (kGranuleBytes / 4);
}
FF_HD constexpr size_t total_words(int nt, int k1s, int n_hidden)
{
    return stream_words(nt, k1s, n_hidden) + (size_t)(n_hidden - 1) * 32 * nt + 32;
}

// LDS map (byte offsets) of a workgroup of 4 wavefronts
struct LdsMap {
    int wbuf;    // kBuffers x 24 KiB weight granules
    int slots;   // Runge-Kutta stage slots + the parked stage input y + the state x: (kSlots + 2) x 2 x 256 threads x 16 B
    int c1;      // 2 x H floats: first-layer bias of the current / next evaluation
    int hbias;   // (NH-1) x H floats + 32: hidden->hidden and output biases
    int zero;    // H floats of zeros (what tangent columns read instead of a bias)
    int total;
};
FF_HD constexpr LdsMap lds_map(int H, int n_hidden)
{
    LdsMap m{};
    m.wbuf = 0;
    m.slots = kBuffers * kGranuleBytes;
    m.c1 = m.slots + (7 + 2) * 2 * 256 * 16;
    m.hbias = m.c1 + 2 * H * 4;
    const int nh1 = n_hidden - 1 > 1 ? n_hidden - 1 : 1;
    m.zero = m.hbias + nh1 * H * 4 + H * 4;      // (one spare vector: a tile read of the 32-float output bias stays inside)
    m.total = m.zero + H * 4;
    return m;
}

} // namespace split
} // namespace ff
