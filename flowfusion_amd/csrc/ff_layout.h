// ff_layout.h -- tiling shared by the host-side weight packer and the gfx950 kernels.
//
// Geometry (one wavefront = 64 lanes, v_mfma_f32_32x32x2_f32):
//   * samples (and tangent columns) sit on the MFMA column index  col  = lane & 31;
//   * features sit on the MFMA row index; an accumulator tile of 32 rows x 32 columns
//     is 16 registers per lane, register q of lane-half h = lane >> 5 holding row
//         rho(q, h) = (q & 3) + 8 * (q >> 2) + 4 * h                     (CDNA4 C/D map)
//   * a vector of F features per column is therefore F/2 registers per lane; register
//     r holds feature  feat(r, h) = 32 * (r >> 4) + rho(r & 15, h).
//   * the B operand of the f32 MFMA is one register per lane: lanes 0-31 give k = 0,
//     lanes 32-63 give k = 1.  Feeding activation register r straight back as B means
//     this k-step contracts over the feature pair { feat(r,0), feat(r,1) } -- so the
//     accumulator of one layer IS the operand of the next, with no data movement, as
//     long as the A operand (the weights) is packed in the matching order.  That
//     packing is what this header defines.
//
// Packed A stream of one layer with KR operand registers (k-steps) and NOB output blocks
// of 32 rows: for g in [0, KR/4): for ob in [0, NOB): 64 lanes x 4 floats, lane l, float q:
//       W[ ob*32 + (l & 31) ][ kcol(4*g + q, l >> 5) ]
// i.e. 1 KiB per (g, ob), read by one 16-byte load per lane that feeds 4 MFMAs.
#pragma once
#include <stdint.h>
#include <stddef.h>

#if defined(__HIPCC__) || defined(__CUDACC__)
#define FF_HD __host__ __device__ inline
#else
#define FF_HD inline
#endif

namespace ff {

// feature index held by activation/state register r on lane-half h
FF_HD int feat_of_reg(int r, int h) { return 32 * (r >> 4) + (r & 3) + 8 * ((r & 15) >> 2) + 4 * h; }

// number of lane registers needed for F features (multiple of 4 registers = 8 features)
FF_HD int regs_for(int F) { return 4 * ((F + 7) / 8); }

// output blocks (32 rows) needed to produce `regs` registers
FF_HD int blocks_for_regs(int regs) { return (regs + 15) / 16; }

struct Layout {
    int H;        // hidden width on chip (multiple of 32)
    int NB;       // H / 32
    int dregs;    // state registers
    int cregs;    // conditional registers
    int n_hidden; // hidden layers
    int nob_out;  // output blocks of the last layer

    // sizes in floats
    size_t l1_floats;      // first layer: (dregs+cregs)/4 groups x NB blocks x 256
    size_t hid_w_floats;   // one hidden->hidden layer weights: NB*4 groups x NB x 256
    size_t hid_floats;     // + bias H
    size_t out_w_floats;   // output layer weights: NB*4 groups x nob_out x 256
    size_t out_floats;     // + bias nob_out*32
    size_t total_floats;

    FF_HD size_t off_l1() const { return 0; }
    FF_HD size_t off_hid(int l /* 0-based index of hidden->hidden layer */) const { return l1_floats + (size_t)l * hid_floats; }
    FF_HD size_t off_out() const { return l1_floats + (size_t)(n_hidden - 1) * hid_floats; }
};

FF_HD Layout make_layout(int H, int dregs, int cregs, int n_hidden)
{
    Layout L;
    L.H = H; L.NB = H / 32; L.dregs = dregs; L.cregs = cregs; L.n_hidden = n_hidden;
    L.nob_out = blocks_for_regs(dregs);
    L.l1_floats = (size_t)((dregs + cregs) / 4) * L.NB * 256;
    L.hid_w_floats = (size_t)(L.NB * 4) * L.NB * 256;
    L.hid_floats = L.hid_w_floats + H;
    L.out_w_floats = (size_t)(L.NB * 4) * L.nob_out * 256;
    L.out_floats = L.out_w_floats + (size_t)L.nob_out * 32;
    L.total_floats = L.l1_floats + (size_t)(n_hidden - 1) * L.hid_floats + L.out_floats;
    return L;
}

// Evaluation-row header (FF_ROW_HDR = 32 four-byte words), followed by c1[H].
struct RowHdr {
    float a;            // 0  multiplies the stage state in the RHS
    float b;            // 1  multiplies the network output in the RHS
    float gn;           // 2  noise coefficient g * sqrt(|dt|)
    uint32_t flags;     // 3  FF_ROW_STEP_END | FF_ROW_NOISE
    int32_t slot;       // 4  stage slot that receives this RHS
    int32_t noise_idx;  // 5  slab of the noise array used by this row
    float pad0[2];      // 6,7
    float cin[8];       // 8..15  stage input  y = x + sum_s cin[s] * k[s]   (first 6 used)
    float cout[8];      // 16..23 step update  x += sum_s cout[s] * k[s]     (first 6 used)
    float pad1[8];      // 24..31
};
static_assert(sizeof(RowHdr) == 32 * 4, "row header is 32 words");

} // namespace ff
