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
// A operands are stored as CHUNKS of 1 KiB: 64 lanes x 4 floats; lane l, float q of the chunk
// for (group g, output block ob) of a layer is
//       W[ ob*32 + (l & 31) ][ kcol(4*g + q, l >> 5) ]
// so one 16-byte load per lane feeds 4 MFMAs.  The chunks of one evaluation form a single
// linear stream in the exact order the kernel consumes them (layer 1, hidden layers, output
// layer), which lets the kernel prefetch with one uniform sliding window, across layer and
// evaluation boundaries.  Within a layer of G groups and NOB blocks the order is
//   phase A (groups 0 .. G-5): group-major  -- all NOB accumulators advance together;
//   phase B (last min(G,4) groups): block-major -- accumulator ob completes after its last
//           chunk, so its activation overlaps the MFMAs of block ob+1.
// Each layer's chunk count is padded to a multiple of kChunkPad (zero chunks, loaded but
// unused) so the prefetch ring position is a compile-time constant everywhere.
#pragma once
#include <stdint.h>
#include <stddef.h>

#if defined(__HIPCC__) || defined(__CUDACC__)
#define FF_HD __host__ __device__ inline
#else
#define FF_HD inline
#endif

namespace ff {

constexpr int kChunkFloats = 256;   // 1 KiB
constexpr int kChunkPad = 16;       // layer chunk counts are padded to a multiple of this

// feature index held by activation/state register r on lane-half h
FF_HD constexpr int feat_of_reg(int r, int h) { return 32 * (r >> 4) + (r & 3) + 8 * ((r & 15) >> 2) + 4 * h; }

// number of lane registers needed for F features (multiple of 4 registers = 8 features)
FF_HD constexpr int regs_for(int F) { return 4 * ((F + 7) / 8); }

// output blocks (32 rows) needed to produce `regs` registers
FF_HD constexpr int blocks_for_regs(int regs) { return (regs + 15) / 16; }

// Chunk order of one layer with KR operand registers and NOB output blocks.
struct LayerGeom {
    int G;      // groups of 4 k-steps
    int NOB;    // output blocks
    int GB;     // groups in phase B (block-major tail)
    int GA;     // groups in phase A (group-major)
    int NC;     // real chunks = G * NOB
    int CPAD;   // padded chunk count
};

FF_HD constexpr LayerGeom layer_geom(int KR, int NOB)
{
    LayerGeom g{};
    g.G = KR / 4;
    g.NOB = NOB;
    g.GB = g.G < 4 ? g.G : 4;
    g.GA = g.G - g.GB;
    g.NC = g.G * NOB;
    g.CPAD = (g.NC + kChunkPad - 1) / kChunkPad * kChunkPad;
    return g;
}

// chunk index -> group (consumption order)
FF_HD constexpr int chunk_group(const LayerGeom& L, int c)
{
    return c < L.GA * L.NOB ? c / L.NOB : L.GA + (c - L.GA * L.NOB) % L.GB;
}
// chunk index -> output block
FF_HD constexpr int chunk_block(const LayerGeom& L, int c)
{
    return c < L.GA * L.NOB ? c % L.NOB : (c - L.GA * L.NOB) / L.GB;
}
// true if chunk c is the last one that touches its output block
FF_HD constexpr bool chunk_completes_block(const LayerGeom& L, int c)
{
    return c >= L.GA * L.NOB && (c - L.GA * L.NOB) % L.GB == L.GB - 1;
}

struct Layout {
    int H;        // hidden width on chip (multiple of 32)
    int NB;       // H / 32
    int dregs;    // state registers
    int cregs;    // conditional registers
    int n_hidden; // hidden layers
    int nob_out;  // output blocks of the last layer
    LayerGeom g1, gh, go;   // first layer, hidden->hidden, output layer

    // chunk stream (in chunks)
    int chunks_l1, chunks_hid, chunks_out, chunks_total;
    // sizes in floats
    size_t stream_floats;   // chunks_total * 256
    size_t bias_floats;     // (n_hidden-1) * H + nob_out*32
    size_t total_floats;

    FF_HD int chunk_off_hid(int l) const { return chunks_l1 + l * chunks_hid; }
    FF_HD int chunk_off_out() const { return chunks_l1 + (n_hidden - 1) * chunks_hid; }
    FF_HD size_t bias_off_hid(int l) const { return stream_floats + (size_t)l * H; }
    FF_HD size_t bias_off_out() const { return stream_floats + (size_t)(n_hidden - 1) * H; }
};

FF_HD constexpr Layout make_layout(int H, int dregs, int cregs, int n_hidden)
{
    Layout L{};
    L.H = H; L.NB = H / 32; L.dregs = dregs; L.cregs = cregs; L.n_hidden = n_hidden;
    L.nob_out = blocks_for_regs(dregs);
    L.g1 = layer_geom(dregs + cregs, L.NB);
    L.gh = layer_geom(L.NB * 16, L.NB);
    L.go = layer_geom(L.NB * 16, L.nob_out);
    L.chunks_l1 = L.g1.CPAD;
    L.chunks_hid = L.gh.CPAD;
    L.chunks_out = L.go.CPAD;
    L.chunks_total = L.chunks_l1 + (n_hidden - 1) * L.chunks_hid + L.chunks_out;
    L.stream_floats = (size_t)L.chunks_total * kChunkFloats;
    L.bias_floats = (size_t)(n_hidden - 1) * H + (size_t)L.nob_out * 32;
    L.total_floats = L.stream_floats + L.bias_floats;
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
